#!/bin/bash
# Board power / clocks while the train loop runs (evidence for "the step runs at the power limit" in DESIGN.md).
python bench.py --no_cpu --steps 600 --warmup 10 --sampler_steps 0 --no_profile > gpurun_out/power_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|GPU use" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.7
done | grep -v "(95Mhz)" | tail -14
tail -1 gpurun_out/power_bench.log | cut -c1-160
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (W)"
