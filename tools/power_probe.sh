#!/bin/bash
# Board power / shader clock while (1) the train loop and (2) the sampler loop run: the numbers behind "the step runs at the board's power budget" (DESIGN.md sections 4, 6).
# usage: tools/power_probe.sh   -> gpurun_out/power_probe.txt   (rocm-smi is read-only here; nothing is set)
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
: > $OUT
probe() {   # $1 = label, rest = command
  local label=$1; shift
  "$@" > /tmp/power_cmd.log 2>&1 &
  local BP=$!
  sleep 25                                   # imports, packs, warm-up
  local n=0
  while kill -0 $BP 2>/dev/null && [ $n -lt 40 ]; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' ' | sed "s/^/$label: /" >> $OUT
    echo >> $OUT
    n=$((n+1)); sleep 0.5
  done
  wait $BP
}
probe train python bench.py --no_cpu --steps 900 --warmup 10 --sampler_steps 0 --no_profile --others 0
probe sampler python tools/sampler_probe.py cfg2 3000
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" >> $OUT
python - <<'PY'
import re, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/power_probe.txt"):
    m = re.match(r"(\w+): \((\d+)Mhz\) ([\d.]+)", l)
    if m and int(m.group(2)) > 200: acc[m.group(1)].append((float(m.group(3)), int(m.group(2))))
for k, v in acc.items():
    print(f"{k}: {len(v)} samples, power mean {sum(p for p, _ in v) / len(v):.0f} W (min {min(p for p, _ in v):.0f}, max {max(p for p, _ in v):.0f}), sclk mean {sum(c for _, c in v) / len(v):.0f} MHz")
PY
tail -2 $OUT
# (samples are (sclk) power; the first seconds of each command - imports, packs, warm-up - are skipped)
