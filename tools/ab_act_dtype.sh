#!/bin/bash
# A/B of the 16-bit storage modes on one box: GMK_ACT_DTYPE=bf16 (all-bf16, rounds 1-2) vs fp16 (fp16 forward + bf16 gradients).
# Whole-step numbers (bench.py, interleaved twice) and per-kernel averages (rocprofv3 --kernel-trace --stats, serial train steps).
# usage: tools/ab_act_dtype.sh [cfg]   -> gpurun_out/ab_act/
CFG=${1:-cfg2}
KEEP=gpurun_out/ab_act; OUT=/tmp/gmk_ab; REPO=$(pwd)
mkdir -p $KEEP $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
for rep in 1 2; do for m in bf16 fp16; do
  GMK_ACT_DTYPE=$m python bench.py --config $CFG --others 0 --sampler_steps 200 --no_cpu --no_profile --steps 30 > $OUT/bench_${m}_$rep.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('$OUT/bench_${m}_$rep.json')); print('$m rep $rep:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['sampler']['steps_per_sec'], 'DDIM steps/s', flush=True)" | tee -a $KEEP/summary.txt
done; done
for m in bf16 fp16; do
  GMK_ACT_DTYPE=$m GMK_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$m -o serial -- python bench.py --config $CFG --others 0 --sampler_steps 0 --no_profile --no_cpu --steps 10 --warmup 3 > $OUT/serial_$m.log 2>&1 || exit 1
  cp $(find $OUT/serial_$m -name "*kernel_stats.csv" | head -1) $KEEP/${CFG}_${m}_kernel_stats.csv
done
python - <<PY | tee -a $KEEP/summary.txt
import csv
def load(m):
    return {r["Name"].split("(")[0][:70]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 13 / 1e6) for r in csv.DictReader(open("$KEEP/${CFG}_%s_kernel_stats.csv" % m))}
a, b = load("bf16"), load("fp16")
import re
def key(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    return n[m.end():m.end() + int(m.group(1))] if m else re.sub(r"<.*", "", n)
agg = {}
for tag, d in (("bf16", a), ("fp16", b)):
    for n, (c, ms) in d.items():
        e = agg.setdefault(key(n), {"bf16": 0.0, "fp16": 0.0}); e[tag] += ms
print("ms per step by kernel family (13 steps incl. warm-up):")
for n, e in sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:16]:
    print(f"  {n[:60]:60s} bf16 {e['bf16']:7.3f}  fp16 {e['fp16']:7.3f}  delta {e['fp16'] - e['bf16']:+.3f}")
print("  total", round(sum(e['bf16'] for e in agg.values()), 2), round(sum(e['fp16'] for e in agg.values()), 2))
PY
