#!/bin/bash
# Per-kernel comparison of the SAMPLER forward between the storage modes (rocprofv3 --kernel-trace --stats of tools/sampler_probe.py).
CFG=${1:-cfg2}; N=${2:-40}
KEEP=gpurun_out/ab_act; OUT=/tmp/gmk_abs; REPO=$(pwd)
mkdir -p $KEEP $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
for m in bf16 fp16; do
  GMK_ACT_DTYPE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$m -o s -- python tools/sampler_probe.py $CFG $N > $OUT/s_$m.log 2>&1 || exit 1
  cp $(find $OUT/s_$m -name "*kernel_stats.csv" | head -1) $KEEP/${CFG}_${m}_sampler_kernel_stats.csv
  tail -1 $OUT/s_$m.log
done
python - <<PY
import csv, re
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m: name = name[m.end():m.end() + int(m.group(1))]
    return name.split("(")[0].split("<")[0].strip()
def load(m):
    d = {}
    for r in csv.DictReader(open("$KEEP/${CFG}_%s_sampler_kernel_stats.csv" % m)):
        k = short(r["Name"]); e = d.setdefault(k, [0, 0.0]); e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])
    return d
a, b = load("bf16"), load("fp16")
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, [0, 0])[1], b.get(k, [0, 0])[1]))[:12]:
    ca, xa = a.get(k, [0, 0.0]); cb, xb = b.get(k, [0, 0.0])
    print(f"{k:30s} calls {ca:5d} bf16 {xa / ta * 100:5.1f} %  avg {xa / max(ca, 1) / 1e3:7.1f} us | fp16 avg {xb / max(cb, 1) / 1e3:7.1f} us ({(xb / xa - 1) * 100 if xa else 0:+5.1f} %)")
print(f"total GPU time: fp16 / bf16 = {tb / ta:.4f}")
PY
