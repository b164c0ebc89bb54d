#!/bin/bash
# Idle time between consecutive kernels of the sampler loop (kernel trace of tools/sampler_probe.py): usage tools/gap_probe.sh cfg1 [steps]
CFG=${1:-cfg1}; N=${2:-30}
OUT=/tmp/gmk_gap; KEEP=gpurun_out/gap; mkdir -p $OUT $KEEP
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO"
rocprofv3 --kernel-trace --output-format csv -d $OUT/$CFG -o gap -- python tools/sampler_probe.py $CFG $N > $OUT/$CFG.log 2>&1 || exit 1
python - <<PY
import csv, glob
f = glob.glob("$OUT/$CFG/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# the DDIM guidance-off loop: take the middle third of the trace
n = len(rows); seg = rows[n // 3: 2 * n // 3]
busy = sum(e - s for s, e, _ in seg); span = seg[-1][1] - seg[0][0]
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
pos = [g for g in gaps if g > 0]
print("$CFG kernels %d span %.2f ms busy %.2f ms (%.1f %%) idle %.2f ms; gaps: mean %.2f us median %.2f us max %.1f us" % (
    len(seg), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, sum(pos) / max(1, len(pos)) / 1e3, sorted(pos)[len(pos) // 2] / 1e3, max(pos) / 1e3))
PY
