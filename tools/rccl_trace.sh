#!/bin/bash
# Round 6: RCCL beside the backward pass on the one GPU a box has.  One rank, GMK_FORCE_EXCHANGE=1 (parallel.exchanging): the four bucket
# all-reduces are issued from the exchange stream behind the data-gradient and weight-gradient streams, the persistent kernels run under the
# carved CU limit while they fly.  Writes the kernel-trace summary of `bench.py --config cfg2 --others 0` (program directly after `--`) and
# the bench line with its `exchange` block under gpurun_out/<tag>/; copy the summaries into profiles/.
TAG=${1:-r06_rccl}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/gmk_$TAG
KEEP=$REPO/gpurun_out/$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
cd $REPO
export GMK_FORCE_EXCHANGE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o rccl -- python bench.py --config cfg2 --others 0 --steps 10 --warmup 3 --sampler_steps 0 --no_cpu --no_profile > $KEEP/bench_forced_traced.json 2> $KEEP/bench_forced_traced.err || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $KEEP/rccl_forced_kernel_stats.csv
# every kernel of the trace that is not one of ours (RCCL's, torch's): name, calls, total / average duration
python - <<PY
import csv
rows = list(csv.DictReader(open("$KEEP/rccl_forced_kernel_stats.csv")))
ours = ("conv", "gn_", "gemm", "adam", "q_sample", "v_loss", "temb", "guide", "label", "silu", "expand3x3", "wgrad", "head_", "chansum", "colsum", "pack_", "rng_", "mean", "sumpool", "scale_rows", "stem", "cast", "attn", "sampler", "logsnr", "xform", "reduce")
with open("$KEEP/rccl_forced_foreign_kernels.txt", "w") as f:
    for r in rows:
        n = r["Name"]
        if not any(k in n for k in ours):
            f.write(f'{r["Calls"]:>6} calls  total {float(r["TotalDurationNs"]) / 1e3:10.1f} us  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {n[:160]}\n')
print(open("$KEEP/rccl_forced_foreign_kernels.txt").read())
PY
# the same bench, untraced: the `exchange` block (exposed_ms, both carve-out settings) of an N = 1 line measured over RCCL
python bench.py --config cfg2 --others 0 --steps 20 --warmup 5 --sampler_steps 0 --no_cpu > $KEEP/bench_forced.json 2> $KEEP/bench_forced.err || exit 1
cp gpurun_out/bench_detail.json $KEEP/bench_forced_detail.json
tail -c 600 $KEEP/bench_forced.json
