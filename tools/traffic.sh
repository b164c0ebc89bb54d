#!/bin/bash
# HBM traffic + MFMA-busy of every kernel of the bench step, one rocprofv3 --pmc pass per counter group (the guide's recipe:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; no sys/hip trace next to --pmc).  usage: tools/traffic.sh <outdir>
OUT=${1:-gpurun_out/pmc}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python bench.py --steps 3 --warmup 1 --sampler_steps 1 --no_cpu --no_profile > $OUT/g$i.log 2>&1 || exit 1
  i=$((i+1))
done
python tools/traffic_parse.py $OUT > $OUT/traffic.json && cat $OUT/traffic.json | head -50
