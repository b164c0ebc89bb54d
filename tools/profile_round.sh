#!/bin/bash
# Refresh the judged artefacts of a round: bench line, rocprofv3 kernel-trace summaries (default command and the serial
# train-only variant whose averages are comparable with the bench line's HIP events), PMC traffic.  usage: tools/profile_round.sh <tag>
TAG=${1:-r01x}
OUT=gpurun_out/$TAG
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python bench.py --no_cpu > $OUT/prof_bench.log 2>&1 || exit 1
GMK_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o serial -- python bench.py --sampler_steps 0 --no_profile --no_cpu > $OUT/prof_serial.log 2>&1 || exit 1
bash tools/traffic.sh $OUT/pmc > $OUT/traffic.log 2>&1 || exit 1
cp $OUT/pmc/traffic.json profiles/traffic.json            # the bench line quotes the traffic of THIS code
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
find $OUT -name "*kernel_stats.csv" | head
tail -1 $OUT/bench.json | cut -c1-400
