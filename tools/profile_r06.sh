#!/bin/bash
# Round-6 judged artefacts (PARTS="pmc reports" in ONE call puts the counter passes and the bench line on the same lease: the MFMA-busy x clock of
# the counters and the achieved rate of the line are then of one box).  usage: [PARTS="stats pmc reports"] tools/profile_r06.sh [tag]   (writes gpurun_out/<tag>/..., copies the summaries into
# profiles/; the whole script is ~ 18 minutes of GPU time: PARTS=stats (kernel traces), PARTS=pmc (counter passes -> r06_traffic.json) and
# PARTS=reports (bench line + parity / trajectory reports) run in separate calls)
#   r06_<cfg>_train_serial_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the train steps with non-overlapping launches
#                                             (GMK_WGRAD_STREAM=0): the averages that compare with the bench line's HIP events
#   r06_bench_kernel_stats.csv                the default `python bench.py` command (all configs, samplers, overlapping streams)
#   r06_<cfg>_sampler_serial_kernel_stats.csv  the DDIM loop on ONE stream (shares are read from this one); ..._two_stream_...: the shipped form
#   r06_traffic.json                          HBM bytes per launch of every kernel of the headline config, separate --pmc passes,
#                                             stamped with the kernel-source hash (bench.py quotes it only for the same sources)
#   r06_bench.json                            the bench line of the same build;  r06_parity_report.txt  tests/parity_report.py
#   r06_trajectory_report.txt                 tests/trajectory_report.py: 150 Adam steps, CPU oracle vs HIP fp32 vs HIP 16-bit
TAG=${1:-r06}
OUT=/tmp/gmk_$TAG                 # raw traces are hundreds of MB: they stay on the box; only the summaries travel
KEEP=gpurun_out/$TAG
REPO=$(pwd)
PARTS=${PARTS:-stats pmc reports}
mkdir -p $OUT $KEEP profiles
cd /tmp && export TMPDIR=/tmp
cd $REPO
if [[ " $PARTS " == *" stats "* ]]; then
for cfg in cfg2 cfg1 cfg3; do
  GMK_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$cfg -o serial -- python bench.py --config $cfg --others 0 --sampler_steps 0 --no_profile --no_cpu --steps 10 --warmup 3 > $OUT/serial_$cfg.log 2>&1 || exit 1
  cp $(find $OUT/serial_$cfg -name "*kernel_stats.csv" | head -1) profiles/r06_${cfg}_train_serial_kernel_stats.csv
  echo "serial $cfg done"
done
# the sampler's per-kernel SHARES are read from the one-stream trace (GMK_SAMPLER_STREAMS=1: launches do not overlap, durations add up to the loop's
# time); the shipped two-stream form is traced too, under a name that says so (its durations overlap: no per-kernel shares can be read from it)
for cfg in cfg2 cfg1; do
  GMK_SAMPLER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sampler1_$cfg -o sampler -- python tools/sampler_probe.py $cfg 40 > $OUT/sampler1_$cfg.log 2>&1 || exit 1
  cp $(find $OUT/sampler1_$cfg -name "*kernel_stats.csv" | head -1) profiles/r06_${cfg}_sampler_serial_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sampler2_$cfg -o sampler -- python tools/sampler_probe.py $cfg 40 > $OUT/sampler2_$cfg.log 2>&1 || exit 1
  cp $(find $OUT/sampler2_$cfg -name "*kernel_stats.csv" | head -1) profiles/r06_${cfg}_sampler_two_stream_kernel_stats.csv
  echo "sampler $cfg done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python bench.py --no_cpu --sampler_steps 100 > $OUT/prof_bench.log 2>&1 || exit 1
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) profiles/r06_bench_kernel_stats.csv
echo "bench stats done"
fi
if [[ " $PARTS " == *" stats "* ]]; then cp profiles/r06_*kernel_stats.csv $KEEP/; fi
if [[ " $PARTS " == *" pmc "* ]]; then
for cfg in cfg2 cfg1 cfg3 cfg4; do      # every single-GPU configuration of the bench line gets its own counter passes (cfg4 since round 4)
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rocprofv3 --pmc $grp --kernel-trace --mangled-kernels --output-format csv -d $OUT/pmc_$cfg/g$i -o pmc -- python bench.py --config $cfg --others 0 --steps 3 --warmup 1 --sampler_steps 0 --no_cpu --no_profile > $OUT/pmc_${cfg}_g$i.log 2>&1 || exit 1
    i=$((i+1))
  done
  python tools/traffic_parse.py $OUT/pmc_$cfg > $OUT/pmc_$cfg/kernels.json || exit 1
  echo "pmc $cfg done"
done
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
out = {}
for cfg in ("cfg2", "cfg1", "cfg3", "cfg4"):
    out[cfg] = {"kernel_hash": bench.kernel_hash(), "kernels": json.load(open("$OUT/pmc_%s/kernels.json" % cfg)),
                "provenance": "rocprofv3 --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) of "
                "'bench.py --config %s --others 0 --steps 3 --warmup 1 --sampler_steps 0 --no_cpu --no_profile', kernel names mangled (tools/profile_r06.sh); FETCH_SIZE doubled (gfx950, 16-B/lane "
                "streaming reads: MI355X_MICROARCH.md HBM); bytes averaged over the kernel's launches of 4 train steps (no sampler launches: the sampler runs half-batches); "
                "sclk_ghz_est = GRBM_GUI_ACTIVE per XCD / dispatch time of the same pass (tools/traffic_parse.py)" % cfg}
json.dump(out, open("profiles/r06_traffic.json", "w"), indent=1)
PY
echo "traffic done"
cp profiles/r06_traffic.json $KEEP/
fi
[[ " $PARTS " == *" reports "* ]] || exit 0
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
cp $OUT/bench.json profiles/r06_bench.json                     # the compact line (what the driver parses)
cp gpurun_out/bench_detail.json profiles/r06_bench_detail.json   # the full record of the same run
python tools/wgrad_s2_ab.py 2>/dev/null > profiles/r06_wgrad_stride2_ab.txt || exit 1      # stride-2 weight gradient: four-plane slot form vs im2col, per launch
python tests/parity_report.py > profiles/r06_parity_report.txt 2>/dev/null || exit 1
python tests/trajectory_report.py 150 2>/dev/null | grep -v "^oracle step" > profiles/r06_trajectory_report.txt || exit 1
cp profiles/r06_bench.json profiles/r06_bench_detail.json profiles/r06_parity_report.txt profiles/r06_trajectory_report.txt profiles/r06_wgrad_stride2_ab.txt $KEEP/
cp $OUT/*.log $OUT/bench.err $KEEP/ 2>/dev/null
tail -c 300 $OUT/bench.json
