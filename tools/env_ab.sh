#!/bin/bash
# Same-box A/B of one environment switch over whole bench runs, interleaved:  tools/env_ab.sh VAR "" 1 [cfg] [reps] [sampler_steps]
#   -> gpurun_out/env_ab/<VAR>.txt   (an empty value means "unset")
VAR=$1; A=$2; B=$3; CFG=${4:-cfg2}; REPS=${5:-2}; SS=${6:-0}
KEEP=gpurun_out/env_ab; mkdir -p $KEEP
for rep in $(seq $REPS); do for v in "$A" "$B"; do
  if [ -z "$v" ]; then unset $VAR; else export $VAR="$v"; fi
  python bench.py --config $CFG --others 0 --sampler_steps $SS --no_cpu --no_profile --steps 30 > /tmp/env_ab.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('/tmp/env_ab.json')); s=d.get('sampler') or {}
print('$CFG $VAR=[$v] rep $rep:', d['value'], 'img/s', d['ms_per_step'], 'ms', s.get('steps_per_sec'), 'DDIM steps/s', flush=True)" | tee -a $KEEP/$VAR.txt
done; done
