#!/bin/bash
# tools/debug/skip_elem_patch.py must have been applied and built: one bench line per skipped kernel family
set -e
mkdir -p gpurun_out/skip
for mask in "$@"; do
  CMOOP_DEBUG_SKIP_ELEM=$mask python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/skip/skip_$mask.json 2> gpurun_out/skip/skip_$mask.err || { tail -5 gpurun_out/skip/skip_$mask.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/skip/skip_$mask.json').read().strip().splitlines()[-1])
print('skip mask $mask evals/h', d['value'], 'ms/step', d['ms_per_step'], 'aggregate', d['roofline']['aggregate_timed_region']['achieved'], flush=True)"
done
