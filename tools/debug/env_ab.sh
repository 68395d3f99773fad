#!/bin/bash
# bench lines under different environment settings: tools/debug/env_ab.sh OUTDIR "VAR=val VAR2=val" "..." (use "-" for none)
set -e
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/ab_$i.json" 2> "$OUT/ab_$i.err" || { tail -5 "$OUT/ab_$i.err"; exit 1; }
  python3 -c "
import json
d=json.loads(open('$OUT/ab_$i.json').read().strip().splitlines()[-1])
print('[$setting] evals/h', d['value'], 'ms/step', d['ms_per_step'], 'aggregate', d['roofline']['aggregate_timed_region']['achieved'], flush=True)"
done
