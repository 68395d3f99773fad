set -e
mkdir -p gpurun_out/r2u
for lv in 1 2; do
  CMOOP_DEBUG_SKIP_ELEM=$lv python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2u/skip_$lv.json 2> gpurun_out/r2u/skip_$lv.err || { tail -5 gpurun_out/r2u/skip_$lv.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r2u/skip_$lv.json').read().strip().splitlines()[-1])
print('skip level $lv evals/h', d['value'], 'ms/step', d['ms_per_step'], 'aggregate', d['roofline']['aggregate_timed_region']['achieved'], flush=True)"
done
