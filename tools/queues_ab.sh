#!/bin/bash
# whole-job throughput against the number of HIP hardware queues (GPU_MAX_HW_QUEUES; default 4): one bench line per setting
# usage: tools/queues_ab.sh OUTDIR [bench args...]
set -e
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for q in default 2 8 1; do
    if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/queues_$q.json" 2> "$OUT/queues_$q.err"
    python3 - "$OUT/queues_$q.json" "$q" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("queues", sys.argv[2], "evals/h", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "avg_launch_ms", r.get("avg_launch_ms"),
      "aggregate", r.get("aggregate_timed_region"), flush=True)
PY
done
