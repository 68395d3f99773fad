#!/bin/bash
# Single-stream busy fraction: wall time of candidates evaluated one at a time (slots 1) vs the sum of their kernel
# durations from rocprofv3 --kernel-trace --stats of the same command (launch gaps = wall - kernel time).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--clips 6000 --epochs 2 --pop 6 --slots 1 --no-cpu-baseline --profile-every 0"
python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wall without profiler (s):', d['ms_per_step']/1e3, 'TF/s', d['whole_job_tflops'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o p1 -- python3 bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wall under rocprofv3 (s):', d['ms_per_step']/1e3)"
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("sum of kernel durations (s):", tot / 1e9, "launches:", calls)
PY
rm -rf gpurun_out/prof1
