#!/bin/bash
# The evidence pair for bench.py's roofline: the same command once plain (hipExtLaunchKernelGGL events) and once under
# rocprofv3 --kernel-trace --stats (the tool library crashes on ext launches and with > 4 launching threads in ROCm 7.2,
# so bench.py falls back to event pairs there and the command uses --slots 4).  Writes into gpurun_out/pair/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--clips 3000 --epochs 2 --pop 40 --slots 4 --no-cpu-baseline"
mkdir -p gpurun_out/pair
python bench.py $ARGS 2>/dev/null | tail -1 > gpurun_out/pair/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pair/rocprof -o p -- python3 bench.py $ARGS 2>/dev/null | tail -1 > gpurun_out/pair/bench_line_under_rocprofv3.json
find gpurun_out/pair/rocprof -name "*kernel_trace.csv" -delete     # hundreds of MB; the stats summaries are what is kept
find gpurun_out/pair/rocprof -name "*.csv" | head
