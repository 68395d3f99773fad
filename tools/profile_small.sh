#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py run and its summary (tools/trace_summary.py); the raw trace stays on the box.
#   bash tools/profile_small.sh <outdir under gpurun_out> <bench.py flags...>
# (rocprofv3 of ROCm 7.2 segfaults inside librocprofiler-sdk on the N=30 000 headline run -- profiles/r02_rocprofv3_sigsegv_*.txt --
#  so the committed trace is taken at N=6 000: the per-launch shapes are identical, they depend on the batch, not on N.)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[profile_small] plain run" ; python3 "$ROOT/bench.py" "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err"; echo "rc=$?"
echo "[profile_small] rocprofv3 run"
rocprofv3 --kernel-trace --stats -d "$OUT/raw" -o trace --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench_line_under_rocprofv3.json" 2> "$OUT/bench_rocprof.err"
rc=$?
echo "rocprofv3 rc=$rc"
TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
if [ -n "$TRACE" ]; then
  python3 "$ROOT/tools/trace_summary.py" "$TRACE" "$OUT/trace_summary.json" "$OUT/kernel_stats_recomputed.csv" > "$OUT/trace_summary.txt" 2>&1
  for f in $(find "$OUT/raw" -name '*kernel_stats.csv' -o -name '*domain_stats.csv'); do cp "$f" "$OUT/"; done
fi
rm -rf "$OUT/raw"
head -45 "$OUT/trace_summary.txt"
exit $rc
