#!/bin/bash
# rocprofv3 --kernel-trace --stats of the HEADLINE bench workload (pop 40, N=30 000, E=2) and its bench line, same command.
# The raw kernel trace (millions of dispatches) stays on the box; tools/trace_summary.py reduces it.
#   bash tools/profile_headline.sh <outdir under gpurun_out> [extra bench.py flags]
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-prof_headline}"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/raw" -o trace --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --budget-s 1100 "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rc=$?
echo "rocprofv3 rc=$rc" | tee -a "$OUT/bench.err"
TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
ls -la "$OUT/raw"/*/ 2>/dev/null | tail -20 >> "$OUT/bench.err"
if [ -n "$TRACE" ]; then
  python3 "$ROOT/tools/trace_summary.py" "$TRACE" "$OUT/trace_summary.json" "$OUT/kernel_stats_recomputed.csv" > "$OUT/trace_summary.txt" 2>&1
  for f in $(find "$OUT/raw" -name '*kernel_stats.csv' -o -name '*domain_stats.csv'); do cp "$f" "$OUT/"; done
fi
rm -rf "$OUT/raw"
head -60 "$OUT/trace_summary.txt"
exit $rc
