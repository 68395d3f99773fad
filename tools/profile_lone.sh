#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --lone-only`: the heaviest candidate of the bench's population training ALONE at
# headline N (one launching thread: no tool crash, no CU sharing) -- the trace `roofline.frac` of the bench line must agree with.
#   bash tools/profile_lone.sh <outdir under gpurun_out> [extra bench.py flags]
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-prof_lone}"; shift
mkdir -p "$OUT"
python3 "$ROOT/bench.py" --lone-only --lone-steps 40 "$@" > "$OUT/bench_line_unprofiled.json" 2> "$OUT/bench_unprofiled.err" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/raw" -o trace --output-format csv -- python3 "$ROOT/bench.py" --lone-only --lone-steps 40 "$@" > "$OUT/bench_line_under_rocprofv3.json" 2> "$OUT/bench.err"
rc=$?
echo "rocprofv3 rc=$rc" | tee -a "$OUT/bench.err"
TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
if [ -n "$TRACE" ]; then
  python3 "$ROOT/tools/trace_summary.py" "$TRACE" "$OUT/trace_summary.json" "$OUT/kernel_stats_recomputed.csv" > "$OUT/trace_summary.txt" 2>&1
  for f in $(find "$OUT/raw" -name '*kernel_stats.csv' -o -name '*domain_stats.csv'); do cp "$f" "$OUT/"; done
  python3 "$ROOT/tools/lone_profile_merge.py" "$OUT/bench_line_under_rocprofv3.json" "$OUT/trace_summary.json" "$OUT/bench_line_unprofiled.json" > "$OUT/per_instantiation.txt" 2>&1
fi
rm -rf "$OUT/raw"
head -40 "$OUT/per_instantiation.txt"
exit $rc
