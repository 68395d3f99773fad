#!/bin/bash
# One small candidate alone on the GPU: steps/s, and under rocprofv3 the sum of its kernel durations and launch count
# (stream busy fraction = sum of kernel durations / wall).  usage: bash tools/lone_busy.sh <outdir under gpurun_out> <gene> [<gene>...]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for G in "$@"; do
  echo "== gene $G"
  python3 "$ROOT/tools/lone_candidate.py" "$G" --repeat 3 2>/dev/null | tee -a "$OUT/lone.jsonl"
  rocprofv3 --kernel-trace --stats -d "$OUT/raw" -o t --output-format csv -- python3 "$ROOT/tools/lone_candidate.py" "$G" --repeat 1 > "$OUT/lone_rocprof_$G.json" 2> "$OUT/lone_rocprof_$G.err"
  TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
  python3 "$ROOT/tools/trace_summary.py" "$TRACE" "$OUT/lone_trace_$G.json" > "$OUT/lone_trace_$G.txt" 2>&1
  head -30 "$OUT/lone_trace_$G.txt"
  cat "$OUT/lone_rocprof_$G.json"
  rm -rf "$OUT/raw"
done
