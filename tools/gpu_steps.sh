#!/bin/bash
# Run GPU steps one after another on the box: a step that FAILS (test failures) does not stop the next one, a step that was
# KILLED (timeout / signal: rc >= 124) does -- nothing further touches the GPU after a hang.
#   bash tools/gpu_steps.sh <outdir under gpurun_out> "<seconds> <command>" ...
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
i=0
for step in "$@"; do
  i=$((i+1))
  secs="${step%% *}"; cmd="${step#* }"
  echo "=== step $i (limit ${secs}s): $cmd" | tee -a "$OUT/steps.log"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/step$i.log" 2>&1
  rc=$?
  echo "=== step $i rc=$rc in $(( $(date +%s) - t0 ))s" | tee -a "$OUT/steps.log"
  tail -5 "$OUT/step$i.log"
  if [ $rc -ge 124 ]; then echo "step $i was killed: stopping" | tee -a "$OUT/steps.log"; exit $rc; fi
done
exit 0
