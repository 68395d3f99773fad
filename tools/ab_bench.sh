#!/bin/bash
# whole-job A/B on one box: the same bench (E=2, one timed generation, no CPU leg) and the lone-candidate leg under several
# environments.  usage: bash tools/ab_bench.sh <outdir under gpurun_out> "NAME=VAL ..." "NAME=VAL ..." ...   ("-" = no variables)
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
i=0
for envs in "$@"; do
  i=$((i+1))
  [ "$envs" = "-" ] && envs=""
  BENCH_ARGS=$(echo "$envs" | tr " " "\n" | grep "^ARGS=" | sed "s/^ARGS=//" | tr "," " "); envs=$(echo "$envs" | tr " " "\n" | grep -v "^ARGS=" | tr "\n" " ")
  line=$(env $envs timeout -k 10 400 python bench.py --epochs 2 --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2> "$OUT/run$i.err")
  rc=$?
  echo "$line" > "$OUT/run$i.json"
  python - "$envs" "$OUT/run$i.json" "$rc" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
envs, path, rc = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][0])
    r = d["roofline"]; lone = r["lone_candidate"]
    iso = " ".join(f"{e['fwd_tflops']:.0f}/{e['dgrad_tflops']:.0f}/{e['wgrad_tflops']:.0f}" for e in r.get("isolated_single_stream", []))
    print(f"[{envs or 'default'}] evals/h {d['value']:.1f}  job TF {d['whole_job_tflops']:.2f}  lone ms/step {lone['ms_per_step_incl_sync']:.3f} ({lone['step_tflops']:.1f} TF)  dom {r['achieved']:.1f} TF  iso {iso}  acc {d['mean_val_accuracy']}")
except Exception as e:
    print(f"[{envs}] rc={rc} no line: {e}")
PY
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
done
