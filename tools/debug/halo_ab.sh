bash tools/gpu_steps.sh h24 \
 "900 python -m pytest tests -x -q -m gpu" \
 "300 bash tools/ab_bench.sh h24ab -" \
 "200 python bench.py --lone-only --lone-steps 40 --lone-gene 16,3,1,1,1,1 > gpurun_out/h24/lone_small.json 2>/dev/null"
