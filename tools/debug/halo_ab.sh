bash tools/gpu_steps.sh h6 \
 "500 python -m pytest tests/test_gpu_production_shapes.py -x -q -m gpu" \
 "500 bash tools/ab_bench.sh h6ab CMOOP_HALO=0 -"
