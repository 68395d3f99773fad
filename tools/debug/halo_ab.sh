bash tools/gpu_steps.sh h11 \
 "500 bash tools/ab_bench.sh h11ab - CMOOP_HALO_BAL=1" \
 "900 python -m pytest tests -x -q -m gpu"
