bash tools/gpu_steps.sh h13 \
 "200 python tools/kernel_bench.py '64,64,|128,128,|64,128,'" \
 "200 CMOOP_HALO_WGRAD_SWAP=1 python tools/kernel_bench.py '64,64,|128,128,|64,128,'" \
 "300 CMOOP_HALO_WGRAD_SWAP=1 python -m pytest tests/test_gpu_production_shapes.py -x -q -m gpu -k through"
