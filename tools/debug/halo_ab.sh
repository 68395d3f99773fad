bash tools/gpu_steps.sh h16 \
 "200 CMOOP_HALO_DBG=2 python tools/kernel_bench.py '64,64,'" \
 "200 CMOOP_HALO_DBG=3 python tools/kernel_bench.py '64,64,'"
