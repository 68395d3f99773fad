bash tools/gpu_steps.sh h23 \
 "200 python tools/kernel_bench.py '32,32,3'" \
 "200 CMOOP_HALO_K3_32=1 python tools/kernel_bench.py '32,32,3'" \
 "400 bash tools/ab_bench.sh h23ab - CMOOP_HALO_K3_32=1"
