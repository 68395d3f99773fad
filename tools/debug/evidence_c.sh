mkdir -p gpurun_out/r3h_c
bash tools/gpu_steps.sh ev_c \
 "300 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k 'merged_dense or optimiser_launch or fixed_seed or one_step'" \
 "400 python tools/scaling_emulate.py --plan queue --worlds 1,8 > gpurun_out/r3h_c/scaling.json 2> gpurun_out/r3h_c/scaling.err" \
 "700 python bench.py --protocol reference --steps 1 --warmup 1 --no-cpu-baseline --budget-s 650 > gpurun_out/r3h_c/reference_protocol.json 2> gpurun_out/r3h_c/reference.err"
