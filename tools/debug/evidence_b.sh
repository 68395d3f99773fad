mkdir -p gpurun_out/r3h_job
bash tools/gpu_steps.sh ev_b \
 "300 python bench.py --clips 6000 --epochs 2 --slots 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3h_job/bench_line_plain.json 2> gpurun_out/r3h_job/plain.err" \
 "500 bash tools/profile_headline.sh r3h_job --clips 6000 --epochs 2 --slots 2" \
 "600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h_job/bench_line_driver_cmd.json 2> gpurun_out/r3h_job/driver.err"
