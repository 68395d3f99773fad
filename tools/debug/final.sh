mkdir -p gpurun_out/r3h_final2
bash tools/gpu_steps.sh fin2 \
 "400 bash tools/profile_lone.sh r3h_lone2" \
 "400 bash tools/pmc_clock.sh r3h_clock2" \
 "300 python bench.py --clips 6000 --epochs 2 --slots 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3h_final2/bench_line_plain.json 2> gpurun_out/r3h_final2/plain.err" \
 "500 bash tools/profile_headline.sh r3h_job2 --clips 6000 --epochs 2 --slots 2" \
 "300 python tools/kernel_bench.py > gpurun_out/r3h_final2/kernel_bench.txt" \
 "600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h_final2/bench_line_driver_cmd.json 2> gpurun_out/r3h_final2/driver.err"
