mkdir -p gpurun_out/r3h_final
bash tools/gpu_steps.sh fin \
 "900 python -m pytest tests -x -q -m gpu" \
 "120 python -c 'import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")'" \
 "600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3h_final/bench_line_driver_cmd.json 2> gpurun_out/r3h_final/driver.err"
