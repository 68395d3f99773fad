#!/bin/bash
# quick isolated-kernel A/B on the FLOP-heaviest layer shapes (fwd / dgrad / wgrad TFLOP/s)
for f in "64,64,5" "128,128,5" "256,256,5" "512,512,5" "64,128,5" "128,256,5" "64,64,3" "32,32,5"; do python tools/kernel_bench.py "$f" 2>/dev/null | grep -v "^B,H"; done
