bash tools/gpu_steps.sh lone2 \
 "300 bash tools/profile_lone.sh r3h_lone_bn --lone-gene 64,5,1,3,4,1" \
 "300 bash tools/profile_lone.sh r3h_lone_small --lone-gene 16,3,1,1,1,1"
