bash tools/gpu_steps.sh ev_a \
 "400 bash tools/profile_lone.sh r3h_lone" \
 "400 bash tools/pmc_clock.sh r3h_clock" \
 "600 bash tools/pmc_round3.sh r3h_pmc"
