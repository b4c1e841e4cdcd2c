#!/bin/bash
# round 4, call 21: the profile set r04 once more without the N-rank-path leg in the profiled command (its split launches of
# the shared kernels had entered the per-launch averages of call 9)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_profile_set.sh r04 2>&1 | grep -v "^k_\|^__amd" | tail -12
python3 tools/prof_summary.py r04 2>&1 | tail -3
cp profiles/r04_kernel_stats.txt profiles/r04_pmc_traffic.json $OUT/
head -14 profiles/r04_kernel_stats.txt
