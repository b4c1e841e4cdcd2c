#!/bin/bash
# wave priority during the jobs on the matrix cores; host look every 8 iterations instead of 4
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip_0xED9
E=SKIP_REL4=1
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py default@$E ${V}_prio1.so@$E ${V}_prio3.so@$E default@$E,CUP2D_SOLVE_GROUP=8 2>&1 | tee $OUT/r03_prio.txt
