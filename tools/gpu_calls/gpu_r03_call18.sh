#!/bin/bash
# cache policy of the streams under the two-launch organisation with the zigzag: t store / rhat load allocate or bypass
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip
E=SKIP_REL4=1
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py default@$E ${V}_0xED1.so@$E ${V}_0x6D9.so@$E ${V}_0x6D1.so@$E default@$E,CUP2D_ALLOC_SKEW=2101504 2>&1 | tee $OUT/r03_eab_policy.txt
