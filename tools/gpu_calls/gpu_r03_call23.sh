#!/bin/bash
# run-to-run spread: one hipMalloc per vector against vectors carved out of one slab (CUP2D_POOL_SLAB_MB), with and without skew
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
E=SKIP_REL4=1
REPS=4 timeout 800 python3 tools/gpu_lib_variants.py default@$E default@$E,CUP2D_POOL_SLAB_MB=8192 default@$E,CUP2D_POOL_SLAB_MB=8192,CUP2D_ALLOC_SKEW=4352 default@$E,CUP2D_ALLOC_SKEW=2101504 2>&1 | tee $OUT/r03_slab.txt
