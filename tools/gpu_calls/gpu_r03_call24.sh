#!/bin/bash
# vectors carved out of one slab are reproducibly in the slow mode (call23): which start offsets, if any, leave it?
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
E=SKIP_REL4=1,CUP2D_POOL_SLAB_MB=16384
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py default@SKIP_REL4=1 default@$E default@$E,CUP2D_ALLOC_SKEW=34048 default@$E,CUP2D_ALLOC_SKEW=266496 default@$E,CUP2D_ALLOC_SKEW=1052928 default@$E,CUP2D_ALLOC_SKEW=8917248 2>&1 | tee $OUT/r03_slab2.txt
