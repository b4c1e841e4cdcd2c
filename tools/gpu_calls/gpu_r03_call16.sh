#!/bin/bash
# run-to-run spread of the same build (EAB 238 / 265 us): does it follow where the 2^27-byte vectors start? (CUP2D_ALLOC_SKEW)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
E=CUP2D_FUSED_FORM=eab,SKIP_REL4=1
REPS=3 timeout 800 python3 tools/gpu_lib_variants.py default@$E default@$E,CUP2D_ALLOC_SKEW=4352 default@$E,CUP2D_ALLOC_SKEW=266496 default@$E,CUP2D_ALLOC_SKEW=2101504 2>&1 | tee $OUT/r03_eab_skew.txt
