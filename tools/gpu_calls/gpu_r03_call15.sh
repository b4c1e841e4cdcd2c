#!/bin/bash
# same box: the ascending-only tile loop (variant library) against the loop with a direction, sharing masks
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip_0xED9_oldloop.so
E=CUP2D_FUSED_FORM=eab,SKIP_REL4=1
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py $V@$E,CUP2D_EAB_ZIGZAG=0 default@$E,CUP2D_EAB_ZIGZAG=0 default@$E $V@$E,CUP2D_EAB_ZIGZAG=0,CUP2D_EDGE_SHARE=15 default@$E,CUP2D_EDGE_SHARE=15 default 2>&1 | tee $OUT/r03_eab_ab.txt
