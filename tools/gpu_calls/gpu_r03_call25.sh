#!/bin/bash
# three export buffers per wave (a wave may run two rounds ahead of a sibling) against two; hand-over in C+D' too
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip_0xED9_nxb3.so
E=SKIP_REL4=1
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py default@$E $V@$E default@$E,CUP2D_EDGE_SHARE=15 $V@$E,CUP2D_EDGE_SHARE=15 2>&1 | tee $OUT/r03_nxb.txt
CUP2D_LIB=$V VARIANTS=eab,eab-allshare timeout 300 python3 tools/gpu_edge_check.py check 2>&1 | cut -c1-400
