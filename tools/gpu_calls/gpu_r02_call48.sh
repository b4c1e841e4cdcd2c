#!/bin/bash
# quad advect kernel: knock-out timings (k1 = memory skeleton without the walks, k2 = walks without loads/stores in the loop)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
timeout 300 python tools/gpu_advect_stages.py 4096 20 2>&1 | tail -1
CUP2D_LIB=$V/libcup2d_hip_walk_k1.so timeout 300 python tools/gpu_advect_stages.py 4096 20 2>&1 | tail -1
CUP2D_LIB=$V/libcup2d_hip_walk_k2.so timeout 300 python tools/gpu_advect_stages.py 4096 20 2>&1 | tail -1
CUP2D_ADVECT_WALK=0 timeout 300 python tools/gpu_advect_stages.py 4096 20 2>&1 | tail -1 | sed 's/^/per-block kernel: /'
