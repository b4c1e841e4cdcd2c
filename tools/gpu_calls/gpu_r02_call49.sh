#!/bin/bash
# quad advect kernel: scalar-base addressing (new) against v2; priority modes 0 none, 1 rotating, 2 memory phase first
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
for pr in 1 0 2; do CUP2D_WALK_PRIO=$pr timeout 300 python tools/gpu_advect_only.py 4096 10 2>&1 | tail -1 | sed "s/^/new prio=$pr /"; done
CUP2D_LIB=$V/libcup2d_hip_walk_v2.so timeout 300 python tools/gpu_advect_only.py 4096 10 2>&1 | tail -1 | sed "s/^/v2  prio=1 /"
CUP2D_WALK_PRIO=1 timeout 300 python tools/gpu_advect_only.py 4096 10 check 2>&1 | tail -2 | sed "s/^/new prio=1 /"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advect or rectangular or step_matches or functors_vs_golden" 2>&1 | tail -2
