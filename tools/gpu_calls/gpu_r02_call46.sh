#!/bin/bash
# quad advect kernel: results through the hand-over buffer, stores deferred behind the prefetch (new) against v1
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V1=$PWD/cup2d_amd/variants/libcup2d_hip_walk_v1.so
for ch in 0 4 8; do
  CUP2D_WALK_CHUNK=$ch timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1
  CUP2D_LIB=$V1 CUP2D_WALK_CHUNK=$ch timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1 | sed 's/^/  v1: /'
done
CUP2D_WALK_CHUNK=0 timeout 300 python tools/gpu_advect_only.py 4096 2 check 2>&1 | tail -1
CUP2D_WALK_CHUNK=0 bash tools/gpu_sq_cmd.sh walk python tools/gpu_advect_only.py 4096 2 2>&1 | grep "^k_advect\|rc=" | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advect or rectangular or step_matches or functors_vs_golden" 2>&1 | tail -2
