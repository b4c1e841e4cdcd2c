#!/bin/bash
# quad advect kernel: LDS reads issued ahead (new) against the first version (variants/..._v0), SQ counters of the new one
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V0=$PWD/cup2d_amd/variants/libcup2d_hip_walk_v0.so
for ch in 0 4; do
  CUP2D_WALK_CHUNK=$ch timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1
  CUP2D_LIB=$V0 CUP2D_WALK_CHUNK=$ch timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1 | sed 's/^/  v0: /'
done
CUP2D_WALK_CHUNK=0 timeout 300 python tools/gpu_advect_only.py 4096 2 check 2>&1 | tail -1
CUP2D_WALK_CHUNK=0 bash tools/gpu_sq_cmd.sh walk python tools/gpu_advect_only.py 4096 2 2>&1 | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advect or rectangular or step_matches" 2>&1 | tail -2
