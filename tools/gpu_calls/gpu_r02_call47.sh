#!/bin/bash
# quad advect kernel: rotating wave priority against none, persistent grid
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for pr in 0 1; do
  CUP2D_WALK_PRIO=$pr timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1 | sed "s/^/prio=$pr /"
  CUP2D_WALK_PRIO=$pr CUP2D_WALK_CHUNK=8 timeout 300 python tools/gpu_advect_only.py 4096 5 2>&1 | tail -1 | sed "s/^/prio=$pr /"
done
CUP2D_WALK_PRIO=1 bash tools/gpu_sq_cmd.sh walk python tools/gpu_advect_only.py 4096 2 check 2>&1 | grep "^k_advect\|rc=" | cut -c1-400
