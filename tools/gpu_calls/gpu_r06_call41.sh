#!/bin/bash
# round 6, call 41: C+D': the second half of the tile's own batches requested at the top of the tile (in front of the ring's staging)
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in "" $V/libcup2d_hip_0xED9_top.so "" $V/libcup2d_hip_0xED9_top.so; do
  echo "lib ${L:-default}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in "" $V/libcup2d_hip_0xED9_top.so "" $V/libcup2d_hip_0xED9_top.so; do
  echo "lib ${L:-default}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
