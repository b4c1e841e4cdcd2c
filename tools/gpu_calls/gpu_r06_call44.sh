#!/bin/bash
# round 6, call 44: C+D': r and nu' of the next tile's SECOND half requested early too (CUP2D_CD_AHEAD 3: 253 registers, no spill) against the first half only
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in "" $V/libcup2d_hip_0xED9_a3.so "" $V/libcup2d_hip_0xED9_a3.so "" $V/libcup2d_hip_0xED9_a3.so; do
  echo "lib ${L:-default (AHEAD 1)}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in "" $V/libcup2d_hip_0xED9_a3.so "" $V/libcup2d_hip_0xED9_a3.so; do
  echo "lib ${L:-default (AHEAD 1)}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
