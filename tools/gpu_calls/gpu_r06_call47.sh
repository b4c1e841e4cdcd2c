#!/bin/bash
# round 6, call 47: C+D': the ring list read in one LDS round trip per half (against eight in a row per tile), alternating builds
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in $V/libcup2d_hip_0xED9_old.so "" $V/libcup2d_hip_0xED9_old.so "" $V/libcup2d_hip_0xED9_old.so ""; do
  echo "lib ${L:-new}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in $V/libcup2d_hip_0xED9_old.so "" $V/libcup2d_hip_0xED9_old.so ""; do
  echo "lib ${L:-new}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
timeout 900 python3 -m pytest tests/test_solver_variants_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
