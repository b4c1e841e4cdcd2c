#!/bin/bash
# round 6, call 57: k_edge: the fill of the edge columns of P_inv into LDS behind the first tile's requests (against in front of everything), alternating builds
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in $V/libcup2d_hip_0xED9_ff.so "" $V/libcup2d_hip_0xED9_ff.so "" $V/libcup2d_hip_0xED9_ff.so ""; do
  echo "lib ${L:-new (fill behind)}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in $V/libcup2d_hip_0xED9_ff.so "" $V/libcup2d_hip_0xED9_ff.so "" $V/libcup2d_hip_0xED9_ff.so ""; do
  echo "lib ${L:-new (fill behind)}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
timeout 900 python3 -m pytest tests/test_solver_variants_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
