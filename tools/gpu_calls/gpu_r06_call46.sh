#!/bin/bash
# round 6, call 46: the branch-free epilogue of k_edge MODE 1 / 2 / 3 against the one written with `if` (alternating builds), the forms test, the phases
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in $V/libcup2d_hip_0xED9_br.so "" $V/libcup2d_hip_0xED9_br.so "" $V/libcup2d_hip_0xED9_br.so ""; do
  echo "lib ${L:-new (branch-free)}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in $V/libcup2d_hip_0xED9_br.so "" $V/libcup2d_hip_0xED9_br.so ""; do
  echo "lib ${L:-new (branch-free)}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
timeout 900 python3 -m pytest tests/test_solver_variants_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
CUP2D_LIB=$V/libcup2d_hip_0xED9_ph.so timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "EPHASES mode 3 wg 0 wave 0|EPHASES mode 2 wg 0 wave 0" | tail -4 | cut -c1-330
