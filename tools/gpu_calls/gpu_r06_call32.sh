#!/bin/bash
# round 6, call 32: the placement search with the repair as it ships (default settings): six processes at 4096^2, two at 2048^2
set -u
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-300
done
for i in 1 2; do
  N=2048 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-300
done
