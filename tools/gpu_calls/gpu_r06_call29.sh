#!/bin/bash
# round 6, call 29: which of the eleven vectors decide the mode of a set (hybrids of the fastest and the slowest set)
set -u
export TMPDIR=/tmp
for i in 1 2; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=12 CUP2D_PLACEMENT_HYBRID=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|hybrid)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-260
done
