#!/bin/bash
# round 6, call 30: the placement search with the repair pass (default settings), six processes; then with the first batch cut to 3
# sets (a box without a fast set among the first: does the repair make one?)
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-260
done
for i in 1 2 3; do
  CUP2D_PLACEMENT_TRIES=2 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-260
done
