#!/bin/bash
# round 6, call 34: how far does the exchange search go with more passes and donors (experiment)
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_PLACEMENT_PASSES=4 CUP2D_PLACEMENT_DONORS=7 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-400
done
