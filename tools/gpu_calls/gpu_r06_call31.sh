#!/bin/bash
# round 6, call 31: can a slow set be repaired with vectors of other slow sets only?
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_PLACEMENT_TRIES=12 CUP2D_PLACEMENT_REPAIR_TEST=1 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|repair|search)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-300
done
