#!/bin/bash
# round 6, call 53: the search's log on the box of call 52 (fast sets rare there)
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|repair|search)|EDGE_SHARE" | sed 's/\[cup2d timing\] tune_placement: //; s/ (separate, pad -1, first vector at 0x[0-9a-f]*)//' | awk '/^set/ {printf "%s ", $3; next} {print ""; print}' | cut -c1-330
done
