#!/bin/bash
# round 6, call 26: physically contiguous vectors (hipDeviceMallocContiguous) as placement candidates: is the mode a matter of
# how fragmented the physical pages of a set are (TLB reach with eleven streams in eight windows)?
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=16 CUP2D_PLACEMENT_CONTIG=6 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: set|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-120
done
