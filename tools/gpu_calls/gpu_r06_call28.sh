#!/bin/bash
# round 6, call 28: pseudo-random offsets of the eleven vectors inside ONE arena (contiguous, then plain)
set -u
export TMPDIR=/tmp
for CONTIG in 1 0; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=8 CUP2D_PLACEMENT_SCATTER=40 CUP2D_PLACEMENT_ARENA_CONTIG=$CONTIG timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: s|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-200
done
