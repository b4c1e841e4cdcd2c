#!/bin/bash
# round 6, call 23: is the placement mode a property of single allocations?  16 sets, every vector of every set read alone
set -u
export TMPDIR=/tmp
for i in 1 2; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_SOLO=1 CUP2D_PLACEMENT_TRIES=16 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement|EDGE_SHARE" | cut -c1-330
done
