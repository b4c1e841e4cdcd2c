#!/bin/bash
# round 6, call 25: arenas (one allocation carved at 2^27 + pad) on a box that HAS fast placements (round 5's arena experiment ran on a box without)
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=20 CUP2D_PLACEMENT_ARENA="0,4096,65536,1048576,2097152,2162688,6291456,35651584,69206016,12288,786432,3145728" timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: set|EDGE_SHARE" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-120
done
