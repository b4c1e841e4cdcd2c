#!/bin/bash
# round 6, call 38: arenas with LARGE strides (256 MiB ... 4 GiB between the eleven vectors): is the conflict a matter of distance?
set -u
export TMPDIR=/tmp
PADS="134217728,268435456,402653184,536870912,671088640,939524096,1073741824,1476395008,2013265920,2147483648,3087007744,4160749568"
for i in 1 2; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=14 CUP2D_PLACEMENT_ARENA=$PADS timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (set|repair)|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-200
done
