#!/bin/bash
# round 6, call 45: is the mode a property of the ALLOCATION a vector lies in?  every vector the first 128 MiB of its own allocation of
# 128 / 130 / 192 / 256 / 512 / 1024 MiB: 12 sets each, the times of the complete sets as the allocator hands them out (repair off: TRIES sets only)
set -u
export TMPDIR=/tmp
for MB in 0 130 192 256 512 1024 0 256 1024; do
  echo "== allocation size $MB MiB (0: the pool's 128 MiB)"
  CUP2D_PLACEMENT_BO_MB=$MB CUP2D_PLACEMENT_TRIES=12 CUP2D_PLACEMENT_MAX_GB=200 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: set|EDGE_SHARE" | sed 's/\[cup2d timing\] tune_placement: //' | awk '/^set/ {printf "%s ", $(NF-3)} /EDGE_SHARE/ {print ""; print $0}' | cut -c1-200
done
