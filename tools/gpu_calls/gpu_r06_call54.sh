#!/bin/bash
# round 6, call 54: pair exchanges (two written streams from one donor set) on the SLOWEST set of a batch, single exchanges off (mechanics + reach)
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  CUP2D_PLACEMENT_TEST_PAIRS=1 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: (repair|search)|EDGE_SHARE" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-400
done
