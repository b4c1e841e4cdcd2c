#!/bin/bash
# round 4, call 26: the rows launch of the hybrid operator (k_hyb_rows, 2 x 13 us of a 129 us iteration on the 63 k-block grid):
# workgroups per CU
set -u
export TMPDIR=/tmp
for w in 1 2 4; do
  echo "== CUP2D_ROWS_WG_PER_CU=$w"
  CUP2D_ROWS_WG_PER_CU=$w LFINE=9 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|sweep|scalars|operator:" | head -14
done
