#!/bin/bash
# round 6, call 35: the placement search on the hybrid operator (the three sweeps of the five-launch organisation as the probe)
set -u
export TMPDIR=/tmp
for H in 0 1 0 1; do
  CUP2D_PLACEMENT_HYB=$H CUP2D_HOST_TIMING=1 FORM=auto LFINE=9 NOTIMING=1 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "tune_placement: (set|repair|search)|AMR step|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-300
done
