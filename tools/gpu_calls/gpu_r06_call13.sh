#!/bin/bash
# round 6, call 13: the hand-over mask of the edge-form sweeps on whatever kind of box this is (boxes without a fast placement
# run every kernel that computes AND streams 8-10 % slower while the knocked-out halves run at the same rate: power)
set -u
export TMPDIR=/tmp
for SH in 5 15 13 7 5; do CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1; done
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" | head -6
