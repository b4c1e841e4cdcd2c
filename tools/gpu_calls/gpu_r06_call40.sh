#!/bin/bash
# round 6, call 40: with the early half-batch in C+D' (CUP2D_CD_AHEAD 1): hand-over masks again, alternating
set -u
export TMPDIR=/tmp
for SH in 13 5 13 5; do CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200; done
for SH in 13 5 13 5; do N=2048 CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200; done
