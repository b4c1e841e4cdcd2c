#!/bin/bash
# round 6, call 22: sibling hand-over in C+D' (bit 3 of CUP2D_EDGE_SHARE) against the default, 4096^2 and 2048^2, alternating
set -u
export TMPDIR=/tmp
for SH in 5 13 5 13 15 5; do CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1; done
for SH in 5 13 5 13; do N=2048 CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1; done
