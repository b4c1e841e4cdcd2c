#!/bin/bash
# round 6, call 42: where a tile's time goes in the two launches of an iteration (k_edge built with -DEDGE_PHASES: shader-clock cycles per phase and tile)
set -u
export TMPDIR=/tmp
CUP2D_LIB=cup2d_amd/variants/libcup2d_hip_0xED9_ph.so timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "EPHASES|EDGE_SHARE" | sort | uniq -c | sort -rn | head -40 | cut -c1-400
