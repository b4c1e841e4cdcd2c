#!/bin/bash
# round 6, call 27: ONE physically contiguous arena carved at 2^27 + pad: the pad is then the PHYSICAL spacing of the eleven vectors
set -u
export TMPDIR=/tmp
PADS="0,4096,8192,16384,32768,65536,131072,262144,524288,1048576,2097152,4194304,8388608,16777216,33554432,12288,20480,36864,69632,135168,266240,528384,1052672,2101248,6291456,10485760"
for i in 1 2; do
  CUP2D_HOST_TIMING=1 CUP2D_PLACEMENT_TRIES=28 CUP2D_PLACEMENT_ARENA_CONTIG=1 CUP2D_PLACEMENT_MAX_GB=80 CUP2D_PLACEMENT_ARENA=$PADS timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: set|EDGE_SHARE|rror" | sed 's/\[cup2d timing\] tune_placement: //' | cut -c1-120
done
