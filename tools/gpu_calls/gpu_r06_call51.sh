#!/bin/bash
# round 6, call 51: how often is a complete set fast when every vector lies in an allocation of 128 / 384 / 512 MiB of its own?  24 sets each, three processes each
set -u
export TMPDIR=/tmp
for MB in 0 512 384 0 512 384 0 512; do
  echo "== allocation size $MB MiB (0: the pool's 128 MiB)"
  CUP2D_PLACEMENT_BO_MB=$MB CUP2D_PLACEMENT_TRIES=24 CUP2D_PLACEMENT_MAX_GB=200 CUP2D_HOST_TIMING=1 timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "tune_placement: set" | sed 's/.*): //; s/ us per iteration.*//' | tr '\n' ' '; echo
done
