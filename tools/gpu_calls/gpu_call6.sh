#!/bin/bash
# cache-policy variants of the fused solver (tools/build_policy_variants.sh): step time + per-sweep times at 4096^2
set -u
echo "== default"; SKIP_CHECK=1 VARIANTS=fused1 timeout 120 python tools/gpu_variants.py 2>&1 | grep -E "^TIME|rror"
for f in cup2d_amd/variants/libcup2d_hip_*.so; do
  echo "== $f"; CUP2D_LIB=$PWD/$f SKIP_CHECK=1 VARIANTS=fused1 timeout 120 python tools/gpu_variants.py 2>&1 | grep -E "^TIME|rror"
done
