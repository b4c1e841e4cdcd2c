#!/bin/bash
# fused sweeps: B fragments of the MFMA jobs double-buffered (default: CD only; db3: AB too) against the previous library
set -u
export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
for lib in default fused_v0; do
  if [ $lib = default ]; then unset CUP2D_LIB; else export CUP2D_LIB=$V/libcup2d_hip_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/gpu_ring_check.py time 2>&1 | grep "4096\|sweep\|FAIL\|ALL"
done
