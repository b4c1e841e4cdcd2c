#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "CHECK|TIME|VARIANTS|Error|error" | tail -4; }
{
run "new (nt stores)" X=1
run "new + sibling barrier" CUP2D_FUSED_DBG=8
run "sc1 stores" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_0x40ED9_sc1.so
run "sc1 stores + sibling barrier" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_0x40ED9_sc1.so CUP2D_FUSED_DBG=8
run "base (8-byte accesses)" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_base.so
} 2>&1 | tee $OUT/r02_variants4.log
echo "total $(( $(date +%s) - t0 )) s"
