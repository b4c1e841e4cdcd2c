#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "CHECK|TIME|VARIANTS|Error|error" | tail -4; }
{
run "GE stride 10, DEEP 2 (default build)" X=1
run "GE stride 8" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_ge8.so
run "GE10 DEEP 0" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_0xED9_deep0.so
run "GE10 DEEP 1 (AB only)" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_0xED9_deep1.so
run "GE10 DEEP 3 (both)" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_0xED9_deep3.so
run "GE stride 10 again" X=1
} 2>&1 | tee $OUT/r02_variants6.log
