#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -m gpu -q -k "jacobi or residual or decomposed" > $OUT/r02_pytest12.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r02_pytest12.log
echo "== new (16-byte)"; timeout 200 python tools/gpu_smoother.py 2>&1 | tail -3
echo "== base (8-byte)"; CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_ge8.so timeout 200 python tools/gpu_smoother.py 2>&1 | tail -3
echo "== new again"; timeout 200 python tools/gpu_smoother.py 2>&1 | tail -3
