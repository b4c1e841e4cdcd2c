#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_distributed.py tests/test_baseline_sizes_gpu.py tests/test_comm.py -m gpu -q -k "not functor and not whole_step" > $OUT/r02_pytest14.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r02_pytest14.log
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 SKIP_CHECK=1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "TIME|Error|error" | sed 's/finish_in_kernel=1 n=4096 //; s/advect_stage.*A=/A=/; s/scalars.*//'; }
run "no s store" X=1
run "previous" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_prev.so
run "no s store" X=1
run "previous" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_prev.so
