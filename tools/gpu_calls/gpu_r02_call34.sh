#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | cut -c1-150 | sed -n 2,14p; timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | grep "steps after"
CUP2D_POOL=0 timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | grep "adapt:\|steps after"
timeout 2400 python -m pytest tests/test_amr.py tests/test_distributed.py tests/test_comm.py tests/test_spmat_gpu.py tests/test_solver_variants_gpu.py -m gpu -q > $OUT/r02_pytest34.log 2>&1; echo "rc=$?"; tail -4 $OUT/r02_pytest34.log
