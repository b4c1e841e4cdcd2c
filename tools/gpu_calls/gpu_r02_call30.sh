#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
nproc
timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | cut -c1-150 | head -16
CUP2D_HOST_THREADS=1 timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | cut -c1-150 | sed -n 2,12p
timeout 2400 python -m pytest tests/test_amr.py -m gpu -q > $OUT/r02_pytest30.log 2>&1; echo "amr rc=$?"; tail -3 $OUT/r02_pytest30.log
