#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CUP2D_REGRID_POISON=1 timeout 1500 python -m pytest tests/test_amr.py -m gpu -q -x -k "adapt or regrid" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_amr.py tests/test_distributed.py -m gpu -q -x -k "adapt or regrid or amr" 2>&1 | tail -2
timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | cut -c1-150 | sed -n 2,16p
