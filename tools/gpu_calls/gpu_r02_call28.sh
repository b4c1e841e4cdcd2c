#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_amr.py tests/test_spmat_gpu.py tests/test_comm.py tests/test_distributed.py tests/test_solver_variants_gpu.py -m gpu -q > $OUT/r02_pytest28.log 2>&1; echo "rc=$?"; tail -8 $OUT/r02_pytest28.log
