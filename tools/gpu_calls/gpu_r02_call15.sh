#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed.py tests/test_amr.py tests/test_comm.py -m gpu -x -q --durations=5 > $OUT/r02_pytest15.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/r02_pytest15.log
