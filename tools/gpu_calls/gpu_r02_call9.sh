#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_spmat_gpu.py -m gpu -q -s -k "all_sites" --durations=5 > $OUT/r02_pytest9.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/r02_pytest9.log
