#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amr.py -m gpu -q -k "penalisation" --durations=3 > $OUT/r02_pytest10.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/r02_pytest10.log
