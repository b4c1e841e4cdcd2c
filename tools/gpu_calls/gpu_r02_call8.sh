#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
hostname; getent hosts $(hostname) || echo "hostname does not resolve"
timeout 900 python -m pytest tests/test_comm.py -m gpu -q --durations=4 > $OUT/r02_pytest8.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r02_pytest8.log
