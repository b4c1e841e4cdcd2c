#!/bin/bash
# C++ N-rank AMR driver against the one-rank driver
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_distributed.py -x -q -m gpu -p no:cacheprovider -k "cpp_mpi_driver_amr" -s > $OUT/r03_call6_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^\[cup2d\]\|^$" $OUT/r03_call6_pytest.log | tail -40
