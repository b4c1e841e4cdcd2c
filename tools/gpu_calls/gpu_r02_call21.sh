#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
NCCL_DEBUG=INFO timeout 900 python -m pytest tests/test_comm.py -m gpu -q -k "amr" -s > $OUT/r02_pytest21.log 2>&1; echo "pytest rc=$?"; grep -E "WARN|error|Error|failed" $OUT/r02_pytest21.log | head -20
timeout 900 python -m pytest tests/test_comm.py -m gpu -q > $OUT/r02_pytest21b.log 2>&1; echo "whole file rc=$?"; tail -5 $OUT/r02_pytest21b.log
