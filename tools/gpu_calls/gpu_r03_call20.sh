#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_comm.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8
