#!/bin/bash
# edge form of the fused sweeps: first run on the GPU (parity against the five sweeps, timing against the full form)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 tools/gpu_edge_check.py check time > $OUT/r03_edge_check.log 2>&1; echo "rc=$?"; cat $OUT/r03_edge_check.log | cut -c1-1500
