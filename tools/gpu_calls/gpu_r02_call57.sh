#!/bin/bash
# the ring of the fused sweeps from stored edges (CUP2D_FUSED_RING=stored) against the whole-block ring: parity with the five sweeps, timing
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CUP2D_FUSED_RING=stored timeout 200 python tools/gpu_ring_check.py time 2>&1 | tail -12
timeout 200 python tools/gpu_ring_check.py time 2>&1 | tail -6
