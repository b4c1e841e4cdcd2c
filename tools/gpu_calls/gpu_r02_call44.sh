#!/bin/bash
# the quad ("register walk") advect kernel: parity + A/B timing against the per-block kernel, chunk sizes
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/gpu_advect_only.py 4096 3 check 2>&1 | tail -3
CUP2D_ADVECT_WALK=0 timeout 300 python tools/gpu_advect_only.py 4096 3 2>&1 | tail -1
for ch in 1 2 8 16 0; do CUP2D_WALK_CHUNK=$ch timeout 300 python tools/gpu_advect_only.py 4096 3 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes_gpu.py -m gpu -q -x -k "advect or rectangular or functors_vs_golden or step_matches or consecutive or (functors_vs_live and 2048)" 2>&1 | tail -4
