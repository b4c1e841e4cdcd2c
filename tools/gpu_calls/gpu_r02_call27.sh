#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in 1 2 4; do echo "rows WG per CU $w"; CUP2D_ROWS_WG_PER_CU=$w LFINE=9 timeout 600 python tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|sweep_A|sweep_C"; done
SWEEPS=1 LFINE=9 timeout 600 python tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step"
