#!/bin/bash
# smoother tuning experiments: non-temporal policies (bit 0: b loads, bit 1: out stores, bit 2: x loads), twice each
set -u
for rep in 1 2; do for nt in ${NTS:-0 1 2 4 5 6 7}; do echo "== NT=$nt"; CUP2D_SMOOTHER_NT=$nt timeout 300 python tools/gpu_smoother.py 2>&1 | grep -E "jacobi|residual|rror"; done; done
