#!/bin/bash
# tools/gpu_fused_phases.sh -- where a wave of the fused sweeps spends the cycles of a tile (library built with -DFUSED_PHASES:
# cup2d_amd/variants/libcup2d_hip_phases.so), 4096^2, both ring forms
set -u
export TMPDIR=/tmp
L=$PWD/cup2d_amd/variants/libcup2d_hip_phases.so
for ring in blocks stored; do
CUP2D_FUSED_RING=$ring CUP2D_LIB=$L timeout 200 python - <<'PY' 2>&1 | grep "PHASES" | sort | uniq | head -16
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import cup2d_amd
from cup2d_amd import lib as L
n = 4096
with cup2d_amd.Simulation(n // 8) as s:
    b = np.random.default_rng(3).uniform(-1, 1, (n, n)); b -= b.mean()
    s.set_solver(fused=True, finish_in_kernel=True)
    s.tmp = b
    s.fill(L.PRES, 0.0)
    s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=8)
    s.synchronize()
PY
done
