#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "CHECK|TIME|Error|error" | sed 's/finish_in_kernel=1 n=4096 //; s/advect_stage.*A=/A=/; s/scalars.*//'; }
run "carry" X=1
run "no carry (dbg 16)" CUP2D_FUSED_DBG=16
run "previous lib" CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_prev2.so
run "carry" X=1
run "no carry (dbg 16)" CUP2D_FUSED_DBG=16
python - <<'PY'
import os, subprocess, sys, numpy as np
code = r'''
import numpy as np, sys
sys.path.insert(0, ".")
import cup2d_amd
from cup2d_amd import lib as L
n = 1024
rng = np.random.default_rng(3); b = rng.uniform(-1, 1, (n, n)); b -= b.mean()
with cup2d_amd.Simulation(n // 8) as s:
    s.tmp = b; s.fill(L.PRES, 0.0)
    r = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=40)
    np.save(sys.argv[1], s.pres); print(r)
'''
for tag, env in (("a", {}), ("b", {"CUP2D_FUSED_DBG": "16"})):
    subprocess.check_call([sys.executable, "-c", code, "/tmp/x_%s.npy" % tag], env=dict(os.environ, **env))
a, b = np.load("/tmp/x_a.npy"), np.load("/tmp/x_b.npy")
print("carry vs recompute: bitwise equal =", np.array_equal(a, b), " max diff", np.abs(a - b).max())
PY
timeout 900 python -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_distributed.py -m gpu -q -k "not amr" > $OUT/r02_pytest16.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r02_pytest16.log
