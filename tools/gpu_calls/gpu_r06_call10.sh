#!/bin/bash
# round 6, call 10: what the in-kernel finish of a reduction costs a launch (timing-only build without it for k_edge MODE 2 / 3:
# the scalars stay frozen, the iterations run on garbage at the same durations) at 4096^2 and 2048^2
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/t.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import cup2d_amd, bench
from cup2d_amd import lib as L
for n in (4096, 2048):
    with cup2d_amd.Simulation(n // 8, nu=1e-3, cfl=0.5) as s:
        s.set_math(False)
        s.vel = bench.synthetic_velocity(n, n, 0, 0, n, n, seed=20250117)
        for _ in range(4):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize(); el = (time.perf_counter() - t0) / 20
        s.set_timing(2)
        for _ in range(8):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        tc, te = s.get_timing(L.TIMER_NAMES.index("sweep_C")), s.get_timing(L.TIMER_NAMES.index("sweep_EA"))
        print("%s  %d^2: %.3f ms/step   C+D' %.1f us  E+A+B %.1f us" % (os.path.basename(os.environ.get("CUP2D_LIB", "product")), n, el * 1e3, 1e3 * tc[0] / tc[1], 1e3 * te[0] / te[1]), flush=True)
PY
for i in 1 2; do
  python3 /tmp/t.py 2>&1 | tail -2
  CUP2D_LIB=$PWD/tools/ab/libcup2d_hip_nofin.so python3 /tmp/t.py 2>&1 | tail -2
done
