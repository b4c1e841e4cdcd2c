#!/bin/bash
# round 6, call 11: one GPU, the finish of a reduction in the prologue of the consumer launch (k_edge MERGE 5): solver tests,
# A/B against the finish in the producer (CUP2D_DEFER_SCALARS=0) at 4096^2 and 2048^2 in one box
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_baseline_sizes_gpu.py tests/test_spmat_gpu.py -m gpu -x -q -p no:cacheprovider > $OUT/r06c11_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/r06c11_tests.log | cut -c1-300
grep -n "^E  " $OUT/r06c11_tests.log | head -10 | cut -c1-300
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
            r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize(); el = (time.perf_counter() - t0) / 20
        s.set_timing(2)
        for _ in range(8):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        tc, te = s.get_timing(L.TIMER_NAMES.index("sweep_C")), s.get_timing(L.TIMER_NAMES.index("sweep_EA"))
        pl = s.placement()
        print("DEFER=%s  %d^2: %.3f ms/step = %.1f Mcell-updates/s  C+D' %.1f us  E+A+B %.1f us  iters %d err %.6e placement %.0f" % (os.environ.get("CUP2D_DEFER_SCALARS", "1"), n, el * 1e3, n * n / el / 1e6, 1e3 * tc[0] / tc[1], 1e3 * te[0] / te[1], r["iters"], r["err"], pl.get("kept_us", 0)), flush=True)
PY
for i in 1 2; do
  python3 /tmp/t.py 2>&1 | tail -2
  CUP2D_DEFER_SCALARS=0 python3 /tmp/t.py 2>&1 | tail -2
done
