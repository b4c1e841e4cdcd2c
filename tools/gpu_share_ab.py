#!/usr/bin/env python3
"""A/B of the sibling hand-over mask of the edge-form sweeps (CUP2D_EDGE_SHARE, read once per process: run once per value) at
4096^2: step time and the two launches of an iteration (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd, bench
from cup2d_amd import lib as L
n = int(os.environ.get("N", 4096))
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
    print("EDGE_SHARE=%s  %d^2: %.3f ms/step = %.1f Mcell-updates/s  C+D' %.1f us  E+A+B %.1f us  err %.6e  placement kept %.0f slowest %.0f of %d" % (
        os.environ.get("CUP2D_EDGE_SHARE", "5"), n, el * 1e3, n * n / el / 1e6, 1e3 * tc[0] / tc[1], 1e3 * te[0] / te[1], r["err"],
        pl.get("kept_us", 0), pl.get("slowest_us", 0), pl.get("candidates", 0)), flush=True)
