#!/usr/bin/env python3
"""The fused sweeps with the ring from stored edges (CUP2D_FUSED_RING=stored) against the five sweeps on the same systems:
a few iterations at zero tolerance (round-off apart the same iterate), a converged solve (the reference's criterion against
the oracle-free residual the library recomputes), non-Hilbert block orders (more than 16 ring entries per tile), restarts.
Prints per-iteration time at 4096^2 with `time`.  Run once per ring mode (the mode is read when the library is loaded)."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
from cup2d_amd.grid import BlockGrid  # noqa: E402

mode = os.environ.get("CUP2D_FUSED_RING", "blocks")
ok = True


def rhs(nx, ny, seed):
    b = np.random.default_rng(seed).uniform(-1, 1, (ny, nx))
    return b - b.mean()


def solve(s, b, **kw):
    s.tmp = b
    s.fill(L.PRES, 0.0)
    info = s.poisson_solve(**kw)
    return s.pres, info


for order, nbx, nby in (("hilbert", 32, 32), ("hilbert", 8, 8), ("rowmajor", 12, 10), ("hilbert", 6, 5), ("rowmajor", 1, 1), ("hilbert", 64, 64)):
    g = BlockGrid(nbx, nby, order=order)
    b = rhs(g.nx, g.ny, 17)
    with cup2d_amd.Simulation(nbx, nby, grid=g, h=1.0 / g.nx) as s:
        s.set_precond(L.PRECOND_MFMA)
        s.set_solver(fused=False, finish_in_kernel=False)
        xa, ia = solve(s, b, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=6)
        s.set_solver(fused=True, finish_in_kernel=True)
        xb, ib = solve(s, b, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=6)
        d = np.abs(xa - xb).max() / max(1e-300, np.abs(xa).max())
        good = ia["iters"] == ib["iters"] and abs(ia["err"] - ib["err"]) <= 1e-10 * max(1.0, ia["err_init"]) and d < 1e-9
        # converged solve: the criterion, recomputed from the fields
        xc, ic = solve(s, b, tol=1e-9, max_restarts=100)
        s.tmp = b
        res = s.poisson_residual()
        good2 = ic["err"] <= 1e-9 and res <= 1.05e-9
        s.set_solver(fused=False, finish_in_kernel=False)
        xd, idd = solve(s, b, tol=1e-9, max_restarts=100)
        print("%-8s %2dx%-2d ring=%s: 6 its |x5 - xf|/|x| %.2e err %.3e vs %.3e %s | converged: iters %d (five sweeps %d) err %.2e true residual %.2e %s" % (
            order, nbx, nby, mode, d, ib["err"], ia["err"], "OK" if good else "FAIL", ic["iters"], idd["iters"], ic["err"], res, "OK" if good2 else "FAIL"))
        ok = ok and good and good2
if "time" in sys.argv:
    n = 4096
    with cup2d_amd.Simulation(n // 8) as s:
        b = rhs(n, n, 3)
        s.set_solver(fused=True, finish_in_kernel=True)
        solve(s, b, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=10)
        s.synchronize()
        t0 = time.perf_counter()
        x, info = solve(s, b, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize()
        dt = time.perf_counter() - t0
        s.tmp = b
        res = s.poisson_residual()
        print("4096^2 ring=%s: 50 iterations %.2f ms (%.4f ms per iteration incl. set-up), err %.6e, true residual %.6e, restarts %d" % (
            mode, 1e3 * dt, 1e3 * dt / 50, info["err"], res, info["restarts"]))
        s.set_timing(True)
        solve(s, b, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize()
        for nm, t in (("A", L.T_SWEEP_A), ("C", L.T_SWEEP_C), ("E", L.T_SWEEP_E)):
            ms, calls = s.get_timing(t)
            print("   sweep %s %.1f us over %d launches" % (nm, 1e3 * ms / max(1, calls), calls))
print("ring=%s: %s" % (mode, "ALL OK" if ok else "FAILURES"))
sys.exit(0 if ok else 1)
