#!/usr/bin/env python3
"""Quick GPU sanity + timing of the hot-path kernels (development aid; the judged tests are tests/ -m gpu).
Checks against the CPU oracle at small sizes, then prints HIP-event timings at 4096^2."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
from oracle import oracle as O  # noqa: E402

kind = os.environ.get("PRECOND", "default")  # (applied below through cup2d_set_precond)
ok = True


def check(name, cond, detail=""):
    global ok
    print("%-44s %s %s" % (name, "ok" if cond else "FAIL", detail))
    ok = ok and bool(cond)


n = 64
rng = np.random.default_rng(3)
x = rng.uniform(-1, 1, (n, n))
with cup2d_amd.Simulation(n // 8) as s:
    s.pres = x
    P = s.P_inv()
    s.precond(L.TMP, L.PRES)
    d = np.abs(s.tmp - O.precond(x, P)).max()
    check("precond[%s] vs oracle" % kind, d < 1e-14, "%.2e" % d)
    vel = O.taylor_green(n, noise=0.3, seed=7)
    h, nu = 1.0 / n, 1e-3
    dt = O.compute_dt(h, nu, 0.5, np.abs(vel).max())
    ref = O.advect_diffuse_rhs(vel, h, nu, dt)
    ref2, _ = O.rk2_advect_diffuse(vel, h, nu, dt)
    s.set_math(True)
    s.vel = vel
    s.advect_diffuse_rhs(dt)
    check("advect STRICT rhs bit-exact", np.array_equal(s.tmpV, ref), "%.2e" % np.abs(s.tmpV - ref).max())
    s.advect_diffuse_rk2(dt)
    check("advect STRICT rk2 bit-exact", np.array_equal(s.vel, ref2))
    s.set_math(False)
    s.vel = vel
    s.advect_diffuse_rhs(dt)
    e = np.abs(s.tmpV - ref).max() / np.abs(ref).max()
    check("advect FAST rhs rel err", e <= 2e-13, "%.2e" % e)
    s.advect_diffuse_rk2(dt)
    e = np.abs(s.vel - ref2).max() / np.abs(ref2).max()
    check("advect FAST rk2 rel err", e <= 1e-13, "%.2e" % e)
    # solver
    b = O.laplacian_sub(x, O.pressure_rhs(ref2, h, dt))
    b -= b.mean()
    xo, io = O.bicgstab(b, tol=1e-10, max_restarts=100, max_iter=300)
    s.tmp = b
    s.fill(L.PRES, 0.0)
    info = s.poisson_solve(tol=1e-10, max_restarts=100, max_iter=300)
    res = np.abs(b - O.apply_A(s.pres)).max()
    check("solver iters %d vs oracle %d" % (info["iters"], io["iters"]), abs(info["iters"] - io["iters"]) <= 3 and res <= 1.0001e-10,
          "res %.2e" % res)

if "--time" in sys.argv:
    n = 4096
    with cup2d_amd.Simulation(n // 8) as s:
        xs = (np.arange(n) + 0.5) / n
        X, Y = np.meshgrid(xs, xs, indexing="xy")
        vel = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
        vel += 1e-3 * np.random.default_rng(1).uniform(-1, 1, vel.shape)
        for strict in (False, True):
            s.set_math(strict)
            s.vel = vel
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=10)
            s.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(2):
                s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
            s.synchronize()
            el = (time.perf_counter() - t0) / 2
            out = []
            for i, name in enumerate(L.TIMER_NAMES):
                ms, calls = s.get_timing(i)
                if calls:
                    out.append("%s=%.1fus" % (name, 1e3 * ms / calls))
            s.set_timing(False)
            print("TIMING precond=%s math=%s step=%.2fms  %s" % (kind, "strict" if strict else "fast", el * 1e3, " ".join(out)))
print("QUICK_%s" % ("OK" if ok else "FAILED"))
sys.exit(0 if ok else 1)
