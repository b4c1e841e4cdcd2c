#!/usr/bin/env python3
"""A/B of the solver organisations on the GPU (development aid; the judged tests are tests/ -m gpu):
{five sweeps, tile-fused} x {finish as its own launch, finish by the last workgroup}.
Small-size check of every variant against the CPU oracle, then HIP-event timings at 4096^2 (one process,
one context, so the variants see the same box and the same data)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
from oracle import oracle as O  # noqa: E402

VARIANTS = [("sweeps", False), ("sweeps", True), ("fused", False), ("fused", True)]
if os.environ.get("VARIANTS"):  # e.g. VARIANTS=fused0,sweeps1
    VARIANTS = [(v[:-1], v[-1] == "1") for v in os.environ["VARIANTS"].split(",")]
ok = True
SKIP_CHECK = os.environ.get("SKIP_CHECK") == "1"

n = 128
rng = np.random.default_rng(3)
b = rng.uniform(-1, 1, (n, n))
b -= b.mean()
xo, io = O.bicgstab(b, tol=1e-10, max_restarts=100, max_iter=400)
with cup2d_amd.Simulation(n // 8) as s:
    for kind, fin in ([] if SKIP_CHECK else VARIANTS):
        s.set_solver(fused=kind == "fused", finish_in_kernel=fin)
        s.tmp = b
        s.fill(L.PRES, 0.0)
        try:
            info = s.poisson_solve(tol=1e-10, max_restarts=100, max_iter=400)
            res = np.abs(b - O.apply_A(s.pres)).max()
            good = res <= 1.05e-10 and abs(info["iters"] - io["iters"]) <= max(5, io["iters"] // 4)
            print("CHECK %-6s finish_in_kernel=%d iters=%d (oracle %d) err=%.2e true_res=%.2e %s"
                  % (kind, fin, info["iters"], io["iters"], info["err"], res, "ok" if good else "FAIL"), flush=True)
        except Exception as e:  # noqa: BLE001
            good = False
            print("CHECK %-6s finish_in_kernel=%d EXCEPTION %r" % (kind, fin, e), flush=True)
        ok = ok and good

n = int(os.environ.get("VARIANTS_N", "4096"))
iters = 50
with cup2d_amd.Simulation(n // 8) as s:
    xs = (np.arange(n) + 0.5) / n
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    vel = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
    vel += 1e-3 * np.random.default_rng(20250117).uniform(-1, 1, vel.shape)
    for kind, fin in VARIANTS:
        try:
            s.set_solver(fused=kind == "fused", finish_in_kernel=fin)
            s.vel = vel
            s.fill(L.PRES, 0.0)
            for _ in range(2):
                r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
            s.synchronize()
            plain = (time.perf_counter() - t0) / 4
            s.set_timing(2)
            for _ in range(2):
                s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
            s.synchronize()
            out = []
            for i, name in enumerate(L.TIMER_NAMES):
                ms, calls = s.get_timing(i)
                if calls:
                    out.append("%s=%.1f" % (name.replace("sweep_", ""), 1e3 * ms / calls))
            s.set_timing(0)
            print("TIME %-6s finish_in_kernel=%d n=%d step=%.3f ms (%.1f Mcell/s) iters=%d err=%.3e | us: %s"
                  % (kind, fin, n, plain * 1e3, n * n / plain / 1e6, r["iters"], r["err"], " ".join(out)), flush=True)
        except Exception as e:  # noqa: BLE001
            ok = False
            print("TIME %-6s finish_in_kernel=%d EXCEPTION %r" % (kind, fin, e), flush=True)
print("VARIANTS_%s" % ("OK" if ok else "FAILED"))
sys.exit(0 if ok else 1)
