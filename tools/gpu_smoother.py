#!/usr/bin/env python3
"""Timing of the smoother / residual tile kernels at 4096^2 next to the per-block 5-point kernels (development aid)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402

n = int(os.environ.get("N", "4096"))
with cup2d_amd.Simulation(n // 8) as s:
    rng = np.random.default_rng(1)
    b = rng.uniform(-1, 1, (n, n))
    b -= b.mean()
    s.tmp = b
    s.fill(L.PRES, 0.0)
    s.jacobi_sweeps(5)
    for what, fn in (("jacobi x50", lambda: s.jacobi_sweeps(50)), ("residual x20", lambda: [s.poisson_residual() for _ in range(20)]),
                     ("apply_A x20", lambda: [s.apply_A(L.POLD, L.PRES) for _ in range(20)])):
        s.synchronize()
        s.set_timing(1)
        t0 = time.perf_counter()
        fn()
        s.synchronize()
        wall = time.perf_counter() - t0
        ms, calls = s.get_timing(L.TIMER_NAMES.index("smoother"))
        s.set_timing(0)
        reps = 50 if "jacobi" in what else 20
        per = ms / calls if calls else wall * 1e3 / reps
        bpc = 24 if "apply" not in what else 16
        print("%-14s %8.1f us/launch (events: %s)  %6.0f GB/s on %d B/cell = %.3f of 8 TB/s | wall %.1f us/launch"
              % (what, per * 1e3, bool(calls), bpc * n * n / per / 1e6, bpc, bpc * n * n / per / 1e6 / 8000, wall / reps * 1e6), flush=True)
