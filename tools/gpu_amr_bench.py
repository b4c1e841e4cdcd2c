#!/usr/bin/env python3
"""Timing of the block-AMR path on a three-level grid (BASELINE.json configs[4] shape: finest level = LFINE,
a refined band around a circle) next to the uniform path at the same cell count: development aid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import amr as A, lib as L  # noqa: E402

LF = int(os.environ.get("LFINE", "8"))  # finest level: 2^LF blocks per side (8 -> 2048^2 equivalent, 9 -> 4096^2)
t0 = time.perf_counter()
g = A.circle_band_grid(LF)
blocks = g.blocks
t_grid = time.perf_counter() - t0
nb = g.nblocks
print("grid: %d blocks, levels %s, host regrid+tables %.2f s" % (nb, np.bincount(blocks[:, 0]).tolist(), t_grid), flush=True)
with A.AmrSimulation(g) as s:
    xc, yc = g.cell_centres()
    vel = np.stack([np.sin(2 * np.pi * xc) * np.cos(2 * np.pi * yc), -np.cos(2 * np.pi * xc) * np.sin(2 * np.pi * yc)], -1)
    s.set_field(L.VEL, vel)
    s.set_math(False)
    t0 = time.perf_counter()
    s.install_poisson_matrix()
    print("operator assembly + sliced-ELL upload %.2f s" % (time.perf_counter() - t0), flush=True)
    print("operator:", s.matrix_stats(), flush=True)
    if os.environ.get("SWEEPS"):
        s.set_solver(fused=False, finish_in_kernel=False)
    elif os.environ.get("FORM"):  # full: three sweeps + two rows launches per iteration (k_fused HYB); default: two + two (k_edge HYB)
        s.set_solver(fused=True, finish_in_kernel=True, form=os.environ["FORM"])
    for _ in range(2):
        r = s.step(max_iter=50)
    nst = 5
    t0 = time.perf_counter()
    for _ in range(nst):
        r = s.step(max_iter=50)
    L.check(s.L.cup2d_synchronize(s._ctx)) if hasattr(s.L, "cup2d_synchronize") else None
    el = (time.perf_counter() - t0) / nst
    print("AMR step %.2f ms: %.1f Mcell-updates/s (%d cells), iters=%d err=%.2e  [%s solver, form %s]"
          % (el * 1e3, nb * 64 / el / 1e6, nb * 64, r["iters"], r["err"], s.last_solver(), s.last_solver_form()), flush=True)
    if os.environ.get("NOTIMING"):
        sys.exit(0)
    L.check(s.L.cup2d_set_timing(s._ctx, 1))  # per-launch events: the breakdown below, not the figure above
    for _ in range(3):
        r = s.step(max_iter=50)
    import ctypes
    for i, name in enumerate(L.TIMER_NAMES):
        ms, calls = ctypes.c_double(), ctypes.c_int()
        s.L.cup2d_get_timing(s._ctx, i, ctypes.byref(ms), ctypes.byref(calls))
        if calls.value:
            print("   %-14s %8.1f us avg x %d" % (name, 1e3 * ms.value / calls.value, calls.value))
