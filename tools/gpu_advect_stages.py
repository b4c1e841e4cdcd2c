#!/usr/bin/env python3
"""Times the three forms of the WENO5 advect-diffuse launch separately at n^2 (default 4096): the functor alone (MODE 0),
RK stage 1 (old = the tile itself) and RK stage 2 (old from a second slab).  Library under test: CUP2D_LIB."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
with cup2d_amd.Simulation(n // 8) as s:
    xs = (np.arange(n) + 0.5) / n
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    vel = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
    vel += 1e-3 * np.random.default_rng(1).uniform(-1, 1, vel.shape)
    s.set_math(False)
    s.vel = vel
    dt = s.compute_dt()
    s.advect_diffuse_rk2(dt)
    out = []
    for name, call in (("rhs", lambda: s.advect_diffuse_rhs(dt)),
                       ("stage1", lambda: L.check(s.L.cup2d_advect_diffuse_stage(s._ctx, s.nu, dt, 1, L.BLOCKS_ALL), "stage")),
                       ("stage2", lambda: L.check(s.L.cup2d_advect_diffuse_stage(s._ctx, s.nu, dt, 2, L.BLOCKS_ALL), "stage"))):
        call()
        s.synchronize()
        s.set_timing(1)   # HIP events around every launch on its own stream (the host clock of back-to-back ctypes calls is launch-rate bound)
        for _ in range(reps):
            call()
        s.synchronize()
        ms = n_l = 0
        for t in ("advect_stage", "advect_stage2"):
            a, b = s.get_timing(L.TIMER_NAMES.index(t))
            ms, n_l = ms + a, n_l + b
        s.set_timing(0)
        out.append("%s %.1f" % (name, 1e3 * ms / max(1, n_l)))
    print("%s: us per launch (HIP events on the launch stream, mean of %d launches): %s" % (os.path.basename(os.environ.get("CUP2D_LIB", "default")), reps, "  ".join(out)))
