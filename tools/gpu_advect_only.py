#!/usr/bin/env python3
"""Runs only the fused WENO5 advect-diffuse RK2 stages at n^2 (default 4096) -- target for rocprofv3 --pmc."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
strict = len(sys.argv) > 3 and sys.argv[3] == "strict"
with cup2d_amd.Simulation(n // 8) as s:
    xs = (np.arange(n) + 0.5) / n
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    vel = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
    vel += 1e-3 * np.random.default_rng(1).uniform(-1, 1, vel.shape)
    s.set_math(strict)
    s.vel = vel
    dt = s.compute_dt()
    s.advect_diffuse_rk2(dt)
    s.set_timing(True)
    for _ in range(reps):
        s.advect_diffuse_rk2(dt)
    s.synchronize()
    ms, calls = s.get_timing(L.T_ADVECT_STAGE)
    print("advect_stage n=%d math=%s: %.1f us per launch over %d launches" % (n, "strict" if strict else "fast", 1e3 * ms / calls, calls))
