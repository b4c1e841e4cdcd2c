#!/usr/bin/env python3
"""Runs only the fused WENO5 advect-diffuse RK2 stages at n^2 (default 4096) -- target for rocprofv3 --pmc.
Prints the time per launch and, with `check`, the difference between the FAST result (the quad kernel, or the
per-block one under CUP2D_ADVECT_WALK=0) and the STRICT per-block kernel (bit-identical to the reference) on the
same field."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
from cup2d_amd import lib as L  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
strict = "strict" in sys.argv[3:]
check = "check" in sys.argv[3:]
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
    tag = "walk=%s chunk=%s" % (os.environ.get("CUP2D_ADVECT_WALK", "1"), "persistent")
    print("advect_stage n=%d math=%s %s: %.1f us per launch over %d launches" % (n, "strict" if strict else "fast", tag, 1e3 * ms / calls, calls))
    if check:
        s.set_timing(False)
        out = {}
        for mode in (True, False):
            s.set_math(mode)
            s.vel = vel
            s.advect_diffuse_rhs(dt)
            rhs = s.tmpV
            s.advect_diffuse_rk2(dt)
            out[mode] = (rhs, s.vel)
        d0 = np.abs(out[False][0] - out[True][0]).max() / np.abs(out[True][0]).max()
        d1 = np.abs(out[False][1] - out[True][1]).max() / np.abs(out[True][1]).max()
        print("  fast vs strict at n=%d: rhs %.2e of max|rhs| (tolerance 2e-13), rk2 %.2e of max|vel| (1e-13) -> %s" % (
            n, d0, d1, "OK" if d0 <= 2e-13 and d1 <= 1e-13 else "FAIL"))
