#!/usr/bin/env python3
"""Where does a regrid of the BASELINE.json configs[4]-shaped grid spend its time?  (development aid, GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_amr import _circle_grid
from cup2d_amd import amr as A, lib as L

LF = int(os.environ.get("LFINE", "9"))
g = _circle_grid(LF)
print("grid", g.nblocks, "blocks")
import cProfile, pstats
with A.AmrSimulation(g) as s:
    xc, yc = g.cell_centres()
    # a vortex ring near the refined band, shifted: the tags move the band
    r = np.hypot(xc - 0.53, yc - 0.5)
    w = np.exp(-((r - 0.25) / 0.03) ** 2)
    vel = np.stack([-(yc - 0.5) * w, (xc - 0.53) * w], -1)
    s.set_field(L.VEL, vel)
    s.install_poisson_matrix()
    s.step(max_iter=50)
    s.vorticity()
    om = np.abs(s.get_field(L.TMP)).reshape(g.nblocks, -1).max(1)
    rtol, ctol = np.quantile(om, 0.97), np.quantile(om, 0.5)
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    changed = s.adapt(rtol, ctol, LF + 1)
    pr.disable()
    t1 = time.perf_counter()
    print("adapt: changed=%s -> %d blocks in %.3f s" % (changed, s.grid.nblocks, t1 - t0))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    t0 = time.perf_counter(); s.step(max_iter=50); s.step(max_iter=50); print("2 steps after: %.1f ms each" % ((time.perf_counter() - t0) * 500))
