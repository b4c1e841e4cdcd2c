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
    for qr, qc in ((0.97, 0.5), (0.99, 0.03), (0.99, 0.03), (0.995, 0.02)):  # a start-up regrid, then ones that change a few per cent
        s.vorticity()
        om = np.abs(s.get_field(L.TMP)).reshape(s.grid.nblocks, -1).max(1)
        rtol, ctol = np.quantile(om, qr), np.quantile(om, qc)
        for host in (True, False):
            n0 = s.grid.nblocks
            t0 = time.perf_counter()
            if host is False:
                pr = cProfile.Profile(); pr.enable()
            changed = s.adapt(rtol, ctol, LF + 1, host_fields=host)
            t1 = time.perf_counter()
            print("adapt(host_fields=%s) quantiles %.3f/%.2f: changed=%s %d -> %d blocks in %.3f s" % (host, qr, qc, changed, n0, s.grid.nblocks, t1 - t0), flush=True)
            if host is False:
                pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
            s.step(max_iter=20)
            s.vorticity()
            om = np.abs(s.get_field(L.TMP)).reshape(s.grid.nblocks, -1).max(1)
            rtol, ctol = np.quantile(om, qr), np.quantile(om, qc)
    t0 = time.perf_counter(); s.step(max_iter=50); s.step(max_iter=50); print("2 steps after: %.1f ms each" % ((time.perf_counter() - t0) * 500))
