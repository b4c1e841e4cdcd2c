#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from cup2d_amd import lib as L
from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
lstart, lmax, rtol, ctol, iters = [int(a) if i in (0, 1, 4) else float(a) for i, a in enumerate(sys.argv[1:6])] if len(sys.argv) > 5 else (4, 9, 0.5, 0.1, 50)
def by(blocks, arr): return {tuple(int(v) for v in b): arr[k] for k, b in enumerate(blocks)}
for steps in (1, 2, 3, 4, 6):
    R = O.ref_run_amr(level_start=lstart, level_max=lmax, steps=steps, rtol=rtol, ctol=ctol, nu=1e-3, max_iter=iters, env={"OMP_NUM_THREADS": "1"})
    g = AmrBlockGrid([(lstart, i, j) for j in range(1 << lstart) for i in range(1 << lstart)])
    x, y = g.cell_centres()
    u, v = np.zeros_like(x), np.zeros_like(x)
    for cx, cy, gam in ((0.35, 0.5, 1.0), (0.65, 0.5, -1.0)):
        dx, dy = x - cx, y - cy
        f = gam * np.exp(-(dx * dx + dy * dy) / (0.06 * 0.06)) / 0.06
        u += -dy * f; v += dx * f
    with AmrSimulation(g, nu=1e-3, cfl=0.5) as s:
        s.install_poisson_matrix(); s.set_math(True)
        s.set_field(L.VEL, np.stack([u, v], axis=-1))
        info = []
        for k in range(steps):
            dt = s.compute_dt(); s.adapt(rtol, ctol, lmax)
            r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters, dt=dt); info.append((s.grid.nblocks, r["iters"], "%.1e" % r["err"]))
        same = set(map(tuple, s.grid.blocks.tolist())) == set(map(tuple, R["blocks"].tolist()))
        rv, rp = by(R["blocks"], R["vel"].reshape(len(R["blocks"]), -1)), by(R["blocks"], R["pres"])
        vel, pres = s.get_field(L.VEL).reshape(s.grid.nblocks, -1), s.get_field(L.PRES)
        keys = list(map(tuple, s.grid.blocks.tolist()))
        dv = np.array([np.abs(vel[k] - rv[b]).max() for k, b in enumerate(keys)]) if same else None
        dp = np.array([np.abs(pres[k] - rp[b]).max() for k, b in enumerate(keys)]) if same else None
        lv = s.grid.blocks[:, 0]
        print("steps %d same leaves %s  ref %s  here %s" % (steps, same, [(st["blocks"], st["prev_iters"]) for st in R["steps"]], info))
        if same:
            print("   max|dv| %.2e max|dp| %.2e ; per level dv %s" % (dv.max(), dp.max(), {int(l): "%.1e" % dv[lv == l].max() for l in np.unique(lv)}), flush=True)
