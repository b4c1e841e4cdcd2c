#!/usr/bin/env python3
"""Do the two timing modes of the solver's launches follow the PLACEMENT of the vectors in memory?  (development aid)
K contexts of the bench size created one after the other in ONE process, all kept alive (every context gets memory the others do
not hold; the pool is off), the same capped solve timed on each: if the launches' durations differ from context to context
within a process, it is where the buffers landed and not the box or the process."""
import os, sys, time
os.environ["CUP2D_POOL"] = "0"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd
from cup2d_amd import lib as L

n = int(os.environ.get("N", 4096))
K = int(os.environ.get("K", 10))
rng = np.random.default_rng(1)
b = rng.uniform(-1, 1, (n, n)); b -= b.mean()
sims = []
for k in range(K):
    s = cup2d_amd.Simulation(n // 8, nu=1e-3)
    sims.append(s)
    s.tmp = b
    s.set_solver(fused=True, finish_in_kernel=True)
    out = []
    for rep in range(2):
        s.fill(L.PRES, 0.0)
        s.set_timing(1)
        s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        tm = {}
        for i, name in enumerate(L.TIMER_NAMES):
            ms, cnt = s.get_timing(i)
            if cnt:
                tm[name] = ms / cnt * 1e3
        out.append((tm.get("sweep_C", 0), tm.get("sweep_EA", 0)))
    print("context %2d: C+D' %.1f / %.1f us   E+A+B %.1f / %.1f us   (field ptr %#x)" % (k, out[0][0], out[1][0], out[0][1], out[1][1], s.field_ptr(L.PRES)), flush=True)
# and again on the first contexts: does a context keep its mode?
for k in (0, 1, 2):
    s = sims[k]
    s.fill(L.PRES, 0.0); s.set_timing(1)
    s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    print("context %2d again: C+D' %.1f us  E+A+B %.1f us" % (k, s.get_timing(L.TIMER_NAMES.index("sweep_C"))[0] / s.get_timing(L.TIMER_NAMES.index("sweep_C"))[1] * 1e3,
          s.get_timing(L.TIMER_NAMES.index("sweep_EA"))[0] / s.get_timing(L.TIMER_NAMES.index("sweep_EA"))[1] * 1e3), flush=True)
for s in sims:
    s.close()
