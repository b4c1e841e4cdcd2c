#!/usr/bin/env python3
"""GPU check of the edge form of the fused sweeps (csrc/krylov_edge.h) -- development aid; the judged tests are tests/ -m gpu.
Each configuration runs in a child process (the form is chosen by environment variables read once per process):
  * 4 and 50 capped iterations on several grids: last iterate of {edge+share, edge, full} against the five sweeps;
  * timing of the sweeps at 4096^2 (HIP-event timers of the library), 3 steps of 50 iterations."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, numpy as np
sys.path.insert(0, %r)
import cup2d_amd
from cup2d_amd import lib as L
from cup2d_amd.grid import BlockGrid
what = sys.argv[1]
out = {}
if what == "check":
    rng = np.random.default_rng(5)
    for order, nbx, nby in (("hilbert", 8, 8), ("hilbert", 32, 32), ("rowmajor", 5, 3), ("hilbert", 6, 5), ("hilbert", 64, 32), ("rowmajor", 16, 16), ("hilbert", 128, 128)):
        g = BlockGrid(nbx, nby, order=order)
        b = rng.uniform(-1, 1, (g.ny, g.nx)); b -= b.mean()
        res = {}
        for iters in (4, 50):
            xs = {}
            for fused in (True, False):
                with cup2d_amd.Simulation(nbx, nby, grid=g) as s:
                    s.set_precond(L.PRECOND_MFMA)
                    s.set_solver(fused=fused, finish_in_kernel=True)
                    s.keep_last_iterate(True)
                    s.tmp = b; s.fill(L.PRES, 0.0)
                    info = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
                    s.last_iterate_to(L.POLD)
                    xs[fused] = (s.pold.copy(), info)
            d = float(np.abs(xs[True][0] - xs[False][0]).max() / max(1e-300, np.abs(xs[False][0]).max()))
            res[iters] = {"rel_diff_last_iterate": d, "err": [xs[True][1]["err"], xs[False][1]["err"]], "iters": [xs[True][1]["iters"], xs[False][1]["iters"]]}
        out["%%s %%dx%%d" %% (order, nbx, nby)] = res
    # zero tolerance, many iterations: the breakdown restarts behind convergence to round-off, best iterate returned
    g = BlockGrid(8, 8, order="hilbert")
    b = rng.uniform(-1, 1, (g.ny, g.nx)); b -= b.mean()
    rr = {}
    for fused in (True, False):
        with cup2d_amd.Simulation(8, 8, grid=g) as s:
            s.set_precond(L.PRECOND_MFMA)
            s.set_solver(fused=fused, finish_in_kernel=True)
            s.tmp = b; s.fill(L.PRES, 0.0)
            info = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=300)
            rr[fused] = {k: info[k] for k in ("iters", "restarts", "err")}
    out["restarts 8x8"] = rr
else:
    n = 4096
    from oracle import oracle as O
    with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
        s.vel = O.taylor_green(n, noise=1e-3, seed=1)
        s.set_solver(fused=True, finish_in_kernel=True)
        for _ in range(2):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.set_timing(2)
        import time
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        s.synchronize(); el = time.perf_counter() - t0
        sw = {}
        for name in ("sweep_A", "sweep_C", "sweep_E", "sweep_EA", "advect_stage", "poisson_rhs", "project"):
            ms, calls = s.get_timing(L.TIMER_NAMES.index(name))
            if calls:
                sw[name] = round(ms / calls * 1e3, 1)
        out = {"ms_per_step": round(el / 4 * 1e3, 3), "avg_us": sw}
print("RESULT " + json.dumps(out))
''' % ROOT

def run(what, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, what], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    txt = r.stdout.decode()
    for line in txt.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    return {"rc": r.returncode, "tail": txt[-1500:]}

variants = {"eab": {"CUP2D_FUSED_FORM": "eab"}, "eab-allshare": {"CUP2D_FUSED_FORM": "eab", "CUP2D_EDGE_SHARE": "15"},
            "eab-noshare": {"CUP2D_FUSED_FORM": "eab", "CUP2D_EDGE_SHARE": "0"},
            "edge+share": {"CUP2D_FUSED_FORM": "edge", "CUP2D_EDGE_SHARE": "15"}, "edge": {"CUP2D_FUSED_FORM": "edge", "CUP2D_EDGE_SHARE": "0"}, "full": {"CUP2D_FUSED_FORM": "full"}}
only = os.environ.get("VARIANTS")
if only:
    variants = {k: v for k, v in variants.items() if k in only.split(",")}
which = sys.argv[1:] or ["check", "time"]
for what in which:
    for name, env in variants.items():
        try:
            print(what, name, json.dumps(run(what, env)), flush=True)
        except subprocess.TimeoutExpired:
            print(what, name, "TIMEOUT", flush=True)
