#!/usr/bin/env python3
"""A/B of builds of libcup2d_hip.so on one box (development aid): for every library given (CUP2D_LIB, one child process
each): four capped iterations of the fused solver against the five sweeps on a 1024^2 grid, then the sampled per-kernel
timers of the step at 4096^2.   usage: python tools/gpu_lib_variants.py default path/to/lib_a.so default@CUP2D_FUSED_FORM=edge
(lib@NAME=value,NAME=value: environment of that child)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time, numpy as np
sys.path.insert(0, %r)
import cup2d_amd
from cup2d_amd import lib as L
from oracle import oracle as O
out = {}
n = 1024
rng = np.random.default_rng(3)
b = rng.uniform(-1, 1, (n, n)); b -= b.mean()
xs = {}
import os
for fused in (() if os.environ.get("SKIP_REL4") else (True, False)):
    with cup2d_amd.Simulation(n // 8) as s:
        s.set_precond(L.PRECOND_MFMA)
        s.set_solver(fused=fused, finish_in_kernel=True)
        s.keep_last_iterate(True)
        s.tmp = b; s.fill(L.PRES, 0.0)
        s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=4)
        s.last_iterate_to(L.POLD)
        xs[fused] = s.pold.copy()
if xs: out["rel4"] = float(np.abs(xs[True] - xs[False]).max() / np.abs(xs[False]).max())
n = 4096
with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
    s.vel = O.taylor_green(n, noise=1e-3, seed=1)
    s.set_solver(fused=True, finish_in_kernel=True)
    for _ in range(2):
        s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    s.set_timing(2)
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    s.synchronize(); el = time.perf_counter() - t0
    sw = {}
    for name in ("sweep_A", "sweep_C", "sweep_E", "sweep_EA"):
        ms, calls = s.get_timing(L.TIMER_NAMES.index(name))
        sw[name] = round(ms / max(calls, 1) * 1e3, 1)
    out.update(ms_per_step=round(el / 6 * 1e3, 3), avg_us=sw)
print("RESULT " + json.dumps(out))
''' % ROOT
for rep in range(int(os.environ.get("REPS", "1"))):
    for lib in sys.argv[1:] or ["default"]:
        env = dict(os.environ)
        tag = lib
        lib, _, sets = lib.partition("@")
        for kv in filter(None, sets.split(",")):
            env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        if lib != "default":
            env["CUP2D_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
        res = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")]
        print(os.path.basename(tag), res[0][7:] if res else ("rc %d: " % r.returncode) + r.stdout.decode()[-800:], flush=True)
