#!/usr/bin/env python3
"""Where the solver's eleven vectors lie (csrc/krylov_fused.hip tune_placement), arenas against separate allocations: fresh
processes, each running the placement search of one 4096^2 context with CUP2D_PLACEMENT_ARENA = a list of strides' pads (one
allocation carved into the eleven vectors, stride = 2^27 bytes + pad) in front of separate hipMallocs; prints the library's own
per-set timings (CUP2D_HOST_TIMING).  Development aid: which arrangement is reliably in the fast mode?"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import cup2d_amd
from cup2d_amd import lib as L
n = int(sys.argv[1])
with cup2d_amd.Simulation(n // 8) as s:
    b = np.random.default_rng(1).uniform(-1, 1, (n, n)); b -= b.mean()
    s.tmp = b
    s.fill(L.PRES, 0.0)
    s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=2)
    print("placement", s.placement())
''' % ROOT
pads = os.environ.get("PADS", "0,4096,65536,1048576,2097152,2162688,6291456,35651584,69206016")
n = sys.argv[1] if len(sys.argv) > 1 else "4096"
for trial in range(int(os.environ.get("TRIALS", "3"))):
    env = dict(os.environ, CUP2D_HOST_TIMING="1", CUP2D_PLACEMENT_ARENA=pads, CUP2D_PLACEMENT_TRIES=str(len(pads.split(",")) + 5),
               CUP2D_PLACEMENT_MAX_GB="40")
    r = subprocess.run([sys.executable, "-c", CHILD, n], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    print("---- process %d (rc %d)" % (trial, r.returncode))
    for line in r.stdout.decode().splitlines():
        if "tune_placement: set" in line or line.startswith("placement"):
            print(line.replace("[cup2d timing] tune_placement: ", ""))
    sys.stdout.flush()
