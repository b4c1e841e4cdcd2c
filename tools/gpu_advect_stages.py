#!/usr/bin/env python3
"""The fused WENO5 stage kernel and its two floors at n^2 (default 4096), the way bench.py measures them (north_star_floors):
whole steps with sampled HIP events, the product / the arithmetic alone / the memory skeleton (cup2d_debug_walk_knockout) in
one context on the same data.  Library under test: CUP2D_LIB (development aid: A/B of kernel variants on one box)."""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd  # noqa: E402
import bench  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = argparse.Namespace(iters=int(os.environ.get("ITERS", 50)))
with cup2d_amd.Simulation(n // 8, nu=1e-3, cfl=0.5) as s:
    s.set_math(False)
    s.vel = bench.synthetic_velocity(n, n, 0, 0, n, n, seed=20250117)
    for _ in range(3):
        s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)
    fl = bench.north_star_floors(s, args, nsteps)
print("%s: %s" % (os.path.basename(os.environ.get("CUP2D_LIB", "default")), {k: v for k, v in fl.items() if k != "how"}))
