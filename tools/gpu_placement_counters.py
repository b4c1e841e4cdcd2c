#!/usr/bin/env python3
"""The two launches of an iteration under hardware counters for ONE placement of the solver's vectors (the caller chooses it through
CUP2D_PLACEMENT_TRIES: 1 keeps the context's own set, the default searches and repairs): a few steps at 4096^2.  Run under
rocprofv3 --pmc ... (tools/gpu_calls/gpu_r06_call37.sh); prints the placement it ran on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd, bench
n = 4096
with cup2d_amd.Simulation(n // 8, nu=1e-3, cfl=0.5) as s:
    s.set_math(False)
    s.vel = bench.synthetic_velocity(n, n, 0, 0, n, n, seed=20250117)
    for _ in range(3):
        r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    s.synchronize()
    print("placement", s.placement(), flush=True)
