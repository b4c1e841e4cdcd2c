#!/usr/bin/env python3
"""What the launch in front of RK stage 1 does to its duration (development aid): steps back to back (dt from the maxima the
projection left), with a call in between (the step recomputes max|u|: a pass over the velocity in front of stage 1), with a
device copy of the velocity onto itself in between, and with a pause on the host in between."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd, bench
from cup2d_amd import lib as L
n = 4096
with cup2d_amd.Simulation(n // 8, nu=1e-3, cfl=0.5) as s:
    s.set_math(False)
    s.vel = bench.synthetic_velocity(n, n, 0, 0, n, n, seed=20250117)
    for _ in range(5):
        s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    i1, i2 = L.TIMER_NAMES.index("advect_stage"), L.TIMER_NAMES.index("advect_stage2")
    def between_nothing(): pass
    def between_call(): s.last_solver()
    def between_copy(): L.check(s.L.cup2d_copy_field(s.ctx, L.TMPV, L.VEL), "copy"); L.check(s.L.cup2d_copy_field(s.ctx, L.VEL, L.TMPV), "copy")
    def between_sleep(): s.synchronize(); time.sleep(0.002)
    def between_copy_out(): L.check(s.L.cup2d_copy_field(s.ctx, L.TMPV, L.VEL), "copy")
    def between_unrelated():
        for _ in range(2):
            L.check(s.L.cup2d_copy_field(s.ctx, L.TMP, L.CHI), "copy"); L.check(s.L.cup2d_copy_field(s.ctx, L.POLD, L.CHI), "copy")
    for tag, fn in (("back to back", between_nothing), ("a call in between (max|u| recomputed)", between_call),
                    ("velocity copied to TMPV and back in between", between_copy), ("velocity copied to TMPV only", between_copy_out),
                    ("unrelated scalar fields copied (4 x 134 MB)", between_unrelated), ("2 ms pause on the host in between", between_sleep),
                    ("back to back again", between_nothing)):
        s.set_timing(3)
        for _ in range(10):
            fn()
            s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        (m1, n1), (m2, n2) = s.get_timing(i1), s.get_timing(i2)
        s.set_timing(0)
        print(os.path.basename(os.environ.get("CUP2D_LIB", "product")), "%-48s stage 1 %.1f us (%d)   stage 2 %.1f us (%d)" % (tag, 1e3 * m1 / n1, n1, 1e3 * m2 / n2, n2), flush=True)
