#!/usr/bin/env python3
"""How far apart do two summation orders drift over ten AMR steps of fifty UNCONVERGED BiCGSTAB iterations each?
(the large case of tests/test_amr.py::test_amr_run_with_regridding_vs_reference_gpu: fused vs five sweeps, and vs more iterations)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cup2d_amd import lib as L
from cup2d_amd.amr import AmrBlockGrid, AmrSimulation

def run(fused, iters, lstart=4, lmax=9, steps=10, rtol=0.5, ctol=0.1):
    g = AmrBlockGrid([(lstart, i, j) for j in range(1 << lstart) for i in range(1 << lstart)])
    x, y = g.cell_centres()
    u, v = np.zeros_like(x), np.zeros_like(x)
    for cx, cy, gam in ((0.35, 0.5, 1.0), (0.65, 0.5, -1.0)):
        dx, dy = x - cx, y - cy
        f = gam * np.exp(-(dx * dx + dy * dy) / (0.06 * 0.06)) / 0.06
        u += -dy * f; v += dx * f
    with AmrSimulation(g, nu=1e-3, cfl=0.5) as s:
        s.install_poisson_matrix(); s.set_math(True); s.set_solver(fused=fused, finish_in_kernel=fused)
        s.set_field(L.VEL, np.stack([u, v], axis=-1))
        errs = []
        for k in range(steps):
            dt = s.compute_dt(); s.adapt(rtol, ctol, lmax); s.set_solver(fused=fused, finish_in_kernel=fused)
            r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters, dt=dt); errs.append(r["err"])
        return s.grid.blocks.copy(), s.get_field(L.VEL).copy(), s.get_field(L.PRES).copy(), errs

def cmp(a, b, tag):
    assert np.array_equal(a[0], b[0]), "different leaves"
    print("%-34s max|dv| %.2e  max|dp| %.2e   (max|v| %.2f max|p| %.2f)  residuals last step %.1e / %.1e"
          % (tag, np.abs(a[1] - b[1]).max(), np.abs(a[2] - b[2]).max(), np.abs(a[1]).max(), np.abs(a[2]).max(), a[3][-1], b[3][-1]))

A = run(True, 50); B = run(False, 50); cmp(A, B, "fused vs sweeps, 50 iterations")
C = run(True, 400); D = run(False, 400); cmp(C, D, "fused vs sweeps, 400 iterations")
cmp(A, C, "fused: 50 vs 400 iterations")
