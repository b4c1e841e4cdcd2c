#!/usr/bin/env python3
"""The N-rank code path of a whole time step on ONE GPU with bytes moving through RCCL (development aid).
A patch that is its own W and E neighbour (ncclSend / ncclRecv to self on the communication stream, tests/test_comm.py):
ghost blocks on both x sides, the halo set ordered last, k_halo pack / unpack, whole ghost blocks of the Krylov vectors, the
MERGE 2 kernels, two all-gathers + one-wave kernels per iteration, one host look per iteration -- everything a rank of an
N-rank run does except waiting for another GPU.  Against the plain context on the same patch (walls instead of the
periodic link: other numbers, the same work).  NBX, NBY: blocks (default 512 x 512 = 4096^2 cells)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
import cup2d_amd
from cup2d_amd import lib as L
from test_comm import _self_periodic_sim
from oracle import oracle as O

nbx, nby = int(os.environ.get("NBX", 512)), int(os.environ.get("NBY", 512))
iters, steps = int(os.environ.get("ITERS", 50)), int(os.environ.get("STEPS", 8))
vel = O.taylor_green(nbx * 8, noise=1e-3, ny=nby * 8)


def run(sim, label):
    sim.set_math(False)
    sim.vel = vel
    for _ in range(2):
        sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
    sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
    sim.synchronize()
    el = (time.perf_counter() - t0) / steps
    sim.set_timing(2)
    for _ in range(3):
        sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)
    tm = {}
    for k, name in enumerate(L.TIMER_NAMES):
        ms, n = sim.get_timing(k)
        if n:
            tm[name] = "%.1f us x%d" % (ms / n * 1e3, n)
    print("%-26s %.3f ms/step = %.1f Mcell-updates/s  iters %d  form %s\n    %s" % (
        label, el * 1e3, nbx * nby * 64 / el / 1e6, r["iters"], sim.last_solver_form(), tm), flush=True)
    return el


with cup2d_amd.Simulation(nbx, nby, nu=1e-3) as s:
    t_plain = run(s, "plain context")
s, g = _self_periodic_sim(nbx, nby, os.environ.get("AXES", "xy"))  # "xy": ghost blocks on all four sides (an interior rank)
with s:
    s.nu = 1e-3
    print("self-periodic patch: %d blocks, %d ghost blocks, halo set of %d x %d-block patches, n_inner %d" % (g.nblocks, g.nghost, g.halo_tile, g.halo_tile, g.n_inner))
    if os.environ.get("ORG"):  # cup2d_set_nrank_organisation: "deferred,split", e.g. ORG=1,1 = the overlap organisation
        s.set_nrank_organisation(*[int(v) for v in os.environ["ORG"].split(",")])
    t_self = run(s, "self-periodic (RCCL)%s" % (" ORG=" + os.environ["ORG"] if os.environ.get("ORG") else ""))
    L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")
print("N-rank path / plain = %.3f" % (t_self / t_plain))
