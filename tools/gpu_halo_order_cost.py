#!/usr/bin/env python3
"""What does the block order of a decomposed patch cost the solver's sweeps?  (development aid, one GPU)
A 512 x 256-block patch (BASELINE.json configs[3]'s per-rank size) as rank (0, 0) of a 2 x 2 decomposition -- ghost blocks on its
E and N sides, the blocks that touch them ordered last (grid.py) -- driven on ONE process with a transport that moves
nothing (the ghost blocks keep whatever they hold: timing only), against the same patch without ghost sides."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cup2d_amd
from cup2d_amd import lib as L
from cup2d_amd.distributed import PatchTopology
from cup2d_amd.grid import BlockGrid

lib = L.load_library()
vp = ctypes.c_void_p
nbx, nby = int(os.environ.get("NBX", 512)), int(os.environ.get("NBY", 256))
rng = np.random.default_rng(1)


def timed(sim, label):
    b = rng.uniform(-1, 1, (sim.grid.ny, sim.grid.nx)); b -= b.mean()
    sim.tmp = b
    sim.set_solver(fused=True, finish_in_kernel=True)
    for rep in range(2):
        sim.fill(L.PRES, 0.0)
        sim.set_timing(1)
        t0 = time.perf_counter()
        r = sim.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
        el = time.perf_counter() - t0
    out = {}
    for k, name in enumerate(L.TIMER_NAMES):
        ms, n = sim.get_timing(k)
        if n:
            out[name] = (ms / n * 1e3, n)
    print("%-34s solve %.2f ms, form %s; " % (label, el * 1e3, sim.last_solver_form()) +
          "  ".join("%s %.1f us x%d" % (k, v[0], v[1]) for k, v in out.items() if k.startswith("sweep")), flush=True)


with cup2d_amd.Simulation(nbx, nby, grid=BlockGrid(nbx, nby), h=1.0 / 4096) as s:
    timed(s, "no ghost sides")
for order_note, topo in (("halo blocks last (grid.py)", PatchTopology(nbx, nby, 2, 2, 0, 0)),):
    g = topo.grid
    with cup2d_amd.Simulation(nbx, nby, grid=g, h=1.0 / 4096) as s:
        L.check(lib.cup2d_halo_plan(s.ctx, topo.nsend, topo.send_block.ctypes.data_as(vp), topo.send_face.ctypes.data_as(vp),
                                    topo.nrecv, topo.recv_block.ctypes.data_as(vp), topo.recv_face.ctypes.data_as(vp)), "halo_plan")
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        bufs = []
        for nbytes in (max(1, topo.nsend) * 192 * 8, max(1, topo.nrecv) * 192 * 8, 64):
            q = vp()
            assert hip.hipMalloc(ctypes.byref(q), nbytes) == 0
            hip.hipMemset(q, 0, ctypes.c_size_t(nbytes))
            bufs.append(q)
        cb = (L.EXCHANGE_FN(lambda u, a, b_, sd, st: 0), L.WAIT_FN(lambda u, st: 0), L.ALLREDUCE_FN(lambda u, buf, n, op, st: 0))
        L.check(lib.cup2d_set_comm(s.ctx, cb[0], cb[1], cb[2], None, bufs[0], bufs[1], bufs[2]), "set_comm")
        L.check(lib.cup2d_set_comm_strip_capacity(s.ctx, 192), "cap")
        timed(s, "E, N ghost sides, " + order_note)
