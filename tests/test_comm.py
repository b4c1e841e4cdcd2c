"""The communicator inside the library (csrc/comm.hip, cup2d_comm_*).

CPU  : the C ABI validates its arguments and reports a missing GPU / RCCL as CUP2D_ERR_COMM (no crash, no fallback).
GPU  : (a) RCCL send/recv with real buffers, offsets and streams on ONE rank: the rank is its own W and E neighbour
           (ncclSend / ncclRecv to self inside one group), i.e. a domain periodic in x -- the ghost strips must be the
           opposite edge's cells, bit for bit, for the WENO halo (3 layers, 2 components), a scalar width-1 halo and whole
           ghost blocks;
       (b) a one-rank decomposition through the whole communicator path (comm stream, events, all-reduce, the
           sum+max all-gather) gives the step the plain single-GPU context gives, bit for bit.
The N > 1 arithmetic (ghost blocks, overlapped sweeps, reductions) is covered by tests/test_distributed.py with the
callback transport (gloo): two ranks cannot share one GPU under RCCL.
"""
import ctypes

import numpy as np
import pytest

from cup2d_amd import lib as L
from cup2d_amd.grid import BlockGrid


def test_comm_argument_checks_without_a_gpu():
    lib = L.load_library()
    vp = ctypes.c_void_p
    assert lib.cup2d_comm_unique_id(None) == -1  # CUP2D_ERR_ARG
    ids = ctypes.create_string_buffer(L.COMM_ID_BYTES)
    one = (ctypes.c_int32 * 1)(0)
    assert lib.cup2d_comm_init(None, 1, 0, ids, 0, None, None, None, None, None) == -1  # null context
    assert b"null context" in lib.cup2d_last_error()
    assert lib.cup2d_comm_finalize(None) == -1
    assert lib.cup2d_comm_selftest(None, 1.0, None, 0) == -1
    assert lib.cup2d_halo_exchange(None, L.VEL, 3) == -1
    assert lib.cup2d_comm_stats(None, None, None, None, None, None) == -1
    assert lib.cup2d_halo_plan_cells(None, 0, 0, None, 0, None) == -1 and lib.cup2d_comm_set_cell_counts(None, 0, 0, None, None, None, None) == -1
    assert lib.cup2d_set_nrank_organisation(None, 1, 0) == -1 and b"null context" in lib.cup2d_last_error()
    # the trace of the kernels' ghost reads is host code: bad tables, sets and readers are refused
    import numpy as np
    k, n2, h = np.zeros((2, 4), dtype=np.int32), -np.ones((2, 4, 2), dtype=np.int32), np.zeros((2, 4), dtype=np.int32)
    mask, rd = np.zeros(2, dtype=np.uint64), np.asarray([0, 5], dtype=np.int32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.cup2d_amr_trace_reads(2, P(k), P(n2), P(h), 1, P(rd), 3, P(mask)) == -1   # no such set
    assert lib.cup2d_amr_trace_reads(2, P(k), P(n2), P(h), 2, P(rd), 0, P(mask)) == -1 and b"readers[1]" in lib.cup2d_last_error()
    assert lib.cup2d_amr_trace_reads(2, P(k), P(n2), P(h), 1, P(rd), 0, P(mask)) == 0 and not mask.any()  # walls all round: nothing read
    k[0, 1] = L.AMR_SAME  # a neighbour entry that is missing from the tables: reported, not dereferenced
    assert lib.cup2d_amr_trace_reads(2, P(k), P(n2), P(h), 1, P(rd), 0, P(mask)) == -1 and b"outside the tables" in lib.cup2d_last_error()
    del vp, one


def _hip():
    h = ctypes.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return h


def _self_periodic_sim(nbx, nby, axes="x"):
    """one rank that is its own W and E (axes "y": S and N; "xy": all four) neighbour (cup2d_amd.distributed.self_periodic_simulation)"""
    from cup2d_amd.distributed import self_periodic_simulation
    return self_periodic_simulation(nbx, nby, axes=axes)


@pytest.mark.gpu
@pytest.mark.parametrize("axes", ["x", "y", "xy"])
def test_rccl_send_recv_to_self_fills_the_ghost_strips(gpu_lib, axes):
    """axes "xy": ghost blocks on all four sides (what an interior rank of a decomposition has; the ranks of the 2 x 4 layout of
    BASELINE.json configs[3] have two or three): four send/recv pairs in one ncclGroup"""
    from cup2d_amd.distributed import strip_cells, OPPOSITE
    import os
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap socket on loopback
    nbx, nby = 6, 5
    s, g = _self_periodic_sim(nbx, nby, axes)
    npeers = 2 * len(axes)
    hip = _hip()
    rng = np.random.default_rng(3)
    with s:
        # the checked round the N-rank callers run right after cup2d_comm_init (strips between all peers, all-gather, all-reduce)
        info = ctypes.create_string_buffer(512)
        L.check(s.L.cup2d_comm_selftest(s.ctx, 20.0, info, len(info)), "comm_selftest")
        rep = dict(t.partition("=")[::2] for t in info.value.decode().split())
        assert rep["ranks"] == "1" and rep["rank"] == "0" and rep["peers"] == ",".join("0" * npeers) and "librccl" in rep["rccl"], rep
        for field, dim, width in ((L.VEL, 2, 3), (L.PRES, 1, 1), (L.TMP, 1, 8)):
            a = rng.uniform(-1, 1, (g.ny, g.nx, dim) if dim > 1 else (g.ny, g.nx))
            s.set_field(field, a)
            L.check(s.L.cup2d_halo_exchange(s.ctx, field, width), "halo_exchange")
            s.synchronize()
            slab = np.zeros((g.nblocks + g.nghost, 64, dim))
            assert hip.hipMemcpy(slab.ctypes.data, s.field_ptr(field), slab.nbytes, 2) == 0
            owned = g.to_blocks(a).reshape(g.nblocks, 64, dim)
            assert np.array_equal(slab[:g.nblocks], owned)
            for gi, (side, pos) in enumerate(g.ghost_coords):
                # periodic: the opposite edge's block
                src = owned[g.index_of[pos, nbx - 1 if side == 0 else 0] if side < 2 else g.index_of[nby - 1 if side == 2 else 0, pos]]
                cells = strip_cells(OPPOSITE[side], width)
                assert np.array_equal(slab[g.nblocks + gi, cells], src[cells]), (field, side, pos)
        st = {}
        n, p, e, ar, ag = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        L.check(s.L.cup2d_comm_stats(s.ctx, ctypes.byref(n), ctypes.byref(p), ctypes.byref(e), ctypes.byref(ar), ctypes.byref(ag)), "stats")
        st = dict(nranks=n.value, peers=p.value, exchanges=e.value)
        assert st == dict(nranks=1, peers=npeers, exchanges=3)
        # a new halo plan under a live in-library communicator is refused (its offsets, cell counts and in-place receive targets
        # were derived from the plan it was initialised on): finalize first
        one = np.zeros(1, dtype=np.int32)
        gid = np.asarray([g.nblocks], dtype=np.int32)
        vp = ctypes.c_void_p
        assert s.L.cup2d_halo_plan(s.ctx, 1, one.ctypes.data_as(vp), one.ctypes.data_as(vp), 1, gid.ctypes.data_as(vp), one.ctypes.data_as(vp)) == -1
        assert b"cup2d_comm_finalize first" in s.L.cup2d_last_error()
        L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")
        L.check(s.L.cup2d_halo_plan(s.ctx, 1, one.ctypes.data_as(vp), one.ctypes.data_as(vp), 1, gid.ctypes.data_as(vp), one.ctypes.data_as(vp)), "halo_plan")


@pytest.mark.gpu
@pytest.mark.parametrize("nbx,nby", [(16, 16), (64, 32)])
def test_doubly_periodic_patch_equals_the_oracle_on_the_wrapped_field(gpu_lib, oracle, nbx, nby):
    """A patch with ghost blocks on all FOUR sides, each filled from the opposite edge through ncclSend / ncclRecv to self (a
    doubly periodic domain): the halo set meets in four corners, four peers share one ncclGroup, the in-place receive fills four
    consecutive ghost ranges -- the shape of an interior rank, of which the 2 x 4 layout's ranks have two or three sides.  The
    oracle (walls) runs on the field wrapped by two blocks on every side; its centre is the periodic result: RK2 WENO5
    advect-diffuse (halo 3, inner / halo phases), the Poisson right-hand side, two Jacobi sweeps -- STRICT, bit for bit
    (main.cpp:5441-5503, 6105-6139, 6209-6230)."""
    import os
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    O = oracle
    s, g = _self_periodic_sim(nbx, nby, "xy")
    nx, ny, pad = g.nx, g.ny, 16
    wrap = lambda a: np.ascontiguousarray(np.pad(a, ((pad, pad), (pad, pad)) + ((0, 0),) * (a.ndim - 2), mode="wrap"))
    mid = (slice(pad, pad + ny), slice(pad, pad + nx))
    rng = np.random.default_rng(11)
    xx, yy = np.meshgrid((np.arange(nx) + 0.5) / nx, (np.arange(ny) + 0.5) / ny, indexing="xy")
    vel = np.stack([np.sin(2 * np.pi * xx) * np.cos(2 * np.pi * yy), -np.cos(2 * np.pi * xx) * np.sin(2 * np.pi * yy)], -1)
    vel += 0.05 * rng.uniform(-1, 1, vel.shape)
    pres = rng.uniform(-1, 1, (ny, nx))
    with s:
        assert g.nghost == 2 * (nbx + nby) and g.n_inner < g.nblocks
        h, nu = s.h, 1e-3
        s.set_math(True)
        s.vel = vel
        assert s.max_abs_vel() == np.abs(vel).max()
        dt = s.compute_dt()
        s.advect_diffuse_rk2(dt)
        ref = O.rk2_advect_diffuse(wrap(vel), h, nu, dt)[0][mid]
        assert np.array_equal(s.vel, ref), "rk2 on the doubly periodic patch"
        s.set_math(False)
        s.vel = vel
        s.advect_diffuse_rk2(dt)  # FAST: the quad walk on the inner and the halo plan
        assert np.abs(s.vel - ref).max() <= 2e-13 * np.abs(ref).max()
        s.set_math(True)
        s.vel = ref
        s.pres = pres
        s.poisson_rhs(dt)
        b = O.laplacian_sub(wrap(pres), O.pressure_rhs(wrap(ref), h, dt))[mid]
        assert np.array_equal(s.tmp, b) and np.array_equal(s.pold, pres) and not s.pres.any()
        s.pres = pres
        s.jacobi_sweeps(2, omega=0.8)
        xj, _ = O.jacobi_sweeps(wrap(pres), wrap(b), 0.8, 2)
        assert np.array_equal(s.pres, xj[mid]), "jacobi sweeps on the doubly periodic patch"
        L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")


@pytest.mark.gpu
def test_cell_plan_through_rccl_to_self(gpu_lib):
    """cup2d_halo_plan_cells + cup2d_comm_set_cell_counts with real ncclSend / ncclRecv (the rank is its own W and E neighbour):
    the patch is given topology tables (cup2d_set_amr: every block on level 0, the x sides of the edge blocks same-level
    neighbours of the ghost blocks), so its operators run the adapted-grid kernels and refresh their ghost blocks through the
    cell plan of the halo-1 family -- the 8 edge cells of a ghost block travel, the other 56 stay NaN -- and Lap(p) of the
    x-periodic field comes out bit for bit"""
    import os
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    nbx, nby = 6, 5
    s, g = _self_periodic_sim(nbx, nby)
    hip = _hip()
    vp = ctypes.c_void_p
    nb, ng = g.nblocks, g.nghost
    with s:
        kind = np.zeros((nb + ng, 4), dtype=np.int32)                      # ghost blocks: sides nobody reads (walls)
        nbr2 = -np.ones((nb + ng, 4, 2), dtype=np.int32)
        kind[:nb] = np.where(g.nbr >= 0, L.AMR_SAME, L.AMR_WALL)
        nbr2[:nb, :, 0] = g.nbr
        level, half = np.zeros(nb + ng, dtype=np.int32), np.zeros((nb + ng, 4), dtype=np.int32)
        L.check(s.L.cup2d_set_amr(s.ctx, s.h, level.ctypes.data_as(vp), kind.ctypes.data_as(vp), nbr2.ctypes.data_as(vp),
                                  half.ctypes.data_as(vp)), "set_amr")
        # what the kernels read of the ghost blocks, from the library's own trace: the column that touches the patch
        mask = np.zeros(nb + ng, dtype=np.uint64)
        readers = np.arange(nb, dtype=np.int32)
        L.check(s.L.cup2d_amr_trace_reads(nb + ng, kind.ctypes.data_as(vp), nbr2.ctypes.data_as(vp), half.ctypes.data_as(vp), nb,
                                          readers.ctypes.data_as(vp), L.CELLS_HALO1, mask.ctypes.data_as(vp)), "trace")
        col = lambda x: sum(1 << (8 * y + x) for y in range(8))
        for gi, (side, pos) in enumerate(g.ghost_coords):  # the W ghost is read at its east column, the E ghost at its west one
            assert int(mask[nb + gi]) == col(7 if side == 0 else 0)
        # send list in the order of the block plan (W edge blocks, then E edge blocks); receive list: W ghosts, then E ghosts
        send = [64 * int(g.index_of[pos, 0 if side == 0 else nbx - 1]) + 8 * y + (0 if side == 0 else 7)
                for side in (0, 1) for pos in range(nby) for y in range(8)]
        recv = [64 * int(g._ghost_id[(side, pos)]) + 8 * y + (7 if side == 0 else 0) for side in (0, 1) for pos in range(nby) for y in range(8)]
        sc, rc = np.asarray(send, dtype=np.int32), np.asarray(recv, dtype=np.int32)
        L.check(s.L.cup2d_halo_plan_cells(s.ctx, L.CELLS_HALO1, len(sc), sc.ctypes.data_as(vp), len(rc), rc.ctypes.data_as(vp)), "halo_plan_cells")
        n8 = 8 * nby
        soff, roff, cnt = (np.asarray(a, dtype=np.int32) for a in ([n8, 0], [0, n8], [n8, n8]))  # as the block plan pairs them
        assert s.L.cup2d_comm_set_cell_counts(s.ctx, L.CELLS_HALO1, 1, soff.ctypes.data_as(vp), cnt.ctypes.data_as(vp), roff.ctypes.data_as(vp),
                                              cnt.ctypes.data_as(vp)) == -1  # one entry per peer of cup2d_comm_init: two here
        L.check(s.L.cup2d_comm_set_cell_counts(s.ctx, L.CELLS_HALO1, 2, soff.ctypes.data_as(vp), cnt.ctypes.data_as(vp), roff.ctypes.data_as(vp),
                                               cnt.ctypes.data_as(vp)), "comm_set_cell_counts")
        rng = np.random.default_rng(5)
        p, t = rng.uniform(-1, 1, (g.ny, g.nx)), rng.uniform(-1, 1, (g.ny, g.nx))
        s.set_field(L.POLD, p)
        s.set_field(L.TMP, t)
        nan = np.full((ng, 64), np.nan)  # ghost copies of POLD start as NaN: only what the plan delivers can be a number
        base = ctypes.c_void_p(s.field_ptr(L.POLD) + nb * 64 * 8)
        assert hip.hipMemcpy(base, nan.ctypes.data, nan.nbytes, 1) == 0
        s.laplacian_sub()
        W, E = np.roll(p, 1, axis=1), np.roll(p, -1, axis=1)                 # periodic in x
        S, N = np.vstack([p[:1], p[:-1]]), np.vstack([p[1:], p[-1:]])          # walls in y: ghost = edge cell
        assert np.array_equal(s.get_field(L.TMP), t - ((((W + E) + S) + N) - 4 * p))
        slab = np.zeros((nb + ng, 64))
        assert hip.hipMemcpy(slab.ctypes.data, s.field_ptr(L.POLD), slab.nbytes, 2) == 0
        got = np.isfinite(slab[nb:])
        assert got.sum() == 8 * ng and all(got[gi].reshape(8, 8)[:, 7 if side == 0 else 0].all() for gi, (side, pos) in enumerate(g.ghost_coords))
        L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_one_rank_through_the_communicator_equals_the_plain_context(gpu_lib, oracle, fused, tmp_path):
    """world = 1 under torch.distributed (gloo carries the token): every reduction goes through ncclAllReduce /
    ncclAllGather on the compute stream, the solver runs its N-rank organisation (MERGE 2 + scalar kernels) -- bit for bit
    the plain context's numbers (same launches, same order of every sum)."""
    import os
    import torch.distributed as dist
    import cup2d_amd
    from cup2d_amd.distributed import DistributedSimulation
    n = 256
    vel = oracle.taylor_green(n)
    with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
        s.set_solver(fused=fused, finish_in_kernel=True)
        s.vel = vel
        r0 = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=30)
        v0, p0 = s.vel, s.pres
    # loopback everywhere: gloo and RCCL's bootstrap otherwise resolve the box's hostname, which can take minutes to fail
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + int(fused)), RANK="0", WORLD_SIZE="1",
                      GLOO_SOCKET_IFNAME="lo", NCCL_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with DistributedSimulation(n // 8, n // 8, 1, 1, nu=1e-3, comm="rccl") as d:
            d.set_solver(fused=fused, finish_in_kernel=True)
            d.vel = vel
            r1 = d.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=30)
            st = d.comm_stats()
            assert r1["dt"] == r0["dt"] and r1["iters"] == r0["iters"] == 30
            assert np.array_equal(d.vel, v0) and np.array_equal(d.pres, p0)
            assert r1["err"] == r0["err"]
            # per iteration one all-gather + one scalar kernel per reduction point: two in the default organisation of the
            # fused solver (after C+D, and after the launch that holds sweep E and the next A+B), three in the five sweeps
            assert st["nranks"] == 1 and st["peers"] == 0 and st["allgathers"] >= (60 if fused else 90)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_one_rank_amr_through_the_communicator_equals_the_plain_context(gpu_lib, oracle, tmp_path):
    """an adapted grid on 'one of one' ranks with the in-library communicator: no ghost blocks, but every reduction of the
    AMR step (max|u|, the volume-weighted pressure means, the solver's sums and norms on the assembled operator) goes through
    RCCL -- the step equals the plain context's bit for bit"""
    import os
    import torch.distributed as dist
    from conftest import golden
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    from cup2d_amd.amr_dist import DistributedAmrSimulation
    F = golden("amr_functors.npz")
    G = AmrBlockGrid(F["blocks"])
    with AmrSimulation(G, nu=float(F["nu"])) as ref:
        ref.set_math(True)
        ref.set_field(L.VEL, F["vel"])
        ref.install_poisson_matrix()
        r0 = ref.step(tol=1e-10, rel_tol=0.0, max_restarts=100, max_iter=300)
        v0, p0 = ref.get_field(L.VEL), ref.get_field(L.PRES)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", GLOO_SOCKET_IFNAME="lo",
                      NCCL_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with DistributedAmrSimulation(G, nu=float(F["nu"]), comm="rccl") as s:
            assert s.part.nghost == 0 and s.part.nowned == G.nblocks
            s.set_math(True)
            s.set_field(L.VEL, F["vel"])
            s.install_poisson_matrix()
            r1 = s.step(tol=1e-10, rel_tol=0.0, max_restarts=100, max_iter=300)
            assert r1["dt"] == r0["dt"] and r1["iters"] == r0["iters"] and r1["err"] == r0["err"]
            assert np.array_equal(s.get_field(L.VEL), v0) and np.array_equal(s.get_field(L.PRES), p0)
    finally:
        dist.destroy_process_group()


_DIRECT_CHILD = r'''
import sys, os, json, ctypes, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
from cup2d_amd import lib as L
from test_comm import _self_periodic_sim
out = {}
# (512, 256, "xy"): the per-rank patch of BASELINE.json configs[3] (4096 x 2048 cells) with ghost blocks on all four sides
# (the big patch runs under the default and under round 4's organisation; the other transports are compared on the small ones)
for nbx, nby, axes in ((32, 16, "x"), (64, 64, "x"), (32, 32, "xy")) + (((512, 256, "xy"),) if os.environ.get("CUP2D_TEST_BIG_PATCH", "1") == "1" else ()):
    s, g = _self_periodic_sim(nbx, nby, axes)
    with s:
        rng = np.random.default_rng(7)
        b = rng.uniform(-1, 1, (g.ny, g.nx)); b -= b.mean()
        s.set_solver(fused=True, finish_in_kernel=True)
        s.keep_last_iterate(True)
        s.tmp = b
        s.fill(L.PRES, 0.0)
        r = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=6)
        e = s.last_iterate_to(L.POLD)
        x = s.pold
        s.fill(L.PRES, 0.0)
        cap = 2000 if nbx * nby <= 4096 else 300  # (the big patch: 300 iterations of the same solve, converged or not)
        conv = s.poisson_solve(tol=1e-8, max_restarts=100, max_iter=cap)
        n, p, ex, ar, ag = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        L.check(s.L.cup2d_comm_stats(s.ctx, ctypes.byref(n), ctypes.byref(p), ctypes.byref(ex), ctypes.byref(ar), ctypes.byref(ag)), "stats")
        form = list(s.last_solver_form())
        out_form_overlap = None
        # the same six iterations with the organisation chosen through the API (cup2d_set_nrank_organisation): round 4's reduction
        # points, then the split sweeps -- the first the same bits, the second to round-off (other order of the partial sums)
        split_x = None
        for dfr, spl in ((0, 0), (0, 1), (1, 1)):
            s.set_nrank_organisation(dfr, spl)
            s.fill(L.PRES, 0.0)
            ro = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=6)
            s.last_iterate_to(L.POLD)
            xo = s.pold
            fo = s.last_solver_form()
            assert ro["iters"] == 6 and fo[0] == "eab" and fo[1] in (2, 3, 4), (dfr, spl, fo)
            if spl == 0:
                assert np.array_equal(xo, x), "organisation (0, 0) differs from the default"
            else:
                assert np.abs(xo - x).max() <= 1e-11 * np.abs(x).max(), float(np.abs(xo - x).max())
                # "overlap" (split sweeps + deferred update, merge 4 where the communicator allows it) = the split sweeps of round 4
                if split_x is not None:
                    assert np.array_equal(xo, split_x), "organisation (1, 1) differs from (0, 1)"
                    out_form_overlap = list(fo)
                split_x = xo
            # a converged solve under this organisation as well (the end of the solve is learnt a launch later when deferred)
            if nbx * nby <= 1024:
                s.fill(L.PRES, 0.0)
                rc_ = s.poisson_solve(tol=1e-8, max_restarts=100, max_iter=cap)
                assert rc_["iters"] == conv["iters"] or spl == 1, (dfr, spl, rc_, conv)
                assert rc_["err"] <= 1e-8 or rc_["iters"] >= cap, (dfr, spl, rc_)
        s.set_nrank_organisation(-1, -1)
        d5 = -1.0
        if axes == "xy":  # eight iterations of the two-launch MERGE 2 organisation against the five sweeps on the same periodic operator
            s.set_precond(L.PRECOND_MFMA)
            xs = []
            for fused in (True, False):
                s.set_solver(fused=fused, finish_in_kernel=True)
                s.fill(L.PRES, 0.0)
                r8 = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=8)
                assert r8["iters"] == 8 and s.last_solver() == ("fused" if fused else "sweeps"), (r8, s.last_solver())
                e8 = s.last_iterate_to(L.POLD)
                xs.append((s.pold, e8))
            d5 = float(np.abs(xs[0][0] - xs[1][0]).max() / np.abs(xs[1][0]).max())
            assert abs(xs[0][1] - xs[1][1]) <= 1e-9 * xs[1][1], (xs[0][1], xs[1][1])
            s.set_precond(L.PRECOND_FD)
        out["%%dx%%d%%s" %% (nbx, nby, axes)] = dict(sum=float(np.abs(x).sum()), hash=int(np.frombuffer(x.tobytes(), dtype=np.uint64).sum() %% (1 << 62)),
                                        iters=r["iters"], err=e, conv_err=conv["err"], conv_iters=conv["iters"], cap=cap, form=form, form_overlap=out_form_overlap, exchanges=ex.value, vs_five_sweeps=d5)
        L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_ghost_blocks_received_in_place_equal_the_generic_exchange(gpu_lib):
    """The Krylov ghost-block exchanges of the in-library communicator run on the compute stream and land in the vectors' ghost
    regions directly (comm.hip comm_exchange_blocks; default) -- against the generic path (pack, second stream, receive buffer,
    unpack; CUP2D_COMM_DIRECT=0) on a patch that is its own W and E neighbour, bytes through ncclSend / ncclRecv both ways: six
    iterations of the two-launch MERGE 2 solver leave the same last iterate bit for bit, a converged solve the same counts.
    The same on patches with ghost blocks on all FOUR sides (doubly periodic; four peers in one group, four ghost ranges received
    in place, halo-set patches meeting in corners) up to BASELINE.json configs[3]'s per-rank size, 512 x 256 blocks -- there also
    eight iterations of MERGE 2 against the five sweeps on the same periodic operator, to 1e-10 of max|x|.
    Likewise r' and p'' of the ghost blocks formed by the receiving rank (k_ghost_rp; default) against the three vectors
    travelling (CUP2D_GHOST_LOCAL=0): the same bits."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (direct, local): the default; the generic transport path; three vectors travelling instead of r' and p'' of the ghost
    # blocks being formed by the receiver (k_ghost_rp)
    # (direct, local, deferred): the default -- the reduction records ride in the send/recv group and the scalar updates happen
    # in the consumer sweeps (k_edge MERGE 3) --; round 4's organisation (all-gather + one-wave kernel per reduction point); ...
    for key in (("1", "1", "1"), ("1", "1", "0"), ("0", "1", "1"), ("1", "0", "1"), ("0", "0", "1")):
        env = dict(os.environ, CUP2D_COMM_DIRECT=key[0], CUP2D_GHOST_LOCAL=key[1], CUP2D_DEFER_SCALARS=key[2], NCCL_SOCKET_IFNAME="lo",
                   CUP2D_TEST_BIG_PATCH="1" if key[:2] == ("1", "1") else "0")
        r = subprocess.run([sys.executable, "-c", _DIRECT_CHILD % (root, root)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and lines, r.stdout.decode()[-3000:]
        res[key] = json.loads(lines[0][7:])
    ref = res[("1", "1", "1")]
    for key, other in res.items():
        for k in other:
            a, b = ref[k], other[k]
            # the organisation that ran: two launches per iteration; MERGE 3 (deferred scalar updates) only in the default
            assert a["form"] == ["eab", 3, a["form"][2]] and b["form"] == ["eab", 3 if key == ("1", "1", "1") else 2, a["form"][2]], (key, a, b)
            assert a["iters"] == b["iters"] == 6, (key, a, b)
            # (1, 1) through the API is the overlap organisation (merge 4) where the transport allows the deferred update and
            # the patch splits (halo set a whole number of tiles), else round 4's split sweeps (merge 2)
            # (a patch whose halo set is not a whole number of tiles does not split: the deferred unsplit form, merge 3)
            assert b["form_overlap"][1] in ((2, 3, 4) if key == ("1", "1", "1") else (2,)), (key, k, b["form_overlap"])
            if key == ("1", "1", "1") and k.startswith("512x256"):
                assert b["form_overlap"][1] == 4, (k, b["form_overlap"])   # configs[3]'s rank size does run the overlap organisation
            assert a["hash"] == b["hash"] and a["sum"] == b["sum"] and a["err"] == b["err"], (key, k, a, b)
            assert a["conv_iters"] == b["conv_iters"] and a["conv_err"] == b["conv_err"], (key, a, b)
            assert a["conv_err"] <= 1e-8 or a["conv_iters"] >= a["cap"], (key, a)
            # (the deferred organisation learns of the end of the converged solve one launch later: up to one dead iteration = two
            # exchanges more than the others)
            assert b["exchanges"] > 12 and 0 <= a["exchanges"] - b["exchanges"] <= 2, (key, k, a["exchanges"], b["exchanges"])
            assert b["vs_five_sweeps"] <= 1e-10, (key, k, b)  # (-1: not run on the x-periodic patches)
