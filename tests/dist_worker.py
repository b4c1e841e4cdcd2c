"""Worker for the multi-process tests (launched by torch.distributed.run).

cpu mode : world_size ranks on the CPU (gloo): exercises PatchTopology + TorchComm("host") -- strips
           packed with a numpy restatement of halo.hip's layout land in the right ghost cells, and the
           all-reduce callbacks reduce what the library would hand them.
gpu mode : world_size ranks sharing cuda:0 (gloo, host-staged): the full DistributedSimulation
           (HIP pack/unpack, ghost blocks, overlapped sweeps, reductions through the callbacks)
           against the CPU oracle on the global grid.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_field(gnx, gny, dim, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (gny, gnx, dim) if dim > 1 else (gny, gnx))


def run_cpu(rank, world, px, py, nbx, nby):
    import torch
    import torch.distributed as dist
    from cup2d_amd.distributed import PatchTopology, TorchComm, strip_cells
    cx, cy = rank % px, rank // px
    topo = PatchTopology(nbx, nby, px, py, cx, cy)
    g = topo.grid
    comm = TorchComm(topo, "host")
    for dim, width in ((2, 3), (1, 1), (1, 8)):  # WENO halo, Krylov strips, whole ghost blocks (Krylov vectors)
        G = global_field(px * nbx * 8, py * nby * 8, dim, seed=100 + dim).reshape(py * nby * 8, px * nbx * 8, dim)
        patch = G[cy * nby * 8:(cy + 1) * nby * 8, cx * nbx * 8:(cx + 1) * nbx * 8]
        slab = np.zeros((g.nblocks + g.nghost, 64, dim))
        slab[:g.nblocks] = g.to_blocks(patch if dim > 1 else patch[..., 0]).reshape(g.nblocks, 64, dim)
        sd = 8 * width * dim
        send = np.zeros(topo.nsend * sd)
        for k in range(topo.nsend):  # numpy restatement of k_halo<PACK>
            cells = strip_cells(int(topo.send_face[k]), width)
            send[k * sd:(k + 1) * sd] = slab[topo.send_block[k], cells].ravel()
        comm.send[:send.size] = torch.from_numpy(send)
        comm.exchange(sd)
        comm.wait()
        recv = comm.recv[:topo.nrecv * sd].numpy()
        for k in range(topo.nrecv):  # k_halo<UNPACK>
            cells = strip_cells(int(topo.recv_face[k]), width)
            slab[topo.recv_block[k], cells] = recv[k * sd:(k + 1) * sd].reshape(len(cells), dim)
        # every ghost strip cell must equal the global field at its geometric position
        checked = 0
        for gi, (side, pos) in enumerate(g.ghost_coords):
            gb = g.nblocks + gi
            bx = cx * nbx + (-1 if side == 0 else nbx if side == 1 else pos)
            by = cy * nby + (pos if side < 2 else (-1 if side == 2 else nby))
            blockG = G[by * 8:(by + 1) * 8, bx * 8:(bx + 1) * 8].reshape(64, dim)
            cells = strip_cells((1, 0, 3, 2)[side], width)
            assert np.array_equal(slab[gb, cells], blockG[cells]), (rank, side, pos)
            checked += len(cells)
        assert checked == topo.nrecv * 8 * width
    # two whole-block vectors in one message (halo.hip k_halo_blocks2): strip = [64 cells of f0 | 64 cells of f1]
    G0 = global_field(px * nbx * 8, py * nby * 8, 1, seed=7)
    G1 = global_field(px * nbx * 8, py * nby * 8, 1, seed=8)
    slabs = []
    for G in (G0, G1):
        sl = np.zeros((g.nblocks + g.nghost, 64))
        sl[:g.nblocks] = g.to_blocks(G[cy * nby * 8:(cy + 1) * nby * 8, cx * nbx * 8:(cx + 1) * nbx * 8]).reshape(g.nblocks, 64)
        slabs.append(sl)
    send = np.zeros(topo.nsend * 128)
    for k in range(topo.nsend):
        cells = strip_cells(int(topo.send_face[k]), 8)
        send[k * 128:k * 128 + 64] = slabs[0][topo.send_block[k], cells]
        send[k * 128 + 64:(k + 1) * 128] = slabs[1][topo.send_block[k], cells]
    comm.send[:send.size] = torch.from_numpy(send)
    comm.exchange(128)
    comm.wait()
    recv = comm.recv[:topo.nrecv * 128].numpy()
    for k in range(topo.nrecv):
        cells = strip_cells(int(topo.recv_face[k]), 8)
        slabs[0][topo.recv_block[k], cells] = recv[k * 128:k * 128 + 64]
        slabs[1][topo.recv_block[k], cells] = recv[k * 128 + 64:(k + 1) * 128]
    for gi, (side, pos) in enumerate(g.ghost_coords):
        bx = cx * nbx + (-1 if side == 0 else nbx if side == 1 else pos)
        by = cy * nby + (pos if side < 2 else (-1 if side == 2 else nby))
        for sl, G in zip(slabs, (G0, G1)):
            assert np.array_equal(sl[g.nblocks + gi], G[by * 8:(by + 1) * 8, bx * 8:(bx + 1) * 8].reshape(64)), (rank, side, pos)
    # reductions as the library drives them: sum of 2 at offset 0, max of 1 at offset 2
    comm.red[:] = torch.tensor([rank + 1.0, 2.0 * rank, -5.0 + rank, 0, 0, 0, 0, 0], dtype=torch.float64)
    comm.allreduce(0, 2, 0)
    comm.allreduce(2, 1, 1)
    assert comm.red[0].item() == world * (world + 1) / 2 and comm.red[1].item() == world * (world - 1)
    assert comm.red[2].item() == -5.0 + world - 1
    dist.barrier()


def run_gpu(rank, world, px, py, nbx, nby):
    import torch.distributed as dist
    from cup2d_amd.distributed import DistributedSimulation
    from cup2d_amd import lib as L
    from oracle import oracle as O
    cx, cy = rank % px, rank // px
    gnx, gny = px * nbx * 8, py * nby * 8
    nu = 1e-3
    vel = O.taylor_green(gnx, noise=0.05, seed=77, ny=gny)
    h = 1.0 / max(gnx, gny)
    sl = (slice(cy * nby * 8, (cy + 1) * nby * 8), slice(cx * nbx * 8, (cx + 1) * nbx * 8))
    sim = DistributedSimulation(nbx, nby, px, py, nu=nu, device=0)
    assert sim.h == h
    sim.set_math(True)
    sim.vel = vel[sl]
    umax = sim.max_abs_vel()
    assert umax == np.abs(vel).max()
    dt = sim.compute_dt()
    assert dt == O.compute_dt(h, nu, 0.5, umax)
    # advect-diffuse RK2 with halo-3 exchanges and the inner/halo split: bit-identical to the global oracle
    sim.advect_diffuse_rk2(dt)
    ref, _ = O.rk2_advect_diffuse(vel, h, nu, dt)
    assert np.array_equal(sim.vel, ref[sl]), "rk2 mismatch on rank %d" % rank
    # the same with the FAST policy (bench.py's): the quad kernel on the inner and the halo phase, ghost blocks as
    # the surroundings of quads, leftovers on the per-block kernel -- round-off apart the same numbers
    sim.set_math(False)
    sim.vel = vel[sl]
    sim.advect_diffuse_rk2(dt)
    assert np.abs(sim.vel - ref[sl]).max() <= 1e-13 * np.abs(ref).max(), "FAST rk2 mismatch on rank %d" % rank
    sim.set_math(True)
    sim.vel = ref[sl]
    # Poisson rhs (halo-1 exchanges of vel and pold)
    rng = np.random.default_rng(5)
    pres = rng.uniform(-1, 1, (gny, gnx))
    sim.pres = pres[sl]
    sim.poisson_rhs(dt)
    bref = O.laplacian_sub(pres, O.pressure_rhs(ref, h, dt))
    assert np.array_equal(sim.tmp, bref[sl]), "poisson rhs mismatch"
    # smoother sweeps and residual (csrc/smoother.hip): a width-1 exchange per sweep, max norm through the all-reduce
    # callback; bit-identical to the global oracle
    x0 = rng.uniform(-1, 1, (gny, gnx))
    sim.pres = x0[sl]
    e = sim.jacobi_sweeps(3, omega=0.8)
    xj, ej = O.jacobi_sweeps(x0, bref, 0.8, 3)
    assert np.array_equal(sim.pres, xj[sl]) and e == ej, "jacobi mismatch on rank %d" % rank
    assert sim.poisson_residual() == O.poisson_residual(xj, bref)[1]
    sim.pold = pres[sl]
    sim.fill(L.PRES, 0.0)
    # solve: reductions through the all-reduce callback, Krylov halos overlapped
    # (eight ranks time-slice ONE GPU here and every exchange needs all of them: an iteration costs tens of milliseconds, so the
    # 2 x 4 layout gets one solve to 1e-7 -- a few hundred iterations --; the second organisation and the whole steps below are
    # world 2 / 4's, where the same code runs)
    tol = 1e-9 if world < 8 else 1e-7
    info = sim.poisson_solve(tol=tol, rel_tol=0.0, max_restarts=100)
    # the default (tile-fused) solver with ghost blocks: z edges of the boundary blocks exchanged per sweep
    assert sim.last_solver() == "fused"
    xo, io = O.bicgstab(bref, tol=tol, max_restarts=100)
    # BiCGSTAB's iteration count is chaotic in the round-off of its dot products (the decomposition
    # changes their summation order; the reference's cuBLAS order is itself unspecified): demand the
    # same convergence, not the same count
    # (8 ranks, 487 iterations in the oracle, 642 here with one restart more: a restart throws the Krylov space away, counts move
    # by whole restart cycles -- what is demanded is the solution, checked below against the global operator)
    # ... which is relaxed for the time-sliced 8-rank layout only: on 2 / 4 ranks the counts agree to a quarter
    slack = 2.0 if world >= 8 else 1.25
    assert info["iters"] <= slack * io["iters"] + 5 and io["iters"] <= slack * info["iters"] + 5, (info, io)
    assert info["err"] <= tol
    gathered = [None] * world
    dist.all_gather_object(gathered, (cx, cy, sim.pres))
    X = np.zeros((gny, gnx))
    for (ax, ay, xl) in gathered:
        X[ay * nby * 8:(ay + 1) * nby * 8, ax * nbx * 8:(ax + 1) * nbx * 8] = xl
    assert np.abs(bref - O.apply_A(X)).max() <= 1.05 * tol  # x = x0 + P_inv y: recurrence vs true residual differ by round-off
    # projection: global mean removal via all-reduce + halo-1 exchange of pres
    sim.project(dt)
    pnew = O.pressure_update(X, pres, h)
    vnew = O.add_scaled(ref, O.pressure_correction(pnew, h, dt), h)
    assert np.abs(sim.pres - pnew[sl]).max() < 1e-9
    assert np.abs(sim.vel - vnew[sl]).max() < 1e-9
    if world >= 8:
        assert not sim.comm_errors, sim.comm_errors
        dist.barrier()
        sim.close()
        return
    # whole steps
    sim.vel = vel[sl]
    sim.fill(L.PRES, 0.0)
    v, p = vel.copy(), np.zeros((gny, gnx))
    for _ in range(2):
        r = sim.step(tol=1e-9, rel_tol=0.0, max_restarts=100)
        v, p, dt2, _i = O.step(v, p, h, nu, 0.5, tol=1e-9, max_restarts=100)
        assert abs(r["dt"] - dt2) < 1e-7 * dt2  # step 2's dt follows a projection solved to 1e-9
    assert np.abs(sim.vel - v[sl]).max() < 1e-7
    # the five-sweep organisation on the same decomposition
    sim.set_solver(fused=False, finish_in_kernel=False)
    sim.tmp = bref[sl]
    sim.fill(L.PRES, 0.0)
    info5 = sim.poisson_solve(tol=1e-9, rel_tol=0.0, max_restarts=100)
    assert sim.last_solver() == "sweeps" and info5["err"] <= 1e-9
    assert info5["iters"] <= 1.25 * io["iters"] + 5 and io["iters"] <= 1.25 * info5["iters"] + 5, (info5, io)
    assert not sim.comm_errors, sim.comm_errors
    dist.barrier()
    sim.close()


def run_gpu_big(rank, world, px, py, nbx, nby):
    """The N-rank path at the per-rank size of BASELINE.json configs[3] (8192^2 over 2 x 4: 4096 x 2048 cells = 512 x 256 blocks
    per rank + a ghost ring), ranks sharing the one GPU (gloo, host-staged).  The reference is the single context on the whole
    (px nbx) x (py nby) block grid on the same GPU -- itself pinned bit for bit to the live reference at 4096^2
    (tests/test_baseline_sizes_gpu.py).  What 16 x 16-block patches cannot reach: 131 072-block neighbour tables with ghost
    ids, the inner / halo quad plans of k_advect_walk, the MERGE 2 instances of k_edge over thousands of rounds, three-block
    strips for r', p'', nu'', the moved last tile of the tile kernels on a patch whose halo blocks are ordered last."""
    import torch.distributed as dist
    import cup2d_amd
    from cup2d_amd.distributed import DistributedSimulation
    from cup2d_amd import lib as L
    from oracle import oracle as O
    cx, cy = rank % px, rank // px
    gbx, gby = px * nbx, py * nby
    gnx, gny = gbx * 8, gby * 8
    nu = 1e-3
    vel = O.taylor_green(gnx, noise=1e-3, seed=20250117, ny=gny)
    sl = (slice(cy * nby * 8, (cy + 1) * nby * 8), slice(cx * nbx * 8, (cx + 1) * nbx * 8))
    rng = np.random.default_rng(5)
    x = (np.arange(gnx) + 0.5) / max(gnx, gny)
    y = (np.arange(gny) + 0.5) / max(gnx, gny)
    X, Y = np.meshgrid(x, y, indexing="xy")
    pres = np.cos(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 1e-2 * rng.uniform(-1, 1, (gny, gnx))
    del X, Y
    ref = cup2d_amd.Simulation(gbx, gby, nu=nu)
    sim = DistributedSimulation(nbx, nby, px, py, nu=nu, device=0)
    assert sim.h == ref.h and sim.grid.nblocks == nbx * nby and sim.grid.nghost > 0
    # ---- (a) every functor of the step, STRICT: bit for bit; FAST: round-off ----
    for s_, v in ((ref, vel), (sim, vel[sl])):
        s_.set_math(True)
        s_.vel = v
    assert sim.max_abs_vel() == ref.max_abs_vel()
    dt = ref.compute_dt()
    assert sim.compute_dt() == dt
    ref.advect_diffuse_rk2(dt)
    sim.advect_diffuse_rk2(dt)      # halo-3 exchange per stage, inner blocks while the strips travel, then the halo blocks
    vadv = ref.vel
    assert np.array_equal(sim.vel, vadv[sl]), "STRICT rk2 differs on rank %d" % rank
    sim.set_math(False)
    sim.vel = vel[sl]
    sim.advect_diffuse_rk2(dt)      # FAST: k_advect_walk on the inner and the halo plan, leftovers on the per-block kernel
    e = np.abs(sim.vel - vadv[sl]).max() / np.abs(vadv).max()
    assert e <= 2e-13, ("FAST rk2", rank, e)
    for s_, v, p in ((ref, vadv, pres), (sim, vadv[sl], pres[sl])):
        s_.set_math(True)
        s_.vel = v
        s_.pres = p
        s_.poisson_rhs(dt)
    b = ref.tmp
    assert np.array_equal(sim.tmp, b[sl]), "poisson rhs differs on rank %d" % rank
    assert np.array_equal(sim.pold, pres[sl]) and not sim.pres.any()
    # tile kernels (16-block tiles, moved last tile; a width-1 exchange per sweep, the max norm over the ranks)
    for s_, p in ((ref, pres), (sim, pres[sl])):
        s_.pres = p
    eg, el = ref.jacobi_sweeps(2, omega=0.8), sim.jacobi_sweeps(2, omega=0.8)
    assert el == eg and np.array_equal(sim.pres, ref.pres[sl]), "jacobi sweeps differ on rank %d" % rank
    assert sim.poisson_residual() == ref.poisson_residual() and np.array_equal(sim.pold, ref.pold[sl])
    # projection: two volume-weighted mean removals over the ranks (summation order is the only freedom), gradient + update
    for s_, v, p in ((ref, vadv, pres), (sim, vadv[sl], pres[sl])):
        s_.vel = v
        s_.pres = p
        s_.fill(L.POLD, 0.0)
        s_.project(dt)
    assert np.abs(sim.pres - ref.pres[sl]).max() <= 1e-12
    assert np.abs(sim.vel - ref.vel[sl]).max() <= 1e-12 * np.abs(vadv).max()
    # ---- (b) eight iterations of the default N-rank organisation = the five sweeps on one context ----
    ref.set_precond(L.PRECOND_MFMA)
    sim.set_precond(L.PRECOND_MFMA)
    out = {}
    for name, s_, rhs_, fused in (("five", ref, b, False), ("two", ref, b, True), ("nrank", sim, b[sl], True)):
        s_.set_solver(fused=fused, finish_in_kernel=True)
        s_.keep_last_iterate(True)
        s_.tmp = rhs_
        s_.fill(L.PRES, 0.0)
        info = s_.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=8)
        assert info["iters"] == 8, (name, info)
        err_last = s_.last_iterate_to(L.POLD)
        out[name] = (s_.pold, info, err_last, s_.last_solver(), s_.last_solver_form())
    assert out["five"][3] == "sweeps" and out["two"][3] == "fused" and out["nrank"][3] == "fused"
    form, merge, handover = out["nrank"][4]
    assert form == "eab" and merge == 2, out["nrank"][4]     # two launches per iteration, reductions over the ranks
    assert out["two"][4][0] == "eab" and out["two"][4][1] == 1
    x5 = out["five"][0]
    scale = np.abs(x5).max()
    d_two = np.abs(out["two"][0] - x5).max() / scale
    d_n = np.abs(out["nrank"][0] - x5[sl]).max() / scale
    assert d_two <= 1e-10 and d_n <= 1e-10, (rank, d_two, d_n)
    assert abs(out["nrank"][2] - out["five"][2]) <= 1e-9 * out["five"][2], (out["nrank"][2], out["five"][2])
    assert out["nrank"][1]["err_init"] == out["five"][1]["err_init"]
    # and the residual the N-rank recurrence carries is the residual of the iterate it holds (assembled over the ranks)
    gathered = [None] * world
    dist.all_gather_object(gathered, (cx, cy, out["nrank"][0]))
    Xn = np.zeros((gny, gnx))
    for (ax, ay, xl) in gathered:
        Xn[ay * nby * 8:(ay + 1) * nby * 8, ax * nbx * 8:(ax + 1) * nbx * 8] = xl
    true = np.abs(b - O.apply_A(Xn)).max()
    assert abs(true - out["nrank"][2]) <= 1e-6 * true + 1e-9, (true, out["nrank"][2])
    del Xn, gathered
    # ---- a whole step as bench.py runs it (FAST, 50 iterations at zero tolerance, default organisation) ----
    for s_, v in ((ref, vel), (sim, vel[sl])):
        s_.set_precond(L.PRECOND_FD)
        s_.set_solver(fused=True, finish_in_kernel=True)
        s_.keep_last_iterate(False)
        s_.set_math(False)
        s_.vel = v
        s_.fill(L.PRES, 0.0)
        s_.fill(L.POLD, 0.0)
    # (a small grid -- the 2 x 4 layout at 16 x 16 blocks per rank -- is far into convergence after 50 iterations, where the
    # round-off of two summation orders has been amplified to the size of the residual itself (measured: best residuals 2.5e-6
    # and 1.8e-6); and eight ranks time-slicing one GPU pay tens of milliseconds per iteration: there the step is capped at
    # eight iterations, after which two correct organisations still agree to round-off)
    small = nbx * nby < 4096
    cap = 8 if small else 50
    rg = ref.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=cap)
    rl = sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=cap)
    assert rl["dt"] == rg["dt"] == dt and rl["iters"] == rg["iters"] == cap
    pg, vg = ref.pres, ref.vel
    dp = np.abs(sim.pres - pg[sl]).max() / max(np.abs(pg).max(), 1e-300)
    dv = np.abs(sim.vel - vg[sl]).max()
    assert abs(rl["err"] - rg["err"]) <= 1e-6 * rg["err"], (rl, rg)
    assert dp <= 2e-9 and dv <= 1e-9, (rank, dp, dv)
    assert not sim.comm_errors, sim.comm_errors
    if rank == 0:
        print("gpu_big %dx%d ranks of %dx%d blocks: hand-over mask %d; 8 iterations vs five sweeps: one context %.1e, N ranks %.1e of max|x|; "
              "step: dp %.1e of max|p|, dv %.1e" % (px, py, nbx, nby, handover, d_two, d_n, dp, dv), flush=True)
    dist.barrier()
    sim.close()
    ref.close()


def run_amr_gpu(rank, world):
    """BASELINE.json configs[4] on N ranks sharing the one GPU (gloo, host-staged): the three-level golden grid split into
    contiguous Hilbert ranges; every block operator on a rank's owned blocks (ghost blocks refreshed whole, flux-correction
    faces exchanged) equals the reference's own functors bit for bit -- the same golden vectors the single-GPU test uses --
    a time step lands on the single-context step, and a regrid across the ranks reproduces the single-context regrid."""
    import torch.distributed as dist
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    from cup2d_amd.amr_dist import DistributedAmrSimulation
    F = dict(np.load(os.path.join(ROOT, "tests", "golden", "amr_functors.npz")))
    G = AmrBlockGrid(F["blocks"])
    dt = float(F["dt"])
    with DistributedAmrSimulation(G, nu=float(F["nu"]), device=0) as s:
        lo, hi = s.part.lo, s.part.hi
        own = slice(lo, hi)
        s.set_math(True)
        s.set_field(L.POLD, F["pold"][own])
        s.set_field(L.TMP, F["tmp_in"][own])
        s.laplacian_sub()  # ghosts of pold + the fine faces of remote blocks
        assert np.array_equal(s.get_field(L.TMP), F["tmp_out"][own]), "laplacian_sub rank %d" % rank
        s.set_field(L.VEL, F["vel"][own])
        s.vorticity()
        assert np.array_equal(s.get_field(L.TMP), F["vort"][own]), "vorticity rank %d" % rank
        s.set_field(L.TMPV, F["udef"][own])
        s.set_field(L.CHI, F["chi"][own])
        s.pressure_rhs(dt)
        assert np.array_equal(s.get_field(L.TMP), F["prhs"][own]), "pressure_rhs rank %d" % rank
        s.set_field(L.PRES, F["pres"][own])
        s.pressure_correction(dt)
        assert np.array_equal(s.get_field(L.TMPV), F["pcorr"][own]), "pressure_correction rank %d" % rank
        s.set_field(L.VEL, F["vel"][own])
        s.advect_diffuse_rhs(dt)  # halo-3 tile: two rings of ghost blocks, dim-2 face exchange
        assert np.array_equal(s.get_field(L.TMPV), F["advdiff"][own]), "advect_diffuse rank %d" % rank
        # computeA's split by hand (main.cpp:3035-3057): the blocks that read no ghost block, the ghost copies refreshed (whole
        # blocks through the block plan), then the others + the flux correction -- the same bits
        s.set_field(L.POLD, F["pold"][own])
        s.set_field(L.TMP, F["tmp_in"][own])
        L.check(s.L.cup2d_laplacian_sub(s._ctx, L.BLOCKS_INNER), "laplacian_sub inner")
        L.check(s.L.cup2d_halo_exchange(s._ctx, L.POLD, 8), "halo_exchange")
        L.check(s.L.cup2d_laplacian_sub(s._ctx, L.BLOCKS_HALO), "laplacian_sub halo")
        assert np.array_equal(s.get_field(L.TMP), F["tmp_out"][own]), "laplacian_sub in two phases, rank %d" % rank
        s.set_field(L.VEL, F["vel"][own])
        L.check(s.L.cup2d_advect_diffuse_rhs(s._ctx, s.nu, dt, L.BLOCKS_INNER), "advect inner")
        L.check(s.L.cup2d_halo_exchange(s._ctx, L.VEL, 8), "halo_exchange")
        L.check(s.L.cup2d_advect_diffuse_rhs(s._ctx, s.nu, dt, L.BLOCKS_HALO), "advect halo")
        assert np.array_equal(s.get_field(L.TMPV), F["advdiff"][own]), "advect_diffuse in two phases, rank %d" % rank
        # ---- a whole step against the single-context path on the same grid ----
        with AmrSimulation(G, nu=float(F["nu"])) as ref:
            ref.set_math(True)
            ref.set_field(L.VEL, F["vel"])
            ref.install_poisson_matrix()
            s.set_field(L.VEL, F["vel"][own])
            s.set_field(L.PRES, np.zeros((hi - lo, 64)))
            s.set_field(L.POLD, np.zeros((hi - lo, 64)))
            s.set_field(L.CHI, np.zeros((hi - lo, 64)))
            s.set_field(L.TMPV, np.zeros((hi - lo, 64, 2)))
            ref.set_field(L.VEL, F["vel"])
            for f in (L.PRES, L.POLD, L.CHI):
                ref.set_field(f, np.zeros((G.nblocks, 64)))
            ref.set_field(L.TMPV, np.zeros((G.nblocks, 64, 2)))
            rr = ref.step(tol=1e-10, rel_tol=0.0, max_restarts=100, max_iter=400)
            vref, pref = ref.get_field(L.VEL), ref.get_field(L.PRES)
            s.install_poisson_matrix()
            r = s.step(tol=1e-10, rel_tol=0.0, max_restarts=100, max_iter=400)
            assert r["dt"] == rr["dt"], (r, rr)
            dv, dp = np.abs(s.get_field(L.VEL) - vref[own]).max(), np.abs(s.get_field(L.PRES) - pref[own]).max()
            assert dv < 1e-8 and dp < 1e-7, (rank, dv, dp, r, rr)
            from cup2d_amd.amr_dist import strips_enabled
            assert (min(s.cell_exchanges) > 0 and s.sent_doubles[0] * 4 <= s.sent_doubles[1]) if strips_enabled() else sum(s.cell_exchanges) == 0, \
                (s.cell_exchanges, s.sent_doubles)
            # ---- regrid across the ranks = the single-context regrid (same leaves, bit-identical fields) ----
            ref.set_field(L.VEL, F["vel"])
            s.set_field(L.VEL, F["vel"][own])
            changed_ref = ref.adapt(1.0, 0.2, 5)
            blocks_ref, vel_ref = ref.grid.blocks.copy(), ref.get_field(L.VEL)
        changed = s.adapt(1.0, 0.2, 5)
        assert changed == changed_ref
        assert np.array_equal(s.global_grid.blocks, blocks_ref)
        assert np.array_equal(s.get_field(L.VEL), vel_ref[s.part.lo:s.part.hi])
        # the re-partitioned grid is balanced to one block and still steps
        counts = [None] * world
        dist.all_gather_object(counts, s.part.nowned)
        assert max(counts) - min(counts) <= 1 and sum(counts) == len(blocks_ref)
        r2 = s.step(tol=1e-9, rel_tol=0.0, max_restarts=100, max_iter=400)
        assert np.isfinite(r2["err"]) and r2["err"] <= 1e-9
        assert not s.comm_errors, s.comm_errors
    dist.barrier()


def run_amr_cpu(rank, world):
    """the AMR plan in a real exchange (gloo, host tensors): every rank packs the blocks its peers list as ghosts, one
    batched send/receive per peer with the plan's offsets and counts (different in the two directions), and every ghost
    block that arrives is the owner's block of that global id; then the face arrays (4 faces x 8 x dim per sent block) and
    the two reductions of the AMR step (max, and the {sum, sum} pair of the volume-weighted mean); then the three cell plans
    (what actually travels when a field is refreshed) through the same transport."""
    import torch
    import torch.distributed as dist
    from cup2d_amd.amr import AmrBlockGrid, circle_band_grid
    from cup2d_amd.amr_dist import AmrPartition
    from cup2d_amd.distributed import TorchComm
    for G in (AmrBlockGrid(np.load(os.path.join(ROOT, "tests", "golden", "amr_functors.npz"))["blocks"]), circle_band_grid(6)):
        P = AmrPartition(G, world, rank)
        cm = TorchComm(P, "host")
        nb = G.nblocks
        for unit, seed in ((64, 1), (128, 2), (32, 3)):  # a scalar block, two scalar blocks / one vector block, a scalar face array
            field = np.random.default_rng(seed).uniform(-1, 1, (nb, unit))  # the same global field on every rank
            slab = np.zeros((P.nowned + P.nghost, unit))
            slab[:P.nowned] = field[P.lo:P.hi]
            cm.send[:P.nsend * unit] = torch.from_numpy(slab[P.send_block].ravel())
            cm.exchange(unit)
            cm.wait()
            got = cm.recv[:P.nrecv * unit].numpy().reshape(P.nrecv, unit)
            slab[P.recv_block] = got
            assert np.array_equal(slab, field[P.local_ids]), (rank, unit)
        # the cell plans through the same transport: what arrives in a ghost block are the owner's values of exactly the cells
        # the plan lists, in the places the kernels read them; every other cell of the ghost blocks is left alone
        for which, (sc, rc, T) in enumerate(P.cells):
            for dim, seed in ((1, 11), (2, 12)):
                field = np.random.default_rng(seed + which).uniform(-1, 1, (nb * 64, dim))
                slab = np.full(((P.nowned + P.nghost) * 64, dim), np.nan)
                slab[:P.nowned * 64] = field[P.lo * 64:P.hi * 64]
                cm.send[:T.nsend * dim] = torch.from_numpy(slab[sc].ravel())
                cm.exchange(dim, topo=T)
                cm.wait()
                slab[rc] = cm.recv[:T.nrecv * dim].numpy().reshape(T.nrecv, dim)
                want = field.reshape(nb, 64, dim)[P.local_ids].reshape(-1, dim)
                got_cells = np.zeros(len(slab), dtype=bool)
                got_cells[rc] = True
                assert np.array_equal(slab[rc], want[rc]) and np.isnan(slab[P.nowned * 64:][~got_cells[P.nowned * 64:]]).all(), (rank, which, dim)
            assert T.nsend * 4 <= 64 * P.nsend or which == 1, (which, T.nsend, P.nsend)
        # reductions as the AMR step issues them
        cm.red[0] = float(rank + 1)
        cm.allreduce(0, 1, 1)  # max
        assert cm.red[0].item() == float(world)
        cm.red[0], cm.red[1] = 0.5 * (rank + 1), 2.0
        cm.allreduce(0, 2, 0)  # sums
        assert cm.red[0].item() == 0.5 * world * (world + 1) / 2 and cm.red[1].item() == 2.0 * world
        # the ghost tables are closed under what the kernels read: every non-wall side of an owned block is held
        k = P.kind[:P.nowned]
        assert (P.nbr2[:P.nowned][k != 0][:, 0] >= 0).all() and (P.nbr2[:P.nowned][k == 3] >= 0).all()
    dist.barrier()


def run_amr_regrid_cpu(rank, world):
    """The data side of a regrid on N ranks without a device (cup2d_amd/amr_dist.py fetch_new_range; gloo): the fields of a
    4 084-block grid live in numpy arrays of the OWNED blocks only; every rank fetches what its new range is made of from
    the old owners (requests by id, whole blocks back), computes its prolonged / restricted blocks with
    cup2d_amr_regrid_local on compact arrays, and ends up with exactly the blocks of the single-process regrid on its new
    range, bit for bit -- and nobody ever held more than its own share plus what the plan names."""
    import torch.distributed as dist
    from cup2d_amd import lib as L
    from cup2d_amd import amr as A
    from cup2d_amd.amr_dist import AmrPartition, fetch_new_range, FIELDS, partition_bounds
    G = A.circle_band_grid(7)
    nb = G.nblocks
    rng = np.random.default_rng(5)          # the same global fields on every rank (only the owned slice is "on the device")
    glob = {k: rng.uniform(-1, 1, (nb, 64 * L.FIELD_DIM[f])) for k, f in FIELDS}
    for seed, frac in ((1, 0.1), (2, 0.5)):
        tag = np.random.default_rng(seed)
        st = np.where(tag.uniform(0, 1, nb) < frac, A.REFINE, np.where(tag.uniform(0, 1, nb) < 0.5, A.COMPRESS, A.LEAVE)).astype(np.int32)
        st = A.validate_states(G.blocks, st, 9, G.bpdx, G.bpdy)
        assert (st == A.REFINE).any()
        P = AmrPartition(G, world, rank)
        fetched = []

        def download_units(ids, P=P, fetched=fetched):
            ids = np.asarray(ids, dtype=np.int64)
            assert ((ids >= 0) & (ids < P.nowned)).all(), "a block this rank does not own was asked of it"
            fetched.append(len(ids))
            return np.concatenate([glob[k][P.lo + ids] for k, _ in FIELDS], axis=1) if len(ids) else np.zeros((0, 448))

        R = fetch_new_range(G, P, st, 9, rank, world, download_units, None)
        full_blocks, full = A.regrid(G.blocks, st, {k: (glob[k], L.FIELD_DIM[f], L.FIELD_DIM[f] == 2) for k, f in FIELDS}, 9, G.bpdx, G.bpdy)
        lo, hi = R["lo"], R["hi"]
        assert np.array_equal(R["new_blocks"], full_blocks)
        assert np.array_equal(partition_bounds(len(full_blocks), world)[rank:rank + 2], [lo, hi])
        src, kept, slot = R["my_src"], R["kept"], R["slot"]
        for k, _ in FIELDS:
            rows = R["data"][k].copy()
            for q in np.flatnonzero(kept):
                o = int(src[q])
                rows[q] = glob[k][o] if P.lo <= o < P.hi else R["comp"][k][0][slot[o]]   # stays on the device | migrated here
            assert np.array_equal(rows, full[k][lo:hi]), (k, rank, seed)
        stt = R["stats"]
        assert stt["host_blocks_held"] == stt["blocks_received"] + stt["own_blocks_downloaded"]
        tot = [None] * world
        dist.all_gather_object(tot, stt)
        moved = sum(t["blocks_received"] for t in tot)
        assert moved == sum(t["blocks_sent"] for t in tot)
        if frac < 0.2:  # a regrid that touches a tenth of the grid moves a fraction of it
            assert moved < 0.6 * nb and max(t["host_blocks_held"] for t in tot) < 0.6 * nb, tot
        if rank == 0:
            print("amr_regrid_cpu: %d -> %d blocks on %d ranks, %d blocks moved between ranks, at most %d held on a host"
                  % (nb, len(full_blocks), world, moved, max(t["host_blocks_held"] for t in tot)), flush=True)
    dist.barrier()


def run_amr_big_gpu(rank, world):
    """the same on a grid of 4 084 blocks (three levels, a band around a circle, Hilbert order): the single context on the
    whole grid is the reference here -- every block operator on a rank's owned blocks bit for bit, a step to the solve
    tolerance, the regrid across the ranks bit for bit.  Contiguous Hilbert ranges of ~1 400 blocks: interior tiles take the
    fused sweeps, rank boundaries cut through all three levels."""
    import torch.distributed as dist
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrSimulation, circle_band_grid
    from cup2d_amd.amr_dist import DistributedAmrSimulation
    # CUP2D_TEST_LFINE / _MAXITER: the configs[4]-scale variant on 8 ranks (LFINE = 8: 16 k blocks; the solves capped -- eight ranks
    # time-slice the one GPU and every iteration needs all of them, tens of milliseconds each)
    lfine = int(os.environ.get("CUP2D_TEST_LFINE", "7"))
    cap = int(os.environ.get("CUP2D_TEST_MAXITER", "0"))
    G = circle_band_grid(lfine)
    nb = G.nblocks
    rng = np.random.default_rng(77)
    x, y = G.cell_centres()
    vel = np.stack([np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)], -1)\
        .reshape(nb, 64, 2) + 1e-2 * rng.uniform(-1, 1, (nb, 64, 2))
    pold, tmp_in, pres = (rng.uniform(-1, 1, (nb, 64)) for _ in range(3))
    chi = rng.uniform(0, 1, (nb, 64)) * (rng.uniform(0, 1, (nb, 1)) < 0.2)
    udef = rng.uniform(-1, 1, (nb, 64, 2)) * (chi[:, :, None] > 0)
    nu, dt = 1e-3, 1e-4
    ref_out = {}
    with AmrSimulation(G, nu=nu) as ref, DistributedAmrSimulation(G, nu=nu, device=0) as s:
        own = slice(s.part.lo, s.part.hi)
        n_own = s.part.nowned
        for sim, sl, out in ((ref, slice(None), ref_out), (s, own, None)):
            got = {}
            sim.set_math(True)
            sim.set_field(L.POLD, pold[sl]); sim.set_field(L.TMP, tmp_in[sl]); sim.laplacian_sub(); got["lap"] = sim.get_field(L.TMP).copy()
            sim.set_field(L.VEL, vel[sl]); sim.vorticity(); got["vort"] = sim.get_field(L.TMP).copy()
            sim.set_field(L.TMPV, udef[sl]); sim.set_field(L.CHI, chi[sl]); sim.pressure_rhs(dt); got["prhs"] = sim.get_field(L.TMP).copy()
            sim.set_field(L.PRES, pres[sl]); sim.pressure_correction(dt); got["pcorr"] = sim.get_field(L.TMPV).copy()
            sim.set_field(L.VEL, vel[sl]); sim.advect_diffuse_rhs(dt); got["advdiff"] = sim.get_field(L.TMPV).copy()
            if out is not None:
                out.update(got)
            else:
                for k, v in got.items():
                    assert np.array_equal(v, ref_out[k][own]), "%s rank %d" % (k, rank)
        # ---- a whole step ----
        for sim, sl, n in ((ref, slice(None), nb), (s, own, n_own)):
            sim.set_field(L.VEL, vel[sl])
            for f in (L.PRES, L.POLD, L.CHI):
                sim.set_field(f, np.zeros((n, 64)))
            sim.set_field(L.TMPV, np.zeros((n, 64, 2)))
            sim.install_poisson_matrix()
        # (capped: the same number of iterations of the same recurrences on both sides -- the fields agree to round-off, and
        # the ranks' STRICT advected velocity enters the comparison through them bit for bit)
        tol_s = 0.0 if cap else 1e-10
        rr = ref.step(tol=tol_s, rel_tol=0.0, max_restarts=100, max_iter=cap or 1000)
        import time
        dist.barrier()
        t_step = time.perf_counter()
        r = s.step(tol=tol_s, rel_tol=0.0, max_restarts=100, max_iter=cap or 1000)
        t_step = time.perf_counter() - t_step
        assert r["dt"] == rr["dt"] and s.last_solver() == ref.last_solver() == "fused", (r, rr)
        assert not cap or r["iters"] == rr["iters"] == cap, (r, rr)
        dv = np.abs(s.get_field(L.VEL) - ref.get_field(L.VEL)[own]).max()
        dp = np.abs(s.get_field(L.PRES) - ref.get_field(L.PRES)[own]).max()
        assert dv < 1e-8 and dp < 1e-6, (rank, dv, dp, r, rr)
        # ---- what travelled: the cells the kernels read (cell plans), not the blocks ----
        from cup2d_amd.amr_dist import strips_enabled
        if strips_enabled():
            sent, whole = s.sent_doubles
            assert min(s.cell_exchanges) > 0 and sent * 4 <= whole, (s.cell_exchanges, s.sent_doubles)
            if rank == 0:
                P = s.part
                print("amr_big strips (rank 0): exchanges through the cell plans halo1 / halo3 / matrix = %s, block-plan exchanges (face "
                      "arrays) = %d; doubles sent %d where whole ghost blocks are %d (1/%.1f); cells per plan %s of %d in the %d sent blocks"
                      % (s.cell_exchanges, s.block_exchanges, sent, whole, whole / sent, [c[2].nsend for c in P.cells], 64 * P.nsend, P.nsend),
                      flush=True)
        else:
            assert s.part.cells is None and sum(s.cell_exchanges) == 0
        # ---- regrid across the ranks = the single-context regrid ----
        ref.set_field(L.VEL, vel); s.set_field(L.VEL, vel[own])
        ref.vorticity()
        om = np.abs(ref.get_field(L.TMP)).reshape(nb, -1).max(1)
        rt, ct = float(np.quantile(om, 0.9)), float(np.quantile(om, 0.3))
        changed_ref = ref.adapt(rt, ct, lfine + 1)
        blocks_ref, vel_ref = ref.grid.blocks.copy(), ref.get_field(L.VEL)
        changed = s.adapt(rt, ct, lfine + 1)   # per-rank regrid + block migration (fetch_new_range)
        assert changed and changed_ref
        assert np.array_equal(s.global_grid.blocks, blocks_ref)
        assert np.array_equal(s.get_field(L.VEL), vel_ref[s.part.lo:s.part.hi])
        stt = s.regrid_stats
        assert stt["host_blocks_held"] < 0.6 * nb, stt   # nobody gathered the grid
        if rank == 0:
            print("amr_big regrid stats (rank 0):", stt, flush=True)
        counts = [None] * world
        dist.all_gather_object(counts, s.part.nowned)
        assert max(counts) - min(counts) <= 1 and sum(counts) == len(blocks_ref)
        r2 = s.step(tol=0.0 if cap else 1e-9, rel_tol=0.0, max_restarts=100, max_iter=cap or 1000)
        assert np.isfinite(r2["err"]) and (cap or r2["err"] <= 1e-9)
        assert not s.comm_errors, s.comm_errors
        if rank == 0:
            print("amr_big: %d blocks on %d ranks; step dv %.1e dp %.1e (%d iterations, %.0f ms on rank 0: ranks share the GPU, host-staged "
                  "gloo, strips %s); regrid -> %d blocks" % (nb, world, dv, dp, r["iters"], 1e3 * t_step, os.environ.get("CUP2D_AMR_STRIPS", "1"),
                                                            len(blocks_ref)), flush=True)
    dist.barrier()


def main():
    import torch.distributed as dist
    mode = sys.argv[1]
    px, py, nbx, nby = (int(a) for a in sys.argv[2:6])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode in ("amr", "amr_big", "amr_cpu", "amr_regrid_cpu"):
        {"amr": run_amr_gpu, "amr_big": run_amr_big_gpu, "amr_cpu": run_amr_cpu, "amr_regrid_cpu": run_amr_regrid_cpu}[mode](rank, world)
        if rank == 0:
            print("DIST_OK mode=%s world=%d" % (mode, world))
        dist.destroy_process_group()
        return
    assert world == px * py
    {"cpu": run_cpu, "gpu": run_gpu, "gpu_big": run_gpu_big}[mode](rank, world, px, py, nbx, nby)
    if rank == 0:
        print("DIST_OK mode=%s world=%d" % (mode, world))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
