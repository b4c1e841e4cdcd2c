"""Multi-process tests of the domain-decomposed path.
CPU (gloo, world_size 2 and 4): plan + exchange + all-reduce callbacks.
GPU (-m gpu; 2 and 4 ranks sharing the one GPU of the box, gloo host-staged): full HIP path vs the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(mode, world, px, py, nbx, nby, port, timeout=600, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, mode, str(px), str(py), str(nbx), str(nby)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=timeout, cwd=ROOT)
    out = r.stdout.decode()
    if r.returncode != 0 or "DIST_OK" not in out:
        # the first worker traceback (the launcher's own summary fills the tail and names no cause)
        lines = out.splitlines()
        first = next((i for i, l in enumerate(lines) if "Traceback (most recent call last)" in l and "torch/distributed/run.py" not in "".join(lines[i:i + 8])), None)
        cause = "\n".join(lines[first:first + 40]) if first is not None else ""
        errs = "\n".join(l for l in lines if ("Error" in l or "error" in l or "assert" in l) and "elastic" not in l)[:3000]
        raise AssertionError("rc %d\n---- first worker traceback ----\n%s\n---- error lines ----\n%s\n---- tail ----\n%s" % (r.returncode, cause, errs, out[-1500:]))
    for line in out.splitlines():  # what rank 0 measured (shown with -s)
        if line.startswith(("gpu_big", "amr_big", "amr_regrid_cpu")):
            print(line)


@pytest.mark.parametrize("world,px,py,nbx,nby", [(2, 2, 1, 4, 6), (2, 1, 2, 5, 3), (4, 2, 2, 4, 4), (8, 2, 4, 4, 4), (8, 4, 2, 3, 5)])
def test_plan_and_exchange_cpu_gloo(world, px, py, nbx, nby):
    """(8, 2, 4): BASELINE.json configs[3]'s layout (main.cpp:6494-6504 for the ranges) -- four of its eight ranks have THREE
    ghost sides (S and N at once + one x side), the other four two; (8, 4, 2): the transposed layout (W and E at once)."""
    launch("cpu", world, px, py, nbx, nby, 29611 + world + px)


def test_cartesian_dims():
    from cup2d_amd.distributed import cartesian_dims
    assert cartesian_dims(1) == (1, 1) and cartesian_dims(2) == (1, 2)
    assert cartesian_dims(4) == (2, 2) and cartesian_dims(8) == (2, 4)
    for w in (3, 6, 12):
        px, py = cartesian_dims(w)
        assert px * py == w


@pytest.mark.gpu
@pytest.mark.parametrize("world,px,py,nbx,nby", [(2, 2, 1, 8, 16), (2, 1, 2, 16, 8), (4, 2, 2, 8, 8), (8, 2, 4, 16, 16)])
def test_decomposed_step_matches_global_oracle_gpu(world, px, py, nbx, nby):
    """(8, 2, 4, 16, 16): the partitioning BASELINE.json configs[3] names (2 x 4 over 8 ranks, main.cpp:6494-6504), the eight
    ranks sharing the one GPU: four of them have three ghost sides (S and N at once + one x side: halo-set patches meeting at
    two corners, three peers per exchange), every functor STRICT bit for bit against the global CPU oracle"""
    launch("gpu", world, px, py, nbx, nby, 29711 + world + px, timeout=900)


@pytest.mark.gpu
def test_two_by_four_ranks_against_the_single_context_gpu():
    """dist_worker.run_gpu_big on the 2 x 4 layout at 16 x 16 blocks per rank (a 256 x 512-cell global grid): STRICT functors
    bit for bit against the single context on the whole grid, eight iterations of the two-launch MERGE 2 solver on 8 ranks =
    the five sweeps on one context to 1e-10 of max|x|, the residual the N-rank recurrence carries = max|b - A x| of the
    iterate assembled over the ranks, one bench step"""
    launch("gpu_big", 8, 2, 4, 16, 16, 29761, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("px,py,share,split", [(1, 2, "5", "0"), (2, 1, "5", "1"), (1, 2, "0", "0")])
def test_decomposed_path_at_configs3_rank_size_gpu(px, py, share, split):
    """BASELINE.json configs[3]'s per-rank patch -- 4096 x 2048 cells = 512 x 256 blocks plus a ghost ring -- on two ranks
    sharing the GPU (dist_worker.run_gpu_big): every functor STRICT bit for bit (FAST to 2e-13) against the single context on
    the whole grid, eight iterations of the two-launch MERGE 2 solver = the five sweeps to 1e-10 of max|x|, one whole bench
    step.  (1, 2): the global grid is 4096^2; (2, 1): 8192 x 2048 cells, a rectangle whose Hilbert order is not the
    square's.  share: CUP2D_EDGE_SHARE, the hand-over between sibling waves on (where the grid allows it) and off.  split:
    CUP2D_SWEEP_SPLIT, the sweeps as halo-set-first + inner launches with the ghost blocks travelling in between (opt-in)."""
    launch("gpu_big", 2, px, py, 512, 256, 29741 + 2 * px + py + (7 if share == "0" else 0), timeout=850, CUP2D_EDGE_SHARE=share,
           CUP2D_SWEEP_SPLIT=split)


@pytest.mark.parametrize("nranks", [2, 3, 5, 8])
def test_amr_partition_plans_are_consistent(nranks):
    """host planning of an adapted grid on N ranks (cup2d_amd/amr_dist.py): contiguous balanced Hilbert ranges, ghost sets
    that cover every table entry the kernels read and every matrix column, send lists that are the peers' receive lists"""
    import numpy as np
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid
    from cup2d_amd.amr_dist import AmrPartition
    G = AmrBlockGrid(np.load(os.path.join(ROOT, "tests", "golden", "amr_functors.npz"))["blocks"])
    coo = G.poisson_coo()
    parts = [AmrPartition(G, nranks, r, coo=coo) for r in range(nranks)]
    assert sum(P.nowned for P in parts) == G.nblocks and max(P.nowned for P in parts) - min(P.nowned for P in parts) <= 1
    nnz = 0
    for P in parts:
        for (p, so, ro, ns, nr) in P.peers:
            Q = parts[p]
            q = [x for x in Q.peers if x[0] == P.rank][0]
            assert (ns, nr) == (q[4], q[3])
            assert np.array_equal(P.local_ids[P.send_block[so:so + ns]], Q.local_ids[Q.recv_block[q[2]:q[2] + q[4]]])
        k = P.kind[:P.nowned]
        assert np.array_equal(k, G.kind[P.lo:P.hi])  # nothing of an owned block's topology is lost
        assert np.array_equal(P.local_ids[P.nbr2[:P.nowned][k != L.AMR_WALL][:, 0]], G.nbr2[P.lo:P.hi][k != L.AMR_WALL][:, 0])
        # across a coarser neighbour the halo-3 tile reads that neighbour's tangential sides: present for ghosts too
        for b in range(P.nowned):
            for s in range(4):
                if k[b, s] == L.AMR_COARSER:
                    n0 = P.nbr2[b, s, 0]
                    assert np.array_equal(P.kind[n0], G.kind[P.local_ids[n0]]), "coarse neighbour's table"
        assert P.col.max() < 64 * (P.nowned + P.nghost) and P.row.max() < 64 * P.nowned
        nnz += len(P.val)
    assert nnz == len(coo[2])
    # ---- the cell plans: both ends of a link list the same cells of the same blocks, inside the block plan ----
    for which in range(3):
        for P in parts:
            sc, rc, T = P.cells[which]
            assert (sc < 64 * P.nowned).all() and (rc >= 64 * P.nowned).all() and (rc < 64 * (P.nowned + P.nghost)).all()
            assert [t[0] for t in T.peers] == [t[0] for t in P.peers]
            for (q, so, ro, ns, nr), (_q, bso, bro, bns, bnr) in zip(T.peers, P.peers):
                Q = parts[q]
                qs, qr, QT = Q.cells[which]
                e = [x for x in QT.peers if x[0] == P.rank][0]
                assert (ns, nr) == (e[4], e[3])
                mine = P.local_ids[sc[so:so + ns] // 64] * 64 + sc[so:so + ns] % 64          # global cells I send to q
                theirs = Q.local_ids[qr[e[2]:e[2] + e[4]] // 64] * 64 + qr[e[2]:e[2] + e[4]] % 64
                assert np.array_equal(mine, theirs) and (np.diff(mine) > 0).all()
                assert np.isin(sc[so:so + ns] // 64, P.send_block[bso:bso + bns]).all()
            if which == L.CELLS_MATRIX:  # exactly the ghost columns of this rank's rows
                assert np.array_equal(np.unique(P.col[P.col >= 64 * P.nowned]), np.sort(rc))
                assert np.array_equal(P.gather, sc)
        cells, blocks = sum(P.cells[which][2].nsend for P in parts), 64 * sum(P.nsend for P in parts)
        assert cells * (2 if which == L.CELLS_HALO3 else 4) <= blocks, (which, cells, blocks)


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_amr_inner_and_halo_blocks_of_a_rank(nranks):
    """computeA's inner / halo split on an adapted grid (cup2d_amr_blocks_reading_ghosts, the lists behind CUP2D_BLOCKS_INNER /
    _HALO): a rank's owned blocks whose operators read a ghost block, by the kernels' own expressions.  The library skips the
    trace for blocks further than three neighbour steps from every ghost block; here every owned block of every rank is traced
    on its own (cup2d_amr_trace_reads with one reader) and must get the same verdict -- an inner block that read a ghost cell
    would be swept before the cell has arrived.  Halo-1 operators and the halo-3 tile, the golden grid and a 1 000-block band."""
    import ctypes
    import numpy as np
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, circle_band_grid
    from cup2d_amd.amr_dist import AmrPartition
    lib = L.load_library()
    vp = ctypes.c_void_p
    for G in (AmrBlockGrid(np.load(os.path.join(ROOT, "tests", "golden", "amr_functors.npz"))["blocks"]), circle_band_grid(6)):
        for rank in range(nranks):
            P = AmrPartition(G, nranks, rank)
            nt = P.nowned + P.nghost
            k, n2, h = (np.ascontiguousarray(a, dtype=np.int32) for a in (P.kind, P.nbr2, P.half))
            for which in (L.CELLS_HALO1, L.CELLS_HALO3):
                got = np.full(P.nowned, -1, dtype=np.int32)
                L.check(lib.cup2d_amr_blocks_reading_ghosts(P.nowned, nt, k.ctypes.data_as(vp), n2.ctypes.data_as(vp), h.ctypes.data_as(vp), which,
                                                            got.ctypes.data_as(vp)), "blocks_reading_ghosts")
                want = np.zeros(P.nowned, dtype=np.int32)
                for b in range(P.nowned):
                    mask = np.zeros(nt, dtype=np.uint64)
                    rd = np.asarray([b], dtype=np.int32)
                    L.check(lib.cup2d_amr_trace_reads(nt, k.ctypes.data_as(vp), n2.ctypes.data_as(vp), h.ctypes.data_as(vp), 1, rd.ctypes.data_as(vp),
                                                      which, mask.ctypes.data_as(vp)), "trace")
                    want[b] = int(mask[P.nowned:].any())
                assert np.array_equal(got, want), (G.nblocks, nranks, rank, which, np.flatnonzero(got != want)[:10])
                if nranks > 1 and P.nghost:
                    assert want.any() and (G.nblocks < 200 or not want.all()), (rank, which, int(want.sum()), P.nowned)
    assert lib.cup2d_amr_blocks_reading_ghosts(1, 0, None, None, None, 0, None) == -1


def test_amr_trace_reads_on_small_grids():
    """cup2d_amr_trace_reads (the kernels' own ghost expressions with a recording accessor) on grids small enough to count by
    hand: two same-level blocks; one coarse block next to four fine ones"""
    import ctypes
    import numpy as np
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid
    lib = L.load_library()
    vp = ctypes.c_void_p

    def trace(G, readers, which):
        mask = np.zeros(G.nblocks, dtype=np.uint64)
        k, n, h = (np.ascontiguousarray(a, dtype=np.int32) for a in (G.kind, G.nbr2, G.half))
        r = np.ascontiguousarray(readers, dtype=np.int32)
        L.check(lib.cup2d_amr_trace_reads(G.nblocks, k.ctypes.data_as(vp), n.ctypes.data_as(vp), h.ctypes.data_as(vp), len(r), r.ctypes.data_as(vp),
                                          which, mask.ctypes.data_as(vp)), "trace")
        return [sorted(c for c in range(64) if int(m) >> c & 1) for m in mask]
    # two level-0 blocks side by side: (level, i, j)
    G = AmrBlockGrid(np.array([[0, 0, 0], [0, 1, 0]], dtype=np.int32), bpdx=2, bpdy=1)
    west_col = lambda depth: sorted(8 * y + x for y in range(8) for x in range(depth))
    assert trace(G, [0], L.CELLS_HALO1) == [[], west_col(1)]
    assert trace(G, [0], L.CELLS_HALO3) == [[], west_col(3)]
    assert trace(G, [0], L.CELLS_MATRIX) == [[], west_col(1)]
    # the left root block refined once, the right one not
    G = AmrBlockGrid(np.array([[1, 0, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1], [0, 1, 0]], dtype=np.int32), bpdx=2, bpdy=1)
    lv = G.blocks[:, 0]
    coarse = int(np.nonzero(lv == 0)[0][0])
    fine_e = [b for b in range(G.nblocks) if lv[b] == 1 and G.blocks[b, 1] == 1]  # the two fine blocks that touch the coarse one
    for which, deep_f, n_c in ((L.CELLS_HALO1, 2, 4), (L.CELLS_HALO3, 6, None), (L.CELLS_MATRIX, 2, None)):
        m = trace(G, [coarse], which)  # the coarse block reads 2 x 2 means (rows) / the two fine cells (matrix): `deep_f` columns deep
        # (the kernels never read fine row 1 across a W/E face: the reference's unrolled branch pairs rows 0 and 2 there,
        # main.cpp:2528-2531 -- the trace knows, because it IS the kernel's expression)
        rows = range(8) if which == L.CELLS_MATRIX else (0, 2, 3, 4, 5, 6, 7)
        for b in fine_e:
            assert m[b] == sorted(8 * y + x for y in rows for x in range(8 - deep_f, 8)), (which, b, m[b])
        m = trace(G, fine_e[:1], which)  # a fine block reads the coarse block's west column along its half of the face (+ more)
        col0 = [c for c in m[coarse] if c % 8 == 0]
        assert len(col0) >= 4 and (n_c is None or len(m[coarse]) == n_c), (which, m[coarse])


@pytest.mark.gpu
@pytest.mark.parametrize("world,strips", [(2, "1"), (3, "1"), (2, "0"), (8, "1")])
def test_amr_on_n_ranks_matches_the_reference_functors_gpu(world, strips):
    """strips = "1" (default): the ghost blocks are refreshed through the cell plans -- only the cells the kernels read travel --
    and start as NaN (CUP2D_POISON_GHOSTS), so a kernel that read a cell no plan delivered could not equal the reference's
    functors bit for bit; "0": whole blocks through the block plan"""
    launch("amr", world, 0, 0, 0, 0, 29811 + world + 4 * int(strips), timeout=900, CUP2D_AMR_STRIPS=strips, CUP2D_POISON_GHOSTS="1")


@pytest.mark.gpu
@pytest.mark.parametrize("strips", ["1", "0"])
def test_amr_4084_blocks_on_3_ranks_matches_the_single_context_gpu(strips):
    """the circle-band grid (three levels, Hilbert order) on three ranks: block operators bit for bit, step, regrid; with the
    cell plans (strips) what is sent is at most a quarter of the whole ghost blocks"""
    launch("amr_big", 3, 0, 0, 0, 0, 29831 + int(strips), timeout=900, CUP2D_AMR_STRIPS=strips, CUP2D_POISON_GHOSTS="1")


@pytest.mark.gpu
def test_amr_16k_blocks_on_8_ranks_matches_the_single_context_gpu():
    """BASELINE.json configs[4]'s shape at a size eight ranks sharing one GPU can step (finest level 2048^2-equivalent, 16 k
    blocks on three levels, 2 k per rank): ranges, ghost blocks, cell plans (NaN-poisoned ghosts), every block operator STRICT
    bit for bit against the single context on the whole grid, a capped step, one regrid with migration between the eight
    ranks = the single-context regrid bit for bit, a step on the re-partitioned grid (main.cpp:6494-6504, 5055-5424)"""
    launch("amr_big", 8, 0, 0, 0, 0, 29841, timeout=1500, CUP2D_AMR_STRIPS="1", CUP2D_POISON_GHOSTS="1", CUP2D_TEST_LFINE="8",
           CUP2D_TEST_MAXITER="8")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_amr_whole_block_exchange_cpu_gloo(world):
    """the adapted-grid plan driven through a real multi-process exchange on the CPU (gloo): ghost blocks, face arrays,
    reductions"""
    launch("amr_cpu", world, 0, 0, 0, 0, 29851 + world)


@pytest.mark.parametrize("world", [2, 3])
def test_amr_regrid_on_n_ranks_moves_only_what_the_plan_names_cpu_gloo(world):
    """per-rank regrid + block migration (amr_dist.fetch_new_range, cup2d_amr_regrid_local) with numpy standing in for the
    device: every rank's new range equals the single-process regrid bit for bit; no rank is asked for a block it does not
    own, and a regrid that touches a tenth of a 4 084-block grid moves a fraction of it"""
    launch("amr_regrid_cpu", world, 0, 0, 0, 0, 29871 + world)


@pytest.mark.gpu
@pytest.mark.parametrize("px,py,comm", [(2, 1, "mpi"), (1, 2, "mpi"), (2, 2, "mpi"), (2, 4, "mpi"), (1, 1, "rccl")])
def test_cpp_mpi_driver_matches_the_single_rank_run_gpu(tmp_path, px, py, comm):
    """csrc/cup2d_run_mpi.cpp -- the N-rank time loop with the host side in C++ (Cartesian plan, cup2d_halo_plan, the
    callback transport over MPI; with one GPU per rank the same program takes the in-library RCCL communicator) -- on
    ranks sharing the one GPU: three steps of the Taylor-Green vortex land on the single-context run (dt to round-off,
    fields to the solve tolerance).  (1, 1, rccl): the program's RCCL branch -- token over MPI_Bcast, cup2d_comm_init,
    every reduction an ncclAllGather -- on the one rank a one-GPU box allows."""
    import shutil
    import numpy as np
    import cup2d_amd
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run_mpi")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    assert os.path.exists(exe) and os.path.exists(mpiexec), "cup2d_run_mpi / mpiexec not shipped"
    n, steps, iters = 128, 3, 400
    cmd = [mpiexec, "-n", str(px * py), exe, "-n", str(n), "-px", str(px), "-py", str(py), "-steps", str(steps), "-maxiter", str(iters),
           "-comm", comm, "-math", "strict", "-state", str(tmp_path / "s")]
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT, env=env)
    out = r.stdout.decode()
    assert r.returncode == 0 and "done: %d steps on %d ranks" % (steps, px * py) in out, out[-3000:]
    dts = [float(line.split()[5]) for line in out.splitlines() if line.startswith("step ")]
    h = 1.0 / n
    c = (np.arange(n) + 0.5) * h
    X, Y = np.meshgrid(c, c, indexing="xy")
    vel0 = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
    with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
        s.set_math(True)
        s.vel = vel0
        ref_dt = [s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters)["dt"] for _ in range(steps)]
        vref, pref = s.vel, s.pres
    assert np.allclose(dts, ref_dt, rtol=1e-10, atol=0), (dts, ref_dt)
    pnx, pny = n // px, n // py
    vel, pres = np.empty((n, n, 2)), np.empty((n, n))
    for rank in range(px * py):
        cx, cy = rank % px, rank // px
        vel[cy * pny:(cy + 1) * pny, cx * pnx:(cx + 1) * pnx] = np.fromfile(tmp_path / ("s.%d.vel.f64" % rank)).reshape(pny, pnx, 2)
        pres[cy * pny:(cy + 1) * pny, cx * pnx:(cx + 1) * pnx] = np.fromfile(tmp_path / ("s.%d.pres.f64" % rank)).reshape(pny, pnx)
    dv, dp = np.abs(vel - vref).max(), np.abs(pres - pref).max()
    print("cup2d_run_mpi %dx%d: max|dv| %.2e max|dp| %.2e" % (px, py, dv, dp))
    assert dv < 1e-9 and dp < 1e-8, (dv, dp)


@pytest.mark.gpu
@pytest.mark.parametrize("px,py", [(2, 1), (1, 2)])
def test_cpp_mpi_driver_at_configs3_rank_size_gpu(tmp_path, px, py):
    """cup2d_run_mpi -n 4096 on two ranks sharing the GPU -- 2048 x 4096 or 4096 x 2048 cells per rank, the per-rank patch of
    BASELINE.json configs[3] -- against cup2d_run on one rank: one step as bench.py runs it (FAST, 50 iterations at zero
    tolerance, main.cpp:7028-7030): dt exact, velocity to 1e-9, pressure to 1e-8 of max|p| (fifty unconverged BiCGSTAB
    iterations amplify the two summation orders of the dot products)."""
    import shutil
    import numpy as np
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run_mpi")
    one = os.path.join(ROOT, "cup2d_amd", "cup2d_run")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    assert os.path.exists(exe) and os.path.exists(one) and os.path.exists(mpiexec), "drivers / mpiexec not shipped"
    n = 4096
    common = ["-n", str(n), "-steps", "1", "-maxiter", "50", "-math", "fast"]
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([one] + common + ["-state", str(tmp_path / "one")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    assert r1.returncode == 0, r1.stdout.decode()[-3000:]
    rn = subprocess.run([mpiexec, "-n", "2", exe] + common + ["-px", str(px), "-py", str(py), "-comm", "mpi", "-state", str(tmp_path / "n")],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=800, cwd=ROOT, env=env)
    out = rn.stdout.decode()
    assert rn.returncode == 0 and "done: 1 steps on 2 ranks" in out, out[-3000:]
    s1 = [l.split() for l in r1.stdout.decode().splitlines() if l.startswith("step ")]
    sn = [l.split() for l in out.splitlines() if l.startswith("step ")]
    assert len(s1) == len(sn) == 1 and float(s1[0][5]) == float(sn[0][5]), (s1, sn)    # dt from max|u| of the same field: exact
    assert int(s1[0][7]) == int(sn[0][7]) == 50
    vref = np.fromfile(tmp_path / "one.vel.f64").reshape(n, n, 2)
    pref = np.fromfile(tmp_path / "one.pres.f64").reshape(n, n)
    pnx, pny = n // px, n // py
    dv = dp = 0.0
    for rank in range(2):
        cx, cy = rank % px, rank // px
        sl = (slice(cy * pny, (cy + 1) * pny), slice(cx * pnx, (cx + 1) * pnx))
        v = np.fromfile(tmp_path / ("n.%d.vel.f64" % rank)).reshape(pny, pnx, 2)
        p = np.fromfile(tmp_path / ("n.%d.pres.f64" % rank)).reshape(pny, pnx)
        dv, dp = max(dv, np.abs(v - vref[sl]).max()), max(dp, np.abs(p - pref[sl]).max())
    dp /= max(np.abs(pref).max(), 1e-300)
    print("cup2d_run_mpi -n 4096 %dx%d vs cup2d_run: max|dv| %.2e  max|dp| / max|p| %.2e" % (px, py, dv, dp))
    assert dv < 1e-9 and dp < 1e-8, (dv, dp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,comm", [(2, "mpi"), (3, "mpi"), (1, "rccl")])
def test_cpp_mpi_driver_amr_matches_the_single_rank_driver_gpu(tmp_path, world, comm):
    """csrc/cup2d_run_mpi.cpp -levelMax: the block-AMR time loop on N ranks with the host side in C++ -- contiguous Hilbert
    ranges, two rings of ghost blocks, per-rank regrid with block migration over MPI (cup2d_amr_regrid_local), the assembled
    coarse-fine operator per rank -- against csrc/cup2d_run.cpp on one rank: the same dt, block count and leaves at every
    step, fields to the solve tolerance (the solves run to round-off: 200 iterations at zero tolerance)."""
    import shutil
    import numpy as np
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run_mpi")
    one = os.path.join(ROOT, "cup2d_amd", "cup2d_run")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    assert os.path.exists(exe) and os.path.exists(one) and os.path.exists(mpiexec), "drivers / mpiexec not shipped"
    common = ["-levelStart", "3", "-levelMax", "6", "-Rtol", "2", "-Ctol", "0.5", "-steps", "5", "-maxiter", "200", "-math", "strict"]
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([one] + common + ["-state", str(tmp_path / "one")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    assert r1.returncode == 0, r1.stdout.decode()[-3000:]
    # (the ranks' ghost blocks start as NaN: what the cell plans do not deliver must be what no kernel reads)
    rn = subprocess.run([mpiexec, "-n", str(world), exe] + common + ["-comm", comm, "-state", str(tmp_path / "n")], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, timeout=900, cwd=ROOT, env=dict(env, CUP2D_POISON_GHOSTS="1"))
    out = rn.stdout.decode()
    assert rn.returncode == 0 and "done: 5 steps" in out, out[-3000:]
    s1 = [l.split() for l in r1.stdout.decode().splitlines() if l.startswith("step ")]
    sn = [l.split() for l in out.splitlines() if l.startswith("step ")]
    assert len(s1) == len(sn) == 5
    for a, b in zip(s1, sn):
        assert abs(float(a[5]) - float(b[5])) <= 1e-8 * float(a[5]) and a[11] == b[11], (a, b)   # dt (follows max|u| of the solves), blocks
    b1 = np.fromfile(tmp_path / "one.blocks.i32", dtype=np.int32).reshape(-1, 3)
    bn = np.fromfile(tmp_path / "n.blocks.i32", dtype=np.int32).reshape(-1, 3)
    assert len(b1) > 64 and set(map(tuple, b1.tolist())) == set(map(tuple, bn.tolist()))
    v1 = np.fromfile(tmp_path / "one.vel.f64").reshape(len(b1), 128)
    p1 = np.fromfile(tmp_path / "one.pres.f64").reshape(len(b1), 64)
    bounds = [int(x) for x in open(tmp_path / "n.meta").read().split()]
    assert bounds[0] == 0 and bounds[-1] == len(bn) and len(bounds) == world + 1
    vn = np.concatenate([np.fromfile(tmp_path / ("n.%d.vel.f64" % r)).reshape(-1, 128) for r in range(world)])
    pn = np.concatenate([np.fromfile(tmp_path / ("n.%d.pres.f64" % r)).reshape(-1, 64) for r in range(world)])
    assert len(vn) == len(bn) == len(pn)
    where = {tuple(b): k for k, b in enumerate(b1.tolist())}
    order = np.array([where[tuple(b)] for b in bn.tolist()])
    dv, dp = np.abs(vn - v1[order]).max(), np.abs(pn - p1[order]).max()
    print("cup2d_run_mpi -levelMax on %d rank(s): %d blocks, max|dv| %.2e max|dp| %.2e; %s" % (world, len(bn), dv, dp, out.splitlines()[-1]))
    assert dv < 1e-8 and dp < 1e-6, (dv, dp)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_cpp_amr_partition_equals_the_python_plan(tmp_path, world):
    """csrc/cup2d_run_mpi.cpp AmrPart (-planOnly: no GPU) against cup2d_amd/amr_dist.py AmrPartition on the 4 084-block
    three-level grid: ranges, ghost closure (two rings), local topology tables, whole-block links with their offsets and
    counts per direction, the gather list, the three cell plans (which cells of those blocks travel) -- table by table, rank by
    rank"""
    import shutil
    import numpy as np
    from cup2d_amd import amr as A
    from cup2d_amd.amr_dist import AmrPartition
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run_mpi")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("cup2d_run_mpi / mpiexec not built here")
    G = A.circle_band_grid(7)
    np.ascontiguousarray(G.blocks, dtype=np.int32).tofile(tmp_path / "blocks.i32")
    r = subprocess.run([mpiexec, "-n", str(world), exe, "-planOnly", str(tmp_path / "blocks.i32"), "-state", str(tmp_path / "p")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    for rank in range(world):
        P = AmrPartition(G, world, rank)
        got = lambda name: np.fromfile(tmp_path / ("p.%d.%s" % (rank, name)), dtype=np.int32)
        assert got("range").tolist() == [P.lo, P.hi, P.nghost]
        assert np.array_equal(got("ghost_ids"), P.ghost_ids)
        for name in ("level", "kind", "nbr2", "half", "nbr", "send_block", "recv_block", "gather"):
            assert np.array_equal(got(name), np.asarray(getattr(P, name)).ravel()), (rank, name)
        assert got("links").reshape(-1, 5).tolist() == [list(p) for p in P.peers], rank
        for which in range(3):  # the cell plans: what of those blocks travels, per operator family
            sc, rc, T = P.cells[which]
            assert np.array_equal(got("send_cell%d" % which), sc) and np.array_equal(got("recv_cell%d" % which), rc), (rank, which)
            assert got("cell_links%d" % which).reshape(-1, 5).tolist() == [list(p) for p in T.peers], (rank, which)
