"""Block-AMR (BASELINE.json configs[4]), halo-1 block operators: ghost cells across coarse-fine faces and the flux
correction of the coarse side.  Three layers, all BIT FOR BIT:
  golden  tests/golden/amr_functors.npz -- written by the reference itself (oracle/_ref/ref_harness 'amr': its own
          adapt() builds a 3-level, 76-block grid around a vortex pair; pressure_rhs1, pressure_rhs (both with
          prepare0/fillcases), pressureCorrectionKernel and KernelVorticity run on analytic fields)
  oracle  oracle/amr.py, the numpy restatement of BlockLab::load/post_load + fillcases for halo-1 stencils
  HIP     csrc/amr.hip through the C ABI (cup2d_set_amr + the usual entry points)      [-m gpu]"""
import numpy as np
import pytest

from conftest import golden


def _grid_cases(oracle):
    yield "golden", golden("amr_functors.npz")
    if oracle.have_reference():
        yield "live-118", oracle.ref_amr_functors(1, 5, 6, 1.0, 0.2)
        yield "live-232", oracle.ref_amr_functors(2, 6, 7, 3.0, 1.0)


def test_amr_oracle_bit_exact_vs_reference(oracle):
    from oracle import amr as A
    seen = set()
    for name, F in _grid_cases(oracle):
        g = A.AmrGrid(F["blocks"])
        seen |= {(s, g.neighbour(b, s)[0]) for b in range(len(g.blocks)) for s in range(4)}
        assert np.array_equal(A.laplacian_sub_amr(g, F["pold"], F["tmp_in"]), F["tmp_out"]), name
        assert np.array_equal(A.vorticity_amr(g, F["vel"]), F["vort"]), name
        assert np.array_equal(A.pressure_rhs_amr(g, F["vel"], F["udef"], F["chi"], float(F["dt"])), F["prhs"]), name
        assert np.array_equal(A.pressure_correction_amr(g, F["pres"], float(F["dt"])), F["pcorr"]), name
    # every side meets every kind of neighbour in the fixtures
    assert seen == {(s, k) for s in range(4) for k in ("wall", "same", "coarse", "fine")}


def test_amr_halo3_tile_and_advect_diffuse_vs_reference(oracle):
    """KernelAdvectDiffuse's ghosted tile (Stencil{-3,-3,4,4,true}): the literal transcription of BlockLab
    (oracle/amr_lab.py) reproduces the reference's 14x14x2 tiles completely, the closed forms (oracle/amr.py lab3_cross,
    what csrc/amr.hip implements) on the cross the functor reads; then the functor with its dim-2 flux correction."""
    from oracle import amr as A
    from oracle import amr_lab as AL
    cross = np.zeros((14, 14), bool)
    cross[3:11, :] = True
    cross[:, 3:11] = True
    for name, F in _grid_cases(oracle):
        g = A.AmrGrid(F["blocks"])
        t = AL.Tree(F["blocks"])
        lab = AL.BlockLab(2, (-3, -3, 4, 4, True), True)
        flat = F["vel"].reshape(len(F["blocks"]), -1)
        for b in range(len(g.blocks)):
            assert np.array_equal(lab.load(t, flat, b), F["lab3"][b]), (name, b)
            m = A.lab3_cross(g, F["vel"], b)
            assert np.array_equal(m[cross], F["lab3"][b][cross]), (name, b)
        assert np.array_equal(A.advect_diffuse_amr(g, F["vel"], float(F["nu"]), float(F["dt"])), F["advdiff"]), name


def _by_block(blocks, arr):
    return {tuple(int(v) for v in b): arr[k] for k, b in enumerate(blocks)}


def test_amr_regrid_vs_reference_adapt(oracle):
    """cup2d_amd.amr tag_states / validate_states / regrid against the reference's own adapt() (main.cpp:4657-5440) run
    on analytic fields: same leaf blocks afterwards (refinement by |vorticity|, 2:1 balance across faces and corners,
    sibling groups), prolonged and restricted fields bit-identical.  Golden: 76 -> 94 blocks; live: every field."""
    from cup2d_amd import amr as P
    from oracle import amr as A
    G = golden("amr_adapt.npz")
    cases = [("golden", dict(blocks=G["pre_blocks"], vel=G["pre_vel"], pres=G["pre_pres"]),
              dict(blocks=G["post_blocks"], vel=G["post_vel"], pres=G["post_pres"]), float(G["rtol"]), float(G["ctol"]), int(G["level_max"]))]
    if oracle.have_reference():
        for args in ((1, 5, 6, 1.0, 0.2), (2, 6, 7, 3.0, 1.0)):
            pre, post = oracle.ref_amr_adapt(*args)
            cases.append(("live%s" % (args,), pre, post, args[3], args[4], args[1]))
    for name, pre, post, rtol, ctol, lmax in cases:
        nb = len(pre["blocks"])
        linf = np.abs(A.vorticity_amr(A.AmrGrid(pre["blocks"]), pre["vel"])).max(axis=1)
        st = P.validate_states(pre["blocks"], P.tag_states(linf, pre["blocks"][:, 0], rtol, ctol, lmax), lmax)
        assert (st == P.REFINE).any() and (st == P.COMPRESS).any(), name  # the fixture exercises both
        fields = {k: (pre[k].reshape(nb, -1), 2 if pre[k].ndim == 3 else 1, pre[k].ndim == 3) for k in pre if k != "blocks"}
        blocks, data = P.regrid(pre["blocks"], st, fields, lmax)
        assert set(map(tuple, blocks.tolist())) == set(map(tuple, post["blocks"].tolist())), name
        for k in data:
            ref = _by_block(post["blocks"], post[k].reshape(len(post["blocks"]), -1))
            mine = _by_block(blocks, data[k])
            assert all(np.array_equal(ref[b], mine[b]) for b in ref), (name, k)
        P.AmrBlockGrid(blocks)  # the result is a valid (2:1 balanced) tiling


def test_amr_host_regrid_library_vs_python_statement():
    """the library's host regridding (cup2d_amr_validate_states, cup2d_amr_regrid: leaf table + closed-form sides and
    corners) against the Python statement on the general-stencil BlockLab (amr_lab.py, itself pinned to tiles dumped
    by the reference): random tags on random balanced grids, bit for bit, every kind of side and corner visited"""
    from cup2d_amd import amr as A
    from oracle.amr_lab import Tree
    from oracle import amr_regrid as R
    seen = dict(coarse_corner=0, fine_corner=0, coarse_face=0, fine_face=0, compress=0)
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        level_max, l0 = 4 + seed % 3, 1 + seed % 2
        blocks = np.array([(l0, i, j) for j in range(1 << l0) for i in range(1 << l0)], dtype=np.int64)
        vel, pres = rng.uniform(-1, 1, (len(blocks), 128)), rng.uniform(-1, 1, (len(blocks), 64))
        pr, pc = [0.15, 0.3, 0.1][seed % 3], [0.3, 0.15, 0.5][seed % 3]
        for it in range(6):
            nb = len(blocks)
            if nb > 300:
                break
            st0 = rng.choice([0, 1, 2], size=nb, p=[1 - pr - pc, pr, pc]).astype(np.int32)
            st = A.validate_states(blocks, st0, level_max)
            assert np.array_equal(st, R.validate_states_py(blocks, st0, level_max)), (seed, it)
            tree = Tree(blocks)
            for k, (l, i, j) in enumerate(blocks):
                if st[k] != A.REFINE:
                    continue
                for cx in (-1, 0, 1):
                    for cy in (-1, 0, 1):
                        if (cx or cy) and 0 <= i + cx < 1 << l and 0 <= j + cy < 1 << l:
                            t = tree.state(int(l), int(i + cx), int(j + cy))
                            what = "corner" if cx and cy else "face"
                            seen["coarse_" + what] += t == -2
                            seen["fine_" + what] += t == -1
            seen["compress"] += int((st == A.COMPRESS).sum())
            fields = {"vel": (vel, 2, True), "pres": (pres, 1, False)}
            b_c, d_c = A.regrid(blocks, st, fields, level_max)
            b_py, d_py = R.regrid_py(blocks, st, fields, level_max)
            assert np.array_equal(b_c, b_py), (seed, it)
            for k in d_py:
                assert np.array_equal(d_c[k], d_py[k]) and not np.isnan(d_c[k]).any(), (seed, it, k)
            # the plan of the same regrid for a host that keeps the fields on the device: same leaves; an unchanged block is
            # the old block the plan names; the changed ones come out the same when every block the plan does NOT name as
            # needed is poisoned -- nothing outside the plan is read
            b_p, src, needed = A.regrid_plan(blocks, st, level_max)
            assert np.array_equal(b_p, b_c), (seed, it)
            kept = src >= 0
            assert np.array_equal(d_c["vel"][kept], vel[src[kept]]) and np.array_equal(d_c["pres"][kept], pres[src[kept]])
            vel_p, pres_p = np.where(needed[:, None], vel, np.nan), np.where(needed[:, None], pres, np.nan)
            b_x, d_x = A.regrid_changed(blocks, st, {"vel": (vel_p, 2, True), "pres": (pres_p, 1, False)}, level_max)
            assert np.array_equal(b_x, b_c)
            for k in d_x:
                assert np.array_equal(d_x[k][~kept], d_c[k][~kept]), (seed, it, k)
            blocks, vel, pres = b_c, d_c["vel"], d_c["pres"]
            g = A.AmrBlockGrid(blocks)  # stays 2:1 balanced (raises otherwise)
            assert g.nblocks == len(blocks)
    assert all(v > 10 for v in seen.values()), seen


def test_amr_regrid_kernel_tables_replayed_on_the_cpu():
    """cup2d_amr_regrid_device's HOST half without a GPU: the job and corner tables it hands to k_amr_regrid
    (cup2d_amr_regrid_jobs) replayed in numpy exactly as the kernel reads them -- COPY / RESTRICT / PROLONG from the parent's
    tensorial halo-1 tile, side cells by the closed forms of the block operators (oracle.amr.lab1 = amr_ghost), corner cells from
    the 12-int descriptors (wall / same-level cell / 2 x 2 mean of a finer block / TestInterp on nine coarse cells, leaf or
    averaged) -- against the library's host regrid, bit for bit, on random balanced grids: a wrong index in a table cannot
    hide behind the GPU."""
    import ctypes
    from cup2d_amd import amr as A, lib as L
    from oracle import amr as OA
    lib = L.load_library()
    vp = ctypes.c_void_p
    kinds = {0: 0, 1: 0, 2: 0, 3: 0}
    for seed in range(4):
        rng = np.random.default_rng(500 + seed)
        level_max, l0 = 4 + seed % 2, 1 + seed % 2
        blocks = np.array([(l0, i, j) for j in range(1 << l0) for i in range(1 << l0)], dtype=np.int64)
        vel, pres = rng.uniform(-1, 1, (len(blocks), 128)), rng.uniform(-1, 1, (len(blocks), 64))
        for it in range(5):
            nb = len(blocks)
            if nb > 250:
                break
            st = A.validate_states(blocks, rng.choice([0, 1, 2], size=nb, p=[0.55, 0.2, 0.25]).astype(np.int32), level_max)
            if not (st != A.LEAVE).any():
                continue
            b_host, d_host = A.regrid(blocks, st, {"vel": (vel, 2, True), "pres": (pres, 1, False)}, level_max)
            b32, st32 = np.ascontiguousarray(blocks, dtype=np.int32), np.ascontiguousarray(st, dtype=np.int32)
            npro = ctypes.c_longlong()
            nj = lib.cup2d_amr_regrid_jobs(nb, b32.ctypes.data_as(vp), 1, 1, level_max, st32.ctypes.data_as(vp), 0, None, 0, None, ctypes.byref(npro))
            assert nj > 0
            jobs, corners = np.zeros((nj, 8), np.int32), np.zeros((max(1, npro.value), 4, 12), np.int32)
            assert lib.cup2d_amr_regrid_jobs(nb, b32.ctypes.data_as(vp), 1, 1, level_max, st32.ctypes.data_as(vp), nj, jobs.ctypes.data_as(vp),
                                             npro.value, corners.ctypes.data_as(vp), None) == nj
            assert (jobs[:npro.value, 0] == 2).all() and (jobs[npro.value:, 0] != 2).all()   # prolong jobs first
            grid = OA.AmrGrid(blocks)
            for name, fld, dim in (("vel", vel, 2), ("pres", pres, 1)):
                f = fld.reshape(nb, 64, dim)
                out = np.full((len(b_host), 64, dim), np.nan)
                for k, J in enumerate(jobs):
                    if J[0] == 0:
                        out[J[1]] = f[J[2]]
                    elif J[0] == 1:
                        for lane in range(64):
                            X, Y = lane & 7, lane >> 3
                            kid = f[J[2 + 2 * (Y >> 2) + (X >> 2)]].reshape(8, 8, dim)
                            x, y = X & 3, Y & 3
                            out[J[1], lane] = (kid[2 * y, 2 * x] + kid[2 * y + 1, 2 * x] + kid[2 * y, 2 * x + 1] + kid[2 * y + 1, 2 * x + 1]) / 4
                    else:
                        b, i0, j0 = int(J[1]), int(J[6]), int(J[7])
                        T = np.empty((10, 10, dim))
                        for d in range(dim):
                            sign = None if dim == 1 else ((-1.0, 1.0) if d == 0 else (1.0, -1.0))
                            T[:, :, d] = OA.lab1(grid, f[:, :, d], b, sign)
                        for cn in range(4):
                            cx, cy = (1 if cn & 1 else -1), (1 if cn >> 1 else -1)
                            gx, gy = (0 if cx < 0 else 9), (0 if cy < 0 else 9)       # tile indices [row][col]
                            ex, ey = (1 if cx < 0 else 8), (1 if cy < 0 else 8)
                            C = corners[k, cn]
                            kinds[min(int(C[0]), 3)] += 1
                            if C[0] == 0:
                                ywall = bool(C[1] & 2)
                                for d in range(dim):
                                    v = T[ey, gx, d] if ywall else T[gy, ex, d]
                                    if dim == 2 and ((ywall and d == 1) or (not ywall and d == 0)):
                                        v = -v
                                    T[gy, gx, d] = v
                            elif C[0] == 1:
                                cell = (7 if cy < 0 else 0) * 8 + (7 if cx < 0 else 0)
                                T[gy, gx] = f[C[2], cell]
                            elif C[0] == 2:
                                XX, YY = 4 * i0 + (-1 if cx < 0 else 4), 4 * j0 + (-1 if cy < 0 else 4)
                                Cc = [[np.nan] * 3 for _ in range(3)]
                                for a in range(3):
                                    for c in range(3):
                                        e, GX, GY = int(C[2 + 3 * a + c]), XX - 1 + a, YY - 1 + c
                                        if e < 0:
                                            continue
                                        blk = f[e >> 1].reshape(8, 8, dim)
                                        if e & 1:
                                            x, y = 2 * (GX & 3), 2 * (GY & 3)
                                            Cc[a][c] = (blk[y, x, 0] + blk[y + 1, x, 0] + blk[y, x + 1, 0] + blk[y + 1, x + 1, 0]) / 4
                                        else:
                                            Cc[a][c] = blk[GY & 7, GX & 7, 0]
                                T[gy, gx, :] = OA._test_interp(Cc, 0.25 if cx < 0 else -0.25, 0.25 if cy < 0 else -0.25)
                            elif C[0] == 3:
                                blk = f[C[2]].reshape(8, 8, dim)
                                x, y = (6 if cx < 0 else 0), (6 if cy < 0 else 0)
                                T[gy, gx] = (blk[y, x] + blk[y + 1, x] + blk[y, x + 1] + blk[y + 1, x + 1]) / 4
                            else:
                                T[gy, gx] = np.nan
                        for lane in range(64):
                            pi, pj = lane & 7, lane >> 3
                            I, Jc, i, j = pi >> 2, pj >> 2, 2 * (pi & 3), 2 * (pj & 3)
                            u = lambda dj, di: T[pj + 1 + dj, pi + 1 + di]  # noqa: E731
                            l00, l0p, l0m, lm0, lmm, lmp = u(0, 0), u(1, 0), u(-1, 0), u(0, -1), u(-1, -1), u(1, -1)
                            lp0, lpm, lpp = u(0, 1), u(-1, 1), u(1, 1)
                            x, y = 0.5 * (lp0 - lm0), 0.5 * (l0p - l0m)
                            x2, y2 = (lp0 + lm0) - 2.0 * l00, (l0p + l0m) - 2.0 * l00
                            xy = 0.25 * ((lpp + lmm) - (lpm + lmp))
                            c2 = 0.03125 * x2 + 0.03125 * y2
                            kid = out[J[2 + 2 * Jc + I]].reshape(8, 8, dim)
                            kid[j, i] = (l00 + (-0.25 * x - 0.25 * y)) + (c2 + 0.0625 * xy)
                            kid[j, i + 1] = (l00 + (+0.25 * x - 0.25 * y)) + (c2 - 0.0625 * xy)
                            kid[j + 1, i] = (l00 + (-0.25 * x + 0.25 * y)) + (c2 - 0.0625 * xy)
                            kid[j + 1, i + 1] = (l00 + (+0.25 * x + 0.25 * y)) + (c2 + 0.0625 * xy)
                assert not np.isnan(out).any(), (seed, it, name)
                assert np.array_equal(out.reshape(len(b_host), -1), d_host[name]), (seed, it, name)
            blocks, vel, pres = b_host, d_host["vel"], d_host["pres"]
    assert all(v > 0 for v in kinds.values()), kinds   # every kind of corner was replayed


def test_amr_host_routines_reject_bad_input():
    """the regrid-time host routines (no GPU): argument checks and the error text, through the C ABI"""
    import ctypes
    from cup2d_amd import lib as L
    lib = L.load_library()
    vp = ctypes.c_void_p

    def ptr(a):
        return a.ctypes.data_as(vp)
    blocks = np.array([(1, i, j) for j in range(2) for i in range(2)], dtype=np.int32)
    kind, nbr2, half = np.zeros((4, 4), np.int32), np.zeros((4, 4, 2), np.int32), np.zeros((4, 4), np.int32)
    assert lib.cup2d_amr_tables(4, ptr(blocks), 1, 1, ptr(kind), ptr(nbr2), ptr(half)) == 0
    assert (kind[0] == [L.AMR_WALL, L.AMR_SAME, L.AMR_WALL, L.AMR_SAME]).all() and nbr2[0, 1, 0] == 1 and nbr2[0, 3, 0] == 2
    bad = blocks.copy()
    bad[3] = (1, 2, 0)  # outside the 2 x 2 level-1 grid
    assert lib.cup2d_amr_tables(4, ptr(bad), 1, 1, ptr(kind), ptr(nbr2), ptr(half)) == -1
    assert b"outside" in lib.cup2d_last_error()
    hole = np.array([(1, 0, 0), (1, 1, 0), (1, 0, 1)], dtype=np.int32)  # (1, 1, 1) missing: not a tiling
    assert lib.cup2d_amr_tables(3, ptr(hole), 1, 1, ptr(kind), ptr(nbr2), ptr(half)) == -1
    assert b"balanced" in lib.cup2d_last_error()
    assert lib.cup2d_amr_tables(0, ptr(blocks), 1, 1, ptr(kind), ptr(nbr2), ptr(half)) == -1
    st = np.array([2, 0, 0, 0], dtype=np.int32)  # one sibling alone cannot compress
    assert lib.cup2d_amr_validate_states(4, ptr(blocks), 1, 1, 3, ptr(st)) == 0 and (st == 0).all()
    st = np.array([2, 2, 2, 0], dtype=np.int32)
    f = np.zeros((4, 64))
    srcs = (vp * 1)(f.ctypes.data)
    dims, vec = np.array([1], np.int32), np.array([0], np.int32)
    assert lib.cup2d_amr_regrid(4, ptr(blocks), 1, 1, 3, ptr(st), 1, srcs, ptr(dims), ptr(vec), 0, None, None) == -1
    assert b"siblings" in lib.cup2d_last_error()
    st[:] = 2
    assert lib.cup2d_amr_regrid(4, ptr(blocks), 1, 1, 3, ptr(st), 1, srcs, ptr(dims), ptr(vec), 0, None, None) == 1  # one parent
    st[:] = 1
    assert lib.cup2d_amr_regrid(4, ptr(blocks), 1, 1, 3, ptr(st), 1, srcs, ptr(dims), ptr(vec), 0, None, None) == 16
    nb2 = np.zeros((16, 3), np.int32)
    out = np.zeros((16, 64))
    dsts = (vp * 1)(out.ctypes.data)
    assert lib.cup2d_amr_regrid(4, ptr(blocks), 1, 1, 3, ptr(st), 1, srcs, ptr(dims), ptr(vec), 8, ptr(nb2), dsts) == -1  # capacity
    f[:] = 2.5  # a constant field prolongs to the same constant
    assert lib.cup2d_amr_regrid(4, ptr(blocks), 1, 1, 3, ptr(st), 1, srcs, ptr(dims), ptr(vec), 16, ptr(nb2), dsts) == 16
    assert (out == 2.5).all() and (nb2[:, 0] == 2).all() and len({tuple(b) for b in nb2.tolist()}) == 16
    assert lib.cup2d_amr_poisson_coo(0, ptr(kind), ptr(nbr2), ptr(half), 0, None, None, None) == -1


def test_amr_topology_tables(oracle):
    """cup2d_amd.amr.AmrBlockGrid (product) against the oracle's neighbour logic; level jumps are 2:1"""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid
    from oracle import amr as A
    F = golden("amr_functors.npz")
    g, o = AmrBlockGrid(F["blocks"]), A.AmrGrid(F["blocks"])
    name = {L.AMR_WALL: "wall", L.AMR_SAME: "same", L.AMR_COARSER: "coarse", L.AMR_FINER: "fine"}
    for b in range(g.nblocks):
        for s in range(4):
            t = o.neighbour(b, s)
            assert name[int(g.kind[b, s])] == t[0]
            if t[0] in ("same", "coarse"):
                assert g.nbr2[b, s, 0] == t[1]
            if t[0] == "fine":
                assert tuple(g.nbr2[b, s]) == tuple(t[1:])
            if t[0] == "coarse":
                assert g.half[b, s] == (F["blocks"][b][2] % 2 if s < 2 else F["blocks"][b][1] % 2)
    x, y = g.cell_centres()
    assert x.min() > 0 and x.max() < 1 and y.min() > 0 and y.max() < 1
    with pytest.raises(ValueError):
        AmrBlockGrid([(0, 0, 0), (2, 0, 0)])  # not a tiling


def test_amr_poisson_matrix_vs_reference(oracle):
    """the coarse-fine rows of the Poisson matrix (Solver::makeFlux / interpolate, main.cpp:5915-5997) assembled by
    cup2d_amd.amr.AmrBlockGrid.poisson_coo, against the matrix the REFERENCE assembled for the same grid, through
    its action on a field (the harness applies the reference's own triplets)"""
    import scipy.sparse as sp
    from cup2d_amd.amr import AmrBlockGrid
    for name, F in _grid_cases(oracle):
        g = AmrBlockGrid(F["blocks"])
        r, c, v = g.poisson_coo()  # the library's host routine (C++)
        from oracle import amr_regrid as R
        rp, cp, vp = R.poisson_coo_py(g)  # the same algorithm in Python: bit for bit
        assert np.array_equal(r, rp) and np.array_equal(c, cp) and np.array_equal(v, vp), name
        n = 64 * g.nblocks
        A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
        assert np.abs(A @ F["pres"].ravel() - F["Ax"].ravel()).max() < 1e-14, name
        assert np.abs(A @ np.ones(n)).max() < 1e-14  # constants are in the null space (homogeneous Neumann walls)
        assert A.getnnz(axis=1).max() <= 13 and (A.getnnz(axis=1) >= 3).all()


@pytest.mark.gpu
def test_amr_poisson_solve_gpu(gpu_lib, oracle):
    """config 5 through the C ABI: the assembled coarse-fine operator installed with cup2d_set_matrix_coo, applied and
    solved on the GPU (sliced-ELL sweeps, block-Jacobi BiCGSTAB of cuda.cu:403-548)"""
    import ctypes
    import scipy.sparse as sp
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    g = AmrBlockGrid(F["blocks"])
    r, c, v = g.poisson_coo()
    n = 64 * g.nblocks
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    vp = ctypes.c_void_p
    with AmrSimulation(g) as s:
        L.check(s.L.cup2d_set_matrix_coo(s._ctx, 0, len(v), r.ctypes.data_as(vp), c.ctypes.data_as(vp), v.ctypes.data_as(vp)), "coo")
        s.set_field(L.PRES, F["pres"])
        s.apply_A(L.TMP, L.PRES)
        assert np.abs(s.get_field(L.TMP) - F["Ax"]).max() < 1e-13
        b = F["Ax"].copy()  # in the range of the singular operator
        s.set_field(L.TMP, b)
        s.set_field(L.PRES, np.zeros_like(b))
        it, rs = ctypes.c_int(), ctypes.c_int()
        e, e0 = ctypes.c_double(), ctypes.c_double()
        L.check(s.L.cup2d_poisson_solve(s._ctx, 1e-9, 0.0, 100, 1000, ctypes.byref(it), ctypes.byref(rs), ctypes.byref(e),
                                        ctypes.byref(e0)), "solve")
        x = s.get_field(L.PRES)
        assert e.value <= 1e-9 and 0 < it.value < 1000
        assert np.abs(b.ravel() - A @ x.ravel()).max() <= 1.05e-9
        d = x - F["pres"]
        assert np.abs(d - d.mean()).max() < 1e-6  # the solution up to the constant


@pytest.mark.gpu
def test_amr_kernels_bit_exact_gpu(gpu_lib, oracle):
    for name, F in _grid_cases(oracle):
        _check_amr_kernels(oracle, name, F)


def _check_amr_kernels(oracle, name, F):
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    if True:
        dt = float(F["dt"])
        with AmrSimulation(AmrBlockGrid(F["blocks"])) as s:
            # tmp -= Lap5(pold) with the flux correction (main.cpp:7022-7027)
            s.set_field(L.POLD, F["pold"])
            s.set_field(L.TMP, F["tmp_in"])
            s.laplacian_sub()
            assert np.array_equal(s.get_field(L.TMP), F["tmp_out"]), name
            # the operator alone: y = A x without the correction equals tmp_in - tmp_out away from coarse blocks' edges
            s.set_field(L.PRES, F["pold"])
            s.apply_A(L.TMP, L.PRES)
            fine_side = np.asarray(s.grid.kind == L.AMR_FINER).any(axis=1)
            assert np.array_equal(s.get_field(L.TMP)[~fine_side], (F["tmp_in"] - F["tmp_out"])[~fine_side]) or \
                np.abs(s.get_field(L.TMP)[~fine_side] - (F["tmp_in"] - F["tmp_out"])[~fine_side]).max() < 1e-13
            # KernelVorticity
            s.set_field(L.VEL, F["vel"])
            s.vorticity()
            assert np.array_equal(s.get_field(L.TMP), F["vort"]), name
            # pressure_rhs with chi / udef and its flux correction (main.cpp:7007-7013)
            s.set_field(L.TMPV, F["udef"])
            s.set_field(L.CHI, F["chi"])
            s.pressure_rhs(dt)
            assert np.array_equal(s.get_field(L.TMP), F["prhs"]), name
            # pressureCorrectionKernel (main.cpp:7178)
            s.set_field(L.PRES, F["pres"])
            s.pressure_correction(dt)
            assert np.array_equal(s.get_field(L.TMPV), F["pcorr"]), name
            # KernelAdvectDiffuse on the interpolated halo-3 tile + its dim-2 flux correction (main.cpp:6611-6617)
            s.nu = float(F["nu"])
            s.set_field(L.VEL, F["vel"])
            s.set_math(True)
            s.advect_diffuse_rhs(dt)
            assert np.array_equal(s.get_field(L.TMPV), F["advdiff"]), name
            s.set_math(False)
            s.advect_diffuse_rhs(dt)
            fast = s.get_field(L.TMPV)
            assert np.abs(fast - F["advdiff"]).max() <= 2e-13 * np.abs(F["advdiff"]).max() + 1e-18, name
            # dt uses the finest cell size (main.cpp:6580-6595)
            umax = np.abs(F["vel"]).max()
            hmin = s.grid.h(F["blocks"][:, 0].max())
            assert s.compute_dt() == oracle.compute_dt(hmin, 1e-3, 0.5, umax)
            return s.grid.nblocks


@pytest.mark.gpu
def test_amr_kernels_bit_exact_at_configs4_scale_gpu(gpu_lib, oracle):
    """The same functor-by-functor comparison on a grid of the size BASELINE.json configs[4] names, built by the REFERENCE's
    own adapt() in a live run on the box (vortex pair, levels 3..9 = up to 4096^2-equivalent, ~39 k blocks on seven
    levels): block indices, neighbour tables and face arrays at a scale the 76..232-block fixtures cannot reach.  Then the
    reference-assembled matrix: its action on the reference's pressure, and a solve with the tile-fused sweeps checked
    against that action."""
    if not oracle.have_reference():
        pytest.skip("reference harness not built")
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = oracle.ref_amr_functors(4, 10, 8, 0.5, 0.1)
    nb = len(F["blocks"])
    levels = np.bincount(F["blocks"][:, 0])
    print("reference grid: %d blocks, per level %s" % (nb, levels.tolist()))
    assert nb > 20000 and (levels > 0).sum() >= 6
    assert _check_amr_kernels(oracle, "live-configs4", F) == nb
    g = AmrBlockGrid(F["blocks"])
    with AmrSimulation(g) as s:
        s.install_poisson_matrix()
        st = s.matrix_stats()
        print("operator:", st)
        s.set_field(L.PRES, F["pres"])
        s.apply_A(L.TMP, L.PRES)
        assert np.abs(s.get_field(L.TMP) - F["Ax"]).max() <= 1e-12 * np.abs(F["Ax"]).max()  # vs the reference-assembled matrix
        b = F["Ax"].copy()
        s.set_field(L.TMP, b)
        s.set_field(L.PRES, np.zeros_like(b))
        info = s.poisson_solve(tol=0.0, rel_tol=1e-6, max_restarts=100, max_iter=3000)
        assert s.last_solver() == "fused" and info["err"] <= 1e-6 * info["err_init"]
        x = s.get_field(L.PRES).copy()
        s.set_field(L.PRES, x)
        s.apply_A(L.TMP, L.PRES)
        assert np.abs(s.get_field(L.TMP) - b).max() <= 1.05 * info["err"] + 1e-12 * np.abs(b).max()


@pytest.mark.gpu
def test_amr_time_step_vs_reference_gpu(gpu_lib, oracle):
    """One whole time step on the adapted grid through the C ABI (cup2d_step: dt, RK2 WENO5 with coarse-fine tiles and
    flux correction, Poisson rhs, BiCGSTAB on the assembled coarse-fine operator, volume-weighted projection) against the
    REFERENCE's own time loop: its state after step 5 goes in, its state after step 6 must come out (the grid is settled
    at 76 blocks on three levels by then, so adapt() changes nothing in the reference's step 6)."""
    if not oracle.have_reference():
        pytest.skip("needs oracle/_ref/ref_harness for the before/after states")
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    # one OpenMP thread: the reference's reductions then have a fixed order and its two runs share their first five steps
    # bit for bit (with several threads they differ in the last digits, and a zero-tolerance BiCGSTAB run occasionally
    # amplifies that to the level of the comparison below)
    kw = dict(level_start=2, level_max=5, rtol=2.0, ctol=0.5, nu=1e-3, max_iter=200, env={"OMP_NUM_THREADS": "1"})
    A = oracle.ref_run_amr(steps=5, **kw)
    B = oracle.ref_run_amr(steps=6, **kw)
    assert np.array_equal(A["blocks"], B["blocks"]) and B["steps"][-1]["blocks"] == len(A["blocks"])
    with AmrSimulation(AmrBlockGrid(A["blocks"]), nu=1e-3, cfl=0.5) as s:
        s.install_poisson_matrix()
        s.set_math(True)
        s.set_field(L.VEL, A["vel"])
        s.set_field(L.PRES, A["pres"])
        r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=200)
        assert abs(r["dt"] - B["steps"][-1]["dt"]) <= 1e-12 * r["dt"]
        assert np.abs(s.get_field(L.VEL) - B["vel"]).max() < 1e-9 * max(1.0, np.abs(B["vel"]).max())
        assert np.abs(s.get_field(L.PRES) - B["pres"]).max() < 1e-8 * max(1.0, np.abs(B["pres"]).max())


@pytest.mark.gpu
def test_block_linf_is_the_tagging_norm_gpu(gpu_lib):
    """cup2d_block_linf (the per-block max norm adapt() tags by, main.cpp:4671-4690) against numpy, bit for bit, on an
    adapted grid and on a uniform one"""
    import ctypes
    import cup2d_amd
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    rng = np.random.default_rng(4)
    with AmrSimulation(AmrBlockGrid(F["blocks"])) as s:
        s.set_field(L.VEL, F["vel"])
        s.vorticity()
        w = s.get_field(L.TMP)
        out = np.empty(s.grid.nblocks)
        ptr = out.ctypes.data_as(ctypes.c_void_p)
        L.check(s.L.cup2d_block_linf(s._ctx, L.TMP, ptr))
        assert np.array_equal(out, np.abs(w).reshape(s.grid.nblocks, -1).max(axis=1))
        assert s.L.cup2d_block_linf(s._ctx, L.VEL, ptr) == -1  # scalar fields only
    with cup2d_amd.Simulation(5, 3) as s:
        a = rng.uniform(-1, 1, (24, 40))
        s.tmp = a
        out = s.block_linf(L.TMP)
        assert out.shape == (15,) and np.isclose(out.max(), np.abs(a).max(), rtol=0, atol=0)
        assert np.array_equal(np.sort(out), np.sort(np.abs(a).reshape(3, 8, 5, 8).max(axis=(1, 3)).ravel()))


@pytest.mark.gpu
def test_amr_regrid_kernels_vs_host_regrid_gpu(gpu_lib):
    """SURVEY.md row a21 on the device (cup2d_amr_regrid_device: k_amr_regrid -- copy / restrict main.cpp:5149-5166 / prolong
    4981-5032 from the tensorial halo-1 tile, side cells by the block operators' closed forms, corner cells from host-built
    descriptors) against the library's host regrid, which test_amr_host_regrid_library_vs_python_statement pins to the
    statement-for-statement BlockLab and test_amr_regrid_vs_reference_adapt to the reference's adapt(): random tags on random
    balanced grids, all five fields of a regrid, bit for bit, every kind of side and corner visited (walls, same level,
    coarser with TestInterp on leaf and averaged coarse cells, finer)."""
    import ctypes
    from cup2d_amd import amr as A, lib as L
    lib = L.load_library()
    names = {"chi": L.CHI, "vel": L.VEL, "vold": L.VOLD, "pres": L.PRES, "pold": L.POLD}
    seen = dict(refine=0, compress=0, regrids=0)
    for seed in range(6):
        rng = np.random.default_rng(300 + seed)
        level_max, l0 = 4 + seed % 3, 1 + seed % 2
        blocks = np.array([(l0, i, j) for j in range(1 << l0) for i in range(1 << l0)], dtype=np.int64)
        data = {k: rng.uniform(-1, 1, (len(blocks), 64 * L.FIELD_DIM[f])) for k, f in names.items()}
        pr, pc = [0.15, 0.3, 0.1][seed % 3], [0.3, 0.4, 0.6][seed % 3]
        for it in range(6):
            nb = len(blocks)
            if nb > 600:
                break
            st = A.validate_states(blocks, rng.choice([0, 1, 2], size=nb, p=[1 - pr - pc, pr, pc]).astype(np.int32), level_max)
            if not (st != A.LEAVE).any():
                continue
            fields = {k: (data[k], L.FIELD_DIM[f], L.FIELD_DIM[f] == 2) for k, f in names.items()}
            b_host, d_host = A.regrid(blocks, st, fields, level_max)
            b32 = np.ascontiguousarray(blocks, dtype=np.int32)
            st32 = np.ascontiguousarray(st, dtype=np.int32)
            flds = np.array(list(names.values()), dtype=np.int32)
            with A.AmrSimulation(A.AmrBlockGrid(blocks)) as old, A.AmrSimulation(A.AmrBlockGrid(b_host)) as new:
                for k, f in names.items():
                    old.set_field(f, data[k])
                    new.set_field(f, np.full_like(d_host[k], np.nan))
                L.check(lib.cup2d_amr_regrid_device(new.ctx_ptr(), old.ctx_ptr(), nb, b32.ctypes.data_as(ctypes.c_void_p), 1, 1, level_max,
                                                    st32.ctypes.data_as(ctypes.c_void_p), len(flds), flds.ctypes.data_as(ctypes.c_void_p)),
                        "amr_regrid_device")
                for k, f in names.items():
                    got = new.get_field(f).reshape(len(b_host), -1)
                    assert not np.isnan(got).any(), (seed, it, k)
                    assert np.array_equal(got, d_host[k]), (seed, it, k, int((got != d_host[k]).any(axis=1).sum()))
                # argument checks: a destination of the wrong size, a source without tables
                assert lib.cup2d_amr_regrid_device(old.ctx_ptr(), old.ctx_ptr(), nb, b32.ctypes.data_as(ctypes.c_void_p), 1, 1, level_max,
                                                   st32.ctypes.data_as(ctypes.c_void_p), 0, None) == L.ERR_ARG
            seen["refine"] += int((st == A.REFINE).sum())
            seen["compress"] += int((st == A.COMPRESS).sum())
            seen["regrids"] += 1
            blocks, data = b_host, d_host
    assert seen["regrids"] >= 12 and seen["refine"] > 100 and seen["compress"] >= 12, seen   # (compress counts blocks whose whole sibling group compresses)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["device", "changed", "host"])
def test_amr_adapt_then_step_gpu(gpu_lib, oracle, route):
    """AmrSimulation.adapt (vorticity tags on the GPU, regrid, new context + operator) lands on the reference's post-adapt
    grid and fields, by every route -- 'device': prolongation / restriction kernels, no field crosses PCIe; 'changed': the
    changed blocks computed on the host; 'host': everything through host memory --; a time step on the new grid then runs
    and stays finite and divergence-reducing"""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    G = golden("amr_adapt.npz")
    with AmrSimulation(AmrBlockGrid(G["pre_blocks"])) as s:
        s.set_field(L.VEL, G["pre_vel"])
        s.set_field(L.PRES, G["pre_pres"])
        assert s.adapt(float(G["rtol"]), float(G["ctol"]), int(G["level_max"]), route=route)
        assert set(map(tuple, s.grid.blocks.tolist())) == set(map(tuple, G["post_blocks"].tolist()))
        ref_v = _by_block(G["post_blocks"], G["post_vel"].reshape(len(G["post_blocks"]), -1))
        ref_p = _by_block(G["post_blocks"], G["post_pres"])
        vel, pres = s.get_field(L.VEL).reshape(s.grid.nblocks, -1), s.get_field(L.PRES)
        for k, b in enumerate(map(tuple, s.grid.blocks.tolist())):
            assert np.array_equal(vel[k], ref_v[b]) and np.array_equal(pres[k], ref_p[b]), b
        r = s.step(tol=1e-9, rel_tol=0.0, max_restarts=100, max_iter=500)
        assert r["dt"] > 0 and r["err"] <= 1e-9 and np.isfinite(s.get_field(L.VEL)).all()
        assert not s.adapt(1e30, 0.0, int(G["level_max"]))  # nothing to do with these thresholds


@pytest.mark.gpu
@pytest.mark.parametrize("lstart,lmax,steps,rtol,ctol,iters,tv,tp,threads", [(2, 5, 6, 2.0, 0.5, 200, 1e-8, 1e-7, 1),
                                                                              (4, 9, 8, 0.5, 0.1, 400, 1e-8, 1e-5, 16)])
def test_amr_run_with_regridding_vs_reference_gpu(gpu_lib, oracle, lstart, lmax, steps, rtol, ctol, iters, tv, tp, threads):
    """BASELINE.json configs[4] end to end on one GPU: from a uniform grid and the analytic vortex pair, passes of
    [adapt(); time step] against the reference's own time loop (ref_harness 'amr': its adapt(), labs, flux correction,
    matrix; solver = CPU restatement of cuda.cu).  Small: level 2, six passes, 16 -> 40 -> 76 blocks on three levels.
    Large: level 4, eight passes, 256 -> 292 -> 868 -> 2 884 -> 10 096 -> 10 132 blocks on six levels (3..8).  Same block
    count and dt at every step, same leaves at the end, fields to the solve tolerance.  The solves run to round-off
    (200 / 400 iterations): iterates of an UNCONVERGED BiCGSTAB solve are not comparable between two summation orders --
    with the bench's fifty iterations the two runs agree to 1e-11 for four steps and then drift to the size of the
    unconverged error itself, exactly as the fused and the five-sweep solver drift from each other
    (tools/gpu_amr_sensitivity.py, tools/gpu_amr_debug_steps.py)."""
    if not oracle.have_reference():
        pytest.skip("needs oracle/_ref/ref_harness")
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    R = oracle.ref_run_amr(level_start=lstart, level_max=lmax, steps=steps, rtol=rtol, ctol=ctol, nu=1e-3, max_iter=iters,
                           env={"OMP_NUM_THREADS": str(threads)})  # 1: fixed reduction order in the reference's run
    g = AmrBlockGrid([(lstart, i, j) for j in range(1 << lstart) for i in range(1 << lstart)])
    x, y = g.cell_centres()
    u, v = np.zeros_like(x), np.zeros_like(x)
    for cx, cy, gam in ((0.35, 0.5, 1.0), (0.65, 0.5, -1.0)):  # ref_harness.cpp inject()
        dx, dy = x - cx, y - cy
        f = gam * np.exp(-(dx * dx + dy * dy) / (0.06 * 0.06)) / 0.06
        u += -dy * f
        v += dx * f
    counts, dts = [], []
    with AmrSimulation(g, nu=1e-3, cfl=0.5) as s:
        s.install_poisson_matrix()
        s.set_math(True)
        s.set_field(L.VEL, np.stack([u, v], axis=-1))
        for k in range(steps):
            dt = s.compute_dt()  # before the regrid, as main.cpp:6579-6603 orders them
            s.adapt(rtol, ctol, lmax)
            counts.append(s.grid.nblocks)
            dts.append(s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=iters, dt=dt)["dt"])
        assert counts == [st["blocks"] for st in R["steps"][1:steps + 1]], (counts, [st["blocks"] for st in R["steps"]])
        assert np.allclose(dts, [st["dt"] for st in R["steps"][1:steps + 1]], rtol=1e-8, atol=0)
        assert set(map(tuple, s.grid.blocks.tolist())) == set(map(tuple, R["blocks"].tolist()))
        ref_v = _by_block(R["blocks"], R["vel"].reshape(len(R["blocks"]), -1))
        ref_p = _by_block(R["blocks"], R["pres"])
        vel, pres = s.get_field(L.VEL).reshape(s.grid.nblocks, -1), s.get_field(L.PRES)
        dv = max(np.abs(vel[k] - ref_v[b]).max() for k, b in enumerate(map(tuple, s.grid.blocks.tolist())))
        dp = max(np.abs(pres[k] - ref_p[b]).max() for k, b in enumerate(map(tuple, s.grid.blocks.tolist())))
        print("blocks per step %s; max|dv| = %.2e (max|v| %.2f), max|dp| = %.2e (max|p| %.2f)"
              % (counts, dv, np.abs(R["vel"]).max(), dp, np.abs(R["pres"]).max()))
        assert dv < tv * max(1.0, np.abs(R["vel"]).max()) and dp < tp * max(1.0, np.abs(R["pres"]).max()), (dv, dp)


@pytest.mark.gpu
def test_cpp_host_driver_amr_run_matches_python_gpu(gpu_lib, tmp_path):
    """the block-AMR flow of the C++ host driver (csrc/cup2d_run.cpp -levelMax: dt, adapt() through the library's host
    routines, context rebuild, operator re-assembly, step) against the same sequence driven from Python: same leaves in
    the same order, same dt and iteration count per step, bit-identical fields"""
    import os
    import subprocess
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps, lstart, lmax, rtol, ctol = 4, 2, 5, 2.0, 0.5
    g = AmrBlockGrid([(lstart, i, j) for j in range(1 << lstart) for i in range(1 << lstart)])
    x, y = g.cell_centres()
    u, v = np.zeros_like(x), np.zeros_like(x)
    for cx, cy, gam in ((0.35, 0.5, 1.0), (0.65, 0.5, -1.0)):
        dx, dy = x - cx, y - cy
        f = gam * np.exp(-(dx * dx + dy * dy) / (0.06 * 0.06)) / 0.06
        u += -dy * f
        v += dx * f
    vel0 = np.stack([u, v], axis=-1)
    init = str(tmp_path / "vel0.f64")
    np.ascontiguousarray(vel0).tofile(init)
    r = subprocess.run([os.path.join(root, "cup2d_amd", "cup2d_run"), "-levelStart", str(lstart), "-levelMax", str(lmax), "-Rtol", str(rtol),
                        "-Ctol", str(ctol), "-steps", str(steps), "-maxiter", "200", "-math", "strict", "-init", init, "-state",
                        str(tmp_path / "cpp")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l.split() for l in r.stdout.decode().splitlines() if l.startswith("step ")]
    assert len(lines) == steps
    with AmrSimulation(g, nu=1e-3, cfl=0.5) as s:
        s.install_poisson_matrix()
        s.set_math(True)
        s.set_field(L.VEL, vel0)
        for k in range(steps):
            dt = s.compute_dt()
            s.adapt(rtol, ctol, lmax)
            info = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=200, dt=dt)
            assert float(lines[k][5]) == dt and int(lines[k][7]) == info["iters"] and int(lines[k][11]) == s.grid.nblocks, (k, lines[k], info)
        blocks = np.fromfile(str(tmp_path / "cpp.blocks.i32"), dtype=np.int32).reshape(-1, 3)
        assert np.array_equal(blocks, s.grid.blocks) and len(blocks) > 16
        assert np.array_equal(np.fromfile(str(tmp_path / "cpp.vel.f64")).reshape(len(blocks), 64, 2), s.get_field(L.VEL))
        assert np.array_equal(np.fromfile(str(tmp_path / "cpp.pres.f64")).reshape(len(blocks), 64), s.get_field(L.PRES))


@pytest.mark.gpu
def test_amr_rk_stages_one_by_one_equal_the_rk2_entry_gpu(gpu_lib):
    """cup2d_advect_diffuse_stage on an adapted grid (the reference's un-fused stage, main.cpp:6607-6642): stage 1 leaves vel in
    VOLD and the mid-point velocity in VEL, stages 1 + 2 = cup2d_advect_diffuse_rk2 bit for bit"""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    dt = float(F["dt"])
    for strict in (True, False):
        with AmrSimulation(AmrBlockGrid(F["blocks"]), nu=float(F["nu"])) as s:
            s.set_math(strict)
            s.set_field(L.VEL, F["vel"])
            L.check(s.L.cup2d_advect_diffuse_rk2(s._ctx, s.nu, dt), "rk2")
            want = s.get_field(L.VEL)
            s.set_field(L.VEL, F["vel"])
            L.check(s.L.cup2d_advect_diffuse_stage(s._ctx, s.nu, dt, 1, L.BLOCKS_ALL), "stage 1")
            assert np.array_equal(s.get_field(L.VOLD), F["vel"])
            mid = s.get_field(L.VEL)
            assert not np.array_equal(mid, F["vel"]) and not np.array_equal(mid, want)
            L.check(s.L.cup2d_advect_diffuse_stage(s._ctx, s.nu, dt, 2, L.BLOCKS_ALL), "stage 2")
            assert np.array_equal(s.get_field(L.VEL), want)


@pytest.mark.gpu
def test_amr_unsupported_entry_points_say_so(gpu_lib):
    import ctypes
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    with AmrSimulation(AmrBlockGrid(F["blocks"])) as s:
        assert s.L.cup2d_advect_diffuse_stage(s._ctx, 1e-3, 1e-3, 3, L.BLOCKS_ALL) == -1  # a stage that does not exist
        assert s.L.cup2d_advect_diffuse_stage(s._ctx, 1e-3, 1e-3, 1, L.BLOCKS_INNER) == -1  # a whole stage takes all blocks
        assert s.L.cup2d_laplacian_sub(s._ctx, 7) == -1  # no such phase


@pytest.mark.gpu
def test_amr_block_operators_in_two_phases_gpu(gpu_lib):
    """computeA's inner / halo split (main.cpp:3035-3057) on an adapted grid through the public phases: CUP2D_BLOCKS_INNER (the
    functor on the blocks that read no ghost block; on one rank: all of them) followed by CUP2D_BLOCKS_HALO (the functor on the
    others + the flux correction of all blocks) leaves what CUP2D_BLOCKS_ALL leaves, bit for bit -- and that is the reference's
    functor (golden vectors).  The N-rank form of the same (ghost copies travelling in between) is in tests/dist_worker.py."""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    dt = float(F["dt"])
    with AmrSimulation(AmrBlockGrid(F["blocks"]), nu=float(F["nu"])) as s:
        s.set_math(True)
        s.set_field(L.POLD, F["pold"])
        s.set_field(L.TMP, F["tmp_in"])
        L.check(s.L.cup2d_laplacian_sub(s._ctx, L.BLOCKS_INNER), "inner")
        assert not np.array_equal(s.get_field(L.TMP), F["tmp_out"])   # the flux correction is still missing
        L.check(s.L.cup2d_laplacian_sub(s._ctx, L.BLOCKS_HALO), "halo")
        assert np.array_equal(s.get_field(L.TMP), F["tmp_out"])
        s.set_field(L.VEL, F["vel"])
        s.set_field(L.TMPV, F["udef"])
        s.set_field(L.CHI, F["chi"])
        for ph in (L.BLOCKS_INNER, L.BLOCKS_HALO):
            L.check(s.L.cup2d_pressure_rhs(s._ctx, dt, 1, ph), "pressure_rhs")
        assert np.array_equal(s.get_field(L.TMP), F["prhs"])
        for ph in (L.BLOCKS_INNER, L.BLOCKS_HALO):
            L.check(s.L.cup2d_vorticity(s._ctx, ph), "vorticity")
        assert np.array_equal(s.get_field(L.TMP), F["vort"])
        s.set_field(L.PRES, F["pres"])
        for ph in (L.BLOCKS_INNER, L.BLOCKS_HALO):
            L.check(s.L.cup2d_pressure_correction(s._ctx, dt, ph), "pressure_correction")
        assert np.array_equal(s.get_field(L.TMPV), F["pcorr"])
        for ph in (L.BLOCKS_INNER, L.BLOCKS_HALO):
            L.check(s.L.cup2d_advect_diffuse_rhs(s._ctx, s.nu, dt, ph), "advect_diffuse_rhs")
        assert np.array_equal(s.get_field(L.TMPV), F["advdiff"])


@pytest.mark.gpu
def test_amr_solve_without_an_installed_operator_assembles_its_own_gpu(gpu_lib):
    """cup2d_poisson_solve on an adapted grid whose caller installed no matrix: the library assembles the operator of
    main.cpp:7034-7112 from the topology tables -- the same solve, bit for bit, as after an explicit install of the triplets"""
    import ctypes
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    F = golden("amr_functors.npz")
    G = AmrBlockGrid(F["blocks"])
    b = np.random.default_rng(3).uniform(-1, 1, (G.nblocks, 64))
    b -= b.mean()
    out = []
    for explicit in (False, True):
        with AmrSimulation(G) as s:
            if explicit:
                s.install_poisson_matrix(via_triplets=True)
            s.set_field(L.TMP, b)
            s.set_field(L.PRES, np.zeros((G.nblocks, 64)))
            it, err = ctypes.c_int(), ctypes.c_double()
            L.check(s.L.cup2d_poisson_solve(s._ctx, 0.0, 0.0, 100, 40, ctypes.byref(it), None, ctypes.byref(err), None), "poisson_solve")
            assert it.value == 40 and np.isfinite(err.value)
            out.append((s.get_field(L.PRES), err.value))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def _penalize_numpy(vel, CHI, blocks_idx, origin, h, chi, udef, centre, lam, dt):
    """main.cpp:6643-7006 for one body, statement by statement in numpy (same operation order, no contraction):
    returns (uvw, moments, vel after the blend, tmpV)"""
    lamdt = lam * dt
    iy, ix = np.divmod(np.arange(64), 8)
    q = np.zeros(7)
    terms = []
    for k, b in enumerate(blocks_idx):  # the reference's order: block, iy, ix
        X = chi[k]
        V, U = vel[b], udef[k]
        ud0, ud1 = V[:, 0] - U[:, 0], V[:, 1] - U[:, 1]
        Xl = np.where(X >= 0.5, lamdt, 0.0)
        F = h[b] * h[b] * Xl / (1 + Xl)
        p0 = origin[k, 0] + h[b] * (ix + 0.5) - centre[0]
        p1 = origin[k, 1] + h[b] * (iy + 0.5) - centre[1]
        t = np.stack([F, F * (p0 * p0 + p1 * p1), F * p0, F * p1, F * ud0, F * ud1, F * (p0 * ud1 - p1 * ud0)], axis=1)
        t[X <= 0] = 0.0
        terms.append(t)
    for t in terms:
        for j in range(64):
            q += t[j]
    PM, PJ, PX, PY = q[:4]
    A = np.array([[PM, 0, -PY], [0, PM, PX], [-PY, PX, PJ]])
    uvw = np.linalg.solve(A, q[4:])  # the solve itself is pinned by the bit-exact loop test (tests/test_spmat_gpu.py)
    return q, terms


@pytest.mark.gpu
def test_penalisation_kernels_on_an_adapted_grid_gpu(gpu_lib, oracle):
    """cup2d_body_momentum / cup2d_penalize on a three-level grid (cell size per block from the AMR tables): moments and
    fields equal the numpy statement of main.cpp:6643-7006 bit for bit; the 3 x 3 solve to round-off"""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    name, F = next(_grid_cases(oracle))  # the golden three-level grid
    g = AmrBlockGrid(F["blocks"])
    nb = g.nblocks
    rng = np.random.default_rng(11)
    h = g.h0 / (1 << g.blocks[:, 0])
    org = np.stack([g.blocks[:, 1] * 8 * h, g.blocks[:, 2] * 8 * h], axis=1)
    touched = np.sort(rng.choice(nb, size=nb // 3, replace=False))
    chi = rng.uniform(-0.2, 1.0, (len(touched), 64))      # <= 0, (0, 0.5), >= 0.5 all occur
    udef = 0.3 * rng.uniform(-1, 1, (len(touched), 64, 2))
    CHI = rng.uniform(0, 1, (nb, 64))                      # the field chi: sometimes above, sometimes below the body's
    vel = rng.uniform(-1, 1, (nb, 64, 2))
    centre, lam, dt = (0.47, 0.52), 1e7, 1e-3
    q_ref, _ = _penalize_numpy(vel, CHI, touched, org[touched], h, chi, udef, centre, lam, dt)
    with AmrSimulation(g) as s:
        s.set_field(L.VEL, vel)
        s.set_field(L.CHI, CHI)
        s.body_set(0, touched, org[touched], chi, udef, centre)
        uvw, q = s.body_momentum(0, lam, dt)
        assert np.array_equal(q, q_ref), name
        A = np.array([[q[0], 0, -q[3]], [0, q[0], q[2]], [-q[3], q[2], q[1]]])
        assert np.allclose(A @ uvw, q[4:], rtol=1e-12, atol=1e-14 * np.abs(q[4:]).max())
        s.penalize(lam, dt, uvw)
        # main.cpp:6944-6978 / 6979-7006, elementwise
        iy, ix = np.divmod(np.arange(64), 8)
        v_ref, t_ref = vel.copy(), np.zeros_like(vel)
        for k, b in enumerate(touched):
            X = chi[k]
            p0 = org[b, 0] + h[b] * (ix + 0.5) - centre[0]
            p1 = org[b, 1] + h[b] * (iy + 0.5) - centre[1]
            alpha = np.where(X > 0.5, 1 / (1 + lam * dt), 1.0)
            US = uvw[0] - uvw[2] * p1 + udef[k, :, 0]
            VS = uvw[1] + uvw[2] * p0 + udef[k, :, 1]
            on = ~(CHI[b] > X) & ~(X <= 0)
            v_ref[b, on, 0] = (alpha * vel[b, :, 0] + (1 - alpha) * US)[on]
            v_ref[b, on, 1] = (alpha * vel[b, :, 1] + (1 - alpha) * VS)[on]
            dom = ~(X < CHI[b])
            t_ref[b, dom] += udef[k, dom]
        assert np.array_equal(s.get_field(L.VEL), v_ref), name
        assert np.array_equal(s.get_field(L.TMPV), t_ref), name
        s.body_clear()
        s.penalize(lam, dt, np.zeros((0, 3)))
        assert not s.get_field(L.TMPV).any()


def _circle_grid(lfine):
    from cup2d_amd.amr import circle_band_grid
    return circle_band_grid(lfine)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["golden", "circle6", "circle7"])
def test_amr_tile_fused_solver_on_the_hybrid_operator_gpu(gpu_lib, which):
    """The assembled coarse-fine operator takes the tile-fused sweeps (krylov_fused.hip HYB + k_hyb_rows): tiles of plain
    blocks keep z on the chip, the rows of the other tiles are applied from z in memory.  Same recurrences as the five
    sweeps (cuda.cu:403-548): after four iterations at zero tolerance the iterates agree to round-off -- every row of every
    kind of tile has then been applied eight times by either organisation -- and a converged solve satisfies the
    reference's criterion against the matrix itself."""
    import scipy.sparse as sp
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    g = AmrBlockGrid(golden("amr_functors.npz")["blocks"]) if which == "golden" else _circle_grid(int(which[-1]))
    r, c, v = g.poisson_coo()
    n = 64 * g.nblocks
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    rng = np.random.default_rng(5)
    xt = rng.uniform(-1, 1, n)
    b = (A @ xt).reshape(g.nblocks, 8, 8)  # in the range of the singular operator
    with AmrSimulation(g) as s:
        s.install_poisson_matrix()
        st = s.matrix_stats()
        print(which, g.nblocks, "blocks:", st)
        assert 0 < st["general_tile_blocks"] and st["plain_blocks"] < g.nblocks
        if which == "circle7":  # tiles of plain blocks exist: z stays on the chip there
            assert st["general_tile_blocks"] < 0.6 * g.nblocks
        out = {}
        # ("fused", True, "eab"): the two-launch organisation, on request (k_edge HYB: sweep E with the next A+B, C+D' with the sums of the
        # next beginning; rho' and ||r'||^2 from those sums: round-off apart from the others, like the uniform grid's)
        for kind, fin, form in (("sweeps", False, "auto"), ("fused", False, "auto"), ("fused", True, "auto"), ("fused", True, "eab")):
            s.set_solver(fused=kind == "fused", finish_in_kernel=fin, form=form)
            s.set_field(L.TMP, b)
            s.set_field(L.PRES, np.zeros_like(b))
            info = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=4)
            assert s.last_solver() == kind and info["iters"] == 4
            if kind == "fused":
                assert s.last_solver_form()[0] == ("eab" if form == "eab" else "full"), s.last_solver_form()
            out[(kind, fin, form)] = (s.get_field(L.PRES).copy(), info)
        xs, is_ = out[("sweeps", False, "auto")]
        for key in (("fused", False, "auto"), ("fused", True, "auto"), ("fused", True, "eab")):
            xf, if_ = out[key]
            assert abs(if_["err"] - is_["err"]) <= 1e-12 * max(1.0, is_["err_init"]), (key, if_, is_)
            assert np.abs(xf - xs).max() <= 1e-12 * max(1.0, np.abs(xs).max()), key
        assert np.array_equal(out[("fused", False, "auto")][0], out[("fused", True, "auto")][0])  # the finish in the kernel: same order
        print(which, "two launches vs five sweeps:", np.abs(out[("fused", True, "eab")][0] - xs).max())
        for form in ("auto", "eab"):
            s.set_solver(fused=True, finish_in_kernel=True, form=form)
            s.set_field(L.TMP, b)
            s.set_field(L.PRES, np.zeros_like(b))
            info = s.poisson_solve(tol=1e-9, max_restarts=100, max_iter=2000)
            x = s.get_field(L.PRES)
            print(which, form, info)
            assert info["err"] <= 1e-9 and 0 < info["iters"] < 2000
            assert np.abs(b.ravel() - A @ x.ravel()).max() <= 1.05e-9
            d = x.ravel() - xt
            assert np.abs(d - d.mean()).max() < 1e-5  # the solution up to the constant (residual 1e-9 times the conditioning)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["golden", "circle7"])
def test_amr_operator_installed_from_the_tables_equals_the_triplet_route_gpu(gpu_lib, which):
    """cup2d_amr_install_poisson builds rows only for the blocks with a coarse-fine side; the operator is the one
    cup2d_amr_poisson_coo -> cup2d_set_matrix_coo installs: same split into plain blocks and stored entries, same product
    and same solve, bit for bit"""
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    g = AmrBlockGrid(golden("amr_functors.npz")["blocks"]) if which == "golden" else _circle_grid(7)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (g.nblocks, 8, 8))
    out = {}
    with AmrSimulation(g) as s:
        for via in (True, False):
            s.install_poisson_matrix(via_triplets=via)
            st = s.matrix_stats()
            s.set_field(L.PRES, x)
            s.apply_A(L.TMP, L.PRES)
            ax = s.get_field(L.TMP).copy()
            s.set_field(L.TMP, ax)
            s.set_field(L.PRES, np.zeros_like(x))
            info = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=6)
            out[via] = (st, ax, s.get_field(L.PRES).copy(), info)
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1]) and np.array_equal(out[True][2], out[False][2])
    assert out[True][3] == out[False][3]


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["device", "changed"])
def test_amr_adapt_at_configs4_scale_vs_reference_gpu(gpu_lib, oracle, route):
    """AmrSimulation.adapt() -- tags from max|vorticity| per block on the GPU, the library's state validation,
    prolongation / restriction ('device': k_amr_regrid between the old and the new context, row a21 as kernels; 'changed':
    the threaded host routine on the blocks the plan names), new context, operator from the tables -- against the
    reference's own adapt() (main.cpp:4657-5440) on the ~39 k-block, seven-level grid of a live run: same leaves
    afterwards, all five fields bit for bit."""
    if not oracle.have_reference():
        pytest.skip("reference harness not built")
    from cup2d_amd import lib as L
    from cup2d_amd.amr import AmrBlockGrid, AmrSimulation
    pre, post = oracle.ref_amr_adapt(4, 10, 8, 0.5, 0.1)
    names = {"chi": L.CHI, "vel": L.VEL, "vold": L.VOLD, "pres": L.PRES, "pold": L.POLD}
    with AmrSimulation(AmrBlockGrid(pre["blocks"])) as s:
        for k, f in names.items():
            s.set_field(f, pre[k])
        assert s.adapt(0.5, 0.1, 10, route=route)
        blocks = s.grid.blocks
        print("reference adapt: %d -> %d blocks; here %d" % (len(pre["blocks"]), len(post["blocks"]), len(blocks)))
        assert len(blocks) == len(post["blocks"]) != len(pre["blocks"])
        assert set(map(tuple, blocks.tolist())) == set(map(tuple, post["blocks"].tolist()))
        for k, f in names.items():
            ref = _by_block(post["blocks"], post[k].reshape(len(post["blocks"]), -1))
            mine = _by_block(blocks, s.get_field(f).reshape(len(blocks), -1))
            assert all(np.array_equal(ref[b], mine[b]) for b in ref), k
        assert s.matrix_stats()["plain_blocks"] > 0  # the operator of the new grid is installed
        r = s.step(max_iter=20)
        assert np.isfinite(r["err"]) and r["dt"] > 0
