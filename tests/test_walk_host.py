"""CPU tests of the quad ("register walk") form of the WENO5 advect-diffuse kernel (cup2d_amd/csrc/advect_walk.h).

The per-lane functions the GPU kernel k_advect_walk executes are host+device code; tests/walk_emul.cpp runs them lane
by lane on the CPU.  Here that emulation is compared with the oracle (FAST tolerance of tests/test_gpu_parity.py:
2e-13 of max|rhs|, 1e-13 of max|vel| after the RK2 stages), on grids that take every code path: one-sided upwinding
(plus / minus only), mixed signs, domain walls on every side, rectangular grids, ghost blocks, and block orders in which
not every block finds partners (the plan's `singles`)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "walk_emul.cpp")
HDR = os.path.join(ROOT, "cup2d_amd", "csrc", "advect_walk.h")
HDR2 = os.path.join(ROOT, "cup2d_amd", "csrc", "ranges.h")
SO = os.path.join(ROOT, "tests", "_walk_emul.so")
_i32 = ctypes.POINTER(ctypes.c_int32)
_dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, SRC])
    lib = ctypes.CDLL(SO)
    lib.walk_emul_plan.argtypes = [_i32, ctypes.c_int, ctypes.c_int, _i32, _i32, ctypes.POINTER(ctypes.c_int)]
    lib.walk_emul_run.argtypes = [_i32, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    lib.walk_emul_cursor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32]
    lib.walk_emul_cursor.restype = ctypes.c_longlong
    return lib


def _chunked_grid(count, chunk, wpg=4):   # ctx.h chunked_grid
    groups = (count + wpg - 1) // wpg
    return 8 * (((groups + 7) // 8 + chunk - 1) // chunk)


def _persistent_grid(count, per_cu, wpg=4, cus=256):   # api.hip resident_grid
    groups = (count + wpg - 1) // wpg
    g = min(per_cu * cus, groups)
    if g >= 8:
        g -= g % 8
    return max(g, 1)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 31, 33, 64, 100, 767 * 4 + 1, 4096, 65536, 262144])
def test_quad_loop_never_leaves_the_plan_table(emul, nq):
    """The index sequence of k_advect_walk's loop (ranges.h WaveCursor, the code the kernel runs) for every workgroup and
    wave of a launch: every quad exactly once, and no table read -- the read-ahead past a wave's last quad included -- at
    or beyond entry nq.  nq = 65 536 / 262 144 are the plans of 4096^2 / 8192^2 (round 2's kernel read up to
    stride * 192 bytes past the table there)."""
    visits = np.zeros(nq, dtype=np.int32)
    grids = {(_persistent_grid(nq, pc), 0) for pc in (1, 3, 4, 8)} | {(1, 0), (7, 0), (13, 0)}
    grids |= {(_chunked_grid(nq, ch), ch) for ch in (1, 3, 16)}
    for G, chunk in sorted(grids):
        worst = emul.walk_emul_cursor(nq, 4, G, chunk, visits.ctypes.data_as(_i32))
        assert 0 <= worst < nq, (nq, G, chunk, worst)


def plan(lib, nbr, first, count):
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    quads = np.full((count // 4 + 1, 12), -99, dtype=np.int32)
    singles = np.full(count + 1, -99, dtype=np.int32)
    ns = ctypes.c_int(0)
    nq = lib.walk_emul_plan(nbr.ctypes.data_as(_i32), first, count, quads.ctypes.data_as(_i32), singles.ctypes.data_as(_i32), ctypes.byref(ns))
    return quads[:nq].copy(), singles[:ns.value].copy()


def run(lib, quads, vel_slab, vold_slab, out_slab, mode, oldlab, afc, dfc):
    lib.walk_emul_run(quads.ctypes.data_as(_i32), len(quads), vel_slab.ctypes.data_as(_dp), vold_slab.ctypes.data_as(_dp),
                      out_slab.ctypes.data_as(_dp), mode, oldlab, afc, dfc)


def rhs_and_rk2(lib, grid, vel, h, nu, dt):
    """the functor (MODE 0) and both RK2 stages (MODE 1) through the emulation, as launch_advect drives them"""
    quads, singles = plan(lib, grid.nbr, 0, grid.nblocks)
    assert len(singles) == 0
    afac, dfac = -dt * h, nu * dt
    v = grid.to_blocks(vel)
    rhs = np.full_like(v, np.nan)
    run(lib, quads, v, v, rhs, 0, 0, afac, dfac)
    mid = np.full_like(v, np.nan)
    c1 = 0.5 / (h * h)
    run(lib, quads, v, v, mid, 1, 1, c1 * afac, c1 * dfac)          # stage 1: old is the tile itself
    new = v.copy()
    c2 = 1.0 / (h * h)
    run(lib, quads, mid, new, new, 1, 0, c2 * afac, c2 * dfac)      # stage 2: in place on vel, tile from mid
    return grid.from_blocks(rhs, 2), grid.from_blocks(new, 2)


@pytest.mark.parametrize("nx,ny,noise,shift", [
    (64, 64, 1e-3, (0.0, 0.0)),     # Taylor-Green: all four sign combinations, most tiles one-sided
    (32, 32, 0.5, (0.0, 0.0)),      # noise: mixed signs in every tile (the two-sided walk)
    (64, 32, 0.2, (0.0, 0.0)),      # rectangular
    (32, 32, 0.2, (3.0, 3.0)),      # u, v > 0 everywhere: plus only
    (32, 32, 0.2, (-3.0, -3.0)),    # minus only
    (32, 64, 0.2, (3.0, -3.0)),
    (16, 16, 0.3, (0.0, 0.0)),      # one quad, walls on all four sides
])
def test_quad_walk_equals_the_oracle(emul, oracle, nx, ny, noise, shift):
    from cup2d_amd.grid import BlockGrid
    vel = oracle.taylor_green(nx, noise=noise, seed=nx + ny, ny=ny)
    vel[..., 0] += shift[0]
    vel[..., 1] += shift[1]
    h, nu = 1.0 / max(nx, ny), 1e-3
    dt = oracle.compute_dt(h, nu, 0.5, np.abs(vel).max())
    ref = oracle.advect_diffuse_rhs(vel, h, nu, dt)
    ref2, _ = oracle.rk2_advect_diffuse(vel, h, nu, dt)
    grid = BlockGrid(nx // 8, ny // 8)
    rhs, new = rhs_and_rk2(emul, grid, vel, h, nu, dt)
    assert np.isfinite(rhs).all() and np.isfinite(new).all()
    assert np.abs(rhs - ref).max() <= 2e-13 * np.abs(ref).max()
    assert np.abs(new - ref2).max() <= 1e-13 * np.abs(ref2).max()


def check_plan(grid_nbr, nowned, first, count, quads, singles):
    """every block of the range exactly once; quads are 2 x 2 patches with the right surroundings"""
    seen = np.concatenate([quads[:, :4].ravel(), singles])
    assert sorted(seen.tolist()) == list(range(first, first + count))
    W, E, S, N = 0, 1, 2, 3
    for q in quads:
        sw, se, nw, ne = q[:4]
        assert grid_nbr[sw, E] == se and grid_nbr[nw, E] == ne and grid_nbr[sw, N] == nw and grid_nbr[se, N] == ne
        owner = [sw, nw, se, ne, sw, se, nw, ne]
        side = [W, W, E, E, S, S, N, N]
        for k in range(8):
            nb = grid_nbr[owner[k], side[k]]
            assert q[4 + k] == (nb if nb >= 0 else -1 - owner[k])


def test_plan_on_hilbert_rowmajor_and_decomposed_grids(emul):
    from cup2d_amd.grid import BlockGrid
    g = BlockGrid(16, 16)
    quads, singles = plan(emul, g.nbr, 0, g.nblocks)
    assert len(singles) == 0 and len(quads) == 64
    assert all(sorted(q[:4].tolist()) == list(range(4 * i, 4 * i + 4)) for i, q in enumerate(quads))  # aligned runs
    check_plan(g.nbr, g.nblocks, 0, g.nblocks, quads, singles)
    g = BlockGrid(8, 6, order="rowmajor")
    quads, singles = plan(emul, g.nbr, 0, g.nblocks)
    assert len(singles) == 0 and len(quads) == 12
    check_plan(g.nbr, g.nblocks, 0, g.nblocks, quads, singles)
    g = BlockGrid(5, 3, order="rowmajor")  # odd sizes: a column and a row of singles
    quads, singles = plan(emul, g.nbr, 0, g.nblocks)
    assert len(quads) == 2 and len(singles) == 7
    check_plan(g.nbr, g.nblocks, 0, g.nblocks, quads, singles)
    # a rank's patch of a decomposed grid: inner blocks first, halo ring last, ghost blocks outside
    g = BlockGrid(8, 8, ghost_sides=(True, True, False, True))
    for first, count in ((0, g.n_inner), (g.n_inner, g.nblocks - g.n_inner), (0, g.nblocks)):
        quads, singles = plan(emul, g.nbr, first, count)
        check_plan(g.nbr, g.nblocks, first, count, quads, singles)
    quads, singles = plan(emul, g.nbr, 0, g.nblocks)
    assert len(singles) == 0


def test_quad_walk_with_ghost_blocks_and_phases(emul, oracle):
    """a rank's patch with ghost blocks on two sides, inner and halo phases run separately: the owned cells equal
    the oracle on the enclosing grid"""
    from cup2d_amd.grid import BlockGrid
    nb = 8
    big = oracle.taylor_green(8 * (nb + 2), noise=0.3, seed=3)
    h, nu, dt = 1.0 / (8 * (nb + 2)), 1e-3, 1e-3
    ref = oracle.advect_diffuse_rhs(big, h, nu, dt)[8:-8, 8:-8]
    g = BlockGrid(nb, nb, ghost_sides=(True, True, True, True))
    slab = np.zeros((g.nblocks + g.nghost, 128))
    slab[:g.nblocks] = g.to_blocks(big[8:-8, 8:-8])
    for k, (side, pos) in enumerate(g.ghost_coords):
        bx = {0: 0, 1: nb + 1}.get(side, pos + 1)
        by = {2: 0, 3: nb + 1}.get(side, pos + 1)
        slab[g.nblocks + k] = big[8 * by:8 * by + 8, 8 * bx:8 * bx + 8].reshape(128)
    out = np.full((g.nblocks, 128), np.nan)
    for first, count in ((0, g.n_inner), (g.n_inner, g.nblocks - g.n_inner)):
        quads, singles = plan(emul, g.nbr, first, count)
        assert len(singles) == (0 if first == 0 else count)  # 6 x 6 inner blocks: nine quads; the ring is one block thick
        run(emul, quads, slab, slab, out, 0, 0, -dt * h, nu * dt)
        out[singles] = g.to_blocks(ref)[singles]  # the per-block kernel's share
    got = g.from_blocks(out, 2)
    assert np.abs(got - ref).max() <= 2e-13 * np.abs(ref).max()
    # all owned blocks as one range: the ring pairs up with the inner blocks
    quads, singles = plan(emul, g.nbr, 0, g.nblocks)
    assert len(singles) == 0
    out[:] = np.nan
    run(emul, quads, slab, slab, out, 0, 0, -dt * h, nu * dt)
    assert np.abs(g.from_blocks(out, 2) - ref).max() <= 2e-13 * np.abs(ref).max()


def test_fused_sweep_index_logic_emulation(oracle):
    """tools/emulate_fused.py: the index logic of k_fused (ring classification, staging tile, the MFMA fragment maps, the
    ring job on the 32 edge columns of P_inv and its compact write-back, edge gathers, stencil) lane by lane in numpy
    against the oracle's y = A P_inv v, on Hilbert and row-major grids incl. partial tiles -- a wrong index shows up here
    without a GPU"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("emulate_fused", os.path.join(ROOT, "tools", "emulate_fused.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_edge_form_index_logic_emulation(oracle):
    """tools/emulate_edge.py: the index logic of k_edge (csrc/krylov_edge.h: rounds of 8 sibling tiles, slot classification,
    export slots by prefix popcount, the consumers' look-up in a sibling's export, edge products, epilogue y = v + ghosts)
    lane by lane in numpy against the oracle's y = A P_inv v, sharing on and off, Hilbert and row-major grids incl. partial
    tiles and rounds"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("emulate_edge", os.path.join(ROOT, "tools", "emulate_edge.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
