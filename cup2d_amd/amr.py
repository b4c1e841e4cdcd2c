"""Block-AMR grids on the host (BASELINE.json configs[4]): which leaf blocks exist on which level and what lies
across every side of each -- the tables of cup2d_set_amr (include/cup2d_hip.h).  Replaces, for this purpose, the
reference's tree / Info::Znei / Zchild / Zparent lookups (main.cpp:672-738, 2197-2198) with dense arrays built once
per regrid.  Fields of an adapted grid are per-block arrays [nblocks][64 * dim] in the order of `blocks`."""
import ctypes

import numpy as np

from . import lib as _l

BS = 8


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class AmrBlockGrid:
    """blocks: (nb, 3) int array of leaf blocks (level, i, j) of a bpdx x bpdy base grid (level-0 blocks), 2:1
    balanced (the reference's adapt() guarantees it, main.cpp:4734-4861).  h0 = extent / max(bpdx, bpdy) / 8
    (main.cpp:6338)."""

    def __init__(self, blocks, bpdx=1, bpdy=1, extent=1.0):
        self.blocks = np.ascontiguousarray(blocks, dtype=np.int64).reshape(-1, 3)
        self.nblocks = len(self.blocks)
        self.bpdx, self.bpdy = int(bpdx), int(bpdy)
        self.h0 = float(extent) / max(self.bpdx, self.bpdy) / BS
        nb = self.nblocks
        self.level = np.ascontiguousarray(self.blocks[:, 0], dtype=np.int32)
        self.kind = np.zeros((nb, 4), dtype=np.int32)
        self.nbr2 = -np.ones((nb, 4, 2), dtype=np.int32)
        self.half = np.zeros((nb, 4), dtype=np.int32)
        b32 = np.ascontiguousarray(self.blocks, dtype=np.int32)
        L = _l.load_library()
        rc = L.cup2d_amr_tables(nb, _p(b32), self.bpdx, self.bpdy, _p(self.kind), _p(self.nbr2), _p(self.half))
        if rc != 0:
            raise ValueError(L.cup2d_last_error().decode())
        # same-level neighbour table for cup2d_create (sides that are not same-level: wall)
        self.nbr = np.where(self.kind == _l.AMR_SAME, self.nbr2[:, :, 0], -1).astype(np.int32)
        self.nghost, self.n_inner = 0, nb

    def h(self, level):
        return self.h0 / (1 << int(level))

    # ---- Poisson matrix of main.cpp:7034-7112 on this adapted grid -------------------------------
    def poisson_coo(self):
        """COO triplets (row, col, val) of the matrix the reference assembles on this grid, assembled by the library's
        host routine cup2d_amr_poisson_coo (C++, include/cup2d_hip.h); poisson_coo_py is the same algorithm in Python."""
        L = _l.load_library()
        vp = ctypes.c_void_p
        tabs = [np.ascontiguousarray(a, dtype=np.int32) for a in (self.kind, self.nbr2, self.half)]
        ptr = [a.ctypes.data_as(vp) for a in tabs]
        nnz = L.cup2d_amr_poisson_coo(self.nblocks, *ptr, 0, None, None, None)
        if nnz < 0:
            _l.check(int(nnz), "amr_poisson_coo")
        r, c, v = np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.float64)
        got = L.cup2d_amr_poisson_coo(self.nblocks, *ptr, nnz, r.ctypes.data_as(vp), c.ctypes.data_as(vp), v.ctypes.data_as(vp))
        if got != nnz:
            _l.check(int(min(got, -1)), "amr_poisson_coo")
        return r, c, v

    def poisson_coo_py(self):
        """COO triplets (row, col, val) of the matrix the reference assembles (rows/columns numbered 64 * block + 8 * iy
        + ix in the order of `blocks`): 5-point rows inside a block (main.cpp:7075-7087); on block-edge cells, per side:
        nothing at a domain wall, +1/-1 towards a same-level neighbour, and across coarse-fine faces the reference's
        interpolated fluxes (Solver::makeFlux / interpolate / D1 / D2, main.cpp:5915-5997): weights 2/3, -1/5, 8/15 on
        the two fine cells and the coarse cell plus the Taylor corrections along the face.  Duplicate columns of a row
        are summed as SpRowInfo::mapColVal does (cuda.h:1-24).  Host-side, regrid-time code."""
        rows = {}

        def add(r, c, v):
            d = rows.setdefault(r, {})
            d[c] = d.get(c, 0.0) + v

        def cell(b, ix, iy):
            return 64 * b + 8 * iy + ix

        def d1(b, s, ix, iy):
            t = iy if s < 2 else ix  # coordinate along the face
            nei = (lambda d: cell(b, ix, iy + d)) if s < 2 else (lambda d: cell(b, ix + d, iy))
            if t in (7, 3):
                return [(nei(-2), 1. / 8.), (nei(-1), -1. / 2.), (cell(b, ix, iy), 3. / 8.)]
            if t in (0, 4):
                return [(nei(2), -1. / 8.), (nei(1), 1. / 2.), (cell(b, ix, iy), -3. / 8.)]
            return [(nei(-1), -1. / 8.), (nei(1), 1. / 8.), (cell(b, ix, iy), 0.)]

        def d2(b, s, ix, iy):
            t = iy if s < 2 else ix
            nei = (lambda d: cell(b, ix, iy + d)) if s < 2 else (lambda d: cell(b, ix + d, iy))
            if t in (7, 3):
                return [(nei(-2), 1. / 32.), (nei(-1), -1. / 16.), (cell(b, ix, iy), 1. / 32.)]
            if t in (0, 4):
                return [(nei(2), 1. / 32.), (nei(1), -1. / 16.), (cell(b, ix, iy), 1. / 32.)]
            return [(nei(-1), 1. / 32.), (nei(1), 1. / 32.), (cell(b, ix, iy), -1. / 16.)]

        def interpolate(r, bc, s, ixc, iyc, fine_close, fine_far, sign_int, sign_taylor):
            add(r, fine_close, sign_int * 2. / 3.)
            add(r, fine_far, -sign_int * 1. / 5.)
            tf = sign_int * 8. / 15.
            add(r, cell(bc, ixc, iyc), tf)
            for c, w in d1(bc, s, ixc, iyc):
                add(r, c, sign_taylor * tf * w)
            for c, w in d2(bc, s, ixc, iyc):
                add(r, c, tf * w)

        for b, (l, bi, bj) in enumerate(self.blocks):
            bi, bj = int(bi), int(bj)
            for iy in range(BS):
                for ix in range(BS):
                    r = cell(b, ix, iy)
                    if 0 < ix < BS - 1 and 0 < iy < BS - 1:
                        for c, v in ((cell(b, ix, iy - 1), 1.), (cell(b, ix - 1, iy), 1.), (r, -4.), (cell(b, ix + 1, iy), 1.),
                                     (cell(b, ix, iy + 1), 1.)):
                            add(r, c, v)
                        continue
                    inblock = (ix > 0, ix < BS - 1, iy > 0, iy < BS - 1)
                    inner = (cell(b, ix - 1, iy) if ix > 0 else -1, cell(b, ix + 1, iy) if ix < BS - 1 else -1,
                             cell(b, ix, iy - 1) if iy > 0 else -1, cell(b, ix, iy + 1) if iy < BS - 1 else -1)
                    for s in range(4):
                        if inblock[s]:
                            add(r, inner[s], 1.)
                            add(r, r, -1.)
                            continue
                        k = int(self.kind[b, s])
                        if k == _l.AMR_WALL:
                            continue
                        n0, n1 = int(self.nbr2[b, s, 0]), int(self.nbr2[b, s, 1])
                        if k == _l.AMR_SAME:
                            c = cell(n0, 7, iy) if s == 0 else cell(n0, 0, iy) if s == 1 else cell(n0, ix, 7) if s == 2 else cell(n0, ix, 0)
                            add(r, c, 1.)
                            add(r, r, -1.)
                        elif k == _l.AMR_COARSER:
                            ixc = 7 if s == 0 else 0 if s == 1 else (ix // 2 if bi % 2 == 0 else ix // 2 + 4)
                            iyc = 7 if s == 2 else 0 if s == 3 else (iy // 2 if bj % 2 == 0 else iy // 2 + 4)
                            inward = cell(b, ix + 1, iy) if s == 0 else cell(b, ix - 1, iy) if s == 1 else cell(b, ix, iy + 1) if s == 2 \
                                else cell(b, ix, iy - 1)
                            t = iy if s < 2 else ix
                            interpolate(r, n0, s, ixc, iyc, r, inward, 1., -1. if t % 2 == 0 else 1.)
                            add(r, r, -1.)
                        else:  # two finer cells across the face, in the child block that covers this cell
                            t = iy if s < 2 else ix
                            fb = n1 if t >= 4 else n0
                            f = (t % 4) * 2
                            for j, st in ((0, -1.), (1, 1.)):
                                if s == 0:
                                    close, far = cell(fb, 7, f + j), cell(fb, 6, f + j)
                                elif s == 1:
                                    close, far = cell(fb, 0, f + j), cell(fb, 1, f + j)
                                elif s == 2:
                                    close, far = cell(fb, f + j, 7), cell(fb, f + j, 6)
                                else:
                                    close, far = cell(fb, f + j, 0), cell(fb, f + j, 1)
                                add(r, close, 1.)
                                interpolate(r, b, s, ix, iy, close, far, -1., st)
        rr, cc, vv = [], [], []
        for r in sorted(rows):
            for c in sorted(rows[r]):
                rr.append(r)
                cc.append(c)
                vv.append(rows[r][c])
        return np.asarray(rr, dtype=np.int32), np.asarray(cc, dtype=np.int32), np.asarray(vv, dtype=np.float64)

    def cell_centres(self):
        """x, y of every cell, (nb, 64) each, as the reference places them (origin main.cpp:695-696)"""
        l = self.blocks[:, 0]
        h = self.h0 / (1 << l)
        ox = self.blocks[:, 1] * BS * self.h0 / (1 << l)
        oy = self.blocks[:, 2] * BS * self.h0 / (1 << l)
        ix = np.tile(np.arange(BS), BS)
        iy = np.repeat(np.arange(BS), BS)
        return ox[:, None] + (ix[None, :] + 0.5) * h[:, None], oy[:, None] + (iy[None, :] + 0.5) * h[:, None]


class AmrSimulation:
    """Device-resident fields on an adapted grid + the halo-1 block operators in their AMR form.  Fields are set and
    read as per-block arrays (nb, 64) / (nb, 64, 2)."""

    def __init__(self, grid, nu=1e-3, cfl=0.5, device=0):
        self.L = _l.load_library()
        self.grid, self.nu, self.cfl = grid, float(nu), float(cfl)
        self._ctx = ctypes.c_void_p()
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_create(ctypes.byref(self._ctx), grid.nblocks, 0, grid.nblocks,
                                     np.ascontiguousarray(grid.nbr).ctypes.data_as(vp), grid.h0, int(device)), "cup2d_create")
        self._tables = [np.ascontiguousarray(a, dtype=np.int32) for a in (grid.level, grid.kind, grid.nbr2, grid.half)]
        _l.check(self.L.cup2d_set_amr(self._ctx, grid.h0, *[a.ctypes.data_as(vp) for a in self._tables]), "cup2d_set_amr")

    def close(self):
        if self._ctx:
            self.L.cup2d_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_field(self, field, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(self.grid.nblocks, 64 * _l.FIELD_DIM[field])
        _l.check(self.L.cup2d_upload_slab(self._ctx, field, a.ctypes.data_as(ctypes.c_void_p)), "upload_slab")

    def get_field(self, field):
        dim = _l.FIELD_DIM[field]
        a = np.empty((self.grid.nblocks, 64 * dim))
        _l.check(self.L.cup2d_download_slab(self._ctx, field, a.ctypes.data_as(ctypes.c_void_p)), "download_slab")
        return a.reshape(self.grid.nblocks, 64, 2) if dim == 2 else a

    def set_math(self, strict):
        _l.check(self.L.cup2d_set_math(self._ctx, _l.MATH_STRICT if strict else _l.MATH_FAST), "set_math")

    def advect_diffuse_rhs(self, dt):
        """prepare0 / computeA<VectorLab>(KernelAdvectDiffuse(), var.vel, 2) / fillcases (main.cpp:6611-6617): tmpV"""
        _l.check(self.L.cup2d_advect_diffuse_rhs(self._ctx, self.nu, float(dt), _l.BLOCKS_ALL), "advect_diffuse_rhs")

    def laplacian_sub(self):
        """prepare0 / computeA<ScalarLab>(pressure_rhs1(), var.pold, 1) / fillcases (main.cpp:7022-7027)"""
        _l.check(self.L.cup2d_laplacian_sub(self._ctx, _l.BLOCKS_ALL), "laplacian_sub")

    def vorticity(self):
        _l.check(self.L.cup2d_vorticity(self._ctx, _l.BLOCKS_ALL), "vorticity")

    def pressure_rhs(self, dt):
        """computeB<pressure_rhs,..> + fillcases (main.cpp:7007-7013): tmp from vel, tmpV (= udef), chi"""
        _l.check(self.L.cup2d_pressure_rhs(self._ctx, float(dt), 1, _l.BLOCKS_ALL), "pressure_rhs")

    def pressure_correction(self, dt):
        _l.check(self.L.cup2d_pressure_correction(self._ctx, float(dt), _l.BLOCKS_ALL), "pressure_correction")

    def apply_A(self, dst, src):
        _l.check(self.L.cup2d_apply_A(self._ctx, dst, src), "apply_A")

    def install_poisson_matrix(self):
        """assemble the coarse-fine Poisson rows on the host (AmrBlockGrid.poisson_coo) and hand them to the library:
        what the reference does after every regrid (main.cpp:7034-7113)"""
        r, c, v = self.grid.poisson_coo()
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_set_matrix_coo(self._ctx, 0, len(v), r.ctypes.data_as(vp), c.ctypes.data_as(vp), v.ctypes.data_as(vp)),
                 "set_matrix_coo")

    def step(self, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=1000, dt=None):
        """one pass of the time-loop body on the (fixed) adapted grid: dt, RK2 WENO5 advect-diffuse with flux correction,
        Poisson rhs, BiCGSTAB on the assembled operator, volume-weighted mean removal + projection (main.cpp:6576-7187
        without adapt()).  dt: the reference computes dt BEFORE it regrids (main.cpp:6579-6603); a caller that adapts
        between the two passes the value compute_dt() gave on the old grid."""
        it, e = ctypes.c_int(), ctypes.c_double()
        if dt is None:
            d = ctypes.c_double()
            _l.check(self.L.cup2d_step(self._ctx, self.nu, self.cfl, float(tol), float(rel_tol), int(max_restarts), int(max_iter),
                                       ctypes.byref(d), ctypes.byref(it), ctypes.byref(e)), "step")
            return dict(dt=d.value, iters=it.value, err=e.value)
        dt = float(dt)
        _l.check(self.L.cup2d_advect_diffuse_rk2(self._ctx, self.nu, dt), "advect_diffuse_rk2")
        _l.check(self.L.cup2d_poisson_rhs(self._ctx, dt, 0), "poisson_rhs")
        _l.check(self.L.cup2d_poisson_solve(self._ctx, float(tol), float(rel_tol), int(max_restarts), int(max_iter), ctypes.byref(it),
                                            None, ctypes.byref(e), None), "poisson_solve")
        _l.check(self.L.cup2d_project(self._ctx, dt), "project")
        return dict(dt=dt, iters=it.value, err=e.value)

    def adapt(self, rtol, ctol, level_max):
        """The reference's adapt() (main.cpp:4657-5440) for this simulation: tag by max|vorticity| per block (GPU),
        validate the states, prolong / restrict every field on the host (regrid-time work, as in the reference), then
        rebuild the device context on the new grid and re-assemble the Poisson operator.  Returns True if the grid
        changed."""
        self.vorticity()
        linf = np.empty(self.grid.nblocks)
        _l.check(self.L.cup2d_block_linf(self._ctx, _l.TMP, _p(linf)), "block_linf")  # one double per block crosses PCIe
        st = validate_states(self.grid.blocks, tag_states(linf, self.grid.level, rtol, ctol, level_max), level_max,
                             self.grid.bpdx, self.grid.bpdy)
        if not (st != LEAVE).any():
            return False
        nbk = self.grid.nblocks
        names = {"chi": _l.CHI, "vel": _l.VEL, "vold": _l.VOLD, "pres": _l.PRES, "pold": _l.POLD}
        fields = {k: (self.get_field(f).reshape(nbk, -1), _l.FIELD_DIM[f], _l.FIELD_DIM[f] == 2) for k, f in names.items()}
        blocks, data = regrid(self.grid.blocks, st, fields, level_max, self.grid.bpdx, self.grid.bpdy)
        device = 0
        self.close()
        self.__init__(AmrBlockGrid(blocks, self.grid.bpdx, self.grid.bpdy, self.grid.h0 * max(self.grid.bpdx, self.grid.bpdy) * BS),
                      nu=self.nu, cfl=self.cfl, device=device)
        for k, f in names.items():
            self.set_field(f, data[k])
        self.install_poisson_matrix()
        return True

    def compute_dt(self):
        v = ctypes.c_double()
        _l.check(self.L.cup2d_compute_dt(self._ctx, self.nu, self.cfl, ctypes.byref(v)), "compute_dt")
        return v.value


# ---- regridding: the reference's adapt() (main.cpp:4657-5440) for one rank ---------------------------------------------
LEAVE, REFINE, COMPRESS = 0, 1, 2


def tag_states(linf, level, rtol, ctol, level_max):
    """main.cpp:4678-4690: Refine where max|vorticity| of the block exceeds Rtol, Compress where it is below Ctol;
    the finest level cannot refine, level 0 cannot compress"""
    linf = np.asarray(linf, dtype=np.float64)
    level = np.asarray(level)
    st = np.where(linf > rtol, REFINE, np.where(linf < ctol, COMPRESS, LEAVE)).astype(np.int32)
    st[(st == REFINE) & (level == level_max - 1)] = LEAVE
    st[(st == COMPRESS) & (level == 0)] = LEAVE
    return st


def validate_states(blocks, states, level_max, bpdx=1, bpdy=1):
    """The reference's state validation (main.cpp:4718-4861) by the library's host routine cup2d_amr_validate_states;
    validate_states_py states the same algorithm in Python.  Returns the final states."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.array(states, dtype=np.int32)
    L = _l.load_library()
    _l.check(L.cup2d_amr_validate_states(len(b32), _p(b32), bpdx, bpdy, level_max, _p(st)), "amr_validate_states")
    return st


def validate_states_py(blocks, states, level_max, bpdx=1, bpdy=1):
    """The reference's state validation (main.cpp:4718-4861), which keeps the grid 2:1 balanced across faces AND
    corners: from the finest level down, a block next to finer blocks may not compress and refines if one of those is
    refining; a compressing block next to a same-level refining block stays; four siblings compress together or not
    at all.  Returns the final states."""
    blocks = np.asarray(blocks, dtype=np.int64)
    st = np.array(states, dtype=np.int32)
    if not (st != LEAVE).any():
        return st
    index = {tuple(int(v) for v in b): k for k, b in enumerate(blocks)}

    def tree(l, i, j):
        if (l, i, j) in index:
            return 0
        if l > 0 and (l - 1, i // 2, j // 2) in index:
            return -2
        return -1

    for k, (l, i, j) in enumerate(blocks):
        if (st[k] == REFINE and l == level_max - 1) or (st[k] == COMPRESS and l == 0):
            st[k] = LEAVE
    for m in range(level_max - 1, -1, -1):
        for k, (l, i, j) in enumerate(blocks):
            l, i, j = int(l), int(i), int(j)
            if l != m or st[k] == REFINE or l == level_max - 1:
                continue
            nx, ny = bpdx << l, bpdy << l
            done = False
            for x in (-1, 0, 1):
                for y in (-1, 0, 1):
                    if (x == 0 and y == 0) or not (0 <= i + x < nx and 0 <= j + y < ny):
                        continue
                    if tree(l, i + x, j + y) != -1:
                        continue
                    if st[k] == COMPRESS:
                        st[k] = LEAVE
                    bstep = 3 if abs(x) + abs(y) == 2 else 1
                    for B in range(0, 2, bstep):
                        aux = B % 2 if abs(x) == 1 else B // 2
                        fi = 2 * i + max(x, 0) + x + (B % 2) * max(0, 1 - abs(x))
                        fj = 2 * j + max(y, 0) + y + aux * max(0, 1 - abs(y))
                        fk = index.get((m + 1, fi, fj))
                        if fk is not None and st[fk] == REFINE:
                            st[k] = REFINE
                            done = True
                            break
                    if done:
                        break
                if done:
                    break
        if m == 0:
            break
        for k, (l, i, j) in enumerate(blocks):
            l, i, j = int(l), int(i), int(j)
            if l != m or st[k] != COMPRESS:
                continue
            nx, ny = bpdx << l, bpdy << l
            for x in (-1, 0, 1):
                for y in (-1, 0, 1):
                    if (x == 0 and y == 0) or not (0 <= i + x < nx and 0 <= j + y < ny):
                        continue
                    nk = index.get((l, i + x, j + y))
                    if nk is not None and st[nk] == REFINE:
                        st[k] = LEAVE
    for k, (l, i, j) in enumerate(blocks):
        l, i, j = int(l), int(i), int(j)
        sib = [index.get((l, 2 * (i // 2) + a, 2 * (j // 2) + b)) for a in (0, 1) for b in (0, 1)]
        if any(s is None or st[s] != COMPRESS for s in sib):
            for s in sib:
                if s is not None and st[s] == COMPRESS:
                    st[s] = LEAVE
    return st


def _prolong(tile, dim):
    """the four children of a block from its tensorial halo-1 tile (10 x 10 x dim), main.cpp:4981-5032: second-order
    Taylor expansion about the parent cell, operand order kept"""
    um = tile.reshape(10, 10, dim)
    kids = np.zeros((2, 2, BS, BS, dim))
    for J in range(2):
        for I in range(2):
            b = kids[J, I]
            for j in range(0, BS, 2):
                for i in range(0, BS, 2):
                    i0, j0 = i // 2 + 4 * I + 1, j // 2 + 4 * J + 1
                    l00, l0p, l0m = um[j0, i0], um[j0 + 1, i0], um[j0 - 1, i0]
                    lm0, lmm, lmp = um[j0, i0 - 1], um[j0 - 1, i0 - 1], um[j0 + 1, i0 - 1]
                    lp0, lpm, lpp = um[j0, i0 + 1], um[j0 - 1, i0 + 1], um[j0 + 1, i0 + 1]
                    x = 0.5 * (lp0 - lm0)
                    y = 0.5 * (l0p - l0m)
                    x2 = (lp0 + lm0) - 2.0 * l00
                    y2 = (l0p + l0m) - 2.0 * l00
                    xy = 0.25 * ((lpp + lmm) - (lpm + lmp))
                    b[j, i] = (l00 + (-0.25 * x - 0.25 * y)) + ((0.03125 * x2 + 0.03125 * y2) + 0.0625 * xy)
                    b[j, i + 1] = (l00 + (+0.25 * x - 0.25 * y)) + ((0.03125 * x2 + 0.03125 * y2) - 0.0625 * xy)
                    b[j + 1, i] = (l00 + (-0.25 * x + 0.25 * y)) + ((0.03125 * x2 + 0.03125 * y2) - 0.0625 * xy)
                    b[j + 1, i + 1] = (l00 + (+0.25 * x + 0.25 * y)) + ((0.03125 * x2 + 0.03125 * y2) + 0.0625 * xy)
    return kids


def _restrict(kids, dim):
    """the parent of four siblings kids[J][I] (BS x BS x dim each), main.cpp:5149-5166"""
    out = np.empty((BS, BS, dim))
    for J in range(2):
        for I in range(2):
            b = kids[J][I]
            out[4 * J:4 * J + 4, 4 * I:4 * I + 4] = (b[0::2, 0::2] + b[1::2, 0::2] + b[0::2, 1::2] + b[1::2, 1::2]) / 4
    return out


def regrid(blocks, states, fields, level_max, bpdx=1, bpdy=1):
    """Apply final states with the library's host routine cup2d_amr_regrid (include/cup2d_hip.h); regrid_py states the
    same algorithm in Python on the general-stencil BlockLab of amr_lab.py.  fields: {name: (array (nb, 64*dim), dim,
    is_vector)}.  Returns (new_blocks, new_fields)."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    nb = len(b32)
    names = list(fields)
    src = [np.ascontiguousarray(fields[k][0], dtype=np.float64).reshape(nb, -1) for k in names]
    dims = np.array([fields[k][1] for k in names], dtype=np.int32)
    vec = np.array([1 if fields[k][2] else 0 for k in names], dtype=np.int32)
    L = _l.load_library()
    vp = ctypes.c_void_p
    n = len(names)
    srcp = (vp * max(n, 1))(*[a.ctypes.data for a in src])
    n_new = L.cup2d_amr_regrid(nb, _p(b32), bpdx, bpdy, level_max, _p(st), n, srcp, _p(dims), _p(vec), 0, None, None)
    if n_new < 0:
        _l.check(int(n_new), "amr_regrid")
    new_blocks = np.empty((n_new, 3), dtype=np.int32)
    dst = [np.empty((n_new, BS * BS * int(d))) for d in dims]
    dstp = (vp * max(n, 1))(*[a.ctypes.data for a in dst])
    got = L.cup2d_amr_regrid(nb, _p(b32), bpdx, bpdy, level_max, _p(st), n, srcp, _p(dims), _p(vec), n_new, _p(new_blocks), dstp)
    if got != n_new:
        _l.check(int(min(got, -1)), "amr_regrid")
    return new_blocks.astype(np.int64), dict(zip(names, dst))


def regrid_py(blocks, states, fields, level_max):
    """Apply final states: every Refine block becomes its four children (prolonged from the OLD grid's tensorial halo-1
    tile, cup2d_amd/amr_lab.py), every complete Compress sibling group its parent (2x2 means); everything else is
    kept.  fields: {name: (array (nb, 64*dim), dim, is_vector)}.  Returns (new_blocks, new_fields) ordered along the
    Hilbert curve of the finest level (the reference's Info::id2 order, main.cpp:1550-1562)."""
    from .amr_lab import BlockLab, Tree
    from .grid import hilbert_index
    blocks = np.asarray(blocks, dtype=np.int64)
    tree = Tree(blocks)
    index = tree.index
    new_blocks, new_data = [], {k: [] for k in fields}
    labs = {k: BlockLab(dim, (-1, -1, 2, 2, True), vec) for k, (a, dim, vec) in fields.items()}
    done = set()
    for k, (l, i, j) in enumerate(blocks):
        l, i, j = int(l), int(i), int(j)
        if states[k] == REFINE:
            tiles = {f: _prolong(labs[f].load(tree, a.reshape(len(blocks), -1), k), dim) for f, (a, dim, vec) in fields.items()}
            for J in range(2):
                for I in range(2):
                    new_blocks.append((l + 1, 2 * i + I, 2 * j + J))
                    for f in fields:
                        new_data[f].append(tiles[f][J, I].reshape(-1))
        elif states[k] == COMPRESS:
            key = (l, 2 * (i // 2), 2 * (j // 2))
            if key in done:
                continue
            done.add(key)
            sib = [[index[(l, key[1] + I, key[2] + J)] for I in (0, 1)] for J in (0, 1)]
            new_blocks.append((l - 1, i // 2, j // 2))
            for f, (a, dim, vec) in fields.items():
                kids = [[a[sib[J][I]].reshape(BS, BS, dim) for I in (0, 1)] for J in (0, 1)]
                new_data[f].append(_restrict(kids, dim).reshape(-1))
        else:
            new_blocks.append((l, i, j))
            for f, (a, dim, vec) in fields.items():
                new_data[f].append(a[k].reshape(-1))
    nb = np.asarray(new_blocks, dtype=np.int64)
    L = int(max(level_max - 1, nb[:, 0].max()))
    key = hilbert_index(max(L, 1), nb[:, 1] << (L - nb[:, 0]), nb[:, 2] << (L - nb[:, 0]))
    order = np.lexsort((nb[:, 0], key))
    return nb[order], {f: np.asarray(v)[order] for f, v in new_data.items()}
