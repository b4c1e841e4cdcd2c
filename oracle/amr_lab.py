"""Host-side BlockLab for block-AMR grids: the ghosted tile of ANY stencil, written statement for statement after the
reference's BlockLab (main.cpp:2231-2996: same index formulas, loop bounds, operand order, C integer semantics), so that
it reproduces the reference's tiles bit for bit, quirks included (the unrolled finer-neighbour branch, main.cpp:2476-2534;
TestInterp fed component 0, 2753-2763).  Pinned against tiles dumped by the reference itself (tests/test_amr.py: halo-3
tensorial vector tiles of KernelAdvectDiffuse, halo-1 tensorial scalar and vector tiles of adapt()).

Test infrastructure (never imported by the product).  Two users: the Python statement of regridding (oracle/amr_regrid.py:
the tensorial halo-1 tile a refined block is prolonged from, main.cpp:4906-5032), which the library's host routines
must reproduce bit for bit, and the full-tile cross-check of the closed forms the HIP kernels implement (csrc/amr.hip).
Single rank, bpdx = bpdy = 1."""
import numpy as np

BS = 8


def cdiv(a, b):
    """C integer division (truncation toward zero)"""
    return int(a / b)


def cmod(a, b):
    return a - cdiv(a, b) * b


class Tree:
    """the reference's tree states for a (level, i, j) position: rank >= 0 (leaf), -1 (refined), -2 (a coarser leaf covers it)"""

    def __init__(self, blocks):
        self.blocks = np.asarray(blocks, dtype=np.int64)
        self.index = {tuple(int(v) for v in b): k for k, b in enumerate(self.blocks)}

    def state(self, l, i, j):
        if (l, i, j) in self.index:
            return 0
        if l > 0 and (l - 1, i // 2, j // 2) in self.index:
            return -2
        return -1

    def block(self, l, i, j):
        return self.index.get((l, i, j))


def LI(a, b, c):
    kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b
    lam = (b - c) - kappa
    return (4.0 * kappa + 2.0 * lam) + c


def LE(a, b, c):
    kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b
    lam = (b - c) - kappa
    return (9.0 * kappa + 3.0 * lam) + c


class BlockLab:
    def __init__(self, dim, stencil, vector):
        """stencil = (sx, sy, ex, ey, tensorial); vector: VectorLab boundary conditions (else ScalarLab's Neumann)"""
        self.dim, self.vector = dim, vector
        sx, sy, ex, ey, tens = stencil
        self.istensorial = bool(tens)
        self.start = [sx, sy, 0]
        self.end = [ex, ey, 1]
        self.nm = [BS + ex - sx - 1, BS + ey - sy - 1]
        self.m = np.full(self.nm[0] * self.nm[1] * dim, np.nan)
        self.offset = [cdiv(sx - 1, 2) - 1, cdiv(sy - 1, 2) - 1, cdiv(-1, 2)]
        self.nc = [BS // 2 + cdiv(ex, 2) + 1 - self.offset[0], BS // 2 + cdiv(ey, 2) + 1 - self.offset[1]]
        self.c = np.full(self.nc[0] * self.nc[1] * dim, np.nan)
        self.use_averages = self.istensorial or sx < -2 or sy < -2 or ex > 3 or ey > 3

    # ---- load (main.cpp:2270-2687, single rank) -------------------------------------------------------------
    def load(self, tree, field, b):
        dim, m, c, nm, nc, start, end, offset = self.dim, self.m, self.c, self.nm, self.nc, self.start, self.end, self.offset
        m[:] = np.nan
        c[:] = np.nan
        l, ix0, iy0 = (int(v) for v in tree.blocks[b])
        self.level, self.index = l, (ix0, iy0)
        aux = 1 << l
        NX = NY = aux
        self.NX, self.NY = NX, NY
        p = field[b].reshape(-1)
        for iy in range(BS):
            q = dim * (iy - start[1]) * nm[0] - dim * start[0]
            m[q:q + dim * BS] = p[dim * BS * iy:dim * BS * (iy + 1)]
        self.coarsened = False
        xskin = ix0 == 0 or ix0 == NX - 1
        yskin = iy0 == 0 or iy0 == NY - 1
        xskip = -1 if ix0 == 0 else 1
        yskip = -1 if iy0 == 0 else 1
        icodes = []
        self.coarsened_nei_codes = []
        self.myblocks = {}
        for icode in range(9, 18):
            if icode == 13:
                continue
            code = [icode % 3 - 1, (icode // 3) % 3 - 1, icode // 9 - 1]
            if code[0] == xskip and xskin:
                continue
            if code[1] == yskip and yskin:
                continue
            ni, nj = (ix0 + code[0] + NX) % NX, (iy0 + code[1] + NY) % NY
            TreeNei = tree.state(l, ni, nj)
            if TreeNei >= 0:
                icodes.append(icode)
            elif TreeNei == -2:
                self.coarsened_nei_codes.append(icode)
                infoNei_index_true = [ix0 + code[0], iy0 + code[1]]
                bb = tree.block(l - 1, ni // 2, nj // 2)
                if bb is not None:
                    bdat = field[bb].reshape(-1)
                    s = [(offset[0] if code[0] < 0 else 0) if code[0] < 1 else BS // 2,
                         (offset[1] if code[1] < 0 else 0) if code[1] < 1 else BS // 2]
                    e = [(0 if code[0] < 0 else BS // 2) if code[0] < 1 else BS // 2 + cdiv(end[0], 2) + 2 - 1,
                         (0 if code[1] < 0 else BS // 2) if code[1] < 1 else BS // 2 + cdiv(end[1], 2) + 2 - 1]
                    n = (e[0] - s[0]) * dim
                    if n:
                        base = [cmod(ix0 + code[0], 2), cmod(iy0 + code[1], 2)]
                        CoarseEdge = [0, 0]
                        for d in (0, 1):
                            idx = (ix0, iy0)[d]
                            if code[d] != 0 and ((idx % 2 == 0 and infoNei_index_true[d] > idx) or
                                                 (idx % 2 == 1 and infoNei_index_true[d] < idx)):
                                CoarseEdge[d] = 1
                        st = [max(code[d], 0) * BS // 2 + (1 - abs(code[d])) * base[d] * BS // 2 - code[d] * BS +
                              CoarseEdge[d] * code[d] * BS // 2 for d in (0, 1)]
                        i = s[0] - offset[0]
                        for iy in range(s[1], e[1]):
                            i0 = i + (iy - offset[1]) * nc[0]
                            y0 = iy + st[1]
                            x = s[0] + st[0]
                            c[dim * i0:dim * i0 + n] = bdat[dim * (BS * y0 + x):dim * (BS * y0 + x) + n]
            if (not self.istensorial) and (not self.use_averages) and abs(code[0]) + abs(code[1]) > 1:
                continue
            s = [(start[0] if code[0] < 0 else 0) if code[0] < 1 else BS, (start[1] if code[1] < 0 else 0) if code[1] < 1 else BS]
            e = [(0 if code[0] < 0 else BS) if code[0] < 1 else BS + end[0] - 1,
                 (0 if code[1] < 0 else BS) if code[1] < 1 else BS + end[1] - 1]
            if TreeNei >= 0:
                n = (e[0] - s[0]) * dim
                if not n:
                    continue
                bb = tree.block(l, ni, nj)
                self.myblocks[icode] = bb
                bdat = field[bb].reshape(-1)
                i = s[0] - start[0]
                for iy in range(s[1], e[1]):
                    i0 = i + (iy - start[1]) * nm[0]
                    x0 = s[0] - code[0] * BS
                    y0 = iy - code[1] * BS
                    m[dim * i0:dim * i0 + n] = bdat[dim * (BS * y0 + x0):dim * (BS * y0 + x0) + n]
            elif TreeNei == -1:
                cnt = abs(code[0]) * (e[0] - s[0]) + (1 - abs(code[0])) * cdiv(e[0] - s[0], 2)
                if not cnt * dim:
                    continue
                ys = 2 if code[1] == 0 else 1
                mod = cmod(cdiv(e[1] - s[1], ys), 4)
                Bstep = 1
                if abs(code[0]) + abs(code[1]) == 2:
                    Bstep = 3
                for B in range(0, 4, Bstep):
                    auxB = (B % 2) if abs(code[0]) == 1 else (B // 2)
                    bb = tree.block(l + 1, 2 * ix0 + max(code[0], 0) + code[0] + (B % 2) * max(0, 1 - abs(code[0])),
                                    2 * iy0 + max(code[1], 0) + code[1] + auxB * max(0, 1 - abs(code[1])))
                    if bb is None:
                        continue
                    bdat = field[bb].reshape(-1)
                    i = abs(code[0]) * (s[0] - start[0]) + (1 - abs(code[0])) * (s[0] - start[0] + (B % 2) * cdiv(e[0] - s[0], 2))
                    x = s[0] - code[0] * BS + min(0, code[0]) * (e[0] - s[0])

                    def krow(yy):
                        return i + (abs(code[1]) * (yy - start[1]) +
                                    (1 - abs(code[1])) * (cdiv(yy, 2) - start[1] + auxB * cdiv(e[1] - s[1], 2))) * nm[0]

                    def yrow(yy):
                        return 2 * (yy - code[1] * BS) + min(0, code[1]) * BS if abs(code[1]) == 1 else yy

                    iy = s[1]
                    while iy < e[1] - mod:
                        k = [krow(iy + t * ys) for t in range(4)]
                        y = [yrow(iy + t * ys) for t in range(4)]
                        z = [yy + 1 for yy in y]
                        for ee in range(cnt):
                            for d in range(dim):
                                def q(row, col):
                                    return bdat[dim * (BS * row + x) + dim * col + d]
                                # main.cpp:2528-2535: p0 pairs rows y0 and y1 (sic); p1..p3 pair rows y_k and z_k
                                m[dim * k[0] + dim * ee + d] = (q(y[0], 2 * ee) + q(y[1], 2 * ee) + q(y[0], 2 * ee + 1) + q(y[1], 2 * ee + 1)) / 4
                                for t in (1, 2, 3):
                                    m[dim * k[t] + dim * ee + d] = (q(y[t], 2 * ee) + q(z[t], 2 * ee) + q(y[t], 2 * ee + 1) +
                                                                    q(z[t], 2 * ee + 1)) / 4
                        iy += 4 * ys
                    iy = e[1] - mod
                    while iy < e[1]:
                        k = krow(iy)
                        y = yrow(iy)
                        z = y + 1
                        for ee in range(cnt):
                            for d in range(dim):
                                def q(row, col):
                                    return bdat[dim * (BS * row + x) + dim * col + d]
                                m[dim * k + dim * ee + d] = (q(y, 2 * ee) + q(z, 2 * ee) + q(y, 2 * ee + 1) + q(z, 2 * ee + 1)) / 4
                        iy += ys
        if self.coarsened_nei_codes:
            for icode in icodes:
                code = [icode % 3 - 1, (icode // 3) % 3 - 1, icode // 9 - 1]
                infoNei_index = [(ix0 + code[0] + NX) % NX, (iy0 + code[1] + NY) % NY, 0]
                if self.UseCoarseStencil0(infoNei_index):
                    self.FillCoarseVersion(code, field)
                    self.coarsened = True
        self.post_load()
        return self.m.reshape(self.nm[1], self.nm[0], dim)

    def UseCoarseStencil0(self, infoNei_index):
        if self.level == 0 or not self.use_averages:
            return False
        index = [self.index[0], self.index[1], 0]
        aux = 1 << self.level
        blocks = [aux - 1, aux - 1, aux - 1]
        imin, imax = [0, 0, 0], [0, 0, 0]
        for d in range(3):
            imin[d] = 0 if index[d] < infoNei_index[d] else -1
            imax[d] = 0 if index[d] > infoNei_index[d] else +1
            if index[d] == 0 and infoNei_index[d] == 0:
                imin[d] = 0
            if index[d] == blocks[d] and infoNei_index[d] == blocks[d]:
                imax[d] = 0
        for t in self.coarsened_nei_codes:
            for i2 in range(imin[2], imax[2] + 1):
                for i1 in range(imin[1], imax[1] + 1):
                    for i0 in range(imin[0], imax[0] + 1):
                        if t == (i0 + 1) + 3 * (i1 + 1) + 9 * (i2 + 1):
                            return True
        return False

    def FillCoarseVersion(self, code, field):
        dim, c, nc, end, offset = self.dim, self.c, self.nc, self.end, self.offset
        icode = (code[0] + 1) + 3 * (code[1] + 1) + 9
        bb = self.myblocks.get(icode)
        if bb is None:
            return
        b = field[bb].reshape(-1)
        eC = [cdiv(end[0], 2) + 2, cdiv(end[1], 2) + 2]
        s = [(offset[0] if code[0] < 0 else 0) if code[0] < 1 else BS // 2, (offset[1] if code[1] < 0 else 0) if code[1] < 1 else BS // 2]
        e = [(0 if code[0] < 0 else BS // 2) if code[0] < 1 else BS // 2 + eC[0] - 1,
             (0 if code[1] < 0 else BS // 2) if code[1] < 1 else BS // 2 + eC[1] - 1]
        if not (e[0] - s[0]) * dim:
            return
        st = [s[d] + max(code[d], 0) * (BS // 2) - code[d] * BS + min(0, code[d]) * (e[d] - s[d]) for d in (0, 1)]
        i = s[0] - offset[0]
        x = st[0]
        for iy in range(s[1], e[1]):
            i0 = i + (iy - offset[1]) * nc[0]
            y0 = 2 * (iy - s[1]) + st[1]
            y1 = y0 + 1
            for ee in range(e[0] - s[0]):
                for d in range(dim):
                    def q(row, col):
                        return b[dim * (BS * row + x) + dim * col + d]
                    c[dim * i0 + dim * ee + d] = (q(y0, 2 * ee) + q(y1, 2 * ee) + q(y0, 2 * ee + 1) + q(y1, 2 * ee + 1)) / 4

    # ---- boundary conditions (main.cpp:3131-3255) ---------------------------------------------------------
    def _bc_face(self, dr, side, coarse):
        dim = self.dim
        A = 1 - dr
        if not coarse:
            arr, n0, hb = self.m, self.nm[0], BS
            sb = [self.start[0], self.start[1]]
            se = [self.end[0], self.end[1]]
        else:
            arr, n0, hb = self.c, self.nc[0], BS // 2
            se = [cdiv(self.end[0], 2) + 1 + 2 - 1, cdiv(self.end[1], 2) + 1 + 2 - 1]
            sb = [cdiv(self.start[0] - 1, 2) - 1, cdiv(self.start[1] - 1, 2) - 1]
        s = [(sb[0] if side == 0 else hb) if dr == 0 else sb[0], (sb[1] if side == 0 else hb) if dr == 1 else sb[1]]
        e = [(0 if side == 0 else hb + se[0] - 1) if dr == 0 else hb + se[0] - 1,
             (0 if side == 0 else hb + se[1] - 1) if dr == 1 else hb + se[1] - 1]
        for iy in range(s[1], e[1]):
            for ix in range(s[0], e[0]):
                x = ((0 if side == 0 else hb - 1) if dr == 0 else ix) - sb[0]
                y = ((0 if side == 0 else hb - 1) if dr == 1 else iy) - sb[1]
                i0 = ix - sb[0] + n0 * (iy - sb[1])
                i1 = x + n0 * y
                if self.vector:
                    arr[2 * i0 + 1 - A] = -arr[2 * i1 + 1 - A]
                    arr[2 * i0 + A] = arr[2 * i1 + A]
                else:
                    arr[i0] = arr[i1]

    def _apply_bc(self, coarse):
        if self.index[0] == 0:
            self._bc_face(0, 0, coarse)
        if self.index[0] == self.NX - 1:
            self._bc_face(0, 1, coarse)
        if self.index[1] == 0:
            self._bc_face(1, 0, coarse)
        if self.index[1] == self.NY - 1:
            self._bc_face(1, 1, coarse)

    # ---- post_load (main.cpp:2689-2933) ------------------------------------------------------------------------
    def post_load(self):
        dim, m, c, nm, nc, start, end, offset = self.dim, self.m, self.c, self.nm, self.nc, self.start, self.end, self.offset
        if self.coarsened:
            for j in range(BS // 2):
                for i in range(BS // 2):
                    if i > 1 and i < BS // 2 - 2 and j > 2 and j < BS // 2 - 2:
                        continue
                    ix, iy = 2 * i - start[0], 2 * j - start[1]
                    i00, i10 = ix + nm[0] * iy, ix + 1 + nm[0] * iy
                    i01, i11 = ix + nm[0] * (iy + 1), ix + 1 + nm[0] * (iy + 1)
                    j00 = i - offset[0] + nc[0] * (j - offset[1])
                    for d in range(dim):
                        c[dim * j00 + d] = (m[dim * i01 + d] + m[dim * i00 + d] + m[dim * i10 + d] + m[dim * i11 + d]) / 4
        self._apply_bc(True)
        ix0, iy0 = self.index
        xskin = ix0 == 0 or ix0 == self.NX - 1
        yskin = iy0 == 0 or iy0 == self.NY - 1
        xskip = -1 if ix0 == 0 else 1
        yskip = -1 if iy0 == 0 else 1
        for icode in self.coarsened_nei_codes:
            if icode == 13:
                continue
            code = [icode % 3 - 1, (icode // 3) % 3 - 1, (icode // 9) % 3 - 1]
            if code[2] != 0:
                continue
            if code[0] == xskip and xskin:
                continue
            if code[1] == yskip and yskin:
                continue
            if (not self.istensorial) and (not self.use_averages) and abs(code[0]) + abs(code[1]) > 1:
                continue
            s = [(start[0] if code[0] < 0 else 0) if code[0] < 1 else BS, (start[1] if code[1] < 0 else 0) if code[1] < 1 else BS]
            e = [(0 if code[0] < 0 else BS) if code[0] < 1 else BS + end[0] - 1,
                 (0 if code[1] < 0 else BS) if code[1] < 1 else BS + end[1] - 1]
            sC = [(cdiv(start[0] - 1, 2) if code[0] < 0 else 0) if code[0] < 1 else BS // 2,
                  (cdiv(start[1] - 1, 2) if code[1] < 0 else 0) if code[1] < 1 else BS // 2]
            if not (e[0] - s[0]) * dim:
                continue

            def par(v, d):
                return v - s[d] - min(0, code[d]) * cmod(e[d] - s[d], 2)

            if self.use_averages:
                for iy in range(s[1], e[1]):
                    YY = cdiv(par(iy, 1), 2) + sC[1]
                    for ix in range(s[0], e[0]):
                        XX = cdiv(par(ix, 0), 2) + sC[0]
                        i1 = ix - start[0] + nm[0] * (iy - start[1])
                        x, y = abs(par(ix, 0)) % 2, abs(par(iy, 1)) % 2
                        dx, dy = 0.25 * (2 * x - 1), 0.25 * (2 * y - 1)
                        for d in range(dim):
                            def C(i, j):
                                # main.cpp:2753-2763: Test[i][j] = c + dim * i0 carries NO component offset, so every
                                # component of the ghost cell is interpolated from component 0 of the coarse cells (sic)
                                return c[dim * (XX - 1 + i - offset[0] + nc[0] * (YY - 1 + j - offset[1]))]
                            dudx = 0.5 * (C(2, 1) - C(0, 1))
                            dudy = 0.5 * (C(1, 2) - C(1, 0))
                            dudxdy = 0.25 * ((C(0, 0) + C(2, 2)) - (C(2, 0) + C(0, 2)))
                            dudx2 = (C(0, 1) + C(2, 1)) - 2.0 * C(1, 1)
                            dudy2 = (C(1, 0) + C(1, 2)) - 2.0 * C(1, 1)
                            m[dim * i1 + d] = (C(1, 1) + (dx * dudx + dy * dudy)) + \
                                (((0.5 * dx * dx) * dudx2 + (0.5 * dy * dy) * dudy2) + (dx * dy) * dudxdy)
            if abs(code[0]) + abs(code[1]) == 1:
                for iy in range(s[1], e[1], 2):
                    YY = cdiv(par(iy, 1), 2) + sC[1] - offset[1]
                    y = abs(par(iy, 1)) % 2
                    iyp = -1 if abs(iy) % 2 == 1 else 1
                    dy = 0.25 * (2 * y - 1)
                    for ix in range(s[0], e[0], 2):
                        XX = cdiv(par(ix, 0), 2) + sC[0] - offset[0]
                        x = abs(par(ix, 0)) % 2
                        ixp = -1 if abs(ix) % 2 == 1 else 1
                        dx = 0.25 * (2 * x - 1)
                        if ix < -2 or iy < -2 or ix > BS + 1 or iy > BS + 1:
                            continue
                        i0, i1, i2 = XX + nc[0] * (YY + 2), XX + nc[0] * YY, XX + nc[0] * (YY + 1)
                        i3, i4 = XX + nc[0] * (YY - 2), XX + nc[0] * (YY - 1)
                        i5, i6, i7, i8 = XX + 2 + nc[0] * YY, XX + 1 + nc[0] * YY, XX - 1 + nc[0] * YY, XX - 2 + nc[0] * YY
                        j0 = ix - start[0] + nm[0] * (iy - start[1])
                        j1 = ix - start[0] + nm[0] * (iy - start[1] + iyp)
                        j2 = ix - start[0] + ixp + nm[0] * (iy - start[1])
                        j3 = ix - start[0] + ixp + nm[0] * (iy - start[1] + iyp)
                        okx = s[0] <= ix + ixp < e[0]
                        oky = s[1] <= iy + iyp < e[1]
                        for d in range(dim):
                            def cc(i):
                                return c[dim * i + d]
                            if code[0] != 0:
                                if YY + offset[1] == 0:
                                    du = (-0.5 * cc(i0) - 1.5 * cc(i1)) + 2.0 * cc(i2)
                                    du2 = (cc(i0) + cc(i1)) - 2.0 * cc(i2)
                                elif YY + offset[1] == BS // 2 - 1:
                                    du = (0.5 * cc(i3) + 1.5 * cc(i1)) - 2.0 * cc(i4)
                                    du2 = (cc(i3) + cc(i1)) - 2.0 * cc(i4)
                                else:
                                    du = 0.5 * (cc(i2) - cc(i4))
                                    du2 = (cc(i2) + cc(i4)) - 2.0 * cc(i1)
                                m[dim * j0 + d] = cc(i1) + dy * du + (0.5 * dy * dy) * du2
                                if oky:
                                    m[dim * j1 + d] = cc(i1) - dy * du + (0.5 * dy * dy) * du2
                                if okx:
                                    m[dim * j2 + d] = cc(i1) + dy * du + (0.5 * dy * dy) * du2
                                if okx and oky:
                                    m[dim * j3 + d] = cc(i1) - dy * du + (0.5 * dy * dy) * du2
                            else:
                                if XX + offset[0] == 0:
                                    du = (-0.5 * cc(i5) - 1.5 * cc(i1)) + 2.0 * cc(i6)
                                    du2 = (cc(i5) + cc(i1)) - 2.0 * cc(i6)
                                elif XX + offset[0] == BS // 2 - 1:
                                    du = (0.5 * cc(i8) + 1.5 * cc(i1)) - 2.0 * cc(i7)
                                    du2 = (cc(i8) + cc(i1)) - 2.0 * cc(i7)
                                else:
                                    du = 0.5 * (cc(i6) - cc(i7))
                                    du2 = (cc(i6) + cc(i7)) - 2.0 * cc(i1)
                                m[dim * j0 + d] = cc(i1) + dx * du + (0.5 * dx * dx) * du2
                                if oky:
                                    m[dim * j1 + d] = cc(i1) + dx * du + (0.5 * dx * dx) * du2
                                if okx:
                                    m[dim * j2 + d] = cc(i1) - dx * du + (0.5 * dx * dx) * du2
                                if okx and oky:
                                    m[dim * j3 + d] = cc(i1) - dx * du + (0.5 * dx * dx) * du2
                for iy in range(s[1], e[1]):
                    for ix in range(s[0], e[0]):
                        if ix < -2 or iy < -2 or ix > BS + 1 or iy > BS + 1:
                            continue
                        bx, by = ix - start[0], iy - start[1]

                        def K(dxx, dyy):
                            return bx + dxx + nm[0] * (by + dyy)
                        x, y = abs(par(ix, 0)) % 2, abs(par(iy, 1)) % 2
                        for d in range(dim):
                            def M(k):
                                return m[dim * k + d]
                            a = dim * K(0, 0) + d
                            if code[0] == 0 and code[1] == 1:
                                if y == 0:
                                    m[a] = LI(m[a], M(K(0, -1)), M(K(0, -2)))
                                elif y == 1:
                                    m[a] = LE(m[a], M(K(0, -2)), M(K(0, -3)))
                            elif code[0] == 0 and code[1] == -1:
                                if y == 1:
                                    m[a] = LI(m[a], M(K(0, 1)), M(K(0, 2)))
                                elif y == 0:
                                    m[a] = LE(m[a], M(K(0, 2)), M(K(0, 3)))
                            elif code[1] == 0 and code[0] == 1:
                                if x == 0:
                                    m[a] = LI(m[a], M(K(-1, 0)), M(K(-2, 0)))
                                elif x == 1:
                                    m[a] = LE(m[a], M(K(-2, 0)), M(K(-3, 0)))
                            elif code[1] == 0 and code[0] == -1:
                                if x == 1:
                                    m[a] = LI(m[a], M(K(1, 0)), M(K(2, 0)))
                                elif x == 0:
                                    m[a] = LE(m[a], M(K(2, 0)), M(K(3, 0)))
        self._apply_bc(False)
