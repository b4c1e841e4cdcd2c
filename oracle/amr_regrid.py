"""Test infrastructure (never imported by the product): the regrid-time algorithms of the reference's adapt() stated in
Python -- state validation (main.cpp:4718-4861), prolongation / restriction (4981-5032, 5149-5166) on the literal
BlockLab of oracle/amr_lab.py, and the Poisson rows of an adapted grid (7034-7112, 5915-5997).  The library's host
routines (csrc/amr_host.hip: cup2d_amr_validate_states / _regrid / _poisson_coo) are required to reproduce these bit
for bit (tests/test_amr.py); the statements themselves are pinned to the reference's own adapt() and matrix action."""
import numpy as np

from .amr_lab import BlockLab, Tree

BS = 8
LEAVE, REFINE, COMPRESS = 0, 1, 2
AMR_WALL, AMR_SAME, AMR_COARSER, AMR_FINER = 0, 1, 2, 3


def hilbert_index(order_bits, x, y):
    """distance along the 2^order_bits square Hilbert curve (main.cpp:347-360)"""
    x = np.asarray(x, dtype=np.int64).copy()
    y = np.asarray(y, dtype=np.int64).copy()
    n = 1 << order_bits
    d = np.zeros_like(x)
    s = n >> 1
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, n - 1 - x, x)
        y = np.where(flip, n - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


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
                    if k == AMR_WALL:
                        continue
                    n0, n1 = int(self.nbr2[b, s, 0]), int(self.nbr2[b, s, 1])
                    if k == AMR_SAME:
                        c = cell(n0, 7, iy) if s == 0 else cell(n0, 0, iy) if s == 1 else cell(n0, ix, 7) if s == 2 else cell(n0, ix, 0)
                        add(r, c, 1.)
                        add(r, r, -1.)
                    elif k == AMR_COARSER:
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


def regrid_py(blocks, states, fields, level_max):
    """Apply final states: every Refine block becomes its four children (prolonged from the OLD grid's tensorial halo-1
    tile, cup2d_amd/amr_lab.py), every complete Compress sibling group its parent (2x2 means); everything else is
    kept.  fields: {name: (array (nb, 64*dim), dim, is_vector)}.  Returns (new_blocks, new_fields) ordered along the
    Hilbert curve of the finest level (the reference's Info::id2 order, main.cpp:1550-1562)."""
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
