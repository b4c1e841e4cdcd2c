"""Block-grid topology on the host: which 8x8 blocks exist, in what order, and who their
neighbours are.  Replaces the reference's Info/Grid hash maps and space-filling-curve tables
(main.cpp:342-450, 672-738, 2193-2201) for same-level grids with dense index tables.

Block order matters only for locality: the default is the Hilbert order the reference itself
uses for Info::id (main.cpp:1550-1562), under which every aligned run of 4^k consecutive blocks
is a compact 2^k x 2^k patch -- the ghost cells a wavefront reads were fetched by a neighbouring
wavefront of the same workgroup or XCD.
"""
import numpy as np

BS = 8
WALL = -1


def hilbert_index(order_bits, x, y):
    """Distance along the 2^order_bits square Hilbert curve (the classic xy2d the reference's
    SpaceCurve::AxestoTranspose implements, main.cpp:347-360).  Vectorised over numpy arrays."""
    x = np.asarray(x, dtype=np.int64).copy()
    y = np.asarray(y, dtype=np.int64).copy()
    n = 1 << order_bits
    d = np.zeros_like(x)
    s = n >> 1
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate quadrant
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, n - 1 - x, x)
        y = np.where(flip, n - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


class BlockGrid:
    """A rectangular patch of nbx x nby same-level blocks with walls or ghost blocks around it.

    coords[b] = (bx, by) of device block b.  nbr[b] = (W, E, S, N) device indices, WALL at a domain
    wall, >= nblocks for ghost blocks (when `ghost` sides are given, for domain decomposition).
    Blocks that touch a ghost block -- whole aligned patches of `halo_tile` x `halo_tile` blocks around them, see below --
    are ordered last (`n_inner` = first such index), mirroring the inner/halo split of the reference's synchroniser
    (main.cpp:1115-1117).
    """

    def __init__(self, nbx, nby, order="hilbert", ghost_sides=(False, False, False, False)):
        self.nbx, self.nby = int(nbx), int(nby)
        self.nblocks = self.nbx * self.nby
        self.ghost_sides = tuple(bool(g) for g in ghost_sides)  # W, E, S, N
        bx, by = np.meshgrid(np.arange(self.nbx), np.arange(self.nby), indexing="xy")
        bx, by = bx.ravel(), by.ravel()
        if order == "hilbert":
            bits = max(1, int(np.ceil(np.log2(max(self.nbx, self.nby, 2)))))
            key = hilbert_index(bits, bx, by)
        elif order == "rowmajor":
            key = by * self.nbx + bx
        else:
            raise ValueError("order must be 'hilbert' or 'rowmajor'")
        gW, gE, gS, gN = self.ghost_sides
        # Which blocks count as "halo" (ordered last, swept after the exchange has arrived).  Taking exactly the blocks that
        # touch a ghost block would cut single blocks out of the Hilbert order: every 16-block tile of the fused Krylov
        # sweeps behind the first cut then straddles two 4 x 4 patches (20+ perimeter sides instead of 16: two ring passes, no
        # hand-over between sibling waves -- measured on a 512 x 256-block patch: C+D' 57 -> 68 us, E+A+B 110 -> 135 us), and
        # the one-block-thick ring has no 2 x 2 quads for the WENO walk.  So the halo set is made of WHOLE aligned patches of
        # g x g blocks (= aligned runs of g^2 blocks of the Hilbert order): g = 16 where the patch is large enough to keep an
        # interior (rounds of 8 tiles stay 16 x 8 patches), else 4 (tiles stay 4 x 4 patches), else single blocks.
        g = 1
        if order == "hilbert" and any(self.ghost_sides):
            for cand in (16, 4):
                if self.nbx % cand == 0 and self.nby % cand == 0 and min(self.nbx, self.nby) >= 4 * cand:
                    g = cand
                    break
        self.halo_tile = g
        cx, cy = bx // g, by // g
        touches = ((cx == 0) & gW) | ((cx == self.nbx // g - 1) & gE) | ((cy == 0) & gS) | ((cy == self.nby // g - 1) & gN)
        perm = np.lexsort((key, touches.astype(np.int64)))  # inner first, then halo; Hilbert inside each
        self.coords = np.stack([bx[perm], by[perm]], axis=1).astype(np.int64)
        self.n_inner = int(self.nblocks - touches.sum())
        self.index_of = -np.ones((self.nby, self.nbx), dtype=np.int64)
        self.index_of[self.coords[:, 1], self.coords[:, 0]] = np.arange(self.nblocks)
        # ghost blocks: one per boundary block on each ghost side, numbered after the owned blocks
        # in (side, position) order
        self.ghost_coords = []  # (side, pos) ; side 0..3 = W,E,S,N of this patch
        ghost_id = {}
        for side, on in enumerate(self.ghost_sides):
            if not on:
                continue
            for pos in range(self.nby if side < 2 else self.nbx):
                ghost_id[(side, pos)] = self.nblocks + len(self.ghost_coords)
                self.ghost_coords.append((side, pos))
        self.nghost = len(self.ghost_coords)
        self._ghost_id = ghost_id
        nbr = np.empty((self.nblocks, 4), dtype=np.int32)
        for b in range(self.nblocks):
            x, y = self.coords[b]
            nbr[b, 0] = self.index_of[y, x - 1] if x > 0 else (ghost_id[(0, y)] if gW else WALL)
            nbr[b, 1] = self.index_of[y, x + 1] if x < self.nbx - 1 else (ghost_id[(1, y)] if gE else WALL)
            nbr[b, 2] = self.index_of[y - 1, x] if y > 0 else (ghost_id[(2, x)] if gS else WALL)
            nbr[b, 3] = self.index_of[y + 1, x] if y < self.nby - 1 else (ghost_id[(3, x)] if gN else WALL)
        self.nbr = np.ascontiguousarray(nbr)

    @property
    def nx(self):
        return self.nbx * BS

    @property
    def ny(self):
        return self.nby * BS

    # ---- global row-major <-> block slab --------------------------------------------------------
    def to_blocks(self, a):
        """(ny, nx[, dim]) row-major -> (nblocks, 64*dim) in device block order"""
        a = np.asarray(a, dtype=np.float64)
        dim = 1 if a.ndim == 2 else a.shape[2]
        t = a.reshape(self.nby, BS, self.nbx, BS, dim).transpose(0, 2, 1, 3, 4)  # (by, bx, iy, ix, dim)
        t = t[self.coords[:, 1], self.coords[:, 0]]
        return np.ascontiguousarray(t.reshape(self.nblocks, BS * BS * dim))

    def from_blocks(self, slab, dim):
        slab = np.asarray(slab, dtype=np.float64).reshape(self.nblocks, BS, BS, dim)
        out = np.empty((self.nby, self.nbx, BS, BS, dim))
        out[self.coords[:, 1], self.coords[:, 0]] = slab
        out = out.transpose(0, 2, 1, 3, 4).reshape(self.ny, self.nx, dim)
        return np.ascontiguousarray(out[..., 0] if dim == 1 else out)

    # ---- Poisson matrix of main.cpp:7034-7112 on this same-level grid ---------------------------
    def poisson_coo(self):
        """Local COO triplets (row, col, val) of the matrix the reference assembles for this block
        grid, rows/columns numbered 64*block + 8*iy + ix in device block order: 1 towards every
        existing neighbour cell, -(their number) on the diagonal, nothing across a domain wall
        (main.cpp:7075-7105 with the same-level branch of Solver::makeFlux, main.cpp:5946-5951).
        Columns of ghost blocks (index >= nblocks) address halo entries, block-wise."""
        nb = self.nblocks
        cell = np.arange(BS * BS)
        ix, iy = cell % BS, cell // BS
        rows, cols = [], []
        base = (np.arange(nb) * BS * BS)[:, None]
        me = base + cell[None, :]
        # (dx, dy, side of nbr, mirrored cell in the neighbouring block)
        for dx, dy, side in ((-1, 0, 0), (1, 0, 1), (0, -1, 2), (0, 1, 3)):
            jx, jy = ix + dx, iy + dy
            inside = (jx >= 0) & (jx < BS) & (jy >= 0) & (jy < BS)
            # in-block neighbours
            r = me[:, inside]
            c = base + (jy[inside] * BS + jx[inside])[None, :]
            rows.append(r.ravel())
            cols.append(c.ravel())
            # across the face
            edge = ~inside
            nbr = self.nbr[:, side].astype(np.int64)
            has = nbr >= 0
            r = me[has][:, edge]
            c = nbr[has][:, None] * BS * BS + (((jy[edge] + BS) % BS) * BS + (jx[edge] + BS) % BS)[None, :]
            rows.append(r.ravel())
            cols.append(c.ravel())
        rows = np.concatenate(rows)
        cols = np.concatenate(cols)
        vals = np.ones(rows.size)
        deg = np.bincount(rows, minlength=nb * BS * BS).astype(np.float64)
        allr = np.arange(nb * BS * BS)
        rows = np.concatenate([rows, allr])
        cols = np.concatenate([cols, allr])
        vals = np.concatenate([vals, -deg])
        return rows.astype(np.int32), cols.astype(np.int32), vals
