#!/usr/bin/env python3
"""CPU emulation of k_edge's INDEX LOGIC (csrc/krylov_edge.h), lane by lane in numpy: the workgroup's rounds of 8
consecutive tiles, classification of the 64 (block, side) slots of a tile into own / wall / sibling / recompute, the
ring jobs and the tile job as products with the 32 edge columns of P_inv (fragment maps of precond_mfma.h), the export of
perimeter edges in slot order (prefix popcount of the perimeter mask, two buffers by round parity), the consumers' slot
look-up in a sibling's export, the ghost-edge gathers and the epilogue y = v + ghosts.  Checked against the oracle's
y = A P_inv v on Hilbert and row-major grids incl. partial tiles and partial rounds, with sharing on and off, for several
grid sizes (G workgroups); the waves of a round are visited in a random order, exports before consumers (what the kernel's
flag wait enforces), and a consumer asserts that the buffer it reads holds ITS round and that the slot it asks for is in the
sibling's mask.  Development aid (no GPU here): it cannot see compiler, hardware or the timing of the flags, only a wrong
index."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cup2d_amd.grid import BlockGrid  # noqa: E402
from oracle import oracle as O  # noqa: E402

TB, XS, BS, BC, GS, FWAVES, EXP_SLOTS = 16, 66, 8, 64, 10, 8, 16
LANES = np.arange(64)


def edge_cell(s, q):
    return q * BS if s == 0 else q * BS + 7 if s == 1 else q if s == 2 else 56 + q


def ring_precond(S, Pt):
    """krylov_fused.hip ring_precond: S (v, block-major, stride XS) -> S[e*XS + 8*side + q] = z of block e on its edges"""
    ablk, akk = LANES & 15, LANES >> 4
    X = np.zeros((16, 64))
    for ks in range(16):
        xa = S[ablk * XS + 4 * ks + akk]
        for l in range(64):
            X[l & 15, 4 * ks + (l >> 4)] = xa[l]
    cols = [edge_cell(c >> 3, c & 7) for c in range(32)]
    D = np.zeros((16, 32))
    for ks in range(16):
        for nt in range(2):
            for l in range(64):  # PE[(ks*2+nt)*64 + l] = Pt[4ks + l/16][cols[16nt + l%16]]
                D[:, 16 * nt + (l & 15)] += X[:, 4 * ks + (l >> 4)] * Pt[4 * ks + (l >> 4), cols[16 * nt + (l & 15)]]
    for v in range(4):
        for nt in range(2):
            for l in range(64):
                S[((l >> 4) + 4 * v) * XS + 16 * nt + (l & 15)] = D[(l >> 4) + 4 * v, 16 * nt + (l & 15)]


def share_ok(nbr, first, count):
    for b0 in range(first, first + count, TB):
        nv = min(TB, first + count - b0)
        n = 0
        for b in range(b0, b0 + nv):
            for s in range(4):
                nb = nbr[b, s]
                n += nb >= 0 and (nb < b0 or nb >= b0 + nv)
        if n > EXP_SLOTS:
            return False
    return True


class Wave:
    def __init__(self):
        self.S = np.full(TB * XS, np.nan)
        self.GE = np.full(TB * 4 * GS, np.nan)
        self.X = np.full((2, EXP_SLOTS * BS), np.nan)
        self.xmask = [0, 0]
        self.xround = [-1, -1]   # emulation only: which round each export buffer holds


def tile_ranges(ntiles, G, w):
    """t_begin of wave 0, t_end, t_stride of workgroup w (k_edge's prologue)"""
    if G >= 8 and G % 8 == 0:
        xcd, slot, per = w & 7, w >> 3, G >> 3
        lo, hi = ntiles * xcd // 8, ntiles * (xcd + 1) // 8
        return lo + slot * FWAVES, hi, per * FWAVES
    return w * FWAVES, ntiles, G * FWAVES


def edge_form(vin, nbr, Pt, first, count, G, share, order_seed=0, rev=False):
    """rev: the workgroup's rounds in descending tile order (FusedArgs::rev: the C+D launch of the two-launch organisation);
    the rounds are numbered as the kernel numbers them -- a wave without a tile in the partial round (the first one when
    descending) skips it and keeps the workgroup's numbering"""
    y = np.full_like(vin, np.nan)
    ntiles = (count + TB - 1) // TB
    last = first + count
    si, ss = LANES >> 2, LANES & 3
    rng = np.random.default_rng(order_seed)
    for w in range(G):
        t0_begin, t_end, t_stride = tile_ranges(ntiles, G, w)
        waves = [Wave() for _ in range(FWAVES)]
        nrounds = (t_end - 1 - t0_begin) // t_stride + 1 if t0_begin < t_end else 0
        for rnd in range(nrounds):
            t0 = t0_begin + (nrounds - 1 - rnd if rev else rnd) * t_stride
            par = rnd & 1
            tiles = {}
            # ---- every wave with a tile: classification, ring jobs, tile job, export (phase 1).  The waves are visited in
            #      a random order; phase 2 (consumers) runs after all exports of the round: the flag wait of the kernel ----
            order = list(rng.permutation(FWAVES))
            for wave in order:
                t = t0 + wave
                if t >= t_end:
                    continue
                Wv = waves[wave]
                b0 = first + t * TB
                nvalid = min(TB, last - b0)
                nb = np.where((si < nvalid), nbr[np.minimum(b0 + si, last - 1), ss], -1)
                outside = (si < nvalid) & (nb >= 0) & ((nb < b0) | (nb >= b0 + nvalid))
                nt = (nb - first) // TB
                sibling = bool(share) & outside & (nb >= first) & (nb < last) & (nt >= t0) & (nt < t0 + FWAVES) & (nt < t_end)
                sib = np.where(sibling, nt - t0, -1)
                pmask = 0
                for l in range(64):
                    if outside[l]:
                        pmask |= 1 << l
                is_ring = outside & ~sibling
                ring_nb, ring_dst = [], []
                for l in range(64):
                    if is_ring[l]:
                        ring_nb.append(int(nb[l]))
                        ring_dst.append(l)
                nring = len(ring_nb)
                S, GE = Wv.S, Wv.GE
                S[:] = np.nan
                GE[:] = np.nan
                for base in range(0, nring, TB):
                    ne = min(TB, nring - base)
                    for e in range(TB):
                        S[e * XS + LANES] = vin[ring_nb[base + min(e, ne - 1)]]
                    ring_precond(S, Pt)
                    for h in range(2):
                        for l in range(64):
                            idx = l + 64 * h
                            e, q = idx >> 3, idx & 7
                            if e < ne:
                                dst = ring_dst[base + e]
                                GE[dst * GS + q] = S[e * XS + 8 * ((dst & 3) ^ 1) + q]
                V = np.empty((TB, 64))
                for i in range(TB):
                    V[i] = vin[b0 + min(i, nvalid - 1)]
                    S[i * XS + LANES] = V[i]
                ring_precond(S, Pt)
                if share:
                    for l in range(64):
                        if (pmask >> l) & 1:
                            slot = bin(pmask & ((1 << l) - 1)).count("1")
                            assert slot < EXP_SLOTS
                            for q in range(BS):
                                Wv.X[par][slot * BS + q] = S[si[l] * XS + 8 * ss[l] + q]
                    Wv.xmask[par] = pmask
                    Wv.xround[par] = rnd
                for l in range(64):
                    if si[l] < nvalid and not ((pmask >> l) & 1):
                        sblk = si[l] if nb[l] < 0 else nb[l] - b0
                        sside = ss[l] if nb[l] < 0 else ss[l] ^ 1
                        for q in range(BS):
                            GE[l * GS + q] = S[sblk * XS + 8 * sside + q]
                tiles[wave] = (b0, nvalid, nb, sib, V)
            # ---- phase 2: ghost edges from the siblings' exports, epilogue ----
            for wave in list(rng.permutation(FWAVES)):
                if wave not in tiles:
                    continue
                b0, nvalid, nb, sib, V = tiles[wave]
                Wv = waves[wave]
                for u in range(FWAVES):
                    for l in range(64):
                        if sib[l] != u:
                            continue
                        Ou = waves[u]
                        assert Ou.xround[par] == rnd, "consumer read an export of another round"
                        m = Ou.xmask[par]
                        bit = int(nb[l] - (b0 + (u - wave) * TB)) * 4 + int(ss[l] ^ 1)
                        assert (m >> bit) & 1, "the needed slot is not in the sibling's perimeter mask"
                        slot = min(bin(m & ((1 << bit) - 1)).count("1"), EXP_SLOTS - 1)
                        for q in range(BS):
                            Wv.GE[l * GS + q] = Ou.X[par][slot * BS + q]
                hl = LANES & 31
                hf = LANES >> 5
                c0 = 2 * hl
                px, py = c0 & 7, hl >> 2
                for i in range(TB // 2):
                    for l in range(64):
                        blk = 2 * i + hf[l]
                        if blk >= nvalid:
                            continue
                        ge = blk * 4 * GS
                        yx, yy = V[blk][c0[l]], V[blk][c0[l] + 1]
                        if px[l] == 0:
                            yx += Wv.GE[ge + 0 * GS + py[l]]
                        if px[l] == BS - 2:
                            yy += Wv.GE[ge + 1 * GS + py[l]]
                        if py[l] == 0:
                            yx += Wv.GE[ge + 2 * GS + px[l]]
                            yy += Wv.GE[ge + 2 * GS + px[l] + 1]
                        if py[l] == BS - 1:
                            yx += Wv.GE[ge + 3 * GS + px[l]]
                            yy += Wv.GE[ge + 3 * GS + px[l] + 1]
                        assert np.isnan(y[b0 + blk][c0[l]]), "cell written twice"
                        y[b0 + blk][c0[l]] = yx
                        y[b0 + blk][c0[l] + 1] = yy
    return y


def main():
    P = O.P_inv()  # row-major 64x64, symmetric up to round-off
    Pt = P.T.copy()
    rng = np.random.default_rng(11)
    ok = True
    cases = (("hilbert", 8, 8, 1), ("hilbert", 8, 8, 8), ("hilbert", 4, 4, 8), ("rowmajor", 5, 3, 1), ("hilbert", 6, 5, 8),
             ("rowmajor", 1, 1, 1), ("hilbert", 16, 16, 8), ("hilbert", 16, 16, 1), ("hilbert", 32, 16, 8), ("rowmajor", 16, 16, 1),
             ("hilbert", 32, 32, 16))
    for order, nbx, nby, G in cases:
        g = BlockGrid(nbx, nby, order=order)
        v = rng.uniform(-1, 1, (g.ny, g.nx))
        ref = O.apply_A(O.precond(v, P))
        for share in (0, 1):
            for rev in (False, True):
                sh = share and share_ok(g.nbr, 0, g.nblocks)
                got = g.from_blocks(edge_form(g.to_blocks(v), g.nbr, Pt, 0, g.nblocks, G, sh, order_seed=nbx + G, rev=rev), 1)
                err = np.abs(got - ref).max()
                print("%-9s %2dx%-2d G=%-2d share=%d(%s) %s  max|edge form - oracle| = %.2e"
                      % (order, nbx, nby, G, share, "on" if sh else "off", "descending" if rev else "ascending ", err))
                ok = ok and np.isfinite(got).all() and err < 2e-13
    print("EMULATION_%s" % ("OK" if ok else "FAILED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
