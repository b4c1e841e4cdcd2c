#!/usr/bin/env python3
"""CPU emulation of k_fused's INDEX LOGIC (csrc/krylov_fused.hip), lane by lane in numpy: ring classification
(ballot + prefix popcount), the staging tile with stride XS, the A-operand / D-fragment maps of
v_mfma_f64_16x16x4_f64 as precond_mfma.h documents them, the edge gathers and the stencil.  Checked against
the oracle's y = A P_inv v on Hilbert and row-major grids, incl. partial tiles.  Development aid (no GPU here):
it cannot see compiler or hardware behaviour, only a wrong index."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cup2d_amd.grid import BlockGrid  # noqa: E402
from oracle import oracle as O  # noqa: E402

TB, XS, BS, BC = 16, 66, 8, 64


def edge_cell(s, q):
    return q * BS if s == 0 else q * BS + 7 if s == 1 else q if s == 2 else 56 + q


def tile_precond(S, Pt):
    """S (flat LDS) holds v block-major; emulate xa reads, MFMA (D = X P) and the D-fragment write-back.
    Pt[k][n] = coefficient of input k in output n (what d_Pinv holds)."""
    lanes = np.arange(64)
    ablk, akk = lanes & 15, lanes >> 4
    xa = np.empty((64, 16))
    for ks in range(16):
        xa[:, ks] = S[ablk * XS + 4 * ks + akk]
    # A[i = blk][kk] for k-step ks is held by lane l = kk*16 + blk; B[kk][j] = Pt[4ks+kk][16nt+j] by lane kk*16+j
    X = np.zeros((16, 64))
    for ks in range(16):
        for l in range(64):
            X[l & 15, 4 * ks + (l >> 4)] = xa[l, ks]
    D = X @ Pt  # [blk][n]
    # D fragment: lane l, v -> i = (l/16) + 4v, j = l%16 (n-tile nt)
    for v in range(4):
        for nt in range(4):
            for l in range(64):
                S[((l >> 4) + 4 * v) * XS + 16 * nt + (l & 15)] = D[(l >> 4) + 4 * v, 16 * nt + (l & 15)]


def ring_precond(S, Pt):
    """the ring job of the kernel since the end of round 2 (krylov_fused.hip ring_precond): the product with the 32 EDGE
    columns of P_inv only -- B fragments PE[(ks*2+nt)*64 + l] = Pt[4ks + l/16][edge_cell(col >> 3, col & 7)], col = 16 nt +
    l%16 -- written back as S[e*XS + 8*side + q]"""
    lanes = np.arange(64)
    ablk, akk = lanes & 15, lanes >> 4
    X = np.zeros((16, 64))
    for ks in range(16):
        xa = S[ablk * XS + 4 * ks + akk]
        for l in range(64):
            X[l & 15, 4 * ks + (l >> 4)] = xa[l]
    cols = [edge_cell(c >> 3, c & 7) for c in range(32)]
    PE = np.empty((16, 2, 64))
    for ks in range(16):
        for nt in range(2):
            for l in range(64):
                PE[ks, nt, l] = Pt[4 * ks + (l >> 4), cols[16 * nt + (l & 15)]]
    D = np.zeros((16, 32))
    for ks in range(16):
        for nt in range(2):
            for l in range(64):  # D[i][16 nt + j] += sum_kk A[i][4ks+kk] B[kk][j]: lane l holds B[kk = l/16][j = l%16]
                D[:, 16 * nt + (l & 15)] += X[:, 4 * ks + (l >> 4)] * PE[ks, nt, l]
    for v in range(4):
        for nt in range(2):
            for l in range(64):
                S[((l >> 4) + 4 * v) * XS + 16 * nt + (l & 15)] = D[(l >> 4) + 4 * v, 16 * nt + (l & 15)]


def fused(vin, nbr, Pt, count):
    """y[b][cell] = (A P_inv v) via the kernel's tile logic; vin[b][cell]"""
    y = np.zeros_like(vin)
    ntiles = (count + TB - 1) // TB
    lanes = np.arange(64)
    for t in range(ntiles):
        b0 = t * TB
        nvalid = min(TB, count - b0)
        S = np.full(TB * XS, np.nan)
        GE = np.full(TB * 4 * BS, np.nan)
        si, ss = lanes >> 2, lanes & 3
        nb = np.where(si < nvalid, nbr[np.minimum(b0 + si, count - 1), ss], -1)
        is_ring = (si < nvalid) & (nb >= 0) & ((nb < b0) | (nb >= b0 + nvalid))
        ring_nb, ring_dst = [], []
        for l in range(64):
            if is_ring[l]:
                slot = int(is_ring[:l].sum())
                assert slot == len(ring_nb)
                ring_nb.append(int(nb[l]))
                ring_dst.append(l)
        nring = len(ring_nb)
        for base in range(0, nring, TB):
            ne = min(TB, nring - base)
            for e in range(TB):
                blk = ring_nb[base + min(e, ne - 1)]
                S[e * XS + lanes] = vin[blk]
            ring_precond(S, Pt)
            for h in range(2):
                for l in range(64):
                    idx = l + 64 * h
                    e, q = idx >> 3, idx & 7
                    if e < ne:
                        dst = ring_dst[base + e]
                        GE[dst * BS + q] = S[e * XS + 8 * ((dst & 3) ^ 1) + q]
        for i in range(TB):
            S[i * XS + lanes] = vin[b0 + min(i, nvalid - 1)]
        tile_precond(S, Pt)
        for l in range(64):
            if si[l] < nvalid and not is_ring[l]:
                sblk = si[l] if nb[l] < 0 else nb[l] - b0
                sside = ss[l] if nb[l] < 0 else ss[l] ^ 1
                for q in range(BS):
                    GE[l * BS + q] = S[sblk * XS + edge_cell(sside, q)]
        ix, iy = lanes & 7, lanes >> 3
        for i in range(nvalid):
            zb = i * XS + lanes
            ge = i * 4 * BS
            l0 = S[zb]
            l1 = np.where(ix > 0, S[np.maximum(zb - 1, 0)], GE[ge + 0 * BS + iy])
            l2 = np.where(ix < 7, S[np.minimum(zb + 1, TB * XS - 1)], GE[ge + 1 * BS + iy])
            l3 = np.where(iy > 0, S[np.maximum(zb - BS, 0)], GE[ge + 2 * BS + ix])
            l4 = np.where(iy < 7, S[np.minimum(zb + BS, TB * XS - 1)], GE[ge + 3 * BS + ix])
            y[b0 + i] = l1 + l2 + l3 + l4 - 4 * l0
    return y


def main():
    P = O.P_inv()  # row-major 64x64, symmetric up to round-off
    Pt = P.T.copy()
    rng = np.random.default_rng(11)
    ok = True
    for order, nbx, nby in (("hilbert", 8, 8), ("hilbert", 4, 4), ("rowmajor", 5, 3), ("hilbert", 6, 5), ("rowmajor", 1, 1),
                            ("hilbert", 16, 16)):
        g = BlockGrid(nbx, nby, order=order)
        v = rng.uniform(-1, 1, (g.ny, g.nx))
        ref = O.apply_A(O.precond(v, P))
        got = g.from_blocks(fused(g.to_blocks(v), g.nbr, Pt, g.nblocks), 1)
        err = np.abs(got - ref).max()
        print("%-9s %2dx%-2d  max|fused - oracle| = %.2e" % (order, nbx, nby, err))
        ok = ok and np.isfinite(got).all() and err < 1e-13
    print("EMULATION_%s" % ("OK" if ok else "FAILED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
