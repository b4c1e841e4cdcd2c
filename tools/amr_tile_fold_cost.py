#!/usr/bin/env python3
"""What folding the coarse-fine rows into the tile-fused sweeps would ask of a general tile: for the tiling
csrc/api.hip install_sell cuts on an adapted grid, per tile that holds stored rows, how many blocks OUTSIDE the tile its rows
read (their z = P_inv v would have to be formed by the tile's wave: one 64 x 64 product each) and how many cells of them.
Host only (cup2d_amr_trace_reads); LFINE as tools/gpu_amr_bench.py."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cup2d_amd import amr as A, lib as L  # noqa: E402

LF = int(os.environ.get("LFINE", "9"))
TILE = 16
g = A.circle_band_grid(LF)
nb = g.nblocks
lib = L.load_library()
vp = ctypes.c_void_p
kind, nbr2, half = (np.ascontiguousarray(a, dtype=np.int32) for a in (g.kind, g.nbr2, g.half))
stored = ((kind == L.AMR_COARSER) | (kind == L.AMR_FINER)).any(1)
reg = np.where(kind == L.AMR_SAME, nbr2[:, :, 0], -1)
# the tiling of install_sell: 16 plain slices with <= 16 neighbour slots outside the set start a good tile; everything else is
# cut into chunks up to the next good start
plain_run = np.zeros(nb + 1, dtype=np.int64)
for s in range(nb - 1, -1, -1):
    plain_run[s] = 0 if stored[s] else plain_run[s + 1] + 1


def good(s):
    if plain_run[s] < TILE:
        return False
    r = reg[s:s + TILE]
    return int(((r >= 0) & ((r < s) | (r >= s + TILE))).sum()) <= TILE


tiles, s = [], 0
while s < nb:
    if good(s):
        tiles.append((s, s + TILE)); s += TILE; continue
    e = s + 1
    while e < nb and e - s < TILE and not good(e):
        e += 1
    tiles.append((s, e)); s = e
general = [(a, b) for a, b in tiles if stored[a:b].any()]
print("grid: %d blocks, %d with stored rows; %d tiles, %d general (%d blocks in them)" % (nb, stored.sum(), len(tiles), len(general), sum(b - a for a, b in general)))
foreign, cells, ring_plain = [], [], []
for a, b in general:
    rd = np.arange(a, b, dtype=np.int32)
    mask = np.zeros(nb, dtype=np.uint64)
    L.check(lib.cup2d_amr_trace_reads(nb, kind.ctypes.data_as(vp), nbr2.ctypes.data_as(vp), half.ctypes.data_as(vp), len(rd), rd.ctypes.data_as(vp),
                                      L.CELLS_MATRIX, mask.ctypes.data_as(vp)), "trace")
    mask[a:b] = 0
    nz = np.nonzero(mask)[0]
    foreign.append(len(nz))
    cells.append(int(sum(bin(int(m)).count("1") for m in mask[nz])))
foreign, cells = np.asarray(foreign), np.asarray(cells)
print("blocks outside the tile that a general tile's rows read: mean %.1f, median %d, 90%% %d, max %d  (a plain 4 x 4 tile: 16 ring entries, edges only)"
      % (foreign.mean(), np.median(foreign), np.quantile(foreign, 0.9), foreign.max()))
print("cells of them: mean %.0f, max %d  (= %.0f doubles of z per tile if only the read cells were kept)" % (cells.mean(), cells.max(), cells.mean()))
jobs = np.ceil(foreign / TILE)
print("64 x 64 products for them in jobs of 16 blocks: mean %.2f jobs per general tile (a tile's own job: 1), i.e. +%.0f %% MFMA jobs over all %d tiles"
      % (jobs.mean(), 100.0 * jobs.sum() / len(tiles), len(tiles)))
print("loads for them: %d blocks x 3 vectors x 512 B = %.1f MB per sweep A+B (the sweep moves %.0f MB)"
      % (foreign.sum(), foreign.sum() * 3 * 512 / 1e6, nb * 64 * 48 / 1e6))
