// amr_host.hip -- regrid-time HOST code for block-AMR grids (no kernels): the Poisson matrix of an adapted grid.
//
// Reference: the serial row-assembly loop main.cpp:7034-7112 with Solver::makeFlux / interpolate / D1 / D2
// (main.cpp:5915-5997) and SpRowInfo::mapColVal (cuda.h:1-24).  In the reference this is host C++ too; here it works
// on the dense topology tables of cup2d_set_amr instead of the tree / Info hash maps.  (The tests keep a Python
// statement of the same algorithm and require the two to agree bit for bit.)
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

#include "amr_ghost.h"
#include "block.h"
#include "ctx.h"

namespace {

constexpr int BS = CUP2D_BS;

using cup2d::chunk_count;
using cup2d::host_threads;
using cup2d::parallel_chunks;

struct Row {  // a row under construction: duplicate columns are summed in the order they arrive (mapColVal)
  int n = 0;
  long long col[24];
  double val[24];
  void add(long long c, double v) {
    for (int k = 0; k < n; k++)
      if (col[k] == c) { val[k] += v; return; }
    col[n] = c;
    val[n] = v;
    n++;
  }
};
inline long long cell(int b, int ix, int iy) { return 64LL * b + 8 * iy + ix; }

// first / second derivative stencils ALONG the face at coarse cell (ix, iy) of block b (D1, D2 main.cpp:5915-5942)
void taylor(int b, int s, int ix, int iy, bool second, long long (&c)[3], double (&w)[3]) {
  const int t = s < 2 ? iy : ix;
  const auto nei = [&](int d) { return s < 2 ? cell(b, ix, iy + d) : cell(b, ix + d, iy); };
  if (t == 7 || t == 3) {
    c[0] = nei(-2); c[1] = nei(-1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = -1. / 16.; w[2] = 1. / 32.; } else { w[0] = 1. / 8.; w[1] = -1. / 2.; w[2] = 3. / 8.; }
  } else if (t == 0 || t == 4) {
    c[0] = nei(2); c[1] = nei(1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = -1. / 16.; w[2] = 1. / 32.; } else { w[0] = -1. / 8.; w[1] = 1. / 2.; w[2] = -3. / 8.; }
  } else {
    c[0] = nei(-1); c[1] = nei(1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = 1. / 32.; w[2] = -1. / 16.; } else { w[0] = -1. / 8.; w[1] = 1. / 8.; w[2] = 0.; }
  }
}
// Solver::interpolate main.cpp:5943-5960
void interpolate(Row &r, int bc, int s, int ixc, int iyc, long long fine_close, long long fine_far, double sign_int,
                 double sign_taylor) {
  r.add(fine_close, sign_int * 2. / 3.);
  r.add(fine_far, -sign_int * 1. / 5.);
  const double tf = sign_int * 8. / 15.;
  r.add(cell(bc, ixc, iyc), tf);
  long long c[3];
  double w[3];
  taylor(bc, s, ixc, iyc, false, c, w);
  for (int i = 0; i < 3; i++) r.add(c[i], sign_taylor * tf * w[i]);
  taylor(bc, s, ixc, iyc, true, c, w);
  for (int i = 0; i < 3; i++) r.add(c[i], tf * w[i]);
}

}  // namespace

namespace {

// one row of main.cpp:7034-7112: cell (ix, iy) of block b
void build_row(Row &r, int b, int ix, int iy, const int32_t *kind, const int32_t *nbr2, const int32_t *half) {
  const long long me = cell(b, ix, iy);
  if (ix > 0 && ix < BS - 1 && iy > 0 && iy < BS - 1) {  // main.cpp:7075-7087
    r.add(cell(b, ix, iy - 1), 1.);
    r.add(cell(b, ix - 1, iy), 1.);
    r.add(me, -4.);
    r.add(cell(b, ix + 1, iy), 1.);
    r.add(cell(b, ix, iy + 1), 1.);
    return;
  }
  const bool inblock[4] = {ix > 0, ix < BS - 1, iy > 0, iy < BS - 1};
  const long long inner[4] = {cell(b, ix - 1, iy), cell(b, ix + 1, iy), cell(b, ix, iy - 1), cell(b, ix, iy + 1)};
  for (int s = 0; s < 4; s++) {
    if (inblock[s]) {
      r.add(inner[s], 1.);
      r.add(me, -1.);
      continue;
    }
    const int k = kind[4 * b + s];
    if (k == CUP2D_AMR_WALL) continue;
    const int n0 = nbr2[(4 * b + s) * 2], n1 = nbr2[(4 * b + s) * 2 + 1];
    if (k == CUP2D_AMR_SAME) {  // makeFlux, same level
      r.add(s == 0 ? cell(n0, 7, iy) : s == 1 ? cell(n0, 0, iy) : s == 2 ? cell(n0, ix, 7) : cell(n0, ix, 0), 1.);
      r.add(me, -1.);
    } else if (k == CUP2D_AMR_COARSER) {
      const int h = half[4 * b + s];
      const int ixc = s == 0 ? 7 : s == 1 ? 0 : ix / 2 + 4 * h;
      const int iyc = s == 2 ? 7 : s == 3 ? 0 : iy / 2 + 4 * h;
      const long long inward = s == 0 ? cell(b, ix + 1, iy) : s == 1 ? cell(b, ix - 1, iy) : s == 2 ? cell(b, ix, iy + 1)
                                                                                            : cell(b, ix, iy - 1);
      const int t = s < 2 ? iy : ix;
      interpolate(r, n0, s, ixc, iyc, me, inward, 1., t % 2 == 0 ? -1. : 1.);
      r.add(me, -1.);
    } else {  // the two finer cells across the face, in the child that covers this cell
      const int t = s < 2 ? iy : ix;
      const int fb = t >= 4 ? n1 : n0, f = (t % 4) * 2;
      for (int j = 0; j < 2; j++) {
        long long close, far;
        if (s == 0) { close = cell(fb, 7, f + j); far = cell(fb, 6, f + j); }
        else if (s == 1) { close = cell(fb, 0, f + j); far = cell(fb, 1, f + j); }
        else if (s == 2) { close = cell(fb, f + j, 7); far = cell(fb, f + j, 6); }
        else { close = cell(fb, f + j, 0); far = cell(fb, f + j, 1); }
        r.add(close, 1.);
        interpolate(r, b, s, ix, iy, close, far, -1., j == 0 ? -1. : 1.);
      }
    }
  }
}

}  // namespace

extern "C" long long cup2d_amr_poisson_coo(int nblocks, const int32_t *kind, const int32_t *nbr2, const int32_t *half,
                                           long long cap, int32_t *row, int32_t *col, double *val) {
  if (nblocks <= 0 || !kind || !nbr2 || !half) {
    cup2d::set_error("amr_poisson_coo: bad argument");
    return CUP2D_ERR_ARG;
  }
  if ((long long)nblocks * 64 > 2147483647LL) {
    cup2d::set_error("amr_poisson_coo: %d blocks exceed int32 row indices", nblocks);
    return CUP2D_ERR_ARG;
  }
  const bool fill = row && col && val;
  long long nnz = 0;
  std::vector<std::pair<long long, double>> sorted;
  for (int b = 0; b < nblocks; b++)
    for (int iy = 0; iy < BS; iy++)
      for (int ix = 0; ix < BS; ix++) {
        Row r;
        const long long me = cell(b, ix, iy);
        build_row(r, b, ix, iy, kind, nbr2, half);
        if (fill) {
          if (nnz + r.n > cap) {
            cup2d::set_error("amr_poisson_coo: capacity %lld too small", cap);
            return CUP2D_ERR_ARG;
          }
          sorted.clear();
          for (int k = 0; k < r.n; k++) sorted.emplace_back(r.col[k], r.val[k]);
          std::sort(sorted.begin(), sorted.end(), [](const auto &a, const auto &c2) { return a.first < c2.first; });
          for (const auto &e : sorted) {
            row[nnz] = (int32_t)me;
            col[nnz] = (int32_t)e.first;
            val[nnz] = e.second;
            nnz++;
          }
        } else {
          nnz += r.n;
        }
      }
  return nnz;
}

// Which cells of OTHER blocks the operators of the listed blocks read: the very expressions the kernels evaluate (amr_ghost
// / amr_ghost3 of amr_ghost.h, build_row above), run with an accessor that records instead of reading.  An N-rank exchange
// that delivers exactly these cells of a ghost block (instead of the whole block) leaves every kernel's input unchanged --
// what the reference's synchroniser arrives at with its per-stencil strips (main.cpp:2053-2125, 2582-2684).
extern "C" int cup2d_amr_trace_reads(int nblocks, const int32_t *kind, const int32_t *nbr2, const int32_t *half, int nreaders,
                                     const int32_t *readers, int set, uint64_t *mask) {
  if (nblocks <= 0 || !kind || !nbr2 || !half || nreaders < 0 || (nreaders && !readers) || !mask || set < 0 || set > 2) {
    cup2d::set_error("amr_trace_reads: bad argument");
    return CUP2D_ERR_ARG;
  }
  for (int i = 0; i < nreaders; i++)
    if (readers[i] < 0 || readers[i] >= nblocks) {
      cup2d::set_error("amr_trace_reads: readers[%d] = %d", i, readers[i]);
      return CUP2D_ERR_ARG;
    }
  cup2d::AmrDev T;
  T.kind = kind; T.nbr2 = nbr2; T.half = half; T.level = nullptr; T.faces = nullptr; T.h0 = 0.0;
  int bad = 0;
  const auto mark = [&](int self, int blk, int cell) {
    if (blk < 0 || blk >= nblocks || cell < 0 || cell >= 64) { bad++; return; }
    if (blk != self) mask[blk] |= 1ull << cell;
  };
  for (int i = 0; i < nreaders; i++) {
    const int b = readers[i];
    if (set == 0) {  // the halo-1 operators: k_amr_scalar / k_amr_vector (32 cross ghosts)
      for (int s = 0; s < 4; s++)
        for (int q = 0; q < BS; q++) {
          const auto get = [&](int blk, int cell) { mark(b, blk, cell); return 0.0; };
          (void)cup2d::amr_ghost(get, kind[4 * b + s], nbr2[(4 * b + s) * 2], nbr2[(4 * b + s) * 2 + 1], half[4 * b + s], s, q, 0.0,
                                 0.0, 1.0);
        }
    } else if (set == 1) {  // the halo-3 tile of k_amr_advect (96 cross ghosts)
      for (int s = 0; s < 4; s++)
        for (int k = 0; k < 3; k++)
          for (int q = 0; q < BS; q++) {
            const auto get = [&](int blk, int cell) { mark(b, blk, cell); return double2{0.0, 0.0}; };
            (void)cup2d::amr_ghost3(get, T, b, s, k, q, double2{0.0, 0.0}, double2{0.0, 0.0});
          }
    } else {  // the columns of the block's rows of the Poisson matrix (interior rows stay in the block)
      for (int iy = 0; iy < BS; iy++)
        for (int ix = 0; ix < BS; ix++) {
          if (ix > 0 && ix < BS - 1 && iy > 0 && iy < BS - 1) continue;
          Row r;
          build_row(r, b, ix, iy, kind, nbr2, half);
          for (int k = 0; k < r.n; k++) mark(b, (int)(r.col[k] / 64), (int)(r.col[k] % 64));
        }
    }
  }
  if (bad) {
    cup2d::set_error("amr_trace_reads: %d reads outside the tables (a neighbour entry of a reader is missing)", bad);
    return CUP2D_ERR_ARG;
  }
  return CUP2D_OK;
}

// computeA's inner / halo split on an adapted grid (the lists behind CUP2D_BLOCKS_INNER / _HALO, amr.hip amr_phase_lists) as a
// host routine that needs no context: reads_ghost[b] = 1 where the operators of family `set` (0: halo-1 operators, 1: the halo-3
// tile of KernelAdvectDiffuse) of owned block b read a cell of a block >= nowned, else 0
extern "C" int cup2d_amr_blocks_reading_ghosts(int nowned, int ntotal, const int32_t *kind, const int32_t *nbr2, const int32_t *half,
                                               int set, int32_t *reads_ghost) {
  if (nowned < 0 || ntotal < nowned || !kind || !nbr2 || !half || !reads_ghost || set < 0 || set > 1) {
    cup2d::set_error("amr_blocks_reading_ghosts: bad argument");
    return CUP2D_ERR_ARG;
  }
  for (int b = 0; b < ntotal; b++)
    for (int e = 0; e < 8; e++) {
      const int k = kind[4 * b + (e >> 1)], nb = nbr2[8 * b + e];
      if (k < CUP2D_AMR_WALL || k > CUP2D_AMR_FINER || (k != CUP2D_AMR_WALL && (e & 1) == 0 && (nb < 0 || nb >= ntotal)) ||
          (k == CUP2D_AMR_FINER && (nb < 0 || nb >= ntotal))) {
        cup2d::set_error("amr_blocks_reading_ghosts: block %d side %d: kind %d neighbour %d", b, e >> 1, k, nb);
        return CUP2D_ERR_ARG;
      }
    }
  std::vector<int32_t> inner, halo;
  cup2d::amr_blocks_reading_ghosts(nowned, ntotal, kind, nbr2, half, set, inner, halo);
  for (int b : inner) reads_ghost[b] = 0;
  for (int b : halo) reads_ghost[b] = 1;
  return CUP2D_OK;
}

namespace cup2d {

// Which owned blocks read a ghost block (id >= nowned) in the operators of family `set` (0: the 32 cross ghosts of the halo-1
// operators, 1: the 96 of the halo-3 tile): the kernels' own expressions with an accessor that only looks at the block id.
// A block further than three neighbour steps from every ghost block cannot (a halo-3 tile reaches one block across a side
// and, across a coarser one, that block's tangential neighbours), and is not traced.
void amr_blocks_reading_ghosts(int nowned, int ntotal, const int32_t *kind, const int32_t *nbr2, const int32_t *half, int set,
                               std::vector<int32_t> &inner, std::vector<int32_t> &halo) {
  inner.clear();
  halo.clear();
  std::vector<unsigned char> near((size_t)ntotal, 0), next;
  for (int b = nowned; b < ntotal; b++) near[(size_t)b] = 1;
  for (int step = 0; step < 3; step++) {
    next = near;
    for (int b = 0; b < ntotal; b++) {
      if (near[(size_t)b]) continue;
      for (int e = 0; e < 8 && !next[(size_t)b]; e++) {
        if (kind[4 * b + (e >> 1)] == CUP2D_AMR_WALL) continue;
        const int nb = nbr2[8 * b + e];
        if (nb >= 0 && nb < ntotal && near[(size_t)nb]) next[(size_t)b] = 1;
      }
    }
    near.swap(next);
  }
  AmrDev T;
  T.kind = kind; T.nbr2 = nbr2; T.half = half; T.level = nullptr; T.faces = nullptr; T.h0 = 0.0;
  for (int b = 0; b < nowned; b++) {
    bool reads = false;
    if (near[(size_t)b]) {
      if (set == 0) {
        for (int s = 0; s < 4 && !reads; s++)
          for (int q = 0; q < BS && !reads; q++) {
            const auto get = [&](int blk, int) { reads = reads || blk >= nowned; return 0.0; };
            (void)amr_ghost(get, kind[4 * b + s], nbr2[(4 * b + s) * 2], nbr2[(4 * b + s) * 2 + 1], half[4 * b + s], s, q, 0.0, 0.0, 1.0);
          }
      } else {
        for (int s = 0; s < 4 && !reads; s++)
          for (int k = 0; k < 3 && !reads; k++)
            for (int q = 0; q < BS && !reads; q++) {
              const auto get = [&](int blk, int) { reads = reads || blk >= nowned; return double2{0.0, 0.0}; };
              (void)amr_ghost3(get, T, b, s, k, q, double2{0.0, 0.0}, double2{0.0, 0.0});
            }
      }
    }
    (reads ? halo : inner).push_back(b);
  }
}

// The same operator straight in the hybrid form cup2d_set_matrix_coo arrives at (ctx.h SellMatrix) for the `nowned` first
// blocks of tables that cover `nowned` + ghost blocks: a block whose four sides are walls or same-level OWNED blocks is
// plain -- no row is built for it --, every other block gets its 64 rows as sliced-ELL entries, each row's entries in
// column order with duplicate columns summed in arrival order (what the triplet route stores).  At regrid time this is
// the difference between 0.14 s and 0.01 s on a 63 k-block grid: 94 % of the blocks are plain.
void amr_assemble_hybrid(int nowned, const int32_t *kind, const int32_t *nbr2, const int32_t *half,
                         std::vector<int32_t> &reg, std::vector<long long> &ptr, HostStage &stage, int32_t **ecol_out,
                         double **eval_out, int *nregular) {
  reg.assign((size_t)4 * nowned, CUP2D_WALL);
  ptr.assign((size_t)nowned + 1, 0);
  *ecol_out = nullptr;
  *eval_out = nullptr;
  // Pieces of 128 blocks handed out by a counter (the blocks with rows to build sit in a band of the Hilbert order: equal
  // contiguous shares leave most threads without work), each into its own entry lists -- a block's 64 rows are independent of
  // every other block's --; then laid end to end, in block order, by the same threads.
  constexpr int PIECE = 128;
  const int npieces = (nowned + PIECE - 1) / PIECE;
  std::vector<std::vector<int32_t>> ccol((size_t)npieces);
  std::vector<std::vector<double>> cval((size_t)npieces);
  std::vector<int> cplain((size_t)npieces, 0);
  std::vector<long long> width((size_t)nowned, 0);
  std::atomic<int> next{0};
  const int nthreads = host_threads();
  const auto work = [&](long long, long long, int) {
    std::vector<std::pair<long long, double>> sorted;
    std::vector<Row> rows(64);
    for (;;) {
      const int t = next.fetch_add(1);
      if (t >= npieces) break;
      const int lo = t * PIECE, hi = std::min(nowned, lo + PIECE);
      std::vector<int32_t> &oc = ccol[t];
      std::vector<double> &ov = cval[t];
      for (int b = lo; b < hi; b++) {
        bool plain = true;
        for (int s = 0; s < 4; s++) {
          const int k = kind[4 * b + s];
          plain = plain && (k == CUP2D_AMR_WALL || (k == CUP2D_AMR_SAME && nbr2[(4 * b + s) * 2] < nowned));
        }
        if (plain) {
          for (int s = 0; s < 4; s++)
            reg[(size_t)4 * b + s] = kind[4 * b + s] == CUP2D_AMR_SAME ? nbr2[(4 * b + s) * 2] : CUP2D_WALL;
          cplain[t]++;
          continue;
        }
        reg[(size_t)4 * b] = SELL_STORED;
        int w = 0;
        for (int l = 0; l < 64; l++) {
          rows[l] = Row();
          build_row(rows[l], b, l & 7, l >> 3, kind, nbr2, half);
          w = rows[l].n > w ? rows[l].n : w;
        }
        width[b] = w;
        const size_t base = oc.size();
        oc.resize(base + (size_t)w * 64);
        ov.resize(base + (size_t)w * 64, 0.0);
        for (int l = 0; l < 64; l++) {
          sorted.clear();
          for (int k = 0; k < rows[l].n; k++) sorted.emplace_back(rows[l].col[k], rows[l].val[k]);
          std::sort(sorted.begin(), sorted.end(), [](const auto &a, const auto &c2) { return a.first < c2.first; });
          for (int k = 0; k < w; k++) {
            const size_t e = base + (size_t)k * 64 + l;
            if (k < (int)sorted.size()) {
              oc[e] = (int32_t)sorted[k].first;
              ov[e] = sorted[k].second;
            } else {  // padding: own row, coefficient 0
              oc[e] = b * 64 + l;
              ov[e] = 0.0;
            }
          }
        }
      }
    }
  };
  parallel_chunks(nthreads, 1, work);
  int nreg = 0;
  for (int b = 0; b < nowned; b++) ptr[b + 1] = ptr[b] + width[b] * 64;
  for (int t = 0; t < npieces; t++) nreg += cplain[t];
  *nregular = nreg;
  const size_t entries = (size_t)ptr[nowned];
  if (entries == 0) return;
  char *base = static_cast<char *>(host_stage_reserve(stage, entries * (sizeof(int32_t) + sizeof(double))));
  if (!base) return;  // (the caller reports it: entries > 0 and no buffer)
  // values first: 8-byte aligned whatever the number of entries
  double *ev = reinterpret_cast<double *>(base);
  int32_t *ec = reinterpret_cast<int32_t *>(base + entries * sizeof(double));
  next = 0;
  parallel_chunks(nthreads, 1, [&](long long, long long, int) {
    for (;;) {
      const int t = next.fetch_add(1);
      if (t >= npieces) break;
      if (ccol[t].empty()) continue;
      const size_t at = (size_t)ptr[(size_t)t * PIECE];
      memcpy(ec + at, ccol[t].data(), ccol[t].size() * sizeof(int32_t));
      memcpy(ev + at, cval[t].data(), cval[t].size() * sizeof(double));
    }
  });
  *ecol_out = ec;
  *eval_out = ev;
}

}  // namespace cup2d

// ======================================================================================================================
// Regridding on the host: topology tables, state validation, prolongation / restriction.
//
// Reference: adapt() (main.cpp:4657-5440) -- tagging 4671-4703, state validation / 2:1 balance 4718-4861, the
// tensorial halo-1 tile a refined block is prolonged from (BlockLab with Stencil{-1,-1,2,2,true}, 4906-4913),
// prolongation 4981-5032, restriction 5149-5166 -- and the tree / Znei / Zchild look-ups it leans on (672-738).
// Here the leaves live in one open-addressing table keyed by (level, i, j); what lies across a side or a corner of a
// block is a closed form of that table (amr_ghost.h for the sides, the same code the kernels run).
// ======================================================================================================================
#include <cmath>
#include <cstdint>
#include <limits>

#include "amr_ghost.h"
#include "block.h"

namespace {

using cup2d::BC;

struct Leaves {
  const int32_t *blocks;
  int n, bpdx, bpdy;
  std::vector<uint64_t> keys;
  std::vector<int32_t> vals;
  uint64_t mask;
  static uint64_t key(int l, int i, int j) { return ((uint64_t)(l + 1) << 56) | ((uint64_t)(uint32_t)i << 28) | (uint32_t)j; }
  static uint64_t mix(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    return k;
  }
  Leaves(int n_, const int32_t *b, int bx, int by) : blocks(b), n(n_), bpdx(bx), bpdy(by) {
    size_t cap = 16;
    while (cap < 2 * (size_t)n) cap <<= 1;
    mask = cap - 1;
    keys.assign(cap, 0);
    vals.assign(cap, -1);
    for (int k = 0; k < n; k++) {
      const uint64_t kk = key(b[3 * k], b[3 * k + 1], b[3 * k + 2]);
      uint64_t h = mix(kk) & mask;
      while (keys[h]) h = (h + 1) & mask;
      keys[h] = kk;
      vals[h] = k;
    }
  }
  int find(int l, int i, int j) const {
    if (l < 0 || i < 0 || j < 0) return -1;
    const uint64_t kk = key(l, i, j);
    for (uint64_t h = mix(kk) & mask; keys[h]; h = (h + 1) & mask)
      if (keys[h] == kk) return vals[h];
    return -1;
  }
  bool inside(int l, int i, int j) const { return i >= 0 && j >= 0 && i < (bpdx << l) && j < (bpdy << l); }
  int level(int b) const { return blocks[3 * b]; }
  int bi(int b) const { return blocks[3 * b + 1]; }
  int bj(int b) const { return blocks[3 * b + 2]; }
};

struct Side {
  int kind, n0, n1, half;
};
// what lies across side s (W, E, S, N) of block b; kind = -1 when the grid is not 2:1 balanced there
Side side_of(const Leaves &L, int b, int s) {
  static const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
  const int l = L.level(b), i = L.bi(b), j = L.bj(b), ni = i + di[s], nj = j + dj[s];
  Side r = {CUP2D_AMR_WALL, -1, -1, 0};
  if (!L.inside(l, ni, nj)) return r;
  int k = L.find(l, ni, nj);
  if (k >= 0) {
    r.kind = CUP2D_AMR_SAME;
    r.n0 = k;
    return r;
  }
  k = l > 0 ? L.find(l - 1, ni >> 1, nj >> 1) : -1;
  if (k >= 0) {
    r.kind = CUP2D_AMR_COARSER;
    r.n0 = k;
    r.half = s < 2 ? (j & 1) : (i & 1);
    return r;
  }
  // the two children of (l, ni, nj) that touch this side, ordered along the face
  for (int a = 0; a < 2; a++) {
    const int fi = 2 * ni + (s == 0 ? 1 : s == 1 ? 0 : a), fj = 2 * nj + (s < 2 ? a : s == 2 ? 1 : 0);
    (a ? r.n1 : r.n0) = L.find(l + 1, fi, fj);
  }
  r.kind = (r.n0 >= 0 && r.n1 >= 0) ? CUP2D_AMR_FINER : -1;
  return r;
}

bool bad_grid_args(int nblocks, const int32_t *blocks, int bpdx, int bpdy, const char *who) {
  if (nblocks <= 0 || !blocks || bpdx <= 0 || bpdy <= 0) {
    cup2d::set_error("%s: bad argument", who);
    return true;
  }
  for (int b = 0; b < nblocks; b++) {
    const int l = blocks[3 * b];
    if (l < 0 || l > 20 || blocks[3 * b + 1] < 0 || blocks[3 * b + 2] < 0 || blocks[3 * b + 1] >= (bpdx << l) ||
        blocks[3 * b + 2] >= (bpdy << l)) {
      cup2d::set_error("%s: block %d = (%d, %d, %d) outside the %d x %d base grid", who, b, l, blocks[3 * b + 1],
                       blocks[3 * b + 2], bpdx, bpdy);
      return true;
    }
  }
  return false;
}

enum { LEAVE = 0, REFINE = 1, COMPRESS = 2 };

// the tensorial halo-1 tile (10 x 10 x dim, row-major, components interleaved) of block b
// slot (may be NULL): block k's data lie at f + slot[k] * 64 * dim -- a rank that holds only the blocks it needs keeps them
// in a compact array (cup2d_amr_regrid_local); without it block k is at index k
void halo1_tile(const Leaves &L, const double *f, int dim, bool vector, int b, double *T, const int32_t *slot = nullptr) {
  constexpr int BS = CUP2D_BS, W = BS + 2;
  const auto SL = [&](int k) -> size_t { return (size_t)(slot ? slot[k] : k); };
  const double nan = std::numeric_limits<double>::quiet_NaN();
  const int l = L.level(b), i0 = L.bi(b), j0 = L.bj(b);
  const auto at = [&](int ix, int iy, int d) -> double & { return T[((iy + 1) * W + ix + 1) * dim + d]; };
  for (int c = 0; c < BC; c++)
    for (int d = 0; d < dim; d++) at(c & 7, c >> 3, d) = f[(SL(b) * BC + c) * dim + d];
  // sides: the closed forms the kernels use
  for (int s = 0; s < 4; s++) {
    const Side S = side_of(L, b, s);
    for (int d = 0; d < dim; d++) {
      const auto get = [&](int blk, int cell) { return f[(SL(blk) * BC + cell) * dim + d]; };
      const double sign = (vector && d == (s < 2 ? 0 : 1)) ? -1.0 : 1.0;
      for (int q = 0; q < BS; q++) {
        const int gx = s == 0 ? -1 : s == 1 ? BS : q, gy = s == 2 ? -1 : s == 3 ? BS : q;
        const int ex = s == 0 ? 0 : s == 1 ? BS - 1 : q, ey = s == 2 ? 0 : s == 3 ? BS - 1 : q;
        const int fx = s == 0 ? 1 : s == 1 ? BS - 2 : q, fy = s == 2 ? 1 : s == 3 ? BS - 2 : q;
        at(gx, gy, d) = S.kind < 0 ? nan : cup2d::amr_ghost(get, S.kind, S.n0, S.n1, S.half, s, q, at(ex, ey, d), at(fx, fy, d), sign);
      }
    }
  }
  // comp-0 value of cell (GX, GY) of level l - 1 (global cell coordinates) as the coarse copy of the tile holds it:
  // a leaf of that level, or the 2 x 2 mean of a leaf of level l (UseCoarseStencil0 / FillCoarseVersion 2934-2996)
  const auto coarse0 = [&](int GX, int GY) -> double {
    int k = L.find(l - 1, GX >> 3, GY >> 3);
    if (k >= 0) return f[(SL(k) * BC + (GY & 7) * BS + (GX & 7)) * dim];
    k = L.find(l, GX >> 2, GY >> 2);
    if (k < 0) return nan;
    const int x = 2 * (GX & 3), y = 2 * (GY & 3);
    const auto q = [&](int yy, int xx) { return f[(SL(k) * BC + yy * BS + xx) * dim]; };
    return (q(y, x) + q(y + 1, x) + q(y, x + 1) + q(y + 1, x + 1)) / 4;
  };
  const int NX = L.bpdx << l, NY = L.bpdy << l;
  for (int cy = -1; cy <= 1; cy += 2)
    for (int cx = -1; cx <= 1; cx += 2) {
      const int gx = cx < 0 ? -1 : BS, gy = cy < 0 ? -1 : BS, ex = cx < 0 ? 0 : BS - 1, ey = cy < 0 ? 0 : BS - 1;
      const bool xwall = cx < 0 ? i0 == 0 : i0 == NX - 1, ywall = cy < 0 ? j0 == 0 : j0 == NY - 1;
      if (xwall || ywall) {
        // walls (3131-3255): x faces over the whole ghost column first, then y faces over the whole ghost row
        for (int d = 0; d < dim; d++) {
          double v = ywall ? at(gx, ey, d) : at(ex, gy, d);  // ywall: (gx, ey) is the W/E ghost, or the x-wall ghost
          if (vector && ((ywall && d == 1) || (!ywall && d == 0))) v = -v;
          at(gx, gy, d) = v;
        }
        continue;
      }
      const int ni = i0 + cx, nj = j0 + cy;
      int k = L.find(l, ni, nj);
      if (k >= 0) {
        const int cell = (cy < 0 ? BS - 1 : 0) * BS + (cx < 0 ? BS - 1 : 0);
        for (int d = 0; d < dim; d++) at(gx, gy, d) = f[(SL(k) * BC + cell) * dim + d];
      } else if (l > 0 && L.find(l - 1, ni >> 1, nj >> 1) >= 0) {
        // second-order Taylor expansion about the coarse cell under the ghost (TestInterp 2219-2230); the reference
        // hands it component 0 for every component (2753-2763) -- kept
        const int XX = 4 * i0 + (cx < 0 ? -1 : 4), YY = 4 * j0 + (cy < 0 ? -1 : 4);
        const double dx = cx < 0 ? 0.25 : -0.25, dy = cy < 0 ? 0.25 : -0.25;
        double C[3][3];
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 3; c++) C[a][c] = coarse0(XX - 1 + a, YY - 1 + c);
        const double dudx = 0.5 * (C[2][1] - C[0][1]);
        const double dudy = 0.5 * (C[1][2] - C[1][0]);
        const double dudxdy = 0.25 * ((C[0][0] + C[2][2]) - (C[2][0] + C[0][2]));
        const double dudx2 = (C[0][1] + C[2][1]) - 2.0 * C[1][1];
        const double dudy2 = (C[1][0] + C[1][2]) - 2.0 * C[1][1];
        const double v = (C[1][1] + (dx * dudx + dy * dudy)) + (((0.5 * dx * dx) * dudx2 + (0.5 * dy * dy) * dudy2) + (dx * dy) * dudxdy);
        for (int d = 0; d < dim; d++) at(gx, gy, d) = v;
      } else {
        k = L.find(l + 1, 2 * i0 + (cx > 0 ? 2 : -1), 2 * j0 + (cy > 0 ? 2 : -1));
        const int x = cx < 0 ? BS - 2 : 0, y = cy < 0 ? BS - 2 : 0;
        for (int d = 0; d < dim; d++) {
          const auto q = [&](int yy, int xx) { return f[(SL(k < 0 ? b : k) * BC + yy * BS + xx) * dim + d]; };
          at(gx, gy, d) = k < 0 ? nan : (q(y, x) + q(y + 1, x) + q(y, x + 1) + q(y + 1, x + 1)) / 4;
        }
      }
    }
}

// the four children (kid[J][I] -> out[(2 * J + I)][64 * dim]) of a block from its tile: second-order Taylor expansion
// about the parent cell (main.cpp:4981-5032), operand order kept
void prolong(const double *T, int dim, double *out) {
  constexpr int BS = CUP2D_BS, W = BS + 2;
  for (int J = 0; J < 2; J++)
    for (int I = 0; I < 2; I++) {
      double *kid = out + (size_t)(2 * J + I) * BC * dim;
      for (int j = 0; j < BS; j += 2)
        for (int i = 0; i < BS; i += 2) {
          const int ic = i / 2 + 4 * I + 1, jc = j / 2 + 4 * J + 1;
          for (int d = 0; d < dim; d++) {
            const auto u = [&](int dj, int di) { return T[((jc + dj) * W + ic + di) * dim + d]; };
            const double l00 = u(0, 0), l0p = u(1, 0), l0m = u(-1, 0), lm0 = u(0, -1), lmm = u(-1, -1), lmp = u(1, -1);
            const double lp0 = u(0, 1), lpm = u(-1, 1), lpp = u(1, 1);
            const double x = 0.5 * (lp0 - lm0), y = 0.5 * (l0p - l0m);
            const double x2 = (lp0 + lm0) - 2.0 * l00, y2 = (l0p + l0m) - 2.0 * l00;
            const double xy = 0.25 * ((lpp + lmm) - (lpm + lmp));
            const double c2 = 0.03125 * x2 + 0.03125 * y2;
            kid[(j * BS + i) * dim + d] = (l00 + (-0.25 * x - 0.25 * y)) + (c2 + 0.0625 * xy);
            kid[(j * BS + i + 1) * dim + d] = (l00 + (+0.25 * x - 0.25 * y)) + (c2 - 0.0625 * xy);
            kid[((j + 1) * BS + i) * dim + d] = (l00 + (-0.25 * x + 0.25 * y)) + (c2 - 0.0625 * xy);
            kid[((j + 1) * BS + i + 1) * dim + d] = (l00 + (+0.25 * x + 0.25 * y)) + (c2 + 0.0625 * xy);
          }
        }
    }
}

// distance along the Hilbert curve of a 2^bits square (the order of the reference's blocks, main.cpp:347-360, 1550-1562)
uint64_t hilbert(int bits, uint64_t x, uint64_t y) {
  const uint64_t n = 1ull << bits;
  uint64_t d = 0;
  for (uint64_t s = n >> 1; s > 0; s >>= 1) {
    const uint64_t rx = (x & s) ? 1 : 0, ry = (y & s) ? 1 : 0;
    d += s * s * ((3 * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) {
        x = n - 1 - x;
        y = n - 1 - y;
      }
      std::swap(x, y);
    }
  }
  return d;
}

}  // namespace

extern "C" int cup2d_amr_tables(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int32_t *kind, int32_t *nbr2,
                                int32_t *half) {
  if (bad_grid_args(nblocks, blocks, bpdx, bpdy, "amr_tables")) return CUP2D_ERR_ARG;
  if (!kind || !nbr2 || !half) {
    cup2d::set_error("amr_tables: bad argument");
    return CUP2D_ERR_ARG;
  }
  const Leaves L(nblocks, blocks, bpdx, bpdy);
  for (int b = 0; b < nblocks; b++)
    for (int s = 0; s < 4; s++) {
      const Side S = side_of(L, b, s);
      if (S.kind < 0) {
        cup2d::set_error("amr_tables: block (%d, %d, %d) side %d: neither a leaf, a coarser leaf nor two finer leaves across "
                         "(grid not 2:1 balanced?)", L.level(b), L.bi(b), L.bj(b), s);
        return CUP2D_ERR_ARG;
      }
      kind[4 * b + s] = S.kind;
      nbr2[(4 * b + s) * 2] = S.n0;
      nbr2[(4 * b + s) * 2 + 1] = S.n1;
      half[4 * b + s] = S.half;
    }
  return CUP2D_OK;
}

extern "C" int cup2d_amr_validate_states(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max,
                                         int32_t *st) {
  if (bad_grid_args(nblocks, blocks, bpdx, bpdy, "amr_validate_states")) return CUP2D_ERR_ARG;
  if (!st || level_max < 1) {
    cup2d::set_error("amr_validate_states: bad argument");
    return CUP2D_ERR_ARG;
  }
  const Leaves L(nblocks, blocks, bpdx, bpdy);
  bool any = false;
  for (int k = 0; k < nblocks; k++) {
    if ((st[k] == REFINE && L.level(k) == level_max - 1) || (st[k] == COMPRESS && L.level(k) == 0)) st[k] = LEAVE;
    any = any || st[k] != LEAVE;
  }
  if (!any) return CUP2D_OK;
  // blocks of every level (the passes below go level by level, finest first)
  std::vector<std::vector<int>> of_level(level_max);
  for (int k = 0; k < nblocks; k++)
    if (L.level(k) < level_max) of_level[L.level(k)].push_back(k);
  for (int m = level_max - 1; m >= 0; m--) {
    // next to finer blocks: may not compress; refines if one of those is refining (main.cpp:4734-4803)
    // (every block of a pass writes its OWN state only and reads states of the finer level, final since the previous pass:
    // the blocks of a level are independent -- chunks on the host threads)
    if (m < level_max - 1)
      parallel_chunks((long long)of_level[m].size(), 2048, [&](long long lo_, long long hi_, int) {
      for (long long q_ = lo_; q_ < hi_; q_++) {
        const int k = of_level[m][(size_t)q_];
        if (st[k] == REFINE) continue;
        const int i = L.bi(k), j = L.bj(k);
        bool done = false;
        for (int x = -1; x <= 1 && !done; x++)
          for (int y = -1; y <= 1 && !done; y++) {
            if ((x == 0 && y == 0) || !L.inside(m, i + x, j + y)) continue;
            if (L.find(m, i + x, j + y) >= 0 || (m > 0 && L.find(m - 1, (i + x) >> 1, (j + y) >> 1) >= 0)) continue;
            if (st[k] == COMPRESS) st[k] = LEAVE;
            const int bstep = (x != 0 && y != 0) ? 3 : 1;
            for (int B = 0; B < 2; B += bstep) {
              const int aux = x != 0 ? B % 2 : B / 2;
              const int fi = 2 * i + (x > 0 ? x : 0) + x + (B % 2) * (x == 0 ? 1 : 0);
              const int fj = 2 * j + (y > 0 ? y : 0) + y + aux * (y == 0 ? 1 : 0);
              const int fk = L.find(m + 1, fi, fj);
              if (fk >= 0 && st[fk] == REFINE) {
                st[k] = REFINE;
                done = true;
                break;
              }
            }
          }
      }
      });
    if (m == 0) break;
    // a compressing block next to a same-level refining block stays (main.cpp:4804-4830): Refine states of the level are
    // final here; the verdicts are collected in parallel (reads only) and applied afterwards
    std::vector<unsigned char> keep(of_level[m].size(), 0);
    parallel_chunks((long long)of_level[m].size(), 2048, [&](long long lo_, long long hi_, int) {
    for (long long q_ = lo_; q_ < hi_; q_++) {
      const int k = of_level[m][(size_t)q_];
      if (st[k] != COMPRESS) continue;
      const int i = L.bi(k), j = L.bj(k);
      for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
          if ((x == 0 && y == 0) || !L.inside(m, i + x, j + y)) continue;
          const int nk = L.find(m, i + x, j + y);
          if (nk >= 0 && st[nk] == REFINE) keep[(size_t)q_] = 1;
        }
    }
    });
    for (size_t q_ = 0; q_ < keep.size(); q_++)
      if (keep[q_]) st[of_level[m][q_]] = LEAVE;
  }
  // four siblings compress together or not at all (main.cpp:4831-4861): every compressing block looks at its group (reads
  // only, in parallel), then the verdicts are applied
  std::vector<unsigned char> stays((size_t)nblocks, 0);
  parallel_chunks(nblocks, 4096, [&](long long lo_, long long hi_, int) {
    for (int k = (int)lo_; k < (int)hi_; k++) {
      if (st[k] != COMPRESS) continue;
      const int l = L.level(k), i = L.bi(k), j = L.bj(k);
      bool all = true;
      for (int a = 0; a < 2; a++)
        for (int c = 0; c < 2; c++) {
          const int s = L.find(l, 2 * (i >> 1) + a, 2 * (j >> 1) + c);
          all = all && s >= 0 && st[s] == COMPRESS;
        }
      stays[k] = all ? 0 : 1;
    }
  });
  for (int k = 0; k < nblocks; k++)
    if (stays[k]) st[k] = LEAVE;
  return CUP2D_OK;
}

// the new leaf list of a regrid: blocks in the order they are produced from the old list, and where each lands in the
// Hilbert-sorted output
struct New {
  int32_t l, i, j;
  int src, part;  // part -1: copy of src; 0..3: child 2J+I of src; 4: parent of the sibling group whose (0,0) member is src
  uint64_t key;
};
struct RegridPlan {
  std::vector<New> nb;
  std::vector<int> where;  // position of produced block k in the output
  long long n_new = 0;
};
static long long make_plan(const Leaves &L, int nblocks, int bpdx, int bpdy, int level_max, const int32_t *st, bool full, RegridPlan &RP) {
  // a Compress state counts only where the four siblings agree (validated states guarantee it)
  long long n_new = 0;
  for (int k = 0; k < nblocks; k++) {
    if (st[k] == REFINE) n_new += 4;
    else if (st[k] == COMPRESS) {
      const int l = L.level(k), i = L.bi(k), j = L.bj(k);
      for (int a = 0; a < 4; a++) {
        const int s = L.find(l, 2 * (i >> 1) + (a & 1), 2 * (j >> 1) + (a >> 1));
        if (l == 0 || s < 0 || st[s] != COMPRESS) {
          cup2d::set_error("amr_regrid: block %d compresses without its siblings (states not validated)", k);
          return CUP2D_ERR_ARG;
        }
      }
      if ((i & 1) == 0 && (j & 1) == 0) n_new += 1;
    } else n_new += 1;
  }
  RP.n_new = n_new;
  if (!full) return n_new;
  // new blocks in the order they are produced, then sorted along the Hilbert curve of the finest level
  std::vector<New> &nb = RP.nb;
  nb.reserve(n_new);
  int lmax_new = 0;
  for (int k = 0; k < nblocks; k++) {
    const int l = L.level(k), i = L.bi(k), j = L.bj(k);
    if (st[k] == REFINE) {
      for (int J = 0; J < 2; J++)
        for (int I = 0; I < 2; I++) nb.push_back({l + 1, 2 * i + I, 2 * j + J, k, 2 * J + I, 0});
    } else if (st[k] == COMPRESS) {
      if ((i & 1) == 0 && (j & 1) == 0) nb.push_back({l - 1, i >> 1, j >> 1, k, 4, 0});
    } else {
      nb.push_back({l, i, j, k, -1, 0});
    }
  }
  for (const New &b : nb) lmax_new = std::max(lmax_new, (int)b.l);
  const int Lf = std::max(level_max - 1, lmax_new);
  int base_bits = 0;
  while ((1 << base_bits) < std::max(bpdx, bpdy)) base_bits++;
  const int bits = std::max(1, Lf + base_bits);
  for (New &b : nb) b.key = hilbert(bits, (uint64_t)b.i << (Lf - b.l), (uint64_t)b.j << (Lf - b.l));
  std::vector<int> order(nb.size());
  for (size_t k = 0; k < nb.size(); k++) order[k] = (int)k;
  const auto before = [&](int a, int c) { return nb[a].key != nb[c].key ? nb[a].key < nb[c].key : nb[a].l < nb[c].l; };
  // A leaf list that came out of a regrid is in Hilbert order already, and a regrid substitutes in place (the children where
  // the parent was, the parent where its (even, even) child was): only the four children of a refined block need ordering
  // among themselves.  Any other input order (a caller's first grid) takes the full sort.
  for (size_t k = 0; k + 3 < nb.size(); k++)
    if (nb[k].part == 0) std::sort(order.begin() + k, order.begin() + k + 4, before), k += 3;
  bool sorted = true;
  for (size_t p = 1; p < order.size() && sorted; p++) sorted = !before(order[p], order[p - 1]);
  if (!sorted) {
    for (size_t k = 0; k < nb.size(); k++) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), before);
  }
  RP.where.resize(nb.size());
  for (size_t p = 0; p < order.size(); p++) RP.where[order[p]] = (int)p;
  return n_new;
}

// src_of_new [cap] (optional): per new block the old block it is an unchanged copy of, or -1 (prolonged / restricted).
// needed_old [nblocks] (optional): 1 for every old block a prolonged or restricted block is computed from -- the refined
// parents and every leaf that overlaps the 3 x 3 block neighbourhood of one (the tensorial halo-1 tile: sides, corners,
// the coarse cells TestInterp looks at), the compressing siblings.  changed_only: unchanged copies are NOT written to
// new_fields and of `fields` only the needed blocks are read (a host that keeps the fields on the device moves the
// unchanged blocks there, cup2d_copy_blocks).
// Ranged form (cup2d_amr_regrid_local): only the new blocks at positions [new_lo, new_hi) of the new leaf list count --
// needed_old names what THEY are computed from, only they are written, to new_fields[f] + (position - new_lo) * 64 * dim, and
// `fields` are compact arrays addressed through slot_of_old (halo1_tile).  new_hi < 0: everything, positions as indices.
static long long regrid_impl(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *st,
                             int nfields, const double *const *fields, const int32_t *dims, const int32_t *is_vector,
                             long long cap, int32_t *new_blocks, double *const *new_fields, int32_t *src_of_new,
                             int32_t *needed_old, bool changed_only, long long new_lo = 0, long long new_hi = -1,
                             const int32_t *slot_of_old = nullptr) {
  if (bad_grid_args(nblocks, blocks, bpdx, bpdy, "amr_regrid")) return CUP2D_ERR_ARG;
  if (!st || nfields < 0 || (nfields && (!fields || !dims || !is_vector))) {
    cup2d::set_error("amr_regrid: bad argument");
    return CUP2D_ERR_ARG;
  }
  const Leaves L(nblocks, blocks, bpdx, bpdy);
  RegridPlan RP;
  {
    const long long rc = make_plan(L, nblocks, bpdx, bpdy, level_max, st, new_blocks != nullptr, RP);
    if (rc < 0) return rc;
  }
  const long long n_new = RP.n_new;
  if (!new_blocks) return n_new;
  if (cap < n_new || (nfields && !new_fields)) {
    cup2d::set_error("amr_regrid: capacity %lld < %lld new blocks", cap, n_new);
    return CUP2D_ERR_ARG;
  }
  const std::vector<New> &nb = RP.nb;
  const std::vector<int> &where = RP.where;
  for (size_t k = 0; k < nb.size(); k++) {
    const size_t p = (size_t)where[k];
    const New &b = nb[k];
    new_blocks[3 * p] = b.l;
    new_blocks[3 * p + 1] = b.i;
    new_blocks[3 * p + 2] = b.j;
    if (src_of_new) src_of_new[p] = b.part == -1 ? b.src : -1;
  }
  const bool ranged = new_hi >= 0;
  if (ranged && (new_lo < 0 || new_hi > n_new || new_lo > new_hi)) {
    cup2d::set_error("amr_regrid: range [%lld, %lld) of %lld new blocks", new_lo, new_hi, n_new);
    return CUP2D_ERR_ARG;
  }
  const auto in_range = [&](size_t produced) { return !ranged || (where[produced] >= new_lo && where[produced] < new_hi); };
  // the compact form reads fields through slot_of_old: what it will read is computed here even when the caller did not ask
  // for the list, so that a block the caller failed to fetch (slot -1) is an error and not a wild read
  std::vector<int32_t> needed_own;
  if (!needed_old && slot_of_old && nfields > 0) {
    needed_own.resize((size_t)nblocks);
    needed_old = needed_own.data();
  }
  if (needed_old) {
    std::fill(needed_old, needed_old + nblocks, 0);
    for (size_t p = 0; p < nb.size(); p++) {
      const New &b = nb[p];
      if (b.part == -1 || !in_range(p)) continue;
      const int k = b.src, l = L.level(k), i = L.bi(k), j = L.bj(k);
      if (b.part == 4) {  // the parent of four compressing siblings
        for (int a = 0; a < 4; a++) needed_old[L.find(l, i + (a & 1), j + (a >> 1))] = 1;
        continue;
      }
      // a child of the refined block k: k and every leaf overlapping its 3 x 3 block neighbourhood
      for (int dj = -1; dj <= 1; dj++)
        for (int di = -1; di <= 1; di++) {
          const int x = i + di, y = j + dj;
          int s = L.find(l, x, y);
          if (s >= 0) { needed_old[s] = 1; continue; }
          s = l > 0 ? L.find(l - 1, x >> 1, y >> 1) : -1;  // (negative coordinates: outside the domain, no leaf)
          if (x >= 0 && y >= 0 && s >= 0) { needed_old[s] = 1; continue; }
          for (int a = 0; a < 4; a++) {
            const int c = L.find(l + 1, 2 * x + (a & 1), 2 * y + (a >> 1));
            if (c >= 0) needed_old[c] = 1;
          }
        }
    }
  }
  if (slot_of_old && nfields > 0)
    for (int k = 0; k < nblocks; k++)
      if (needed_old[k] && slot_of_old[k] < 0) {
        cup2d::set_error("amr_regrid_local: old block %d (%d, %d, %d) is read by the new blocks [%lld, %lld) but slot_of_old[%d] = %d", k,
                         L.level(k), L.bi(k), L.bj(k), new_lo, new_hi, k, slot_of_old[k]);
        return CUP2D_ERR_ARG;
      }
  constexpr int BS = CUP2D_BS;
  for (int fi = 0; fi < nfields; fi++) {
    const int dim = dims[fi];
    if (dim < 1 || dim > 2 || !fields[fi] || !new_fields[fi]) {
      cup2d::set_error("amr_regrid: field %d: bad dim or pointer", fi);
      return CUP2D_ERR_ARG;
    }
    const double *f = fields[fi];
    double *g = new_fields[fi];
    const size_t bsz = (size_t)BC * dim;
    const auto SL = [&](int k) -> size_t { return (size_t)(slot_of_old ? slot_of_old[k] : k); };
    const auto out = [&](size_t produced) { return g + (size_t)(where[produced] - (ranged ? new_lo : 0)) * bsz; };
    parallel_chunks((long long)nb.size(), 256, [&](long long p_lo, long long p_hi, int) {
    for (long long p = p_lo; p < p_hi; p++) {
      const New &b = nb[p];
      if (b.part == -1) {
        if (!changed_only && in_range(p)) std::copy(f + SL(b.src) * bsz, f + (SL(b.src) + 1) * bsz, out(p));
      } else if (b.part == 4) {  // mean of the 2 x 2 cells of the four siblings (main.cpp:5149-5166)
        if (!in_range(p)) continue;
        const int l = L.level(b.src), i = L.bi(b.src), j = L.bj(b.src);
        double *o = out(p);
        for (int J = 0; J < 2; J++)
          for (int I = 0; I < 2; I++) {
            const double *kid = f + SL(L.find(l, i + I, j + J)) * bsz;
            for (int y = 0; y < BS / 2; y++)
              for (int x = 0; x < BS / 2; x++)
                for (int d = 0; d < dim; d++) {
                  const auto q = [&](int yy, int xx) { return kid[(yy * BS + xx) * dim + d]; };
                  o[((4 * J + y) * BS + 4 * I + x) * dim + d] =
                      (q(2 * y, 2 * x) + q(2 * y + 1, 2 * x) + q(2 * y, 2 * x + 1) + q(2 * y + 1, 2 * x + 1)) / 4;
                }
          }
      } else if (b.part == 0) {  // the four children are produced consecutively: prolong once
        if (!(in_range(p) || in_range(p + 1) || in_range(p + 2) || in_range(p + 3))) continue;
        double T[(BS + 2) * (BS + 2) * 2], kids[4 * BC * 2];
        halo1_tile(L, f, dim, is_vector[fi] != 0, b.src, T, slot_of_old);
        prolong(T, dim, kids);
        for (int c = 0; c < 4; c++)
          if (in_range(p + c)) std::copy(kids + c * bsz, kids + (c + 1) * bsz, out(p + c));
      }
    }
    });
  }
  return n_new;
}

extern "C" long long cup2d_amr_regrid(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max,
                                      const int32_t *st, int nfields, const double *const *fields, const int32_t *dims,
                                      const int32_t *is_vector, long long cap, int32_t *new_blocks,
                                      double *const *new_fields) {
  return regrid_impl(nblocks, blocks, bpdx, bpdy, level_max, st, nfields, fields, dims, is_vector, cap, new_blocks, new_fields,
                     nullptr, nullptr, false);
}
extern "C" long long cup2d_amr_regrid_plan(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max,
                                           const int32_t *st, long long cap, int32_t *new_blocks, int32_t *src_of_new,
                                           int32_t *needed_old) {
  return regrid_impl(nblocks, blocks, bpdx, bpdy, level_max, st, 0, nullptr, nullptr, nullptr, cap, new_blocks, nullptr, src_of_new,
                     needed_old, true);
}
extern "C" long long cup2d_amr_regrid_local(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max,
                                            const int32_t *st, long long new_lo, long long new_hi, long long cap,
                                            int32_t *new_blocks, int32_t *src_of_new, int32_t *needed_old, int nfields,
                                            const double *const *fields, const int32_t *slot_of_old, const int32_t *dims,
                                            const int32_t *is_vector, double *const *new_fields) {
  if (new_hi < 0) {
    cup2d::set_error("amr_regrid_local: range [%lld, %lld)", new_lo, new_hi);
    return CUP2D_ERR_ARG;
  }
  if (nfields > 0 && !slot_of_old) {
    cup2d::set_error("amr_regrid_local: slot_of_old is required with fields");
    return CUP2D_ERR_ARG;
  }
  return regrid_impl(nblocks, blocks, bpdx, bpdy, level_max, st, nfields, fields, dims, is_vector, cap, new_blocks, new_fields,
                     src_of_new, needed_old, true, new_lo, new_hi, slot_of_old);
}
extern "C" long long cup2d_amr_regrid_changed(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max,
                                              const int32_t *st, int nfields, const double *const *fields,
                                              const int32_t *dims, const int32_t *is_vector, long long cap,
                                              int32_t *new_blocks, double *const *new_fields) {
  return regrid_impl(nblocks, blocks, bpdx, bpdy, level_max, st, nfields, fields, dims, is_vector, cap, new_blocks, new_fields,
                     nullptr, nullptr, true);
}

// ---- regrid with the fields on the device (BASELINE.json configs[4] "restriction/prolongation kernels") ----
//
// Reference: refinement main.cpp:4981-5032 (child = second-order Taylor expansion about the parent cell on a tensorial
// halo-1 lab), compression main.cpp:5149-5166 (parent = mean of 2 x 2), unchanged blocks keep their data.  The host side
// above does the same arithmetic on host arrays; here the old and the new slabs are both resident and no field crosses
// PCIe: the host derives, from the LEAF LISTS alone, one job per source unit --
//   COPY      new block p = old block k
//   RESTRICT  new block p = 2 x 2 means of the four siblings s00, s10, s01, s11
//   PROLONG   new blocks d0..d3 = the children of old block k, from its halo-1 tile: the 32 side cells by amr_ghost (the
//             closed forms of the block operators, from the OLD context's device tables), the 4 corner cells by a
//             10-int descriptor each (wall / same-level cell / 2 x 2 mean of a finer corner block / TestInterp on 3 x 3
//             coarse cells, every coarse cell a leaf of level l - 1 or the 2 x 2 mean of a leaf of level l)
// -- and one launch per field runs them, a wave per job.  Operand order as in halo1_tile / prolong above (the translation
// unit is built with -ffp-contract=off): bit-identical to the host routine, which tests/test_amr.py pins to the reference.
namespace cup2d {

enum { JOB_COPY = 0, JOB_RESTRICT = 1, JOB_PROLONG = 2 };
constexpr int JOB_INTS = 8;      // type | COPY: dst, src | RESTRICT: dst, s00, s10, s01, s11 | PROLONG: src, d0..d3, i0, j0
constexpr int CORNER_INTS = 12;  // kind, flags, nine coarse cells (block << 1 | mode, -1: none), pad
enum { CORNER_WALL = 0, CORNER_SAME = 1, CORNER_COARSE = 2, CORNER_FINE = 3, CORNER_NAN = 4 };

template <int DIM>
__global__ __launch_bounds__(WG) void k_amr_regrid(const double *__restrict__ fo, double *__restrict__ fn,
                                                   const int32_t *__restrict__ jobs, const int32_t *__restrict__ corners,
                                                   int njobs, const int32_t *__restrict__ kind, const int32_t *__restrict__ nbr2,
                                                   const int32_t *__restrict__ half, int is_vector) {
  constexpr int W = BS + 2;
  __shared__ double tiles[WPG][W * W * DIM];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *T = tiles[wave];
  const double nan = __builtin_nan("");
  for (int job = blockIdx.x * WPG + wave; job < njobs; job += gridDim.x * WPG) {
    const int32_t *J = jobs + (size_t)job * JOB_INTS;
    const int type = J[0];
    if (type == JOB_COPY) {
#pragma unroll
      for (int d = 0; d < DIM; d++) fn[((size_t)J[1] * BC) * DIM + lane + 64 * d] = fo[((size_t)J[2] * BC) * DIM + lane + 64 * d];
      continue;
    }
    if (type == JOB_RESTRICT) {  // main.cpp:5149-5166
      const int X = lane & 7, Y = lane >> 3, x = X & 3, y = Y & 3;
      const double *kid = fo + (size_t)J[2 + 2 * (Y >> 2) + (X >> 2)] * BC * DIM;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        const auto q = [&](int yy, int xx) { return kid[(yy * BS + xx) * DIM + d]; };
        fn[((size_t)J[1] * BC + lane) * DIM + d] = (q(2 * y, 2 * x) + q(2 * y + 1, 2 * x) + q(2 * y, 2 * x + 1) + q(2 * y + 1, 2 * x + 1)) / 4;
      }
      continue;
    }
    // ---- PROLONG: the tensorial halo-1 tile of old block b (halo1_tile above) ----
    const int b = J[1], i0 = J[6], j0 = J[7];
    const auto at = [&](int ix, int iy, int d) -> double & { return T[((iy + 1) * W + ix + 1) * DIM + d]; };
#pragma unroll
    for (int d = 0; d < DIM; d++) at(lane & 7, lane >> 3, d) = fo[((size_t)b * BC + lane) * DIM + d];
    wave_lds_sync();
    if (lane < 32) {
      const int s = lane >> 3, q = lane & 7;
      const int gx = s == 0 ? -1 : s == 1 ? BS : q, gy = s == 2 ? -1 : s == 3 ? BS : q;
      const int ex = s == 0 ? 0 : s == 1 ? BS - 1 : q, ey = s == 2 ? 0 : s == 3 ? BS - 1 : q;
      const int fx = s == 0 ? 1 : s == 1 ? BS - 2 : q, fy = s == 2 ? 1 : s == 3 ? BS - 2 : q;
      const int kd = kind[4 * b + s], n0 = nbr2[(4 * b + s) * 2], n1 = nbr2[(4 * b + s) * 2 + 1], hf = half[4 * b + s];
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        const auto get = [&](int blk, int cell) { return fo[((size_t)blk * BC + cell) * DIM + d]; };
        const double sign = (is_vector && d == (s < 2 ? 0 : 1)) ? -1.0 : 1.0;
        at(gx, gy, d) = amr_ghost(get, kd, n0, n1, hf, s, q, at(ex, ey, d), at(fx, fy, d), sign);
      }
    }
    wave_lds_sync();
    if (lane < 4) {  // the four corners
      const int cx = (lane & 1) ? 1 : -1, cy = (lane >> 1) ? 1 : -1;
      const int gx = cx < 0 ? -1 : BS, gy = cy < 0 ? -1 : BS, ex = cx < 0 ? 0 : BS - 1, ey = cy < 0 ? 0 : BS - 1;
      const int32_t *C = corners + ((size_t)job * 4 + lane) * CORNER_INTS;  // prolong jobs come first: job index = corner record
      const int ck = C[0];
      if (ck == CORNER_WALL) {
        const bool ywall = (C[1] & 2) != 0;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          double v = ywall ? at(gx, ey, d) : at(ex, gy, d);
          if (is_vector && ((ywall && d == 1) || (!ywall && d == 0))) v = -v;
          at(gx, gy, d) = v;
        }
      } else if (ck == CORNER_SAME) {
        const int cell = (cy < 0 ? BS - 1 : 0) * BS + (cx < 0 ? BS - 1 : 0);
#pragma unroll
        for (int d = 0; d < DIM; d++) at(gx, gy, d) = fo[((size_t)C[2] * BC + cell) * DIM + d];
      } else if (ck == CORNER_COARSE) {  // TestInterp (main.cpp:2219-2230) on component 0 of the 3 x 3 coarse cells (2753-2763)
        const int XX = 4 * i0 + (cx < 0 ? -1 : 4), YY = 4 * j0 + (cy < 0 ? -1 : 4);
        const double dx = cx < 0 ? 0.25 : -0.25, dy = cy < 0 ? 0.25 : -0.25;
        double Cc[3][3];
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 3; c++) {
            const int e = C[2 + 3 * a + c], GX = XX - 1 + a, GY = YY - 1 + c;
            double v = nan;
            if (e >= 0) {
              const size_t k = (size_t)(e >> 1);
              if ((e & 1) == 0) v = fo[(k * BC + (GY & 7) * BS + (GX & 7)) * DIM];
              else {
                const int x = 2 * (GX & 3), y = 2 * (GY & 3);
                const auto q = [&](int yy, int xx) { return fo[(k * BC + yy * BS + xx) * DIM]; };
                v = (q(y, x) + q(y + 1, x) + q(y, x + 1) + q(y + 1, x + 1)) / 4;
              }
            }
            Cc[a][c] = v;
          }
        const double dudx = 0.5 * (Cc[2][1] - Cc[0][1]);
        const double dudy = 0.5 * (Cc[1][2] - Cc[1][0]);
        const double dudxdy = 0.25 * ((Cc[0][0] + Cc[2][2]) - (Cc[2][0] + Cc[0][2]));
        const double dudx2 = (Cc[0][1] + Cc[2][1]) - 2.0 * Cc[1][1];
        const double dudy2 = (Cc[1][0] + Cc[1][2]) - 2.0 * Cc[1][1];
        const double v = (Cc[1][1] + (dx * dudx + dy * dudy)) + (((0.5 * dx * dx) * dudx2 + (0.5 * dy * dy) * dudy2) + (dx * dy) * dudxdy);
#pragma unroll
        for (int d = 0; d < DIM; d++) at(gx, gy, d) = v;
      } else if (ck == CORNER_FINE) {
        const int x = cx < 0 ? BS - 2 : 0, y = cy < 0 ? BS - 2 : 0;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const auto q = [&](int yy, int xx) { return fo[(((size_t)C[2]) * BC + yy * BS + xx) * DIM + d]; };
          at(gx, gy, d) = (q(y, x) + q(y + 1, x) + q(y, x + 1) + q(y + 1, x + 1)) / 4;
        }
      } else {
#pragma unroll
        for (int d = 0; d < DIM; d++) at(gx, gy, d) = nan;
      }
    }
    wave_lds_sync();
    {  // main.cpp:4981-5032: this lane's parent cell -> 2 x 2 cells of child 2J+I
      const int pi = lane & 7, pj = lane >> 3, I = pi >> 2, Jc = pj >> 2, i = 2 * (pi & 3), j = 2 * (pj & 3);
      double *kid = fn + (size_t)J[2 + 2 * Jc + I] * BC * DIM;
      const int ic = pi + 1, jc = pj + 1;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        const auto u = [&](int dj, int di) { return T[((jc + dj) * W + ic + di) * DIM + d]; };
        const double l00 = u(0, 0), l0p = u(1, 0), l0m = u(-1, 0), lm0 = u(0, -1), lmm = u(-1, -1), lmp = u(1, -1);
        const double lp0 = u(0, 1), lpm = u(-1, 1), lpp = u(1, 1);
        const double x = 0.5 * (lp0 - lm0), y = 0.5 * (l0p - l0m);
        const double x2 = (lp0 + lm0) - 2.0 * l00, y2 = (l0p + l0m) - 2.0 * l00;
        const double xy = 0.25 * ((lpp + lmm) - (lpm + lmp));
        const double c2 = 0.03125 * x2 + 0.03125 * y2;
        kid[(j * BS + i) * DIM + d] = (l00 + (-0.25 * x - 0.25 * y)) + (c2 + 0.0625 * xy);
        kid[(j * BS + i + 1) * DIM + d] = (l00 + (+0.25 * x - 0.25 * y)) + (c2 - 0.0625 * xy);
        kid[((j + 1) * BS + i) * DIM + d] = (l00 + (-0.25 * x + 0.25 * y)) + (c2 - 0.0625 * xy);
        kid[((j + 1) * BS + i + 1) * DIM + d] = (l00 + (+0.25 * x + 0.25 * y)) + (c2 + 0.0625 * xy);
      }
    }
    wave_lds_sync();
  }
}

}  // namespace cup2d

// The job and corner tables of a regrid, from the leaf lists alone (host; no field data).  corner descriptors follow
// halo1_tile's four cases in its order of tests.
static long long regrid_jobs(const Leaves &L, const RegridPlan &RP, std::vector<int32_t> &jobs, std::vector<int32_t> &corners) {
  using namespace cup2d;
  const std::vector<New> &nb = RP.nb;
  long long njobs = 0, nprol = 0;
  for (const New &b : nb) {
    njobs += b.part == -1 || b.part == 4 || b.part == 0;
    nprol += b.part == 0;
  }
  jobs.assign((size_t)njobs * JOB_INTS, 0);
  corners.assign((size_t)nprol * 4 * CORNER_INTS, -1);
  // prolong jobs first: the long ones start early
  long long jp = 0, jo = nprol;
  for (size_t p = 0; p < nb.size(); p++) {
    const New &b = nb[p];
    if (b.part == -1) {
      int32_t *J = &jobs[(size_t)jo++ * JOB_INTS];
      J[0] = JOB_COPY; J[1] = RP.where[p]; J[2] = b.src;
    } else if (b.part == 4) {
      int32_t *J = &jobs[(size_t)jo++ * JOB_INTS];
      const int l = L.level(b.src), i = L.bi(b.src), j = L.bj(b.src);
      J[0] = JOB_RESTRICT; J[1] = RP.where[p];
      for (int a = 0; a < 4; a++) J[2 + a] = L.find(l, i + (a & 1), j + (a >> 1));  // [2 J + I]
    } else if (b.part == 0) {
      int32_t *J = &jobs[(size_t)jp * JOB_INTS];
      const int k = b.src, l = L.level(k), i0 = L.bi(k), j0 = L.bj(k);
      J[0] = JOB_PROLONG; J[1] = k;
      for (int c = 0; c < 4; c++) J[2 + c] = RP.where[p + c];  // the four children are produced consecutively, child 2J+I
      J[6] = i0; J[7] = j0;
      const int NX = L.bpdx << l, NY = L.bpdy << l;
      for (int cn = 0; cn < 4; cn++) {
        const int cx = (cn & 1) ? 1 : -1, cy = (cn >> 1) ? 1 : -1;
        int32_t *C = &corners[((size_t)jp * 4 + cn) * CORNER_INTS];
        const bool xwall = cx < 0 ? i0 == 0 : i0 == NX - 1, ywall = cy < 0 ? j0 == 0 : j0 == NY - 1;
        if (xwall || ywall) { C[0] = CORNER_WALL; C[1] = (xwall ? 1 : 0) | (ywall ? 2 : 0); continue; }
        const int ni = i0 + cx, nj = j0 + cy;
        int kk = L.find(l, ni, nj);
        if (kk >= 0) { C[0] = CORNER_SAME; C[2] = kk; continue; }
        if (l > 0 && L.find(l - 1, ni >> 1, nj >> 1) >= 0) {
          C[0] = CORNER_COARSE;
          const int XX = 4 * i0 + (cx < 0 ? -1 : 4), YY = 4 * j0 + (cy < 0 ? -1 : 4);
          for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) {
              const int GX = XX - 1 + a, GY = YY - 1 + c;
              int e = L.find(l - 1, GX >> 3, GY >> 3);
              if (e >= 0) e = e << 1;
              else {
                e = L.find(l, GX >> 2, GY >> 2);
                e = e >= 0 ? ((e << 1) | 1) : -1;
              }
              C[2 + 3 * a + c] = e;
            }
          continue;
        }
        kk = L.find(l + 1, 2 * i0 + (cx > 0 ? 2 : -1), 2 * j0 + (cy > 0 ? 2 : -1));
        if (kk >= 0) { C[0] = CORNER_FINE; C[2] = kk; }
        else C[0] = CORNER_NAN;
      }
      jp++;  // (its corner record is number jp: the prolong jobs come first, in this order)
    }
  }
  return njobs;
}

// The job and corner tables cup2d_amr_regrid_device hands to k_amr_regrid, for a host that wants to look at them (no GPU, no
// context: tests/test_amr.py replays the kernel on them in numpy).  jobs[njobs][8], corners[nprolong][4][12]; with jobs ==
// NULL only the counts are returned (*nprolong may be NULL).  Returns njobs or a negative error.
extern "C" long long cup2d_amr_regrid_jobs(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *st,
                                           long long cap_jobs, int32_t *jobs, long long cap_prolong, int32_t *corners,
                                           long long *nprolong) {
  if (bad_grid_args(nblocks, blocks, bpdx, bpdy, "amr_regrid_jobs")) return CUP2D_ERR_ARG;
  if (!st) { cup2d::set_error("amr_regrid_jobs: bad argument"); return CUP2D_ERR_ARG; }
  const Leaves L(nblocks, blocks, bpdx, bpdy);
  RegridPlan RP;
  const long long n_new = make_plan(L, nblocks, bpdx, bpdy, level_max, st, true, RP);
  if (n_new < 0) return n_new;
  std::vector<int32_t> J, C;
  const long long njobs = regrid_jobs(L, RP, J, C);
  const long long np = (long long)(C.size() / (4 * cup2d::CORNER_INTS));
  if (nprolong) *nprolong = np;
  if (!jobs) return njobs;
  if (cap_jobs < njobs || cap_prolong < np || (np && !corners)) {
    cup2d::set_error("amr_regrid_jobs: capacity %lld / %lld < %lld jobs / %lld refined blocks", cap_jobs, cap_prolong, njobs, np);
    return CUP2D_ERR_ARG;
  }
  std::copy(J.begin(), J.end(), jobs);
  std::copy(C.begin(), C.end(), corners);
  return njobs;
}

extern "C" int cup2d_amr_regrid_device(cup2d_ctx *dst, cup2d_ctx *src, int nblocks, const int32_t *blocks, int bpdx, int bpdy,
                                       int level_max, const int32_t *st, int nfields, const int32_t *fields) {
  using namespace cup2d;
  if (!dst || !src || dst == src) { set_error("amr_regrid_device: two contexts expected"); return CUP2D_ERR_ARG; }
  if (bad_grid_args(nblocks, blocks, bpdx, bpdy, "amr_regrid_device")) return CUP2D_ERR_ARG;
  if (!st || nfields < 0 || (nfields && !fields)) { set_error("amr_regrid_device: bad argument"); return CUP2D_ERR_ARG; }
  if (src->device != dst->device) { set_error("amr_regrid_device: both contexts must live on the same device"); return CUP2D_ERR_ARG; }
  if (!src->amr.active || src->nblocks != nblocks || src->nghost != 0 || dst->nghost != 0) {
    set_error("amr_regrid_device: the source context must hold the %d blocks of the old leaf list with its tables set (cup2d_set_amr), no ghost blocks", nblocks);
    return CUP2D_ERR_ARG;
  }
  for (int f = 0; f < nfields; f++)
    if (!field_ok(fields[f])) { set_error("amr_regrid_device: field %d", fields[f]); return CUP2D_ERR_ARG; }
  StageClock clk("amr_regrid_device");
  const Leaves L(nblocks, blocks, bpdx, bpdy);
  clk.lap("leaf table");
  RegridPlan RP;
  const long long n_new = make_plan(L, nblocks, bpdx, bpdy, level_max, st, true, RP);
  clk.lap("plan");
  if (n_new < 0) return (int)n_new;
  if (n_new != dst->nblocks) {
    set_error("amr_regrid_device: the destination context holds %d blocks, the new leaf list %lld", dst->nblocks, n_new);
    return CUP2D_ERR_ARG;
  }
  std::vector<int32_t> jobs, corners;
  const long long njobs = regrid_jobs(L, RP, jobs, corners);
  clk.lap("jobs");
  CUP2D_HIP_CHECK(hipSetDevice(dst->device));
  int32_t *d_jobs = nullptr, *d_corners = nullptr;
  CUP2D_HIP_CHECK(dev_malloc(&d_jobs, jobs.size() * sizeof(int32_t) + 16));
  CUP2D_HIP_CHECK(dev_malloc(&d_corners, corners.size() * sizeof(int32_t) + 16));
  int rc = CUP2D_OK;
  do {
    if (hipStreamSynchronize(src->stream) != hipSuccess) { rc = CUP2D_ERR_HIP; break; }  // what the old context has enqueued is done
    if (hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(int32_t), hipMemcpyHostToDevice, dst->stream) != hipSuccess) { rc = CUP2D_ERR_HIP; break; }
    if (!corners.empty() && hipMemcpyAsync(d_corners, corners.data(), corners.size() * sizeof(int32_t), hipMemcpyHostToDevice, dst->stream) != hipSuccess) { rc = CUP2D_ERR_HIP; break; }
    int grid = (int)((njobs + WPG - 1) / WPG);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    for (int f = 0; f < nfields && njobs > 0; f++) {
      const int fld = fields[f], dim = dim_of(fld);
      const AmrTopo &A = src->amr;
      if (dim == 1)
        hipLaunchKernelGGL(k_amr_regrid<1>, dim3(grid), dim3(WG), 0, dst->stream, (const double *)src->d_field[fld], dst->d_field[fld],
                           (const int32_t *)d_jobs, (const int32_t *)d_corners, (int)njobs, (const int32_t *)A.d_kind,
                           (const int32_t *)A.d_nbr2, (const int32_t *)A.d_half, 0);
      else
        hipLaunchKernelGGL(k_amr_regrid<2>, dim3(grid), dim3(WG), 0, dst->stream, (const double *)src->d_field[fld], dst->d_field[fld],
                           (const int32_t *)d_jobs, (const int32_t *)d_corners, (int)njobs, (const int32_t *)A.d_kind,
                           (const int32_t *)A.d_nbr2, (const int32_t *)A.d_half, 1);
      if (hipGetLastError() != hipSuccess) { rc = CUP2D_ERR_HIP; break; }
    }
    if (rc == CUP2D_OK && hipStreamSynchronize(dst->stream) != hipSuccess) rc = CUP2D_ERR_HIP;  // the tables go back to the pool
  } while (0);
  dev_free(d_jobs);
  dev_free(d_corners);
  clk.lap("upload + kernels + sync");
  if (rc != CUP2D_OK) set_error("amr_regrid_device: HIP error: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}
