// advect_walk.h -- WENO5 advect-diffuse on a QUAD of 2x2 blocks per wavefront, the reconstruction walking along
// the grid lines in registers (FAST arithmetic policy; KernelAdvectDiffuse main.cpp:5441-5503, weno5_* 162-208).
//
// Why.  The per-block kernel (advect_tile.h) is bound by FP64 issue: 5 reconstructions of 41 operations per cell
// (4 own centres + the block's rim) although consecutive centres of a grid line share four of their five inputs.
// Here a LANE owns one component on a STRIP of 8 cells of one grid line and visits the strip's centres in order:
//   * first differences D, second differences e, 13/3 e^2 + eps and e/3, -e/2 are formed ONCE per cell and stay in
//     registers for the three centres that use them (a centre costs 31 FP64 instructions instead of 41);
//   * the face value of centre c-1 is the previous iterate: no hand-off through LDS, and a strip of 8 cells needs
//     9 centres (one-sided upwinding) -- the "rim" of the per-block form (one extra reconstruction per cell) is gone;
//   * the 5-point Laplacian of a line IS the second difference e the smoothness indicators need anyway.
// A 16x16 tile x 2 components = 512 strip cells per direction = 64 lanes x 8: the x walk runs with lane =
// (component, row, half row), leaves old + c (afac u dc/dx + dfac c_xx) in an LDS buffer, the y walk runs with lane =
// (component, column, half column), adds its part in place, and the buffer goes to memory with coalesced 16-byte
// stores.  161 FP64 / 195 VALU instructions per cell against 241 / 306 (measured: SQ_INSTS_VALU).
//
// The arithmetic is the WenoFast policy (weno.h) on the same differences, with the common factors moved:
//   plus (c) = s_c + D1/2 + [W1 e1/3 + 2 W2 e2 + W3 (3/2 e2 - e3/2)] / (W1 + 6 W2 + 3 W3)
//   minus(c) = s_c - D2/2 + [W3 e3/3 + 2 W2 e2 + W1 (3/2 e2 - e1/2)] / (W3 + 6 W2 + 3 W1)
// with D_j = s_{c-1+j} - s_{c-2+j}, e_j = D_j - D_{j-1}, b_k = 4 beta_k + 4e-6, W1 = b2^2 b3^2, W2 = b1^2 b3^2,
// W3 = b1^2 b2^2 (the weights 0.1, 0.6, 0.3 times ten).  It differs from WenoStrict by round-off only (same
// tolerance as WenoFast, tests/test_gpu_parity.py); the STRICT policy stays on the per-block kernel.
//
// Everything a lane does is written as host+device functions of (lane, LDS image): tests/walk_emul.cpp runs the
// same code lane by lane on the CPU against the CPU restatement of the reference, so that the GPU is needed for timing,
// not for debugging.
#pragma once
#include <stdint.h>

#include <array>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WALK_HD __host__ __device__ __forceinline__
#else
#define WALK_HD inline
#endif

namespace cup2d {
namespace walk {

constexpr int TS = 16;      // tile edge in cells: 2 x 2 blocks of 8 x 8
constexpr int LW = TS + 6;  // ghosted tile edge (halo 3, cross only)
// row strides in cells (16 bytes): the x walk reads 16 rows at the same column (25 = odd: the 16 lanes of a
// ds_read group land in 16 different 16-byte bank groups), the y walk reads consecutive columns
constexpr int LS = 25;
constexpr int TSTR = 17;
constexpr int QINTS = 12;   // ints per quad of the plan
typedef double V2 __attribute__((vector_size(16)));  // (u, v) of a cell: one 16-byte access, a first-class value in registers
struct Lds {
  V2 lab[LW * LS];   // the ghosted tile, cell (X, Y), X, Y in [-3, 19): lab[(Y + 3) * LS + X + 3]
  V2 T[TS * TSTR];   // hand-over between the walks: old value in, old + x part out
};

// ---- the plan: which blocks form quads (host) ---------------------------------------------------------------
// quad[0..3]  = the blocks at (px, py) = (0,0), (1,0), (0,1), (1,1)   (W -> E is +x, S -> N is +y)
// quad[4..11] = what lies outside: W of row 0, W of row 1, E of row 0, E of row 1, S of column 0, S of column 1,
//               N of column 0, N of column 1: a block id (owned or ghost), or -1 - (own block) at a domain wall.
// Blocks of [first, first + count) that found no partners are `singles` (they take the per-block kernel).
static inline void build_plan(const int32_t *nbr, int first, int count, std::vector<int32_t> &quads,
                              std::vector<int32_t> &singles) {
  quads.clear();
  singles.clear();
  if (count <= 0) return;
  auto in = [&](int b) { return b >= first && b < first + count; };
  auto N = [&](int b, int side) { return nbr[4 * b + side]; };  // 0 W, 1 E, 2 S, 3 N
  // block coordinates from the neighbour table alone: flood fill of every connected piece of the range
  std::vector<int> X(count), Y(count), comp(count, -1), stack;
  std::vector<int> quad_of(count, -1);
  std::vector<std::array<int, 4>> found;
  int ncomp = 0;
  for (int seed = first; seed < first + count; seed++) {
    if (comp[seed - first] >= 0) continue;
    std::vector<int> members;
    comp[seed - first] = ncomp;
    X[seed - first] = Y[seed - first] = 0;
    stack.assign(1, seed);
    while (!stack.empty()) {
      const int b = stack.back();
      stack.pop_back();
      members.push_back(b);
      const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
      for (int s = 0; s < 4; s++) {
        const int nb = N(b, s);
        if (nb < 0 || !in(nb) || comp[nb - first] >= 0) continue;
        comp[nb - first] = ncomp;
        X[nb - first] = X[b - first] + dx[s];
        Y[nb - first] = Y[b - first] + dy[s];
        stack.push_back(nb);
      }
    }
    ncomp++;
    // the parity of the 2 x 2 tiling that covers most blocks of this piece (an even-sized rectangle: all of them)
    std::vector<std::array<int, 4>> best;
    for (int par = 0; par < 4; par++) {
      std::vector<std::array<int, 4>> cand;
      for (int b : members) {  // b as the SW corner of a cell of the tiling
        if (((X[b - first] ^ par) & 1) || ((Y[b - first] ^ (par >> 1)) & 1)) continue;
        const int e = N(b, 1), n = N(b, 3);
        if (e < 0 || n < 0 || !in(e) || !in(n)) continue;
        const int ne = N(e, 3);
        if (ne < 0 || !in(ne)) continue;
        const int q[4] = {b, e, n, ne};
        bool good = e != n && ne != b && N(q[0], 1) == q[1] && N(q[1], 0) == q[0] && N(q[2], 1) == q[3] &&
                    N(q[3], 0) == q[2] && N(q[0], 3) == q[2] && N(q[2], 2) == q[0] && N(q[1], 3) == q[3] && N(q[3], 2) == q[1];
        // the flood fill's coordinates agree (they could not on a wrapped topology)
        good = good && X[e - first] == X[b - first] + 1 && Y[e - first] == Y[b - first] && X[n - first] == X[b - first] &&
               Y[n - first] == Y[b - first] + 1 && X[ne - first] == X[b - first] + 1 && Y[ne - first] == Y[b - first] + 1;
        if (good) cand.push_back({q[0], q[1], q[2], q[3]});
      }
      if (cand.size() > best.size()) best.swap(cand);
    }
    for (auto &q : best) {
      for (int k = 0; k < 4; k++) quad_of[q[k] - first] = (int)found.size();
      found.push_back(q);
    }
  }
  // quads in the order of their first block: the caller's (space-filling-curve) locality carries over
  std::vector<char> done(found.size(), 0);
  for (int b = first; b < first + count; b++) {
    const int qi = quad_of[b - first];
    if (qi < 0) {
      singles.push_back(b);
      continue;
    }
    if (done[qi]) continue;
    done[qi] = 1;
    const auto &q = found[qi];
    const int side_of[8] = {0, 0, 1, 1, 2, 2, 3, 3};
    const int owner[8] = {q[0], q[2], q[1], q[3], q[0], q[1], q[2], q[3]};
    for (int k = 0; k < 4; k++) quads.push_back(q[k]);
    for (int s = 0; s < 8; s++) {
      const int nb = N(owner[s], side_of[s]);
      quads.push_back(nb >= 0 ? nb : -1 - owner[s]);
    }
  }
}

// ---- staging -------------------------------------------------------------------------------------------------
// cache policy of the streams (-DWALK_NT=<mask>; bit 0 the old values of RK stage 2, bit 1 the results; measured: both
// non-temporal -7 % on stage 1, -10 % on stage 2 at 4096^2):
// what is read or written exactly once may pass the L2 so that the tile data the neighbouring quads read again stays
#ifndef WALK_NT
#define WALK_NT 3
#endif
WALK_HD V2 load_v2(const V2 *p, bool nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (nt) return __builtin_nontemporal_load(p);
#endif
  return *p;
}
WALK_HD void store_v2(V2 *p, V2 v, bool nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (nt) {
    __builtin_nontemporal_store(v, p);
    return;
  }
#endif
  *p = v;
}

// the 192 ghost cells of a tile (8 (side, half) slots of 8 x 3 cells) are three per lane; what a lane needs to know
// about its i-th one, packed: cell in the neighbour block | cell in the own block (wall) << 6 | slot << 12 |
// LDS position << 16
WALK_HD int ghost_pack(int lane, int i) {
  const int idx = i * 64 + lane;
  const int slot = idx / 24, t = idx - 24 * slot;
  const int side = slot >> 1, half = slot & 1;
  const bool hi = side & 1;
  const int r = t / 3, k = t - 3 * r;  // W / E: 8 rows x 3 columns
  const int j = t >> 3, x = t & 7;     // S / N: 3 rows x 8 columns
  const bool we = side < 2;
  const int cell_nb = we ? r * 8 + (hi ? k : 5 + k) : (hi ? j : 5 + j) * 8 + x;
  const int cell_own = we ? r * 8 + (hi ? 7 : 0) : (hi ? 7 : 0) * 8 + x;
  const int X = we ? (hi ? 16 + k : k - 3) : 8 * half + x;
  const int Y = we ? 8 * half + r : (hi ? 16 + j : j - 3);
  return cell_nb | cell_own << 6 | slot << 12 | ((Y + 3) * LS + X + 3) << 16;
}

// one tile in flight from memory.  fetch() must not look at a loaded value (the first use is where the compiler
// waits for the load): the wall sign flips happen in stage().
struct Regs {
  V2 own[4], gh[3], old[4];
  int nb[3];
};
// The quad's plan entry arrives as twelve VALUES (wave-uniform: the kernel reads the entry a whole quad ahead, so that no
// load of this function waits for another one).  Scalars, not an array: the selects below, written on an array, are
// turned into an indexed read of that array from scratch memory.
struct Entry {
  int b0, b1, b2, b3;                   // the blocks at (0,0), (1,0), (0,1), (1,1)
  int n0, n1, n2, n3, n4, n5, n6, n7;   // W0 W1 E0 E1 S0 S1 N0 N1
};
template <bool NEED_OLD>
WALK_HD void fetch(Regs &R, const V2 *__restrict__ f, const V2 *__restrict__ vold, int b0, int b1, int b2, int b3, int n0,
                   int n1, int n2, int n3, int n4, int n5, int n6, int n7, int lane, const int (&gp)[3]) {
  const int b[4] = {b0, b1, b2, b3};
  // addresses as (slab base) + (32-bit byte offset per lane): the loads take the base from scalar registers and one
  // 32-bit add per load is all the vector arithmetic spent on them (launch_advect sends slabs of 4 GiB and more to the
  // per-block kernel)
  const char *fb = reinterpret_cast<const char *>(f);
  const unsigned lane16 = (unsigned)lane << 4;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const unsigned off = ((unsigned)b[p] << 10) + lane16;
    R.own[p] = *reinterpret_cast<const V2 *>(fb + off);
    if (NEED_OLD) R.old[p] = load_v2(reinterpret_cast<const V2 *>(reinterpret_cast<const char *>(vold) + off), WALK_NT & 1);
  }
  // ghost cell 64 i + lane lies in slot (64 i + lane) / 24
  R.nb[0] = lane < 24 ? n0 : lane < 48 ? n1 : n2;
  R.nb[1] = lane < 8 ? n2 : lane < 32 ? n3 : lane < 56 ? n4 : n5;
  R.nb[2] = lane < 16 ? n5 : lane < 40 ? n6 : n7;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int nb = R.nb[i];
    const bool wall = nb < 0;
    const unsigned blk = wall ? -1 - nb : nb;
    const unsigned cell = wall ? (gp[i] >> 6) & 63 : gp[i] & 63;  // VectorLab::applyBCface main.cpp:3131-3204: every ghost layer repeats the edge cell
    R.gh[i] = *reinterpret_cast<const V2 *>(fb + ((blk << 10) + (cell << 4)));
  }
}
template <bool NEED_OLD>
WALK_HD void fetch(Regs &R, const V2 *__restrict__ f, const V2 *__restrict__ vold, const Entry &e, int lane,
                   const int (&gp)[3]) {
  fetch<NEED_OLD>(R, f, vold, e.b0, e.b1, e.b2, e.b3, e.n0, e.n1, e.n2, e.n3, e.n4, e.n5, e.n6, e.n7, lane, gp);
}
WALK_HD void stage_lab(const Regs &R, Lds &L, int lane, const int (&gp)[3]) {
  const int ix = lane & 7, iy = lane >> 3;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int px = p & 1, py = p >> 1;
    L.lab[(8 * py + iy + 3) * LS + 8 * px + ix + 3] = R.own[p];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    V2 v = R.gh[i];
    if (R.nb[i] < 0) {  // free-slip wall: the wall-normal component changes sign
      if (((gp[i] >> 12) & 7) < 4) v[0] = -v[0]; else v[1] = -v[1];
    }
    L.lab[gp[i] >> 16] = v;
  }
}
// RK stage 2: the old values go to the hand-over buffer (after the previous quad's results have left it, flush())
WALK_HD void stage_old(const Regs &R, Lds &L, int lane) {
  const int ix = lane & 7, iy = lane >> 3;
#pragma unroll
  for (int p = 0; p < 4; p++) L.T[(8 * (p >> 1) + iy) * TSTR + 8 * (p & 1) + ix] = R.old[p];
}
// The results of a quad leave through the hand-over buffer: the y walk writes them to L.T, and the wave stores them with
// four coalesced 16-byte accesses per lane AFTER it has issued the loads of the quad after the next -- the memory
// counter is in order, so a store issued behind the prefetch would have to retire before the prefetched data may be
// touched (measured: a fifth of a wave's cycles went to that wait).
WALK_HD void flush(const Lds &L, int lane, V2 *__restrict__ out, int b0, int b1, int b2, int b3) {
  const int ix = lane & 7, iy = lane >> 3;
  const int b[4] = {b0, b1, b2, b3};
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const unsigned off = ((unsigned)b[p] << 10) + ((unsigned)lane << 4);
    store_v2(reinterpret_cast<V2 *>(reinterpret_cast<char *>(out) + off), L.T[(8 * (p >> 1) + iy) * TSTR + 8 * (p & 1) + ix], WALK_NT & 2);
  }
}
// which upwind sides anybody in the tile asks for (main.cpp:5493-5496: x derivatives follow u > 0, y derivatives
// v > 0): bit 0 some u > 0, bit 1 some u <= 0, bit 2 some v > 0, bit 3 some v <= 0 -- of this lane's four cells
WALK_HD int lane_signs(const Regs &R) {
  int m = 0;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    m |= R.own[p][0] > 0 ? 1 : 2;
    m |= R.own[p][1] > 0 ? 4 : 8;
  }
  return m;
}

// ---- the walk ------------------------------------------------------------------------------------------------
WALK_HD double rcp_newton(double d) {  // weno.h fast_rcp
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
#else
  double r = 1.0 / d;
#endif
  return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// One strip of 8 cells of a grid line.  s -> the strip's cell 0, component of this lane; a -> the advecting
// component of cell 0; consecutive cells are 2 * STRIDE doubles apart.  DOP / DOM: somebody in the tile upwinds with
// plus / minus (wave-uniform).  tail(cell, d, lap, centre, adv) receives the upwind difference (derivative(),
// main.cpp:202-208), the second difference, the value of the cell and its advecting velocity.
// 1 / (a b) from ONE reciprocal: ra = r b, rb = r a with r = rcp_newton(a b).  The weights' denominators are sums of products of
// squares, >= 1e-21 and -- for velocities below 1e18 in the units of the grid -- far below the overflow of their product.
// v_rcp_f64 issues at 3.25 x the cost of an FMA on gfx950 (tools/fp64_peak.hip, round 6): two reciprocals the plain way cost
// 2 x (3.25 + 2), this way 3.25 + 2 + 3.
WALK_HD void rcp_pair(double a, double b, double &ra, double &rb) {
  const double r = rcp_newton(a * b);
  ra = r * b;
  rb = r * a;
}

#ifndef WALK_RCP_PAIRS
#define WALK_RCP_PAIRS 1
#endif
template <bool DOP, bool DOM, int STRIDE, class Tail>
WALK_HD void walk_strip(const double *__restrict__ s, const double *__restrict__ a, Tail &&tail) {
  constexpr double K = 13.0 / 3.0, E4 = 4e-6, THIRD = 1.0 / 3.0;
  constexpr int C0 = DOP ? -1 : 0, C1 = DOM ? 8 : 7, ST = 2 * STRIDE;
  constexpr bool PAIRS = WALK_RCP_PAIRS != 0;
  // every LDS read of the strip is issued here, ahead of the arithmetic: read where they are used, each one exposes
  // its latency (the walk is one dependent chain per lane and only three waves share a SIMD)
  double v[C1 - C0 + 5], adv[8];
#pragma unroll
  for (int i = 0; i < C1 - C0 + 5; i++) v[i] = s[ST * (C0 - 2 + i)];
#pragma unroll
  for (int i = 0; i < 8; i++) adv[i] = a[ST * i];
  // centre C0: s[C0-2 .. C0+1] are in, s[C0+2] arrives in the loop
  double sB = v[1], sC = v[2], sD = v[3];
  double D0 = sB - v[0], D1 = sC - sB, D2 = sD - sC;
  double e1 = D1 - D0, e2 = D2 - D1;
  double m1 = __builtin_fma(K * e1, e1, E4), m2 = __builtin_fma(K * e2, e2, E4);
  double a1 = e1 * THIRD, a2 = e2 * THIRD, h1 = -0.5 * e1, h2 = -0.5 * e2;
  // The upwind difference is formed from the small parts only (the centre values cancel exactly):
  //   plus(c) - plus(c-1)   = D1 + e1/2 + (QP(c) - QP(c-1))      QP, QM = the weighted corrections num / den
  //   minus(c) - minus(c-1) = D1 - e2/2 + (QM(c) - QM(c-1))
  // so that its rounding error scales with the differences, not with the values (the reference subtracts two rounded
  // face values, main.cpp:202-208: its own error of ~1e-16 |u| is what the FAST tolerance has to cover).
  double QPm1 = 0, QMm1 = 0, dPm1 = 0;
  // one-sided upwinding, reciprocals in pairs: the first centre of a pair waits here with what its tail needs
  double hnum = 0, hden = 1, hlin = 0, hlap = 0, hcen = 0;
  bool held = false;
#pragma unroll
  for (int c = C0; c <= C1; c++) {
    const double sE = v[c + 2 - (C0 - 2)];
    const double D3 = sE - sD, e3 = D3 - D2;
    const double m3 = __builtin_fma(K * e3, e3, E4), a3 = e3 * THIRD, h3 = -0.5 * e3;
    const double t2 = __builtin_fma(3.0, D1, -D0), t4 = D1 + D2, t6 = __builtin_fma(-3.0, D2, D3);
    const double b1 = __builtin_fma(t2, t2, m1), b2 = __builtin_fma(t4, t4, m2), b3 = __builtin_fma(t6, t6, m3);
    const double q1 = b1 * b1, q2 = b2 * b2, q3 = b3 * b3;
    const double W1 = q2 * q3, W3 = q1 * q2, X2 = (q1 + q1) * q3;
    const bool doP = DOP && c <= 7, doM = DOM && c >= 0;
    double numP = 0, denP = 1, numM = 0, denM = 1;
    if (doP) {
      numP = __builtin_fma(W1, a1, __builtin_fma(X2, e2, W3 * __builtin_fma(1.5, e2, h3)));
      denP = __builtin_fma(3.0, X2 + W3, W1);
    }
    if (doM) {
      numM = __builtin_fma(W3, a3, __builtin_fma(X2, e2, W1 * __builtin_fma(1.5, e2, h1)));
      denM = __builtin_fma(3.0, X2 + W1, W3);
    }
    if (DOP && DOM) {  // both sides: the two reciprocals of a centre from one
      double QP = 0, QM = 0;
      if (PAIRS && doP && doM) {
        double rp, rm;
        rcp_pair(denP, denM, rp, rm);
        QP = numP * rp;
        QM = numM * rm;
      } else {
        if (doP) QP = numP * rcp_newton(denP);
        if (doM) QM = numM * rcp_newton(denM);
      }
      const double dP = (QP - QPm1) + __builtin_fma(0.5, e1, D1);   // plus(c) - plus(c-1): cell c
      const double dM = (QM - QMm1) + __builtin_fma(-0.5, e2, D1);  // minus(c) - minus(c-1): cell c-1
      if (c >= 1) tail(c - 1, adv[c < 1 ? 0 : c - 1] > 0 ? dPm1 : dM, e1, sB, adv[c < 1 ? 0 : c - 1]);
      QPm1 = QP; QMm1 = QM; dPm1 = dP;
    } else {
      // one side: the tail of centre c needs Q(c) - Q(c-1) + lin, the second difference and the centre value
      //   U > 0 : cell c,     plus(c) - plus(c-1),    lin = D1 + e1/2, lap = e2, centre = sC     (c >= 0)
      //   else  : cell c - 1, minus(c) - minus(c-1),  lin = D1 - e2/2, lap = e1, centre = sB     (c >= 1)
      const double num = DOP ? numP : numM, den = DOP ? denP : denM;
      const double lin = DOP ? __builtin_fma(0.5, e1, D1) : __builtin_fma(-0.5, e2, D1);
      const double lap = DOP ? e2 : e1, cen = DOP ? sC : sB;
      const int cell = DOP ? c : c - 1;
      if (!PAIRS) {
        const double Q = num * rcp_newton(den);
        if (cell >= 0) tail(cell, (Q - QPm1) + lin, lap, cen, adv[cell < 0 ? 0 : cell]);
        QPm1 = Q;
      } else if (!held && c < C1) {  // first of a pair
        hnum = num; hden = den; hlin = lin; hlap = lap; hcen = cen;
        held = true;
      } else if (held) {             // second of a pair: both reciprocals, both tails
        double r0, r1;
        rcp_pair(hden, den, r0, r1);
        const double Q0 = hnum * r0, Q1 = num * r1;
        if (cell - 1 >= 0) tail(cell - 1, (Q0 - QPm1) + hlin, hlap, hcen, adv[cell - 1 < 0 ? 0 : cell - 1]);
        if (cell >= 0) tail(cell, (Q1 - Q0) + lin, lap, cen, adv[cell < 0 ? 0 : cell]);
        QPm1 = Q1;
        held = false;
      } else {                       // the odd centre at the end
        const double Q = num * rcp_newton(den);
        if (cell >= 0) tail(cell, (Q - QPm1) + lin, lap, cen, adv[cell < 0 ? 0 : cell]);
        QPm1 = Q;
      }
    }
    sB = sC; sC = sD; sD = sE;
    D0 = D1; D1 = D2; D2 = D3;
    e1 = e2; e2 = e3; m1 = m2; m2 = m3; a1 = a2; a2 = a3; h1 = h2; h2 = h3;
  }
}

// MODE 0: out = rhs; MODE 1: out = old + coef rhs (coef is inside afc, dfc).  OLDLAB: old is the tile's own
// centre value (RK stage 1: vold = vel).
// x walk: lane = (component, row, half row); leaves old + afc u dc/dx + dfc c_xx in L.T
template <bool DOP, bool DOM, int MODE, bool OLDLAB>
WALK_HD void xwalk(Lds &L, int lane, double afc, double dfc) {
  const int comp = lane & 1, row = (lane >> 1) & 15, seg = lane >> 5;
  const double *lab = reinterpret_cast<const double *>(&L.lab[(row + 3) * LS + 8 * seg + 3]);
  double *T = reinterpret_cast<double *>(&L.T[row * TSTR + 8 * seg]) + comp;
  double told[8];
  if (MODE == 1 && !OLDLAB) {
#pragma unroll
    for (int i = 0; i < 8; i++) told[i] = T[2 * i];
  }
  walk_strip<DOP, DOM, 1>(lab + comp, lab, [&](int cell, double d, double lap, double centre, double adv) {
    const double old = MODE == 0 ? 0.0 : (OLDLAB ? centre : told[cell]);
    const double aa = afc * adv;
    T[2 * cell] = __builtin_fma(aa, d, MODE == 0 ? dfc * lap : __builtin_fma(dfc, lap, old));
  });
}
// y walk: lane = (component, column, half column); adds afc v dc/dy + dfc c_yy: the result replaces the x part in L.T
// (flush() takes it to memory).  PRE: the x parts are read ahead of the arithmetic (eight more live values per lane)
template <bool DOP, bool DOM, bool PRE>
WALK_HD void ywalk(Lds &L, int lane, double afc, double dfc) {
  const int comp = lane & 1, col = (lane >> 1) & 15, seg = lane >> 5;
  const double *lab = reinterpret_cast<const double *>(&L.lab[(8 * seg + 3) * LS + col + 3]);
  double *T = reinterpret_cast<double *>(&L.T[8 * seg * TSTR + col]) + comp;
  double tx[8];
  if (PRE) {
#pragma unroll
    for (int i = 0; i < 8; i++) tx[i] = T[2 * TSTR * i];
  }
  walk_strip<DOP, DOM, LS>(lab + comp, lab + 1, [&](int cell, double d, double lap, double, double adv) {
    const double aa = afc * adv;
    T[2 * TSTR * cell] = __builtin_fma(aa, d, __builtin_fma(dfc, lap, PRE ? tx[cell] : T[2 * TSTR * cell]));
  });
}

}  // namespace walk
}  // namespace cup2d
