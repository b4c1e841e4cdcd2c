// amr.hip -- the halo-1 block operators on a block-AMR grid (BASELINE.json configs[4]; SURVEY.md rows a5, a9,
// a10, a11, a19 with a20's flux correction): ghost cells across coarse-fine faces and the conservative
// correction of the coarse side.
//
// Reference: BlockLab::load / post_load (main.cpp:2270-2687, 2689-2933) for Stencil{-1,-1,2,2, tensorial =
// false} (use_averages is false for it, main.cpp:2265-2267), LI (2203-2210), the wall conditions 3131-3255, the
// face tails of the functors (6140-6206, 6231-6285) and prepare0 / fillcases / fillcase0 / fillcase1
// (1564-1849).  What lies across a block's side is one of
//   wall     ghost = edge cell (vector: wall-normal component negated)
//   same     the neighbour's edge cell
//   finer    mean of the 2x2 fine cells under the ghost cell -- INCLUDING the reference's unrolled W/E branch
//            (main.cpp:2476-2534), whose first row pairs fine rows y and y+2: parity is with the reference as it is
//   coarser  quadratic interpolation ALONG the face through the coarse column (centred, one-sided at the ends of
//            the block's span, main.cpp:2797-2846), then LI() ACROSS the face with the two fine interior cells
// and a functor records, on every side that is not wall/same, the flux it saw; fillcases then gives the edge
// cell of a COARSE block  + (its own flux) + (sum of the two fine fluxes) per fine pair, W/E faces before S/N.
// Every expression keeps the reference's operand order (the translation unit is built with -ffp-contract=off):
// results are bit-identical to the reference's CPU functors (tests/test_amr.py).
//
// One wave per block as everywhere else; the 32 ghost cells of the cross are computed by lanes 0..31.  The
// irregular sides gather from other blocks through global memory -- this is the first, parity-first version of
// the AMR path; its tuning follows the uniform path's.
#include "advect_tile.h"
#include "advect_walk.h"
#include "amr_ghost.h"
#include "block.h"

namespace cup2d {

// lab index of ghost (side, q), of the edge cell and of the next cell inwards
static __device__ __forceinline__ void amr_slot(int s, int q, int &gi, int &e1, int &e2) {
  if (s == 0) { gi = (q + 1) * LAB1; e1 = gi + 1; e2 = gi + 2; }
  else if (s == 1) { gi = (q + 1) * LAB1 + 9; e1 = gi - 1; e2 = gi - 2; }
  else if (s == 2) { gi = q + 1; e1 = gi + LAB1; e2 = gi + 2 * LAB1; }
  else { gi = 9 * LAB1 + q + 1; e1 = gi - LAB1; e2 = gi - 2 * LAB1; }
}

// OP 0: y -= Lap5(x)      pressure_rhs1, faces = ghost - edge                         (main.cpp:6209-6285)
// OP 1: y  = Lap5(x)      the operator alone (no face arrays)
// OP 2: tmpV = pFac grad x  pressureCorrectionKernel; its face arrays are never consumed by the reference's
//                           fillcases (main.cpp:7174-7179 passes tmp's buffers), so none are written (6021-6043)
template <int OP>
__global__ __launch_bounds__(WG) void k_amr_scalar(const double *__restrict__ x, double *__restrict__ y, AmrDev T,
                                                   int nblocks, double dt, const int32_t *__restrict__ list) {
  __shared__ double labs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *lab = labs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  for (int i = blockIdx.x * WPG + wave; i < nblocks; i += gridDim.x * WPG) {
    const int b = list ? uniform(list[i]) : i;  // (list: the inner or the halo blocks of an N-rank patch, amr_phase_lists)
    lab[c0] = x[(size_t)b * BC + lane];
    wave_lds_sync();
    int s = lane >> 3, q = lane & 7, gi = 0, e1 = 0, e2 = 0, kind = AMR_WALL;
    double g = 0.0;
    if (lane < 32) {
      amr_slot(s, q, gi, e1, e2);
      kind = T.kind[4 * b + s];
      const auto get = [&](int blk, int cell) { return x[(size_t)blk * BC + cell]; };
      g = amr_ghost(get, kind, T.nbr2[(4 * b + s) * 2], T.nbr2[(4 * b + s) * 2 + 1], T.half[4 * b + s], s, q, lab[e1],
                    lab[e2], 1.0);
    }
    wave_lds_sync();
    if (lane < 32) lab[gi] = g;
    wave_lds_sync();
    const double l0 = lab[c0], l1 = lab[c0 - 1], l2 = lab[c0 + 1], l3 = lab[c0 - LAB1], l4 = lab[c0 + LAB1];
    const size_t o = (size_t)b * BC + lane;
    if (OP == 0) y[o] -= l1 + l2 + l3 + l4 - 4 * l0;
    if (OP == 1) y[o] = l1 + l2 + l3 + l4 - 4 * l0;
    if (OP == 2) {
      const double h = T.h0 / (double)(1 << T.level[b]);
      const double pFac = -0.5 * dt * h;
      double2 *tv = (double2 *)y;
      double2 r;
      r.x = pFac * (l2 - l1);
      r.y = pFac * (l4 - l3);
      tv[o] = r;
    }
    if (OP == 0 && lane < 32 && kind >= AMR_COARSE) T.faces[(size_t)(4 * b + s) * BS + q] = g - lab[e1];
    wave_lds_sync();
  }
}

// OP 0: tmp = (1/2h)(u_S - u_N + v_E - v_W)                                  KernelVorticity main.cpp:3343-3366
// OP 1: tmp = facDiv (div vel) - facDiv chi (div udef), faces as main.cpp:6140-6206       pressure_rhs 6105-6206
template <int OP>
__global__ __launch_bounds__(WG) void k_amr_vector(const double2 *__restrict__ vel, const double2 *__restrict__ udef,
                                                   const double *__restrict__ chi, double *__restrict__ out, AmrDev T,
                                                   int nblocks, double dt, const int32_t *__restrict__ list) {
  __shared__ double2 vlabs[WPG][LAB1 * LAB1];
  __shared__ double2 ulabs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *vlab = vlabs[wave], *ulab = ulabs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  for (int i = blockIdx.x * WPG + wave; i < nblocks; i += gridDim.x * WPG) {
    const int b = list ? uniform(list[i]) : i;
    vlab[c0] = vel[(size_t)b * BC + lane];
    if (OP == 1) ulab[c0] = udef[(size_t)b * BC + lane];
    wave_lds_sync();
    int s = lane >> 3, q = lane & 7, gi = 0, e1 = 0, e2 = 0, kind = AMR_WALL;
    double2 gv = {0.0, 0.0}, gu = {0.0, 0.0};
    if (lane < 32) {
      amr_slot(s, q, gi, e1, e2);
      kind = T.kind[4 * b + s];
      const int n0 = T.nbr2[(4 * b + s) * 2], n1 = T.nbr2[(4 * b + s) * 2 + 1], half = T.half[4 * b + s];
      // VectorLab::applyBCface (main.cpp:3131-3204): the wall-normal component changes sign
      const double sx = s < 2 ? -1.0 : 1.0, sy = s < 2 ? 1.0 : -1.0;
      const auto vx = [&](int blk, int cell) { return vel[(size_t)blk * BC + cell].x; };
      const auto vy = [&](int blk, int cell) { return vel[(size_t)blk * BC + cell].y; };
      gv.x = amr_ghost(vx, kind, n0, n1, half, s, q, vlab[e1].x, vlab[e2].x, sx);
      gv.y = amr_ghost(vy, kind, n0, n1, half, s, q, vlab[e1].y, vlab[e2].y, sy);
      if (OP == 1) {
        const auto ux = [&](int blk, int cell) { return udef[(size_t)blk * BC + cell].x; };
        const auto uy = [&](int blk, int cell) { return udef[(size_t)blk * BC + cell].y; };
        gu.x = amr_ghost(ux, kind, n0, n1, half, s, q, ulab[e1].x, ulab[e2].x, sx);
        gu.y = amr_ghost(uy, kind, n0, n1, half, s, q, ulab[e1].y, ulab[e2].y, sy);
      }
    }
    wave_lds_sync();
    if (lane < 32) {
      vlab[gi] = gv;
      if (OP == 1) ulab[gi] = gu;
    }
    wave_lds_sync();
    const double h = T.h0 / (double)(1 << T.level[b]);
    const size_t o = (size_t)b * BC + lane;
    if (OP == 0) {
      const double i2h = 0.5 / h;
      out[o] = i2h * (vlab[c0 - LAB1].x - vlab[c0 + LAB1].x + vlab[c0 + 1].y - vlab[c0 - 1].y);
    } else {
      const double facDiv = 0.5 * h / dt;
      const double ch = chi[o];
      out[o] = facDiv * (vlab[c0 + 1].x - vlab[c0 - 1].x + vlab[c0 + LAB1].y - vlab[c0 - LAB1].y) -
               facDiv * ch * (ulab[c0 + 1].x - ulab[c0 - 1].x + ulab[c0 + LAB1].y - ulab[c0 - LAB1].y);
      if (lane < 32 && kind >= AMR_COARSE) {
        // main.cpp:6140-6206: the face-normal component at the ghost and at the edge cell; chi of the edge cell
        const int ecell = s == 0 ? q * BS : s == 1 ? q * BS + 7 : s == 2 ? q : 7 * BS + q;
        const double che = chi[(size_t)b * BC + ecell];
        const double v1 = s < 2 ? gv.x : gv.y, v0 = s < 2 ? vlab[e1].x : vlab[e1].y;
        const double u1 = s < 2 ? gu.x : gu.y, u0 = s < 2 ? ulab[e1].x : ulab[e1].y;
        double f;
        if ((s & 1) == 0) f = facDiv * (v1 + v0) - (facDiv * che) * (u1 + u0);
        else f = -facDiv * (v1 + v0) + (facDiv * che) * (u1 + u0);
        T.faces[(size_t)(4 * b + s) * BS + q] = f;
      }
    }
    wave_lds_sync();
  }
}

// fillcases (main.cpp:1767-1849) for a scalar field: on every side of a block whose neighbours are FINER the edge
// cells get  own flux + (fine flux 2q + fine flux 2q+1)  (fillcase0 then fillcase1), W/E faces before S/N
__global__ __launch_bounds__(WG) void k_amr_fillcases(double *__restrict__ y, AmrDev T, int nblocks) {
  __shared__ double blk[WPG][BC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *B = blk[wave];
  for (int b = blockIdx.x * WPG + wave; b < nblocks; b += gridDim.x * WPG) {
    const int k0 = T.kind[4 * b + 0], k1 = T.kind[4 * b + 1], k2 = T.kind[4 * b + 2], k3 = T.kind[4 * b + 3];
    if (k0 != AMR_FINE && k1 != AMR_FINE && k2 != AMR_FINE && k3 != AMR_FINE) continue;  // wave-uniform
    B[lane] = y[(size_t)b * BC + lane];
    wave_lds_sync();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      if (lane < 16) {
        const int s = 2 * pass + (lane >> 3), q = lane & 7;
        if (T.kind[4 * b + s] == AMR_FINE) {
          const int fb = T.nbr2[(4 * b + s) * 2 + (q >> 2)], qq = q & 3;
          const double *ff = T.faces + (size_t)(4 * fb + (s ^ 1)) * BS;
          double cf = T.faces[(size_t)(4 * b + s) * BS + q];
          cf += ff[2 * qq] + ff[2 * qq + 1];
          const int cell = s == 0 ? q * BS : s == 1 ? q * BS + 7 : s == 2 ? q : 7 * BS + q;
          B[cell] += cf;
        }
      }
      wave_lds_sync();
    }
    y[(size_t)b * BC + lane] = B[lane];
    wave_lds_sync();
  }
}

// ---- halo 3: the tile of KernelAdvectDiffuse: amr_ghost3 of amr_ghost.h on the field in memory ----
struct AmrField2 {
  const double2 *__restrict__ f;
  __device__ __forceinline__ double2 operator()(int blk, int cell) const { return f[(size_t)blk * BC + cell]; }
};

// tmpV = KernelAdvectDiffuse(vel) on every block of the adapted grid + its face arrays dfac (edge - ghost) on the
// coarse-fine faces (main.cpp:5504-5570); faces2: [nblocks][4][8][2]
template <class W>
__global__ __launch_bounds__(WG) void k_amr_advect(const double2 *__restrict__ vel, double2 *__restrict__ out, AmrDev T,
                                                   double *__restrict__ faces2, int nblocks, double nu, double dt,
                                                   const int32_t *__restrict__ list) {
  __shared__ AdvectLds lds[WPG];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  AdvectLds &L = lds[wave];
  RimSlot rim;
  rim.init(L, lane);
  const int ix = lane & 7, iy = lane >> 3;
  for (int i = blockIdx.x * WPG + wave; i < nblocks; i += gridDim.x * WPG) {
    const int b = list ? uniform(list[i]) : i;
    L.lab[(iy + 3) * LABS + ix + 3] = vel[(size_t)b * BC + lane];
    wave_lds_sync();
    double2 g[4];
    if (lane < 24) {
      const int k = lane >> 3, q = lane & 7;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        // the block's edge cell and the next one inwards at position q of side s
        const int e1 = s == 0 ? (q + 3) * LABS + 3 : s == 1 ? (q + 3) * LABS + 10 : s == 2 ? 3 * LABS + q + 3 : 10 * LABS + q + 3;
        const int e2 = s == 0 ? e1 + 1 : s == 1 ? e1 - 1 : s == 2 ? e1 + LABS : e1 - LABS;
        g[s] = amr_ghost3(AmrField2{vel}, T, b, s, k, q, L.lab[e1], L.lab[e2]);
      }
    }
    wave_lds_sync();
    if (lane < 24) {
      const int k = lane >> 3, q = lane & 7;
      L.lab[(q + 3) * LABS + 2 - k] = g[0];
      L.lab[(q + 3) * LABS + 11 + k] = g[1];
      L.lab[(2 - k) * LABS + q + 3] = g[2];
      L.lab[(11 + k) * LABS + q + 3] = g[3];
    }
    wave_lds_sync();
    const double h = T.h0 / (double)(1 << T.level[b]);
    const double dfac = nu * dt, afac = -dt * h;  // main.cpp:5446-5447
    // face arrays before the tile's scratch is reused: nearest ghost layer and edge cell, both components
    if (lane < 32) {
      const int s = lane >> 3, q = lane & 7;
      if (T.kind[4 * b + s] >= AMR_COARSE) {
        const int e1 = s == 0 ? (q + 3) * LABS + 3 : s == 1 ? (q + 3) * LABS + 10 : s == 2 ? 3 * LABS + q + 3 : 10 * LABS + q + 3;
        const int gi = s == 0 ? e1 - 1 : s == 1 ? e1 + 1 : s == 2 ? e1 - LABS : e1 + LABS;
        const double2 ed = L.lab[e1], gh = L.lab[gi];
        double *fa = faces2 + ((size_t)(4 * b + s) * BS + q) * 2;
        fa[0] = dfac * (ed.x - gh.x);
        fa[1] = dfac * (ed.y - gh.y);
      }
    }
    const double2 r = advect_cell<W>(L, rim, lane, afac, dfac);
    out[(size_t)b * BC + lane] = r;
    wave_lds_sync();
  }
}

// fillcases with dim = 2 (main.cpp:1767-1849): as k_amr_fillcases, plus the reference's own arithmetic slip -- fillcase1
// runs once per received face (two per coarse face) and its memset (main.cpp:1660, 1668) clears entries 0..8 of the 16
// only, so entries 9..15 (position 4 component 1, positions 5..7) are added a SECOND time.
__global__ __launch_bounds__(WG) void k_amr_fillcases2(double2 *__restrict__ y, AmrDev T, const double *__restrict__ faces2,
                                                       int nblocks) {
  __shared__ double2 blk[WPG][BC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *B = blk[wave];
  for (int b = blockIdx.x * WPG + wave; b < nblocks; b += gridDim.x * WPG) {
    const int k0 = T.kind[4 * b + 0], k1 = T.kind[4 * b + 1], k2 = T.kind[4 * b + 2], k3 = T.kind[4 * b + 3];
    if (k0 != AMR_FINE && k1 != AMR_FINE && k2 != AMR_FINE && k3 != AMR_FINE) continue;  // wave-uniform
    B[lane] = y[(size_t)b * BC + lane];
    wave_lds_sync();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      if (lane < 16) {
        const int s = 2 * pass + (lane >> 3), q = lane & 7;
        if (T.kind[4 * b + s] == AMR_FINE) {
          const int fb = T.nbr2[(4 * b + s) * 2 + (q >> 2)], qq = q & 3;
          const double *ff = faces2 + (size_t)(4 * fb + (s ^ 1)) * BS * 2;
          const double *cf = faces2 + ((size_t)(4 * b + s) * BS + q) * 2;
          double cx = cf[0], cy = cf[1];
          cx += ff[2 * (2 * qq)] + ff[2 * (2 * qq + 1)];
          cy += ff[2 * (2 * qq) + 1] + ff[2 * (2 * qq + 1) + 1];
          const int cell = s == 0 ? q * BS : s == 1 ? q * BS + 7 : s == 2 ? q : 7 * BS + q;
          double2 v = B[cell];
          v.x += cx;
          v.y += cy;
          if (2 * q >= 9) v.x += cx;      // entries 9..15 once more
          if (2 * q + 1 >= 9) v.y += cy;
          B[cell] = v;
        }
      }
      wave_lds_sync();
    }
    y[(size_t)b * BC + lane] = B[lane];
    wave_lds_sync();
  }
}

static AmrDev amr_dev(const cup2d_ctx *c) {
  AmrDev T;
  T.kind = c->amr.d_kind; T.nbr2 = c->amr.d_nbr2; T.half = c->amr.d_half; T.level = c->amr.d_level;
  T.faces = c->amr.d_faces; T.h0 = c->amr.h0;
  return T;
}
// ---- N ranks: ghost copies ------------------------------------------------------------------------------------------
// whole ghost blocks of a field the next kernel reads (the halo plan lists whole blocks; width 8 = the block)
// -- or, with a cell plan for the kernel's family (cup2d_halo_plan_cells; set 0: the halo-1 operators, 1: the WENO tile), the
// cells of them that kernel reads: the strips of the reference's synchroniser (main.cpp:2053-2125, 2582-2684)
static int amr_refresh(cup2d_ctx *c, const double *field, int dim, int set) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->cells[set].active) return exchange_cells(c, set, const_cast<double *>(field), dim);
  return exchange_halo(c, const_cast<double *>(field), dim, BS);
}
// ... in two halves: pack + the transfer under way (on the communicator's own stream where it has one) | arrival + unpack
static int amr_refresh_begin(cup2d_ctx *c, const double *field, int dim, int set) {
  if (c->cells[set].active) return exchange_cells_begin(c, set, const_cast<double *>(field), dim);
  return exchange_begin(c, field, dim, BS);
}
static int amr_refresh_end(cup2d_ctx *c, const double *field, int dim, int set) {
  if (c->cells[set].active) return exchange_cells_end(c, set, const_cast<double *>(field), dim);
  return exchange_end(c, const_cast<double *>(field), dim, BS);
}
// The inner / halo split of computeA (main.cpp:3035-3057) on an adapted grid: which owned blocks read a ghost block is not a
// matter of position (coarse-fine sides read two rings, the halo-3 tile the tangential sides of a coarser neighbour), so the
// lists come from the kernels' own ghost expressions run with a recording accessor (amr_host.hip amr_blocks_reading_ghosts)
// on the host copies of the tables, once per topology and operator family.
static int amr_phase_lists(cup2d_ctx *c, int set, const AmrTopo::Phase **out) {
  AmrTopo::Phase &P = c->amr.phase[set];
  if (!P.built) {
    std::vector<int32_t> inner, halo;
    amr_blocks_reading_ghosts(c->nblocks, c->ntotal, c->amr.h_kind.data(), c->amr.h_nbr2.data(), c->amr.h_half.data(), set, inner, halo);
    CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
    dev_free(P.d_inner); dev_free(P.d_halo);
    P.d_inner = P.d_halo = nullptr;
    P.n_inner = (int)inner.size(); P.n_halo = (int)halo.size();
    if (P.n_inner) {
      CUP2D_HIP_CHECK(dev_malloc(&P.d_inner, sizeof(int32_t) * inner.size()));
      CUP2D_HIP_CHECK(hipMemcpy(P.d_inner, inner.data(), sizeof(int32_t) * inner.size(), hipMemcpyHostToDevice));
    }
    if (P.n_halo) {
      CUP2D_HIP_CHECK(dev_malloc(&P.d_halo, sizeof(int32_t) * halo.size()));
      CUP2D_HIP_CHECK(hipMemcpy(P.d_halo, halo.data(), sizeof(int32_t) * halo.size(), hipMemcpyHostToDevice));
    }
    P.built = true;
  }
  *out = &P;
  return CUP2D_OK;
}
void amr_phase_release(cup2d_ctx *c) {
  for (auto &P : c->amr.phase) {
    dev_free(P.d_inner); dev_free(P.d_halo);
    P = AmrTopo::Phase();
  }
  for (int32_t *q : c->amr.quads.d_quads) dev_free(q);
  dev_free(c->amr.quads.d_left);
  c->amr.quads = AmrTopo::Quads();
}
// The quads of an adapted grid (ctx.h AmrTopo::Quads): four CONSECUTIVE blocks of one level that form a 2 x 2 group by their own
// links (the four children of a parent are consecutive in the reference's Hilbert order; any such group will do) and whose eight
// outer sides are domain walls or same-level blocks -- the tile of the quad kernel is then exactly the uniform grid's: ghost
// layers copied from the neighbour blocks, the wall-normal component flipped at a wall, no corner read (main.cpp:2270-2687 for
// same-level sides, 3131-3204).  One scan of the tables; the plan entry is advect_walk.h's (build_plan).
static int amr_quads(cup2d_ctx *c, const AmrTopo::Quads **out) {
  AmrTopo &A = c->amr;
  AmrTopo::Quads &Q = A.quads;
  *out = &Q;
  if (Q.built) return CUP2D_OK;
  const int nb = c->nblocks;
  const auto K = [&](int b, int s) { return A.h_kind[(size_t)4 * b + s]; };
  const auto NB = [&](int b, int s) { return A.h_nbr2[((size_t)4 * b + s) * 2]; };
  const auto eligible = [&](int b) {
    for (int s = 0; s < 4; s++)
      if (K(b, s) != CUP2D_AMR_WALL && K(b, s) != CUP2D_AMR_SAME) return false;
    return true;
  };
  std::vector<std::vector<int32_t>> per_level(32);
  std::vector<int32_t> left;
  for (int b = 0; b < nb;) {
    bool ok = b + 3 < nb;
    for (int k = 0; ok && k < 4; k++) ok = A.h_level[(size_t)b + k] == A.h_level[(size_t)b] && eligible(b + k);
    int sw = -1, se = -1, nw = -1, ne = -1;
    const auto in = [&](int x) { return x >= b && x < b + 4; };
    const auto same_in = [&](int o, int s) { return K(o, s) == CUP2D_AMR_SAME && in(NB(o, s)); };
    if (ok) {
      int found = 0;
      for (int k = 0; k < 4; k++)
        if (same_in(b + k, 1) && same_in(b + k, 3)) { sw = b + k; found++; }
      ok = found == 1;
    }
    if (ok) {
      se = NB(sw, 1);
      nw = NB(sw, 3);
      ok = se != nw && same_in(se, 3);
      if (ok) ne = NB(se, 3);
      ok = ok && ne != sw && ne != se && ne != nw && same_in(nw, 1) && NB(nw, 1) == ne && same_in(se, 0) && NB(se, 0) == sw &&
           same_in(nw, 2) && NB(nw, 2) == sw && same_in(ne, 0) && NB(ne, 0) == nw && same_in(ne, 2) && NB(ne, 2) == se;
    }
    if (!ok) {
      left.push_back(b);
      b++;
      continue;
    }
    const auto outer = [&](int o, int s) -> int32_t { return K(o, s) == CUP2D_AMR_SAME ? NB(o, s) : -1 - o; };  // a wall: -1 - (own block)
    const int32_t e[walk::QINTS] = {sw, se, nw, ne, outer(sw, 0), outer(nw, 0), outer(se, 1), outer(ne, 1),
                                    outer(sw, 2), outer(se, 2), outer(nw, 3), outer(ne, 3)};
    std::vector<int32_t> &L = per_level[(size_t)A.h_level[(size_t)b]];
    L.insert(L.end(), e, e + walk::QINTS);
    b += 4;
  }
  for (int l = 0; l < 32; l++) {
    if (per_level[(size_t)l].empty()) continue;
    int32_t *d = nullptr;
    CUP2D_HIP_CHECK(dev_malloc(&d, per_level[(size_t)l].size() * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(d, per_level[(size_t)l].data(), per_level[(size_t)l].size() * sizeof(int32_t), hipMemcpyHostToDevice));
    Q.level.push_back(l);
    Q.nq.push_back((int)(per_level[(size_t)l].size() / walk::QINTS));
    Q.d_quads.push_back(d);
  }
  Q.nleft = (int)left.size();
  if (Q.nleft) {
    CUP2D_HIP_CHECK(dev_malloc(&Q.d_left, left.size() * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(Q.d_left, left.data(), left.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  Q.built = true;
  return CUP2D_OK;
}
// One block functor over the phases of `blocks` (ctx.h): launch(list, n) runs it on n blocks (list == nullptr: blocks 0 .. n-1).
// ALL on N ranks: `field` (the last of the fields the functor reads across block sides; the others were refreshed by the
// caller) travels while the inner blocks are swept.
template <class Launch>
static int amr_phased(cup2d_ctx *c, int blocks, const double *field, int dim, int set, Launch launch) {
  const bool ghosts = c->nghost > 0 && c->exchange;
  if (blocks == CUP2D_BLOCKS_ALL && !ghosts) {
    launch((const int32_t *)nullptr, c->nblocks);
    CUP2D_HIP_CHECK(hipGetLastError());
    return CUP2D_OK;
  }
  const AmrTopo::Phase *P = nullptr;
  CUP2D_TRY(amr_phase_lists(c, set, &P));
  if (blocks == CUP2D_BLOCKS_ALL) CUP2D_TRY(amr_refresh_begin(c, field, dim, set));
  if (blocks != CUP2D_BLOCKS_HALO && P->n_inner) launch((const int32_t *)P->d_inner, P->n_inner);
  if (blocks == CUP2D_BLOCKS_ALL) CUP2D_TRY(amr_refresh_end(c, field, dim, set));
  if (blocks != CUP2D_BLOCKS_INNER && P->n_halo) launch((const int32_t *)P->d_halo, P->n_halo);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
// the face arrays (BlockCase::d, REC doubles per block) of the blocks in the send list -> the peers' ghost slots: the
// coarse side of a coarse-fine face adds the fine side's fluxes, and the fine blocks may live on another rank
// (main.cpp:1819-1825 exchanges exactly these)
__global__ __launch_bounds__(WG) void k_amr_faces_pack(const double *__restrict__ faces, double *__restrict__ buf,
                                                       const int32_t *__restrict__ blocks, int n, int rec, int pack) {
  const size_t total = (size_t)n * rec;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
    const int k = (int)(i / rec), q = (int)(i - (size_t)k * rec);
    double *f = const_cast<double *>(faces) + (size_t)blocks[k] * rec + q;
    if (pack) buf[i] = *f;
    else *f = buf[i];
  }
}
static int amr_exchange_faces(cup2d_ctx *c, double *faces, int rec) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  const HaloPlan &P = c->plan;
  if (P.nsend > 0) {
    int g = (int)(((size_t)P.nsend * rec + WG - 1) / WG);
    hipLaunchKernelGGL(k_amr_faces_pack, dim3(g > c->grid ? c->grid : g), dim3(WG), 0, c->stream, faces, c->d_send, P.d_send_block, P.nsend, rec, 1);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  if (c->exchange(c->comm_user, c->d_send, c->d_recv, rec, c->stream) != 0) { set_error("exchange callback failed"); return CUP2D_ERR_COMM; }
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) { set_error("wait callback failed"); return CUP2D_ERR_COMM; }
  if (P.nrecv > 0) {
    int g = (int)(((size_t)P.nrecv * rec + WG - 1) / WG);
    hipLaunchKernelGGL(k_amr_faces_pack, dim3(g > c->grid ? c->grid : g), dim3(WG), 0, c->stream, faces, c->d_recv, P.d_recv_block, P.nrecv, rec, 0);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  return CUP2D_OK;
}

static int amr_grid(const cup2d_ctx *c, int n = -1) {
  int g = ((n < 0 ? c->nblocks : n) + WPG - 1) / WPG;
  return g > c->grid ? c->grid : (g < 1 ? 1 : g);
}

int amr_laplacian(cup2d_ctx *c, const double *x, double *y, int subtract, int blocks) {
  const AmrDev T = amr_dev(c);
  CUP2D_TRY(amr_phased(c, blocks, x, 1, CUP2D_CELLS_HALO1, [&](const int32_t *list, int n) {
    if (subtract) hipLaunchKernelGGL(k_amr_scalar<0>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, x, y, T, n, 0.0, list);
    else hipLaunchKernelGGL(k_amr_scalar<1>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, x, y, T, n, 0.0, list);
  }));
  if (subtract && blocks != CUP2D_BLOCKS_INNER) {  // the flux correction needs the face arrays of every block (and of the peers' blocks)
    CUP2D_TRY(amr_exchange_faces(c, c->amr.d_faces, 4 * BS));
    hipLaunchKernelGGL(k_amr_fillcases, dim3(amr_grid(c)), dim3(WG), 0, c->stream, y, T, c->nblocks);
  }
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
int amr_pressure_correction(cup2d_ctx *c, const double *pres, double *tmpV, double dt, int blocks) {
  const AmrDev T = amr_dev(c);
  return amr_phased(c, blocks, pres, 1, CUP2D_CELLS_HALO1, [&](const int32_t *list, int n) {
    hipLaunchKernelGGL(k_amr_scalar<2>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, pres, tmpV, T, n, dt, list);
  });
}
int amr_vorticity(cup2d_ctx *c, const double *vel, double *out, int blocks) {
  const AmrDev T = amr_dev(c);
  return amr_phased(c, blocks, vel, 2, CUP2D_CELLS_HALO1, [&](const int32_t *list, int n) {
    hipLaunchKernelGGL(k_amr_vector<0>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, (const double2 *)vel, (const double2 *)nullptr,
                       (const double *)nullptr, out, T, n, 0.0, list);
  });
}
int amr_pressure_rhs(cup2d_ctx *c, const double *vel, const double *udef, const double *chi, double *out, double dt, int blocks) {
  const AmrDev T = amr_dev(c);
  // two fields cross block sides: the first is refreshed at once, the second travels while the inner blocks are swept
  if (blocks == CUP2D_BLOCKS_ALL) CUP2D_TRY(amr_refresh(c, vel, 2, CUP2D_CELLS_HALO1));
  CUP2D_TRY(amr_phased(c, blocks, udef, 2, CUP2D_CELLS_HALO1, [&](const int32_t *list, int n) {
    hipLaunchKernelGGL(k_amr_vector<1>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, (const double2 *)vel, (const double2 *)udef, chi,
                       out, T, n, dt, list);
  }));
  if (blocks != CUP2D_BLOCKS_INNER) {
    CUP2D_TRY(amr_exchange_faces(c, c->amr.d_faces, 4 * BS));
    hipLaunchKernelGGL(k_amr_fillcases, dim3(amr_grid(c)), dim3(WG), 0, c->stream, out, T, c->nblocks);
  }
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

}  // namespace cup2d

namespace cup2d {
// ---- glue of the time step on an adapted grid: every block carries its own h ------------------------------------
// MODE 0: y = x0 + t * (coef / (h * h))     RK stages, main.cpp:6618-6626 (coef 0.5), 6634-6642 (coef 1.0)
// MODE 1: y += t * (1.0 / h / h)            projection, main.cpp:7180-7187
template <int MODE>
__global__ __launch_bounds__(WG) void k_amr_axpy(double2 *__restrict__ y, const double2 *__restrict__ x0,
                                                 const double2 *__restrict__ t, AmrDev T, int nblocks, double coef) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = blockIdx.x * WPG + wave; b < nblocks; b += gridDim.x * WPG) {
    const double h = T.h0 / (double)(1 << T.level[b]);
    const double ih2 = MODE == 0 ? coef / (h * h) : 1.0 / h / h;
    const size_t o = (size_t)b * BC + lane;
    const double2 tv = t[o];
    double2 v = MODE == 0 ? x0[o] : y[o];
    if (MODE == 0) {
      v.x = v.x + tv.x * ih2;
      v.y = v.y + tv.y * ih2;
    } else {
      v.x += tv.x * ih2;
      v.y += tv.y * ih2;
    }
    y[o] = v;
  }
}
// partial sums of p * h^2 and of h^2 over the cells (main.cpp:7126-7135): slots 0 and 1 of the partials
__global__ __launch_bounds__(WG) void k_amr_wsum(const double *__restrict__ p, AmrDev T, int nblocks,
                                                 double *__restrict__ partials) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double acc[2] = {0.0, 0.0};
  for (int b = blockIdx.x * WPG + wave; b < nblocks; b += gridDim.x * WPG) {
    const double h = T.h0 / (double)(1 << T.level[b]);
    const double vv = h * h;
    acc[0] += p[(size_t)b * BC + lane] * vv;
    acc[1] += vv;
  }
  workgroup_reduce_store<2, false>(acc, partials, 0);
}
__global__ __launch_bounds__(WG) void k_amr_wsum_final(const double *__restrict__ partials, int n, double *__restrict__ red) {
  __shared__ double sm[2][WG];
  double a = 0.0, w = 0.0;
  for (int i = threadIdx.x; i < n; i += WG) {
    a += partials[i];
    w += partials[PSTRIDE + i];
  }
  sm[0][threadIdx.x] = a;
  sm[1][threadIdx.x] = w;
  __syncthreads();
  for (int s = WG / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // quantities[0], quantities[1] of main.cpp:7136-7141: summed over the ranks before the division
    red[0] = sm[0][0];
    red[1] = sm[1][0];
  }
}
// MODE 0: p += -avg (main.cpp:7143-7148)   MODE 1: p += pold - avg (main.cpp:7166-7172)
template <int MODE>
__global__ __launch_bounds__(WG) void k_amr_shift(double *__restrict__ p, const double *__restrict__ pold,
                                                  const double *__restrict__ red, size_t n) {
  const double avg = red[0] / red[1];  // avg = avg / avg1, main.cpp:7142
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    if (MODE == 0) p[i] += -avg;
    else p[i] += pold[i] - avg;
  }
}

// main.cpp:6607-6642 on an adapted grid: vold = vel; two stages of [KernelAdvectDiffuse + flux correction; V = Vold +
// c tmpV / h^2] (the reference's own un-fused sequence: the face arrays need tmpV)
// One stage (cup2d_advect_diffuse_stage on an adapted grid).  Stage 1 saves vel in VOLD and leaves the mid-point velocity in
// VEL (the reference's own data flow, main.cpp:6607-6626); stage 2 reads both.
int amr_advect_diffuse_stage(cup2d_ctx *c, double nu, double dt, int stage) {
  const AmrDev T = amr_dev(c);
  if (stage == 1) {
    const size_t bytes = (size_t)c->nblocks * BC * 2 * sizeof(double);
    CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[CUP2D_VOLD], c->d_field[CUP2D_VEL], bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  CUP2D_TRY(amr_advect_diffuse_rhs(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMPV], nu, dt));
  hipLaunchKernelGGL(k_amr_axpy<0>, dim3(amr_grid(c)), dim3(WG), 0, c->stream, (double2 *)c->d_field[CUP2D_VEL],
                     (const double2 *)c->d_field[CUP2D_VOLD], (const double2 *)c->d_field[CUP2D_TMPV], T, c->nblocks,
                     stage == 1 ? 0.5 : 1.0);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
int amr_advect_diffuse_rk2(cup2d_ctx *c, double nu, double dt) {
  CUP2D_TRY(amr_advect_diffuse_stage(c, nu, dt, 1));
  return amr_advect_diffuse_stage(c, nu, dt, 2);
}
// main.cpp:7007-7027: tmp = pressure_rhs (+ flux correction); pold = pres; pres = 0; tmp -= Lap(pold) (+ flux correction)
int amr_poisson_rhs(cup2d_ctx *c, double dt) {
  CUP2D_TRY(amr_pressure_rhs(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMPV], c->d_field[CUP2D_CHI], c->d_field[CUP2D_TMP], dt));
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[CUP2D_POLD], c->d_field[CUP2D_PRES], (size_t)c->nblocks * BC * sizeof(double),
                                 hipMemcpyDeviceToDevice, c->stream));  // a copy: the public slab pointers stay put
  CUP2D_HIP_CHECK(hipMemsetAsync(c->d_field[CUP2D_PRES], 0, (size_t)c->ntotal * BC * sizeof(double), c->stream));
  return amr_laplacian(c, c->d_field[CUP2D_POLD], c->d_field[CUP2D_TMP], 1);
}
// main.cpp:7120-7187 on an adapted grid: volume-weighted means, pressure gradient, V += tmpV / h / h
int amr_project(cup2d_ctx *c, double dt) {
  const AmrDev T = amr_dev(c);
  const size_t n = (size_t)c->nblocks * BC;
  int gs = (int)((n + WG - 1) / WG);
  if (gs > c->grid) gs = c->grid;
  double *pres = c->d_field[CUP2D_PRES], *pold = c->d_field[CUP2D_POLD];
  for (int pass = 0; pass < 2; pass++) {
    const int g = amr_grid(c);
    hipLaunchKernelGGL(k_amr_wsum, dim3(g), dim3(WG), 0, c->stream, pres, T, c->nblocks, c->d_partials);
    hipLaunchKernelGGL(k_amr_wsum_final, dim3(1), dim3(WG), 0, c->stream, c->d_partials, g, c->d_red);
    if (c->allreduce && c->allreduce(c->comm_user, c->d_red, 2, 0, c->stream) != 0) { set_error("allreduce callback failed"); return CUP2D_ERR_COMM; }
    if (pass == 0) hipLaunchKernelGGL(k_amr_shift<0>, dim3(gs), dim3(WG), 0, c->stream, pres, pold, c->d_red, n);
    else hipLaunchKernelGGL(k_amr_shift<1>, dim3(gs), dim3(WG), 0, c->stream, pres, pold, c->d_red, n);
  }
  CUP2D_TRY(amr_pressure_correction(c, pres, c->d_field[CUP2D_TMPV], dt));
  hipLaunchKernelGGL(k_amr_axpy<1>, dim3(amr_grid(c)), dim3(WG), 0, c->stream, (double2 *)c->d_field[CUP2D_VEL], nullptr,
                     (const double2 *)c->d_field[CUP2D_TMPV], T, c->nblocks, 1.0);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// tmpV = KernelAdvectDiffuse(vel) with the flux correction (main.cpp:6611-6617 / 6627-6633)
int amr_advect_diffuse_rhs(cup2d_ctx *c, const double *vel, double *tmpV, double nu, double dt, int blocks) {
  const AmrDev T = amr_dev(c);
  // one rank, FAST arithmetic, all blocks: the quads through the quad kernel, level by level (afac = -dt h), the other blocks
  // through the per-block kernel with the interpolated tile (CUP2D_ADVECT_WALK=0: every block through that one)
  static const bool use_walk = [] { const char *e = getenv("CUP2D_ADVECT_WALK"); return !e || atoi(e) != 0; }();
  if (use_walk && c->math != CUP2D_MATH_STRICT && blocks == CUP2D_BLOCKS_ALL && c->nghost == 0 && ((size_t)c->ntotal << 10) < (1ull << 32)) {
    const AmrTopo::Quads *Q = nullptr;
    CUP2D_TRY(amr_quads(c, &Q));
    for (size_t k = 0; k < Q->level.size(); k++) {
      const double h = c->amr.h0 / (double)(1 << Q->level[k]);
      CUP2D_TRY(launch_advect_walk_rhs(c, vel, tmpV, Q->d_quads[k], Q->nq[k], -dt * h, nu * dt));  // main.cpp:5446-5447
    }
    if (Q->nleft)
      hipLaunchKernelGGL(k_amr_advect<WenoFast>, dim3(amr_grid(c, Q->nleft)), dim3(WG), 0, c->stream, (const double2 *)vel, (double2 *)tmpV, T,
                         c->amr.d_faces2, Q->nleft, nu, dt, (const int32_t *)Q->d_left);
    CUP2D_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_amr_fillcases2, dim3(amr_grid(c)), dim3(WG), 0, c->stream, (double2 *)tmpV, T, c->amr.d_faces2, c->nblocks);
    CUP2D_HIP_CHECK(hipGetLastError());
    return CUP2D_OK;
  }
  CUP2D_TRY(amr_phased(c, blocks, vel, 2, CUP2D_CELLS_HALO3, [&](const int32_t *list, int n) {
    if (c->math == CUP2D_MATH_STRICT)
      hipLaunchKernelGGL(k_amr_advect<WenoStrict>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, (const double2 *)vel, (double2 *)tmpV,
                         T, c->amr.d_faces2, n, nu, dt, list);
    else
      hipLaunchKernelGGL(k_amr_advect<WenoFast>, dim3(amr_grid(c, n)), dim3(WG), 0, c->stream, (const double2 *)vel, (double2 *)tmpV, T,
                         c->amr.d_faces2, n, nu, dt, list);
  }));
  if (blocks != CUP2D_BLOCKS_INNER) {
    CUP2D_TRY(amr_exchange_faces(c, c->amr.d_faces2, 4 * BS * 2));
    hipLaunchKernelGGL(k_amr_fillcases2, dim3(amr_grid(c)), dim3(WG), 0, c->stream, (double2 *)tmpV, T, c->amr.d_faces2, c->nblocks);
  }
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
}  // namespace cup2d
