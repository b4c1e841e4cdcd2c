// krylov_fused.hip -- tile-fused BiCGSTAB sweeps: the block-Jacobi preconditioner on the FP64 matrix
// cores, fused with the 5-point operator that consumes its result.
//
// Reference: BiCGSTABSolver::main cuda.cu:403-548 -- the same recurrences as krylov.hip, which runs them as
// five sweeps moving 184 B/cell per iteration.  A sweep pair (A,B) or (C,D) has no reduction between its two
// halves; it is split into two launches only because nu = A z needs the edge cells of z = P_inv p of the four
// NEIGHBOUR blocks.  P_inv is block-local and dense (64x64), i.e. 4096 FMA per block on the matrix cores,
// which idle on this bandwidth-bound path: so a wave RECOMPUTES the z edges it needs from the neighbours'
// p instead of reading z back from HBM.  One wave owns a tile of 16 consecutive blocks (the N of a
// v_mfma_f64_16x16x4_f64; in the reference's Hilbert order an aligned run of 16 blocks is a 4x4 patch):
//
//   ring     the (block, side) pairs of the tile whose neighbour lies outside it are collected (ballot +
//            prefix popcount); for every 16 of them: load the neighbour blocks, form v (p_new or s), stage
//            it in LDS, Z = V P_inv on the MFMA pipe, keep the 8 edge cells the tile touches
//   tile     the same for the 16 blocks of the tile; v is also written out (p or s)
//   stencil  y = A z from the LDS tile + edges, the dot products of sweeps B / D fused in
//
// z and z2 are never stored.  The x update moves into the preconditioned space -- y += alpha p + omega s,
// x = x0 + P_inv y once at the end (P_inv is linear, z = P_inv p, z2 = P_inv s) -- so sweep E reads p
// instead of z and z2.  Per iteration and cell:
//   AB  reads p, nu, r, rhat (32) writes p', nu' (16)           = 48 B   (+ ring re-reads, served by L2)
//   CD  reads r, nu' (16)         writes s, t (16)              = 32 B
//   E   reads y, p', s, t, rhat (40) writes y, r (16)           = 56 B        total 136 B (five sweeps: 184 B)
// p and nu are double-buffered (a ring recomputation must see the OLD p, nu of a block another wave may
// already have advanced) and s gets its own vector for the same reason.
#include <stdlib.h>
#include <string.h>

#include "block.h"
#include "krylov_common.h"
#include "precond_mfma.h"

namespace cup2d {

constexpr int TB = 16;  // blocks per tile
constexpr int XS = 66;  // LDS stride of one block in the staging tile: the A-operand reads of a half-wave
                        // (block = lane%16, k = 4ks + lane/16) then fall on 32 distinct 8-byte banks

struct FusedLds {
  double S[TB * XS];     // v of 16 blocks, then (same storage) z of those blocks
  double GE[TB * 4 * BS];  // z on the ghost edges of the tile's blocks: [block][W,E,S,N][position]
  int ring_nb[TB * 4];   // neighbour block of ring entry e ...
  int ring_dst[TB * 4];  // ... and the slot (block * 4 + side) it feeds
};

// cell (iy*8+ix) at position q of the edge on side s (W, E, S, N) of a block
static __device__ __forceinline__ int edge_cell(int s, int q) {
  return s == 0 ? q * BS : s == 1 ? q * BS + (BS - 1) : s == 2 ? q : (BS - 1) * BS + q;
}

// S (v, block-major) -> Z = V P_inv -> S (z, block-major).  Same MFMA sequence as k_sweepA_mfma /
// k_sweepC_mfma (precond_tile), so z is bit-identical to the unfused MFMA preconditioner.
static __device__ __forceinline__ void tile_precond(double *S, const PinvFragments &P, int lane) {
  double xa[16];
  const int ablk = lane & 15, akk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ks++) xa[ks] = S[ablk * XS + 4 * ks + akk];
  v4f64 acc[4];
  precond_tile(xa, P, acc);
  wave_lds_sync();  // every lane has read its operands before the tile is overwritten
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) S[(akk + 4 * v) * XS + 16 * nt + ablk] = acc[nt][v];
  wave_lds_sync();
}

struct FusedArgs {
  const double *in0, *in1, *in2;  // AB: p, nu, r     CD: r, nu, -
  double *w;                      // AB: rhat (written on a restart)
  double *vout, *yout;            // AB: p', nu'      CD: s, t
};

// MODE 0 (sweeps A+B): v = p' = beta (p - omega nu) + r   (cuda.cu:478-483; restart: p' = rhat = r, 461-476)
//                      y = nu' = A P_inv p' ; partial(rhat . nu')                       (484-488)
// MODE 1 (sweeps C+D): v = s  = r - alpha nu'                                            (499-502)
//                      y = t  = A P_inv s  ; partial(t . s, t . t)                      (503-509)
template <int MODE, bool MERGE>
__global__ __launch_bounds__(WG, 2) void k_fused(FusedArgs A, const double *__restrict__ Pinv,
                                                 const int *__restrict__ nbr, KrylovScalars *sc, double *partials,
                                                 int count, double *red, unsigned *ticket, int dbg) {
  __shared__ FusedLds lds[WPG];
  if (sc->status != 0) return;
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  FusedLds &L = lds[wave];
  const int ix = lane & 7, iy = lane >> 3;
  PinvFragments P;
  P.load(Pinv, lane);
  const double c1 = MODE == 0 ? -sc->omega : -sc->alpha;  // momega | malpha
  const double beta = sc->beta;
  const bool restart = MODE == 0 && sc->restart_flag != 0;
  constexpr int NDOT = MODE == 0 ? 1 : 2;
  double acc[NDOT];
#pragma unroll
  for (int i = 0; i < NDOT; i++) acc[i] = 0.0;

  // v of block `blk` at this lane's cell -- the arithmetic of k_sweepA_fd / k_sweepC_fd, operation for operation
  const auto form_v = [&](double a, double b, double c) -> double {
    if (MODE == 0) {
      if (restart) return c;
      double v = a + c1 * b;
      v = v * beta;
      return v + c;
    }
    return a + c1 * b;
  };

  const int ntiles = (count + TB - 1) / TB;
  const TileRange tr = tile_range(ntiles);
  for (int t = tr.begin; t < tr.end; t += tr.stride) {
    const int b0 = t * TB;
    const int nvalid = min(TB, count - b0);
    // ---- classify the 64 (block, side) neighbour slots of the tile: lane = block * 4 + side ----
    const int si = lane >> 2, ss = lane & 3;
    const int nb = si < nvalid ? nbr[4 * (b0 + si) + ss] : CUP2D_WALL;
    const bool is_ring = si < nvalid && nb >= 0 && (nb < b0 || nb >= b0 + nvalid);
    const unsigned long long rmask = __ballot(is_ring);
    int nring = __popcll(rmask);
    if (is_ring) {
      const int slot = __popcll(rmask & ((1ull << lane) - 1ull));
      L.ring_nb[slot] = nb;
      L.ring_dst[slot] = lane;
    }
    if (dbg & 1) {  // timing experiment (CUP2D_FUSED_DBG): no ring at all -- WRONG results
      nring = 0;
#pragma unroll
      for (int q = 0; q < BS; q++) L.GE[lane * BS + q] = 0.0;
    }
    wave_lds_sync();
    // ---- ring: z on the edges of the blocks around the tile, 16 entries per pass ----
    for (int base = 0; base < nring; base += TB) {
      const int ne = min(TB, nring - base);
#pragma unroll
      for (int h = 0; h < TB; h += 8) {
        double ra[8], rb[8], rc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int blk = uniform(L.ring_nb[base + min(h + e, ne - 1)]);
          const size_t o = (size_t)blk * BC + lane;
          ra[e] = A.in0[o];
          rb[e] = A.in1[o];
          rc[e] = MODE == 0 ? A.in2[o] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) L.S[(h + e) * XS + lane] = form_v(ra[e], rb[e], rc[e]);
      }
      wave_lds_sync();
      if (!(dbg & 2)) tile_precond(L.S, P, lane);  // dbg 2: ring loads but no ring MFMA -- WRONG results
      // entry e feeds slot dst = block*4 + side with the OPPOSITE edge of the neighbour block
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int idx = lane + 64 * h, e = idx >> 3, q = idx & 7;
        if (e < ne) {
          const int dst = L.ring_dst[base + e];
          L.GE[dst * BS + q] = L.S[e * XS + edge_cell((dst & 3) ^ 1, q)];
        }
      }
      wave_lds_sync();
    }
    // ---- the tile's own blocks ----
#pragma unroll
    for (int h = 0; h < TB; h += 8) {
      double ra[8], rb[8], rc[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const size_t o = (size_t)(b0 + min(h + e, nvalid - 1)) * BC + lane;
        ra[e] = A.in0[o];
        rb[e] = A.in1[o];
        rc[e] = MODE == 0 ? A.in2[o] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const double v = form_v(ra[e], rb[e], rc[e]);
        L.S[(h + e) * XS + lane] = v;
        if (h + e < nvalid) {
          const size_t o = (size_t)(b0 + h + e) * BC + lane;
          A.vout[o] = v;
          if (MODE == 0 && restart) A.w[o] = rc[e];  // rhat = r
        }
      }
    }
    wave_lds_sync();
    if (!(dbg & 4)) tile_precond(L.S, P, lane);  // dbg 4: no tile MFMA -- WRONG results
    // ---- edges inside the tile and at domain walls (ScalarLab::Neumann2D, main.cpp:3210-3255: ghost =
    //      edge cell), from the z tile: this lane's (block, side) slot ----
    if (si < nvalid && !is_ring) {
      const int sblk = nb < 0 ? si : nb - b0, sside = nb < 0 ? ss : ss ^ 1;
#pragma unroll
      for (int q = 0; q < BS; q++) L.GE[lane * BS + q] = L.S[sblk * XS + edge_cell(sside, q)];
    }
    wave_lds_sync();
    // ---- y = A z (operand order of k_sweepBD / pressure_rhs1 main.cpp:6228) + the fused dot products ----
    const double *wsrc = MODE == 0 ? (restart ? A.in2 : A.w) : A.vout;
    for (int i0 = 0; i0 < nvalid; i0 += 4) {  // four blocks per round: their w loads are in flight together
      double wc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) wc[u] = wsrc[(size_t)(b0 + min(i0 + u, nvalid - 1)) * BC + lane];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u;
        if (i < nvalid) {
          const double *zb = L.S + i * XS + lane;
          const double *ge = L.GE + i * 4 * BS;
          const double l0 = zb[0];
          const double l1 = *(ix > 0 ? zb - 1 : ge + 0 * BS + iy);
          const double l2 = *(ix < BS - 1 ? zb + 1 : ge + 1 * BS + iy);
          const double l3 = *(iy > 0 ? zb - BS : ge + 2 * BS + ix);
          const double l4 = *(iy < BS - 1 ? zb + BS : ge + 3 * BS + ix);
          const double yv = l1 + l2 + l3 + l4 - 4 * l0;
          A.yout[(size_t)(b0 + i) * BC + lane] = yv;
          acc[0] = __builtin_fma(yv, wc[u], acc[0]);
          if constexpr (NDOT == 2) acc[1] = __builtin_fma(yv, yv, acc[1]);
        }
      }
    }
    wave_lds_sync();  // the next tile overwrites S, GE and the ring list
  }
  workgroup_reduce_store<NDOT, false, MERGE>(acc, partials, 0, 0);
  if (MERGE && arrive_last(ticket)) finish_reduce<true>(partials, gridDim.x, NDOT, 0, red, sc, MODE + 1, nullptr);
}

// ---- sweep E in the preconditioned space ---------------------------------------------------------
// y += alpha p + omega s ; r = s - omega t ; partial(rhat.r, r.r), max|r|   (cuda.cu:498, 520-525, 440-442
// with x = x0 + P_inv y); best-iterate copy deferred like k_sweepE's (cuda.cu:535-538)
template <bool MERGE>
__global__ __launch_bounds__(WG) void k_sweepE_y(double2 *__restrict__ y, double2 *__restrict__ yopt,
                                                 const double2 *__restrict__ p, double2 *__restrict__ r,
                                                 const double2 *__restrict__ s, const double2 *__restrict__ t,
                                                 const double2 *__restrict__ rhat, KrylovScalars *sc, double *partials,
                                                 size_t n2, double *red, unsigned *ticket, int *host_status) {
  if (sc->status != 0) return;
  const double alpha = sc->alpha, omega = sc->omega, momega = -sc->omega;
  const int save = sc->x_is_best;
  double sm[2] = {0.0, 0.0}, m[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    double2 yv = y[i];
    if (save) yopt[i] = yv;
    const double2 pv = p[i], sv = s[i], tv = t[i], hv = rhat[i];
    yv.x = yv.x + alpha * pv.x;
    yv.y = yv.y + alpha * pv.y;
    yv.x = yv.x + omega * sv.x;
    yv.y = yv.y + omega * sv.y;
    y[i] = yv;
    double2 rv;
    rv.x = sv.x + momega * tv.x;
    rv.y = sv.y + momega * tv.y;
    r[i] = rv;
    sm[0] = __builtin_fma(hv.x, rv.x, sm[0]);
    sm[0] = __builtin_fma(hv.y, rv.y, sm[0]);
    sm[1] = __builtin_fma(rv.x, rv.x, sm[1]);
    sm[1] = __builtin_fma(rv.y, rv.y, sm[1]);
    m[0] = fmax(m[0], fmax(fabs(rv.x), fabs(rv.y)));
  }
  workgroup_reduce_store<2, false, MERGE>(sm, partials, 0);
  workgroup_reduce_store<1, true, MERGE>(m, partials, 2);
  if (MERGE && arrive_last(ticket)) finish_reduce<true>(partials, gridDim.x, 2, 1, red, sc, 3, host_status);
}

bool fused_supported(const cup2d_ctx *c) { return !c->mat.active && c->nghost == 0 && !c->exchange; }

static int ensure_fused_buffers(cup2d_ctx *c) {
  const size_t bytes = (size_t)c->ntotal * BC * sizeof(double);
  double **v[] = {&c->d_p2, &c->d_nu2, &c->d_s, &c->d_y, &c->d_yopt};
  for (double **p : v)
    if (!*p) {
      CUP2D_HIP_CHECK(hipMalloc(p, bytes));
      CUP2D_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, c->stream));
    }
  return CUP2D_OK;
}

int launch_init_residual(cup2d_ctx *c, const double *x, const double *b, int G);  // krylov.hip

// b = TMP, x0 = PRES, result -> PRES (same contract as solve_impl)
int solve_fused_impl(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter, int *iters,
                     int *restarts, double *linf, double *linf_init) {
  CUP2D_TRY(ensure_fused_buffers(c));
  const int nb = c->nblocks;
  const size_t n = (size_t)nb * BC;
  double *x = c->d_field[CUP2D_PRES];
  const double *b = c->d_field[CUP2D_TMP];
  KrylovScalars init;
  ::memset(&init, 0, sizeof init);
  init.alpha = init.beta = init.omega = init.rho_prev = init.rho_curr = 1.0;
  init.eps = 1e-21;  // cuda.cu:409
  init.err = init.err_init = init.err_opt = 1e50;
  init.max_error = max_error; init.max_rel_error = max_rel_error;
  init.max_restarts = max_restarts; init.max_iter = max_iter;
  *c->h_sc = init;
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_sc, c->h_sc, sizeof init, hipMemcpyHostToDevice, c->stream));
  const int G = grid_for(c, nb);
  int gridE = (int)((n / 2 + WG - 1) / WG);
  if (gridE > c->grid) gridE = c->grid;
  // one wave per 16-block tile, 2 workgroups per CU (P_inv lives in 128 VGPRs of every wave)
  const int ntiles = (nb + TB - 1) / TB;
  int gridF = (ntiles + WPG - 1) / WPG;
  const int capF = 2 * (c->num_cus > 0 ? c->num_cus : 256);
  if (gridF > capF) gridF = capF;
  if (gridF >= 8) gridF -= gridF % 8;
  if (gridF < 1) gridF = 1;
  const bool merge = c->finish_in_kernel && !c->allreduce;
  static const int dbg = [] { const char *e = getenv("CUP2D_FUSED_DBG"); return e ? atoi(e) : 0; }();

  {
    ProfScope prof(c, CUP2D_T_INIT_RESIDUAL);
    CUP2D_TRY(launch_init_residual(c, x, b, G));
  }
  CUP2D_TRY(finish(c, G, 1, 1, 0, false));
  // p, nu start at zero (cuda.cu:436-437); so does the accumulated correction
  CUP2D_HIP_CHECK(hipMemsetAsync(c->d_p, 0, n * sizeof(double), c->stream));
  CUP2D_HIP_CHECK(hipMemsetAsync(c->d_nu, 0, n * sizeof(double), c->stream));
  CUP2D_HIP_CHECK(hipMemsetAsync(c->d_y, 0, n * sizeof(double), c->stream));

  static const int AHEAD = [] {
    const char *e = getenv("CUP2D_SOLVE_AHEAD");
    const int v = e ? atoi(e) : 4;
    return v < 1 ? 1 : (v > cup2d_ctx::SOLVE_AHEAD ? cup2d_ctx::SOLVE_AHEAD : v);
  }();
  for (int i = 0; i < AHEAD; i++) c->h_status[i] = 0;
  for (int k = 0; k <= max_iter + AHEAD; k++) {
    const int slot = k % AHEAD;
    if (k >= AHEAD) {
      CUP2D_HIP_CHECK(hipEventSynchronize(c->solve_ev[slot]));
      if (*(volatile int *)&c->h_status[slot] != 0) break;
    }
    c->prof_sample = (k % 8 == 0) && k < max_iter;
    double *p_in = (k & 1) ? c->d_p2 : c->d_p, *p_out = (k & 1) ? c->d_p : c->d_p2;
    double *nu_in = (k & 1) ? c->d_nu2 : c->d_nu, *nu_out = (k & 1) ? c->d_nu : c->d_nu2;
    {
      ProfScope prof(c, CUP2D_T_SWEEP_A);
      const FusedArgs a = {p_in, nu_in, c->d_r, c->d_rhat, p_out, nu_out};
      if (merge)
        hipLaunchKernelGGL((k_fused<0, true>), dim3(gridF), dim3(WG), 0, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc,
                           c->d_partials, nb, c->d_red, c->d_ticket, dbg);
      else
        hipLaunchKernelGGL((k_fused<0, false>), dim3(gridF), dim3(WG), 0, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc,
                           c->d_partials, nb, c->d_red, c->d_ticket, dbg);
    }
    CUP2D_HIP_CHECK(hipGetLastError());
    if (!merge) CUP2D_TRY(finish(c, gridF, 1, 0, 1, true));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_C);
      const FusedArgs a = {c->d_r, nu_out, nullptr, nullptr, c->d_s, c->d_t};
      if (merge)
        hipLaunchKernelGGL((k_fused<1, true>), dim3(gridF), dim3(WG), 0, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc,
                           c->d_partials, nb, c->d_red, c->d_ticket, dbg);
      else
        hipLaunchKernelGGL((k_fused<1, false>), dim3(gridF), dim3(WG), 0, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc,
                           c->d_partials, nb, c->d_red, c->d_ticket, dbg);
    }
    CUP2D_HIP_CHECK(hipGetLastError());
    if (!merge) CUP2D_TRY(finish(c, gridF, 2, 0, 2, true));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_E);
      if (merge)
        hipLaunchKernelGGL(k_sweepE_y<true>, dim3(gridE), dim3(WG), 0, c->stream, (double2 *)c->d_y, (double2 *)c->d_yopt,
                           (const double2 *)p_out, (double2 *)c->d_r, (const double2 *)c->d_s, (const double2 *)c->d_t,
                           (const double2 *)c->d_rhat, c->d_sc, c->d_partials, n / 2, c->d_red, c->d_ticket,
                           &c->h_status[slot]);
      else
        hipLaunchKernelGGL(k_sweepE_y<false>, dim3(gridE), dim3(WG), 0, c->stream, (double2 *)c->d_y, (double2 *)c->d_yopt,
                           (const double2 *)p_out, (double2 *)c->d_r, (const double2 *)c->d_s, (const double2 *)c->d_t,
                           (const double2 *)c->d_rhat, c->d_sc, c->d_partials, n / 2, c->d_red, c->d_ticket,
                           &c->h_status[slot]);
    }
    CUP2D_HIP_CHECK(hipGetLastError());
    if (!merge) CUP2D_TRY(finish(c, gridE, 2, 1, 3, true, &c->h_status[slot]));
    CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
  }
  c->prof_sample = true;
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof init, hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  // cuda.cu:546-547: return x_opt = x0 + P_inv y_opt
  const double *ybest = c->h_sc->x_is_best ? c->d_y : c->d_yopt;
  CUP2D_TRY(launch_precond(c, ybest, c->d_s, 0, nb));
  CUP2D_TRY(launch_axpy_field(c, x, c->d_s, 1.0, n));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (iters) *iters = c->h_sc->iter;
  if (restarts) *restarts = c->h_sc->restarts;
  if (linf) *linf = c->h_sc->err_opt;
  if (linf_init) *linf_init = c->h_sc->err_init;
  return CUP2D_OK;
}

}  // namespace cup2d
