// krylov.hip -- block-Jacobi preconditioned BiCGSTAB for the pressure Poisson equation, matrix-free
// on the block grid (SURVEY.md rows a16, a17 and the scalar kernels cuda.cu:303-330).
//
// Reference: BiCGSTABSolver::main cuda.cu:403-548.  The reference spends, per iteration, two COO
// cuSPARSE SpMVs (96 B/row), two cuBLAS DGEMMs, ~17 cuBLAS level-1 passes, four host
// synchronisations and four MPI_Allreduce.  Here the same recurrences run as FIVE fused sweeps:
//   A  p = beta*(p - omega*nu) + r ; z = P_inv p                      (cuda.cu:478-486)
//   B  nu = A z ; partial(rhat.nu)                                     (cuda.cu:487-488)
//   C  r -= alpha*nu ; z2 = P_inv r                                    (cuda.cu:498-505, x update deferred)
//   D  t = A z2 ; partial(t.r, t.t)                                    (cuda.cu:506-509)
//   E  x += alpha*z + omega*z2 ; r -= omega*t ; partial(rhat.r, r.r, max|r|)   (cuda.cu:498, 520-525, 440-442)
// separated by single-workgroup scalar kernels that finish the reductions and keep alpha, beta,
// omega, the breakdown/restart logic and the best-iterate bookkeeping ON THE DEVICE; the host
// reads one small struct per iteration to learn whether to stop.
#include <string.h>

#include <stdlib.h>

#include "block.h"
#include "krylov_common.h"
#include "precond_mfma.h"

namespace cup2d {

// Three implementations of the block-Jacobi preconditioner z_b = P_inv p_b, selectable with
// cup2d_set_precond (all three apply the same operator; they differ by round-off only):
//   fd   (default) fast diagonalisation: A_loc = T (x) I + I (x) T with T = tridiag(-1,2,-1) = Q diag(lam) Q^T,
//                  so P_inv = -(Q (x) Q) diag(1/(lam_i+lam_j)) (Q (x) Q)^T: four 8x8x8 products per block
//                  (32 FMAs per cell instead of 64), one wave per block, coalesced loads
//   mfma           dense 64x64 product on v_mfma_f64_16x16x4_f64 (precond_mfma.h), 16 blocks per wave
//   lds            dense product with P_inv in LDS and scalar FMAs (the first version)
// (enum PRECOND_* in ctx.h; cup2d_set_precond selects)
static bool use_mfma(const cup2d_ctx *c) { return c->precond == PRECOND_MFMA; }
// grid of the MFMA sweeps: 2 workgroups (8 waves) per CU; every wave keeps P_inv in 128 VGPRs
static int mfma_grid(const cup2d_ctx *c, int count) {
  const int ntiles = (count + 15) / 16;
  int g = (ntiles + WPG - 1) / WPG;
  if (g > 512) g = 512;
  if (g >= 8) g -= g % 8;
  return g < 1 ? 1 : g;
}

// ---- preconditioner: z_b = P_inv p_b, 64x64 symmetric (cuda.cu:484-486 Dgemm(T,N)) ----------
// LDS-resident P_inv (32 KiB per workgroup, loaded once per persistent workgroup); lane i
// accumulates row i: P_inv[j][i] is read with lane-consecutive addresses (conflict-free
// ds_read_b64), p[j] is a wave-uniform broadcast read.
static __device__ __forceinline__ void load_Pinv(const double *__restrict__ Pinv, double *sP) {
  for (int i = threadIdx.x; i < BC * BC; i += WG) sP[i] = Pinv[i];
  __syncthreads();
}
static __device__ __forceinline__ double precond_row(const double *sP, const double *sv, int lane) {
  double acc = 0.0;
#pragma unroll 16
  for (int j = 0; j < BC; j++) acc = __builtin_fma(sP[j * BC + lane], sv[j], acc);
  return acc;
}

__global__ __launch_bounds__(WG) void k_precond(const double *__restrict__ in, double *__restrict__ out,
                                                const double *__restrict__ Pinv, int first, int count) {
  __shared__ double sP[BC * BC];
  __shared__ double sv[WPG][BC];
  load_Pinv(Pinv, sP);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const size_t o = (size_t)(first + rel) * BC + lane;
      sv[wave][lane] = in[o];
      wave_lds_sync();
      out[o] = precond_row(sP, sv[wave], lane);
      wave_lds_sync();
    }
  }
}

// ---- fast-diagonalisation preconditioner ------------------------------------------------------
// fd[0..63] = Q[i][k] = sqrt(2/9) sin((i+1)(k+1)pi/9) (symmetric, orthogonal), fd[64..71] = lam_k =
// 2 - 2cos((k+1)pi/9): the eigen-decomposition of the 8x8 Dirichlet second-difference matrix, built
// on the host in cup2d_create.  A lane (ix, iy) keeps row ix and row iy of Q in registers.
struct FdBasis {
  double qx[BS], qy[BS], sc;
  __device__ __forceinline__ void load(const double *__restrict__ fd, int ix, int iy) {
#pragma unroll
    for (int k = 0; k < BS; k++) {
      qx[k] = fd[ix * BS + k];
      qy[k] = fd[iy * BS + k];
    }
    sc = -1.0 / (fd[BC + ix] + fd[BC + iy]);
  }
};
// z = P_inv v for the block held one cell per lane; bufA / bufB: 64 doubles of LDS each, private
// to the wave.  Four small products X Q, Q^T (.), scale, Q (.), (.) Q^T; rows come back as
// ds_read_b128, columns as ds_read_b64 at immediate offsets.
static __device__ __forceinline__ double precond_fd(double v, const FdBasis &B, double *bufA, double *bufB, int ix,
                                                    int iy, int lane) {
  bufA[lane] = v;
  wave_lds_sync();
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < BS; j++) t = __builtin_fma(bufA[iy * BS + j], B.qx[j], t);
  bufB[lane] = t;
  wave_lds_sync();
  double y = 0.0;
#pragma unroll
  for (int i = 0; i < BS; i++) y = __builtin_fma(bufB[i * BS + ix], B.qy[i], y);
  y *= B.sc;
  bufA[lane] = y;
  wave_lds_sync();
  double w = 0.0;
#pragma unroll
  for (int k = 0; k < BS; k++) w = __builtin_fma(bufA[k * BS + ix], B.qy[k], w);
  bufB[lane] = w;
  wave_lds_sync();
  double z = 0.0;
#pragma unroll
  for (int k = 0; k < BS; k++) z = __builtin_fma(bufB[iy * BS + k], B.qx[k], z);
  wave_lds_sync();
  return z;
}

// ADD: out += P_inv in (the solver's last step x = x0 + P_inv y, cuda.cu:546-547, without a vector in between)
template <bool ADD>
__global__ __launch_bounds__(WG) void k_precond_fd(const double *__restrict__ in, double *__restrict__ out,
                                                   const double *__restrict__ fd, int first, int count) {
  __shared__ double buf[WPG][2][BC];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ix = lane & 7, iy = lane >> 3;
  FdBasis B;
  B.load(fd, ix, iy);
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const size_t o = (size_t)(first + rel) * BC + lane;
      const double z = precond_fd(in[o], B, buf[wave][0], buf[wave][1], ix, iy, lane);
      out[o] = ADD ? out[o] + z : z;
    }
  }
}

// The fused solver's last step x = x0 + P_inv y_opt (cuda.cu:546-547) with the choice of y_opt -- which of the three y buffers
// holds the best iterate, or none of them when no iterate beat x0 -- read from the scalars ON THE DEVICE: the host does not
// have to wait for the solve to end before it can enqueue this launch (and whatever follows the solve in a step).
template <bool ADD>
__global__ __launch_bounds__(WG) void k_final_x_fd(const double *__restrict__ y0, const double *__restrict__ y1,
                                                   const double *__restrict__ y2, double *__restrict__ out,
                                                   const double *__restrict__ fd, const KrylovScalars *__restrict__ sc, int count) {
  __shared__ double buf[WPG][2][BC];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ix = lane & 7, iy = lane >> 3;
  const int best = sc->ybest;
  const bool keep_x0 = sc->best_is_x0 != 0;  // x = x0: nothing to add (ADD), zero (x0 = 0 and known to be)
  if (keep_x0 && ADD) return;
  const double *__restrict__ in = best == 0 ? y0 : (best == 1 ? y1 : y2);
  FdBasis B;
  B.load(fd, ix, iy);
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const size_t o = (size_t)rel * BC + lane;
      if (keep_x0) { out[o] = 0.0; continue; }  // (wave-uniform)
      const double z = precond_fd(in[o], B, buf[wave][0], buf[wave][1], ix, iy, lane);
      out[o] = ADD ? out[o] + z : z;
    }
  }
}
// true if it was launched: the built-in preconditioner in its fast-diagonalisation form (the default)
bool launch_final_x_on_device(cup2d_ctx *c, const double *y0, const double *y1, const double *y2, double *x, bool x0_zero) {
  if (c->precond != PRECOND_FD) return false;
  const int nb = c->nblocks;
  if (x0_zero)
    hipLaunchKernelGGL(k_final_x_fd<false>, dim3(grid_for(c, nb)), dim3(WG), 0, c->stream, y0, y1, y2, x, c->d_fd, c->d_sc, nb);
  else
    hipLaunchKernelGGL(k_final_x_fd<true>, dim3(grid_for(c, nb)), dim3(WG), 0, c->stream, y0, y1, y2, x, c->d_fd, c->d_sc, nb);
  return true;
}

// sweep A, fast-diagonalisation preconditioner; the next block's p, nu, r are in flight while the
// current block is transformed
__global__ __launch_bounds__(WG) void k_sweepA_fd(double *__restrict__ p, const double *__restrict__ nu,
                                                  const double *__restrict__ r, double *__restrict__ rhat,
                                                  double *__restrict__ z, const double *__restrict__ fd,
                                                  const KrylovScalars *__restrict__ sc, int count) {
  __shared__ double buf[WPG][2][BC];
  if (sc->status != 0) return;
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ix = lane & 7, iy = lane >> 3;
  FdBasis B;
  B.load(fd, ix, iy);
  const double beta = sc->beta, momega = -sc->omega;
  const int restart = sc->restart_flag;
  const GroupRange gr = group_range(count);
  int g = gr.begin;
  bool have = g < gr.end && g * WPG + wave < count;
  if (!have) return;
  // loads are unconditional and never merged with other values at a join, so that the wave does not
  // wait for its prefetch (a wave past its last block re-reads the current one)
  size_t o = (size_t)(g * WPG + wave) * BC + lane;
  double pv = p[o], nv = nu[o], rv = r[o];
  while (have) {
    double v;
    if (restart) {  // cuda.cu:461-476: rhat = r, nu = p = 0  =>  p = r
      rhat[o] = rv;
      v = rv;
    } else {        // cuda.cu:478-483: p += (-omega) nu ; p *= beta ; p += r
      v = pv + momega * nv;
      v = v * beta;
      v = v + rv;
    }
    g += gr.stride;
    have = g < gr.end && g * WPG + wave < count;
    const size_t on = have ? (size_t)(g * WPG + wave) * BC + lane : o;
    pv = p[on];
    nv = nu[on];
    rv = r[on];
    p[o] = v;
    z[o] = precond_fd(v, B, buf[wave][0], buf[wave][1], ix, iy, lane);
    o = on;
  }
}

__global__ __launch_bounds__(WG) void k_sweepC_fd(double *__restrict__ r, const double *__restrict__ nu,
                                                  double *__restrict__ z2, const double *__restrict__ fd,
                                                  const KrylovScalars *__restrict__ sc, int count) {
  __shared__ double buf[WPG][2][BC];
  if (sc->status != 0) return;
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ix = lane & 7, iy = lane >> 3;
  FdBasis B;
  B.load(fd, ix, iy);
  const double malpha = -sc->alpha;
  const GroupRange gr = group_range(count);
  int g = gr.begin;
  bool have = g < gr.end && g * WPG + wave < count;
  if (!have) return;
  size_t o = (size_t)(g * WPG + wave) * BC + lane;
  double rv = r[o], nv = nu[o];
  while (have) {
    const double v = rv + malpha * nv;  // cuda.cu:499-502
    g += gr.stride;
    have = g < gr.end && g * WPG + wave < count;
    const size_t on = have ? (size_t)(g * WPG + wave) * BC + lane : o;
    rv = r[on];
    nv = nu[on];
    r[o] = v;
    z2[o] = precond_fd(v, B, buf[wave][0], buf[wave][1], ix, iy, lane);
    o = on;
  }
}

__global__ void k_precond_mfma(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Pinv,
                               int first, int count);
int launch_precond(cup2d_ctx *c, const double *in, double *out, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  if (c->precond == PRECOND_FD)
    hipLaunchKernelGGL(k_precond_fd<false>, dim3(grid_for(c, count)), dim3(WG), 0, c->stream, in, out, c->d_fd, first, count);
  else if (use_mfma(c))
    hipLaunchKernelGGL(k_precond_mfma, dim3(mfma_grid(c, count)), dim3(WG), 0, c->stream, in, out, c->d_Pinv, first, count);
  else
    hipLaunchKernelGGL(k_precond, dim3(grid_for(c, count)), dim3(WG), 0, c->stream, in, out, c->d_Pinv, first, count);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// x += P_inv y over the owned blocks; tmp is a work vector for the preconditioner forms that have no fused kernel
int launch_precond_add(cup2d_ctx *c, const double *y, double *x, double *tmp) {
  const int nb = c->nblocks;
  if (c->precond == PRECOND_FD) {
    hipLaunchKernelGGL(k_precond_fd<true>, dim3(grid_for(c, nb)), dim3(WG), 0, c->stream, y, x, c->d_fd, 0, nb);
    CUP2D_HIP_CHECK(hipGetLastError());
    return CUP2D_OK;
  }
  CUP2D_TRY(launch_precond(c, y, tmp, 0, nb));
  return launch_axpy_field(c, x, tmp, 1.0, (size_t)nb * BC);
}

// ---- sweep A --------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_sweepA(double *__restrict__ p, const double *__restrict__ nu,
                                               const double *__restrict__ r, double *__restrict__ rhat,
                                               double *__restrict__ z, const double *__restrict__ Pinv,
                                               const KrylovScalars *__restrict__ sc, int count) {
  __shared__ double sP[BC * BC];
  __shared__ double sv[WPG][BC];
  if (sc->status != 0) return;
  load_Pinv(Pinv, sP);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double beta = sc->beta, momega = -sc->omega;
  const int restart = sc->restart_flag;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const size_t o = (size_t)rel * BC + lane;
      const double rv = r[o];
      double pv;
      if (restart) {  // cuda.cu:461-476: rhat = r, nu = p = 0  =>  p = r
        rhat[o] = rv;
        pv = rv;
      } else {        // cuda.cu:478-483: p += (-omega) nu ; p *= beta ; p += r
        pv = p[o] + momega * nu[o];
        pv = pv * beta;
        pv = pv + rv;
      }
      p[o] = pv;
      sv[wave][lane] = pv;
      wave_lds_sync();
      z[o] = precond_row(sP, sv[wave], lane);
      wave_lds_sync();
    }
  }
}


// ---- MFMA variants of precond / sweep A / sweep C -------------------------------------------------
__global__ __launch_bounds__(WG, 2) void k_precond_mfma(const double *__restrict__ in, double *__restrict__ out,
                                                        const double *__restrict__ Pinv, int first, int count) {
  const int lane = threadIdx.x & 63;
  PinvFragments P;
  P.load(Pinv, lane);
  const int ntiles = (count + 15) / 16;
  const TileRange tr = tile_range(ntiles);
  for (int t = tr.begin; t < tr.end; t += tr.stride) {
    const size_t base = ((size_t)first + (size_t)t * 16) * BC;
    const int nvalid = min(16, count - t * 16);
    const bool ok = (lane & 15) < nvalid;
    double xa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ks++) xa[ks] = ok ? in[base + a_offset(lane, ks)] : 0.0;
    v4f64 acc[4];
    precond_tile(xa, P, acc);
    store_tile(out, base, nvalid, lane, acc);
  }
}

__global__ __launch_bounds__(WG, 2) void k_sweepA_mfma(double *__restrict__ p, const double *__restrict__ nu,
                                                       const double *__restrict__ r, double *__restrict__ rhat,
                                                       double *__restrict__ z, const double *__restrict__ Pinv,
                                                       const KrylovScalars *__restrict__ sc, int count) {
  if (sc->status != 0) return;
  const int lane = threadIdx.x & 63;
  PinvFragments P;
  P.load(Pinv, lane);
  const double beta = sc->beta, momega = -sc->omega;
  const int restart = sc->restart_flag;
  const int ntiles = (count + 15) / 16;
  const TileRange tr = tile_range(ntiles);
  for (int t = tr.begin; t < tr.end; t += tr.stride) {
    const size_t base = (size_t)t * 16 * BC;
    const int nvalid = min(16, count - t * 16);
    const bool ok = (lane & 15) < nvalid;
    double xa[16];
    if (restart) {  // cuda.cu:461-476
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        const size_t o = base + a_offset(lane, ks);
        const double rv = ok ? r[o] : 0.0;
        if (ok) { rhat[o] = rv; p[o] = rv; }
        xa[ks] = rv;
      }
    } else {        // cuda.cu:478-483
      double pv[16], nv[16], rv[16];
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        const size_t o = base + a_offset(lane, ks);
        pv[ks] = ok ? p[o] : 0.0;
        nv[ks] = ok ? nu[o] : 0.0;
        rv[ks] = ok ? r[o] : 0.0;
      }
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        double v = pv[ks] + momega * nv[ks];
        v = v * beta;
        v = v + rv[ks];
        xa[ks] = v;
        if (ok) p[base + a_offset(lane, ks)] = v;
      }
    }
    v4f64 acc[4];
    precond_tile(xa, P, acc);
    store_tile(z, base, nvalid, lane, acc);
  }
}

__global__ __launch_bounds__(WG, 2) void k_sweepC_mfma(double *__restrict__ r, const double *__restrict__ nu,
                                                       double *__restrict__ z2, const double *__restrict__ Pinv,
                                                       const KrylovScalars *__restrict__ sc, int count) {
  if (sc->status != 0) return;
  const int lane = threadIdx.x & 63;
  PinvFragments P;
  P.load(Pinv, lane);
  const double malpha = -sc->alpha;
  const int ntiles = (count + 15) / 16;
  const TileRange tr = tile_range(ntiles);
  for (int t = tr.begin; t < tr.end; t += tr.stride) {
    const size_t base = (size_t)t * 16 * BC;
    const int nvalid = min(16, count - t * 16);
    const bool ok = (lane & 15) < nvalid;
    double rv[16], nv[16], xa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
      const size_t o = base + a_offset(lane, ks);
      rv[ks] = ok ? r[o] : 0.0;
      nv[ks] = ok ? nu[o] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
      xa[ks] = rv[ks] + malpha * nv[ks];  // cuda.cu:499-502
      if (ok) r[base + a_offset(lane, ks)] = xa[ks];
    }
    v4f64 acc[4];
    precond_tile(xa, P, acc);
    store_tile(z2, base, nvalid, lane, acc);
  }
}

// ---- sweeps B and D: y = A x with fused dot products -----------------------------------------
// NDOT = 1: partial(w.y)            (B: w = rhat)
// NDOT = 2: partial(y.w, y.y)       (D: w = r)
// MERGE: the last workgroup to arrive finishes the reduction and runs the scalar update (stage NDOT) in
// this launch (krylov_common.h arrive_last); only for a single launch over all blocks (poff == 0)
template <int NDOT, bool MERGE>
__global__ __launch_bounds__(WG) void k_sweepBD(const double *__restrict__ x, double *__restrict__ y,
                                                const double *__restrict__ w, const int *__restrict__ nbr,
                                                KrylovScalars *sc, double *partials, int first, int count, int poff,
                                                double *red, unsigned *ticket) {
  __shared__ double slabs[WPG][LAB1 * LAB1];
  if (sc->status != 0) return;
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double *slab = slabs[wave];
  const int4 *nbr4 = (const int4 *)nbr;
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  double acc[NDOT];
#pragma unroll
  for (int i = 0; i < NDOT; i++) acc[i] = 0.0;
  // Two blocks of this wave are in flight while a third is computed (register double buffer, loop
  // unrolled by two): 24 B/cell of streaming needs more bytes in flight per wave than one 8x8 block.
  const GroupRange gr = group_range(count);
  int g = gr.begin;
  const auto valid = [&](int gg) { return gg < gr.end && gg * WPG + wave < count; };
  if (valid(g)) {
    int b0 = first + g * WPG + wave;
    int g1 = g + gr.stride;
    int b1 = valid(g1) ? first + g1 * WPG + wave : b0;
    ScalarLab1Regs R0 = fetch_scalar_lab1(x, nbr4, b0, lane);
    double w0 = w[(size_t)b0 * BC + lane];
    ScalarLab1Regs R1 = fetch_scalar_lab1(x, nbr4, b1, lane);
    double w1 = w[(size_t)b1 * BC + lane];
    bool have0 = true, have1 = valid(g1);
    const auto compute = [&](int b, double wc) {
      const double l0 = slab[c0], l1 = slab[c0 - 1], l2 = slab[c0 + 1], l3 = slab[c0 - LAB1], l4 = slab[c0 + LAB1];
      const double yv = l1 + l2 + l3 + l4 - 4 * l0;
      y[(size_t)b * BC + lane] = yv;
      acc[0] = __builtin_fma(yv, wc, acc[0]);
      if constexpr (NDOT == 2) acc[NDOT - 1] = __builtin_fma(yv, yv, acc[NDOT - 1]);
    };
    while (have0) {
      {
        store_scalar_lab1(R0, lane, slab);
        const double wc = w0;
        const int bc = b0;
        wave_lds_sync();
        const int gn = g + 2 * gr.stride;
        have0 = valid(gn);
        b0 = have0 ? first + gn * WPG + wave : bc;  // past the end: re-read, branch-free
        R0 = fetch_scalar_lab1(x, nbr4, b0, lane);
        w0 = w[(size_t)b0 * BC + lane];
        compute(bc, wc);
        wave_lds_sync();
      }
      if (!have1) break;
      {
        store_scalar_lab1(R1, lane, slab);
        const double wc = w1;
        const int bc = b1;
        wave_lds_sync();
        const int gn = g + 3 * gr.stride;
        have1 = valid(gn);
        b1 = have1 ? first + gn * WPG + wave : bc;
        R1 = fetch_scalar_lab1(x, nbr4, b1, lane);
        w1 = w[(size_t)b1 * BC + lane];
        compute(bc, wc);
        wave_lds_sync();
      }
      g += 2 * gr.stride;
    }
  }
  workgroup_reduce_store<NDOT, false, MERGE>(acc, partials, 0, poff);
  if (MERGE && arrive_last(ticket)) finish_reduce<true>(partials, gridDim.x, NDOT, 0, red, sc, NDOT, nullptr);
}

// ---- sweep C ----------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_sweepC(double *__restrict__ r, const double *__restrict__ nu,
                                               double *__restrict__ z2, const double *__restrict__ Pinv,
                                               const KrylovScalars *__restrict__ sc, int count) {
  __shared__ double sP[BC * BC];
  __shared__ double sv[WPG][BC];
  if (sc->status != 0) return;
  load_Pinv(Pinv, sP);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double malpha = -sc->alpha;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const size_t o = (size_t)rel * BC + lane;
      const double rv = r[o] + malpha * nu[o];  // cuda.cu:499-502
      r[o] = rv;
      sv[wave][lane] = rv;
      wave_lds_sync();
      z2[o] = precond_row(sP, sv[wave], lane);
      wave_lds_sync();
    }
  }
}

// ---- sweep E ----------------------------------------------------------------------------------
template <bool MERGE>
__global__ __launch_bounds__(WG) void k_sweepE(double *__restrict__ x, double *__restrict__ xopt,
                                               const double *__restrict__ z, const double *__restrict__ z2,
                                               double *__restrict__ r, const double *__restrict__ t,
                                               const double *__restrict__ rhat, KrylovScalars *sc, double *partials,
                                               size_t n, double *red, unsigned *ticket, int *host_status) {
  if (sc->status != 0) return;
  const double alpha = sc->alpha, omega = sc->omega, momega = -sc->omega;
  const int save = sc->x_is_best;
  double s[2] = {0.0, 0.0}, m[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    double xv = x[i];
    if (save) xopt[i] = xv;  // cuda.cu:535-538, deferred: the iterate about to be overwritten was the best
    xv = xv + alpha * z[i];  // cuda.cu:498
    xv = xv + omega * z2[i]; // cuda.cu:520
    x[i] = xv;
    const double rv = r[i] + momega * t[i];  // cuda.cu:521-524
    r[i] = rv;
    s[0] = __builtin_fma(rhat[i], rv, s[0]);
    s[1] = __builtin_fma(rv, rv, s[1]);
    m[0] = fmax(m[0], fabs(rv));
  }
  workgroup_reduce_store<2, false, MERGE>(s, partials, 0);
  workgroup_reduce_store<1, true, MERGE>(m, partials, 2);
  if (MERGE && arrive_last(ticket)) finish_reduce<true>(partials, gridDim.x, 2, 1, red, sc, 3, host_status);
}

// ---- initial residual: r = b - A x0, rhat = r, partial(r.r, max|r|) (cuda.cu:412-436) ---------
// X0ZERO: x is known to be zero and is not read: r = b - 0 (what the stencil over zeros gives, bit for bit)
template <bool X0ZERO>
__global__ __launch_bounds__(WG) void k_init_residual(const double *__restrict__ x, const double *__restrict__ b,
                                                      double *__restrict__ r, double *__restrict__ rhat,
                                                      const int *__restrict__ nbr, double *__restrict__ partials,
                                                      int first, int count, int poff) {
  __shared__ double slabs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *slab = slabs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  double s[1] = {0.0}, m[1] = {0.0};
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int bk = first + rel;
      double lap = 0.0;
      if (!X0ZERO) {
        load_scalar_lab1(x, nbr, bk, lane, slab);
        wave_lds_sync();
        const double l0 = slab[c0], l1 = slab[c0 - 1], l2 = slab[c0 + 1], l3 = slab[c0 - LAB1], l4 = slab[c0 + LAB1];
        lap = l1 + l2 + l3 + l4 - 4 * l0;
      }
      const size_t o = (size_t)bk * BC + lane;
      const double rv = b[o] - lap;
      r[o] = rv;
      rhat[o] = rv;
      s[0] = __builtin_fma(rv, rv, s[0]);
      m[0] = fmax(m[0], fabs(rv));
      if (!X0ZERO) wave_lds_sync();
    }
  }
  workgroup_reduce_store<1, false>(s, partials, 0, poff);
  workgroup_reduce_store<1, true>(m, partials, 2, poff);
}

// ---- general sparse operator (sliced ELL, ctx.h SellMatrix) --------------------------------------
// The same sweeps with y = A x taken from the assembled matrix instead of the 5-point stencil: what
// the reference does with cusparseSpMV on its COO arrays (cuda.cu:344-402).  One wave per slice (= per
// 8x8 block), lane = row; columns/values are read with unit stride, x is gathered (L2-resident on a
// block grid: a row's columns are its own block and the four blocks around it).
//   MODE 0: y = A x                                      (cup2d_apply_A)
//   MODE 1: y = A x ; partial(w.y)                       (sweep B, w = rhat)
//   MODE 2: y = A x ; partial(y.w, y.y)                  (sweep D, w = r)
//   MODE 3: r = rhat = b - A x ; partial(r.r), max|r|    (initial residual; y = r, w = b, y2 = rhat)
template <int MODE>
__global__ __launch_bounds__(WG) void k_sell(const double *__restrict__ x, double *__restrict__ y,
                                             const double *__restrict__ w, double *__restrict__ y2,
                                             const long long *__restrict__ sptr, const int32_t *__restrict__ col,
                                             const double *__restrict__ val, const int32_t *__restrict__ reg,
                                             const KrylovScalars *__restrict__ sc,
                                             double *__restrict__ partials, int count, int poff) {
  if ((MODE == 1 || MODE == 2) && sc->status != 0) return;
  const int4 *reg4 = (const int4 *)reg;
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double acc[2] = {0.0, 0.0}, mx[1] = {0.0};
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int s = g * WPG + wave;
    if (s < count) {
      const double a = sell_row(x, s, lane, sptr, col, val, reg4);
      const size_t o = (size_t)s * BC + lane;
      if (MODE == 3) {
        const double rv = w[o] - a;
        y[o] = rv;
        y2[o] = rv;
        acc[0] = __builtin_fma(rv, rv, acc[0]);
        mx[0] = fmax(mx[0], fabs(rv));
      } else {
        y[o] = a;
        if (MODE == 1) acc[0] = __builtin_fma(a, w[o], acc[0]);
        if (MODE == 2) {
          acc[0] = __builtin_fma(a, w[o], acc[0]);
          acc[1] = __builtin_fma(a, a, acc[1]);
        }
      }
    }
  }
  if (MODE == 1) {
    double a1[1] = {acc[0]};
    workgroup_reduce_store<1, false>(a1, partials, 0, poff);
  }
  if (MODE == 2) workgroup_reduce_store<2, false>(acc, partials, 0, poff);
  if (MODE == 3) {
    double a1[1] = {acc[0]};
    workgroup_reduce_store<1, false>(a1, partials, 0, poff);
    workgroup_reduce_store<1, true>(mx, partials, 2, poff);
  }
}

// send_buff_pack (cuda.cu:338-343): buf[i] = vec[idx[i]]
__global__ __launch_bounds__(WG) void k_gather(const double *__restrict__ vec, const int32_t *__restrict__ idx,
                                               double *__restrict__ buf, int n) {
  for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) buf[i] = vec[idx[i]];
}

__global__ __launch_bounds__(WG) void k_scatter(double *__restrict__ vec, const int32_t *__restrict__ idx,
                                                const double *__restrict__ buf, int n) {
  for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) vec[idx[i]] = buf[i];
}

// Halo of a Krylov vector in matrix mode (cuda.cu:356-380): gather the entries the neighbour ranks
// need, hand them to the exchange callback, which delivers the entries this rank needs straight into
// vec[m .. m+halo) (the "device_recv" argument of the callback), and wait for them.
int matrix_exchange(cup2d_ctx *c, double *vec) {
  const SellMatrix &M = c->mat;
  if (!M.active || !c->exchange || (M.ngather == 0 && M.halo == 0)) return CUP2D_OK;
  ProfScope prof(c, CUP2D_T_HALO);
  if (M.ngather > 0) {
    int grid = (M.ngather + WG - 1) / WG;
    if (grid > c->grid) grid = c->grid;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(WG), 0, c->stream, vec, M.d_gather, c->d_send, M.ngather);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  // cell plan of the matrix columns (cup2d_halo_plan_cells, adapted grids on N ranks): the gathered cells travel as cells and
  // are scattered into the vector's ghost blocks -- the columns of the rows keep their meaning
  const CellPlan &CP = c->cells[CUP2D_CELLS_MATRIX];
  if (CP.active) {
    if (!c->d_recv) { set_error("matrix_exchange: a cell plan needs the receive buffer of cup2d_set_comm"); return CUP2D_ERR_ARG; }
    if (M.ngather != CP.nsend) { set_error("matrix_exchange: the gather list (%d) is not the cell plan's send list (%d)", M.ngather, CP.nsend); return CUP2D_ERR_ARG; }
    if (c->exchange(c->comm_user, c->d_send, c->d_recv, CUP2D_CELL_STRIP(CUP2D_CELLS_MATRIX, 1), c->stream) != 0) {
      set_error("exchange callback failed (matrix cells)");
      return CUP2D_ERR_COMM;
    }
    if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
      set_error("wait callback failed");
      return CUP2D_ERR_COMM;
    }
    if (CP.nrecv > 0) {
      int g2 = (CP.nrecv + WG - 1) / WG;
      if (g2 > c->grid) g2 = c->grid;
      hipLaunchKernelGGL(k_scatter, dim3(g2), dim3(WG), 0, c->stream, vec, CP.d_recv, (const double *)c->d_recv, CP.nrecv);
      CUP2D_HIP_CHECK(hipGetLastError());
    }
    return CUP2D_OK;
  }
  // the gather list is the halo plan's send blocks, whole (adapted grids on N ranks): the message unit is a block, the
  // transports count strips of the plan; otherwise single entries (cuda.h's send_pack_idx_ protocol)
  const bool whole_blocks = c->plan.nsend > 0 && M.ngather == c->plan.nsend * BC && M.halo == c->plan.nrecv * BC;
  if (c->exchange(c->comm_user, c->d_send, vec + (size_t)c->nblocks * BC, whole_blocks ? BC : 1, c->stream) != 0) {
    set_error("exchange callback failed");
    return CUP2D_ERR_COMM;
  }
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
    set_error("wait callback failed");
    return CUP2D_ERR_COMM;
  }
  return CUP2D_OK;
}

int launch_matvec(cup2d_ctx *c, double *x, double *y) {
  const SellMatrix &M = c->mat;
  CUP2D_TRY(matrix_exchange(c, x));
  hipLaunchKernelGGL(k_sell<0>, dim3(grid_for(c, c->nblocks)), dim3(WG), 0, c->stream, x, y, nullptr, nullptr, M.d_ptr,
                     M.d_col, M.d_val, M.d_reg, nullptr, nullptr, c->nblocks, 0);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// r = rhat = b - A x over all owned blocks on the neighbour table (the fused solver's entry): inner blocks
// while the face strips of x are in flight, halo blocks after unpack (main.cpp:3035-3057).  *GP = number of
// per-workgroup partials written.
int launch_init_residual(cup2d_ctx *c, double *x, const double *b, int *GP, bool x0_zero) {
  const int nb = c->nblocks;
  if (c->mat.active) {  // assembled operator: halo entries of x, then one sweep over all slices
    const SellMatrix &M = c->mat;
    const int G = grid_for(c, nb);
    CUP2D_TRY(matrix_exchange(c, x));
    hipLaunchKernelGGL(k_sell<3>, dim3(G), dim3(WG), 0, c->stream, x, c->d_r, b, c->d_rhat, M.d_ptr, M.d_col, M.d_val, M.d_reg,
                       c->d_sc, c->d_partials, nb, 0);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = G;
    return CUP2D_OK;
  }
  const int n_in = overlapped(c) ? c->n_inner : nb, n_ha = nb - n_in;
  const int G_in = n_in > 0 ? grid_for(c, n_in) : 0, G_ha = n_ha > 0 ? grid_for(c, n_ha) : 0;
  if (x0_zero) {  // x = 0 and known to be: no exchange, no stencil; the same launches, so the same partial sums
    if (n_in > 0)
      hipLaunchKernelGGL(k_init_residual<true>, dim3(G_in), dim3(WG), 0, c->stream, x, b, c->d_r, c->d_rhat, c->d_nbr,
                         c->d_partials, 0, n_in, 0);
    if (n_ha > 0)
      hipLaunchKernelGGL(k_init_residual<true>, dim3(G_ha), dim3(WG), 0, c->stream, x, b, c->d_r, c->d_rhat, c->d_nbr,
                         c->d_partials, n_in, n_ha, G_in);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = G_in + G_ha;
    return CUP2D_OK;
  }
  CUP2D_TRY(exchange_begin(c, x, 1, 1));
  if (n_in > 0)
    hipLaunchKernelGGL(k_init_residual<false>, dim3(G_in), dim3(WG), 0, c->stream, x, b, c->d_r, c->d_rhat, c->d_nbr,
                       c->d_partials, 0, n_in, 0);
  CUP2D_TRY(exchange_end(c, x, 1, 1));
  if (n_ha > 0)
    hipLaunchKernelGGL(k_init_residual<false>, dim3(G_ha), dim3(WG), 0, c->stream, x, b, c->d_r, c->d_rhat, c->d_nbr,
                       c->d_partials, n_in, n_ha, G_in);
  CUP2D_HIP_CHECK(hipGetLastError());
  *GP = G_in + G_ha;
  return CUP2D_OK;
}

// ---- scalar kernels ---------------------------------------------------------------------------
// (the recurrences themselves: krylov_common.h)
// finish the per-workgroup partials of slots [0,nsum) (sums) and slot 2 (max) into red[0..2]
// fused_stage >= 0: also run the scalar update of that stage (single-GPU: no all-reduce in between)
__global__ __launch_bounds__(WG) void k_finish_partials(const double *__restrict__ partials, int G, int nsum,
                                                        int with_max, double *__restrict__ red, KrylovScalars *sc,
                                                        int guarded, int fused_stage, int *host_status) {
  if (guarded && sc->status != 0) return;
  finish_reduce<false>(partials, G, nsum, with_max, red, sc, fused_stage, host_status);
}

__global__ void k_scalars(KrylovScalars *sc, const double *__restrict__ red, int stage, int *host_status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (stage != 0 && sc->status != 0) {  // the solve is over; a group's last iteration still tells the host
    if ((stage == 3 || stage == 4) && host_status) __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  scalars_update(sc, red, stage);
  if ((stage == 3 || stage == 4) && host_status) __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The rank-local values in d_red[0 .. nsum) (sums) and d_red[2] (max) reduced over the ranks, then the scalar update of
// `stage` (cuda.cu:445-449, 491-493, 513-515, 533-534; the reference issues one MPI_Allreduce per quantity, four per
// iteration).  In-library communicator (comm.hip): ONE all-gather and ONE single-wave kernel per reduction point -- three
// collectives per BiCGSTAB iteration.  Callback communicators (cup2d_set_comm): an all-reduce per operator, then k_scalars.
static int reduce_over_ranks_and_update(cup2d_ctx *c, int nsum, int with_max, int stage, int *host_status) {
  if (c->rccl && c->comm_user == (void *)c->rccl) {
    if (comm_reduce_scalars(c, nsum, with_max, stage, host_status) != 0) return CUP2D_ERR_COMM;
    return CUP2D_OK;
  }
  if (c->allreduce) {
    if (nsum > 0 && c->allreduce(c->comm_user, c->d_red, nsum, 0, c->stream) != 0) return CUP2D_ERR_COMM;
    if (with_max && c->allreduce(c->comm_user, c->d_red + 2, 1, 1, c->stream) != 0) return CUP2D_ERR_COMM;
  }
  hipLaunchKernelGGL(k_scalars, dim3(1), dim3(64), 0, c->stream, c->d_sc, c->d_red, stage, host_status);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

int finish(cup2d_ctx *c, int G, int nsum, int with_max, int stage, bool guarded, int *host_status) {
  ProfScope prof(c, CUP2D_T_SCALARS);
  const bool split = c->allreduce != nullptr;  // N GPUs: local sums -> reduction over the ranks -> scalar update
  hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(WG), 0, c->stream, c->d_partials, G, nsum, with_max, c->d_red,
                     c->d_sc, guarded ? 1 : 0, split ? -1 : stage, host_status);
  CUP2D_HIP_CHECK(hipGetLastError());
  if (split) CUP2D_TRY(reduce_over_ranks_and_update(c, nsum, with_max, stage, host_status));
  return CUP2D_OK;
}

// after a sweep whose last workgroup already summed this rank's partials into d_red (krylov_fused.hip, MERGE 2):
// reduction over the ranks, then the scalar update
int finish_local(cup2d_ctx *c, int nsum, int with_max, int stage, int *host_status) {
  ProfScope prof(c, CUP2D_T_SCALARS);
  return reduce_over_ranks_and_update(c, nsum, with_max, stage, host_status);
}

// b = TMP, x0 = PRES, result -> PRES
int solve_impl(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter, int *iters,
               int *restarts, double *linf, double *linf_init) {
  c->have_last = false;  // a failed solve must not hand out the previous solve's last iterate
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
  int gridE = (int)((n + WG - 1) / WG);
  if (gridE > c->grid) gridE = c->grid;

  // y = A x over all owned blocks with the halo exchange of x overlapped: inner blocks while the
  // face strips are in flight, halo blocks after unpack (computeA's split, main.cpp:3035-3057).
  // Returns the number of per-workgroup partials the launches wrote.
  const int n_in = overlapped(c) ? c->n_inner : nb, n_ha = nb - n_in;
  const int G_in = n_in > 0 ? grid_for(c, n_in) : 0, G_ha = n_ha > 0 ? grid_for(c, n_ha) : 0;
  const bool matrix = c->mat.active;
  const SellMatrix &M = c->mat;
  auto stencil_sweep = [&](double *xin, auto launch) -> int {
    if (matrix) {  // assembled operator: halo entries first, then one sweep over all slices
      CUP2D_TRY(matrix_exchange(c, xin));
      launch(0, nb, 0, G);
      CUP2D_HIP_CHECK(hipGetLastError());
      return CUP2D_OK;
    }
    CUP2D_TRY(exchange_begin(c, xin, 1, 1));
    if (n_in > 0) launch(0, n_in, 0, G_in);
    CUP2D_TRY(exchange_end(c, xin, 1, 1));
    if (n_ha > 0) launch(n_in, n_ha, G_in, G_ha);
    CUP2D_HIP_CHECK(hipGetLastError());
    return CUP2D_OK;
  };
  const int GP = matrix ? G : G_in + G_ha;
  // finish of the fused reductions inside the sweep itself (last workgroup to arrive) instead of a
  // single-workgroup kernel: one launch over all blocks, no all-reduce in between
  const bool merge = c->finish_in_kernel && !c->allreduce && !matrix && n_ha == 0;

  {
    ProfScope prof(c, CUP2D_T_INIT_RESIDUAL);
    CUP2D_TRY(stencil_sweep(x, [&](int first, int count, int poff, int g) {
      if (matrix)
        hipLaunchKernelGGL(k_sell<3>, dim3(g), dim3(WG), 0, c->stream, x, c->d_r, b, c->d_rhat, M.d_ptr, M.d_col, M.d_val,
                           M.d_reg, c->d_sc, c->d_partials, count, poff);
      else
        hipLaunchKernelGGL(k_init_residual<false>, dim3(g), dim3(WG), 0, c->stream, x, b, c->d_r, c->d_rhat, c->d_nbr,
                           c->d_partials, first, count, poff);
    }));
  }
  CUP2D_TRY(finish(c, GP, 1, 1, 0, false));
  // p, nu start at zero (cuda.cu:436-437)
  CUP2D_TRY(launch_zero(c, c->d_p, n));
  CUP2D_TRY(launch_zero(c, c->d_nu, n));

  // The loop is driven by the device-side status word: every kernel of an iteration returns at once
  // when it is non-zero, so iterations may be enqueued speculatively.  The host never drains the
  // stream inside the loop (the reference synchronises four times PER iteration, cuda.cu:445, 491,
  // 513, 533): it stays at most AHEAD iterations in front of the GPU by waiting on the event recorded
  // AHEAD iterations ago, and learns the outcome of that iteration from a pinned word its last
  // scalar kernel wrote.  At most AHEAD iterations of early-returning kernels are wasted.
  constexpr int AHEAD = 4;
  static_assert(AHEAD <= cup2d_ctx::SOLVE_AHEAD, "event ring of the context");
  for (int i = 0; i < AHEAD; i++) c->h_status[i] = 0;
  for (int k = 0; k < max_iter; k++) {  // (exactly max_iter: iterations behind the cap could only return at once)
    const int slot = k % AHEAD;
    if (k >= AHEAD) {
      CUP2D_HIP_CHECK(hipEventSynchronize(c->solve_ev[slot]));
      if (*(volatile int *)&c->h_status[slot] != 0) break;
    }
    c->prof_sample = (k % 32 == 0) && k < max_iter;  // sampled timing (ctx.h prof_outer)
    {
      ProfScope prof(c, CUP2D_T_SWEEP_A);
      if (c->precond == PRECOND_FD)
        hipLaunchKernelGGL(k_sweepA_fd, dim3(G), dim3(WG), 0, c->stream, c->d_p, c->d_nu, c->d_r, c->d_rhat, c->d_z,
                           c->d_fd, c->d_sc, nb);
      else if (use_mfma(c))
        hipLaunchKernelGGL(k_sweepA_mfma, dim3(mfma_grid(c, nb)), dim3(WG), 0, c->stream, c->d_p, c->d_nu, c->d_r,
                           c->d_rhat, c->d_z, c->d_Pinv, c->d_sc, nb);
      else
        hipLaunchKernelGGL(k_sweepA, dim3(G), dim3(WG), 0, c->stream, c->d_p, c->d_nu, c->d_r, c->d_rhat, c->d_z,
                           c->d_Pinv, c->d_sc, nb);
    }
    {
      ProfScope prof(c, CUP2D_T_SWEEP_B);
      CUP2D_TRY(stencil_sweep(c->d_z, [&](int first, int count, int poff, int g) {
        if (matrix)
          hipLaunchKernelGGL(k_sell<1>, dim3(g), dim3(WG), 0, c->stream, c->d_z, c->d_nu, c->d_rhat, nullptr, M.d_ptr,
                             M.d_col, M.d_val, M.d_reg, c->d_sc, c->d_partials, count, poff);
        else if (merge)
          hipLaunchKernelGGL((k_sweepBD<1, true>), dim3(g), dim3(WG), 0, c->stream, c->d_z, c->d_nu, c->d_rhat, c->d_nbr,
                             c->d_sc, c->d_partials, first, count, poff, c->d_red, c->d_ticket);
        else
          hipLaunchKernelGGL((k_sweepBD<1, false>), dim3(g), dim3(WG), 0, c->stream, c->d_z, c->d_nu, c->d_rhat, c->d_nbr,
                             c->d_sc, c->d_partials, first, count, poff, c->d_red, c->d_ticket);
      }));
    }
    if (!merge) CUP2D_TRY(finish(c, GP, 1, 0, 1, true));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_C);
      if (c->precond == PRECOND_FD)
        hipLaunchKernelGGL(k_sweepC_fd, dim3(G), dim3(WG), 0, c->stream, c->d_r, c->d_nu, c->d_z2, c->d_fd, c->d_sc, nb);
      else if (use_mfma(c))
        hipLaunchKernelGGL(k_sweepC_mfma, dim3(mfma_grid(c, nb)), dim3(WG), 0, c->stream, c->d_r, c->d_nu, c->d_z2,
                           c->d_Pinv, c->d_sc, nb);
      else
        hipLaunchKernelGGL(k_sweepC, dim3(G), dim3(WG), 0, c->stream, c->d_r, c->d_nu, c->d_z2, c->d_Pinv, c->d_sc, nb);
    }
    {
      ProfScope prof(c, CUP2D_T_SWEEP_D);
      CUP2D_TRY(stencil_sweep(c->d_z2, [&](int first, int count, int poff, int g) {
        if (matrix)
          hipLaunchKernelGGL(k_sell<2>, dim3(g), dim3(WG), 0, c->stream, c->d_z2, c->d_t, c->d_r, nullptr, M.d_ptr, M.d_col,
                             M.d_val, M.d_reg, c->d_sc, c->d_partials, count, poff);
        else if (merge)
          hipLaunchKernelGGL((k_sweepBD<2, true>), dim3(g), dim3(WG), 0, c->stream, c->d_z2, c->d_t, c->d_r, c->d_nbr,
                             c->d_sc, c->d_partials, first, count, poff, c->d_red, c->d_ticket);
        else
          hipLaunchKernelGGL((k_sweepBD<2, false>), dim3(g), dim3(WG), 0, c->stream, c->d_z2, c->d_t, c->d_r, c->d_nbr,
                             c->d_sc, c->d_partials, first, count, poff, c->d_red, c->d_ticket);
      }));
    }
    if (!merge) CUP2D_TRY(finish(c, GP, 2, 0, 2, true));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_E);
      if (merge)
        hipLaunchKernelGGL(k_sweepE<true>, dim3(gridE), dim3(WG), 0, c->stream, x, c->d_xopt, c->d_z, c->d_z2, c->d_r,
                           c->d_t, c->d_rhat, c->d_sc, c->d_partials, n, c->d_red, c->d_ticket, &c->h_status[slot]);
      else
        hipLaunchKernelGGL(k_sweepE<false>, dim3(gridE), dim3(WG), 0, c->stream, x, c->d_xopt, c->d_z, c->d_z2, c->d_r,
                           c->d_t, c->d_rhat, c->d_sc, c->d_partials, n, c->d_red, c->d_ticket, &c->h_status[slot]);
    }
    CUP2D_HIP_CHECK(hipGetLastError());
    if (!merge) CUP2D_TRY(finish(c, gridE, 2, 1, 3, true, &c->h_status[slot]));
    CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
  }
  c->prof_sample = c->prof_outer;
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof init, hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->have_last = false;
  if (c->keep_last) {  // the last iterate (x is updated in place), for cup2d_solver_last_iterate
    CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_z, x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->have_last = true;
  }
  // cuda.cu:546-547: return x_opt (with the synchronisation the reference omits)
  if (!c->h_sc->x_is_best)
    CUP2D_HIP_CHECK(hipMemcpyAsync(x, c->d_xopt, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (iters) *iters = c->h_sc->iter;
  if (restarts) *restarts = c->h_sc->restarts;
  if (linf) *linf = c->h_sc->err_opt;
  if (linf_init) *linf_init = c->h_sc->err_init;
  return CUP2D_OK;
}

}  // namespace cup2d
