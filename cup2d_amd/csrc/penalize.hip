// penalize.hip -- Brinkman penalisation with host-supplied bodies (main.cpp:6643-7006).
//
// The reference keeps, per shape and per block the shape touches, an Obstacle with the shape's own indicator chi[8][8]
// and deformation velocity udef[8][8][2] (main.cpp:3283-3286, filled on the host by the shape model -- fish midlines,
// out of scope here) and per time step
//   (1) integrates seven moments of the penalisation force over the shape's cells and solves a 3 x 3 system for the
//       rigid-body velocities u, v, omega                                              main.cpp:6643-6702
//   (2) blends the fluid velocity towards the body velocity where the shape's chi dominates   main.cpp:6944-6978
//   (3) sets tmpV = sum of udef of the shapes that dominate a cell (the u_def of pressure_rhs) main.cpp:6979-7006
// (collisions between shapes, main.cpp:6703-6943, are host logic on a handful of scalars and stay with the caller).
// Here the bodies' block lists live on the device next to the fields; (2) and (3) are one kernel launch per body;
// for (1) a kernel writes the seven integrands of every cell of the body's blocks and a one-wave kernel adds them up in the
// reference's order (block by block, row by row, one serial chain per integral: k_body_sums) -- a few hundred blocks per
// body, and the sums, the 3 x 3 LU solve and therefore u, v, omega are bit-identical to the reference's single-threaded
// loop; seven doubles per body cross PCIe (round 2 downloaded every integrand and summed on the host).  No FMA contraction in this translation unit (-ffp-contract=off): every expression below keeps
// the reference's operation order.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "block.h"

namespace cup2d {

struct Body {
  int nblk = 0;
  int32_t *d_blocks = nullptr;  // [nblk] block index of the context, ascending
  double *d_origin = nullptr;   // [nblk][2] Info::origin of those blocks
  double *d_chi = nullptr;      // [nblk][64]
  double *d_udef = nullptr;     // [nblk][64][2]
  double *d_terms = nullptr;    // [nblk][64][7] integrands of (1)
  double cx = 0, cy = 0;
  void release() {
    (void)hipFree(d_blocks); (void)hipFree(d_origin); (void)hipFree(d_chi); (void)hipFree(d_udef); (void)hipFree(d_terms);
    *this = Body();
  }
};
struct Bodies {
  std::vector<Body> list;
};

static __device__ __forceinline__ double block_h(double h_uniform, double h0, const int32_t *__restrict__ level, int b) {
  return level ? h0 / (double)(1 << level[b]) : h_uniform;
}

// (1) integrands PM, PJ, PX, PY, UM, VM, AM of one cell (main.cpp:6659-6679); zero where the shape's chi <= 0
__global__ __launch_bounds__(WG) void k_body_terms(const double2 *__restrict__ vel, const int32_t *__restrict__ blocks,
                                                   const double *__restrict__ origin, const double *__restrict__ chi,
                                                   const double2 *__restrict__ udef, double *__restrict__ terms, int nblk,
                                                   double cx, double cy, double lambdt, double h_uniform, double h0,
                                                   const int32_t *__restrict__ level) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * WPG + (threadIdx.x >> 6); k < nblk; k += gridDim.x * WPG) {
    const int b = blocks[k];
    const double h = block_h(h_uniform, h0, level, b), hsq = h * h;
    const int ix = lane & 7, iy = lane >> 3;
    const double X = chi[(size_t)k * BC + lane];
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    if (X > 0) {
      const double2 V = vel[(size_t)b * BC + lane], U = udef[(size_t)k * BC + lane];
      const double ud0 = V.x - U.x, ud1 = V.y - U.y;
      const double Xlamdt = X >= 0.5 ? lambdt : 0.0;
      const double F = hsq * Xlamdt / (1 + Xlamdt);
      double p0 = origin[2 * k] + h * (ix + 0.5), p1 = origin[2 * k + 1] + h * (iy + 0.5);
      p0 -= cx;
      p1 -= cy;
      t[0] = F;
      t[1] = F * (p0 * p0 + p1 * p1);
      t[2] = F * p0;
      t[3] = F * p1;
      t[4] = F * ud0;
      t[5] = F * ud1;
      t[6] = F * (p0 * ud1 - p1 * ud0);
    }
    double *dst = terms + ((size_t)k * BC + lane) * 7;
#pragma unroll
    for (int q = 0; q < 7; q++) dst[q] = t[q];
  }
}

// (1), the sums: the seven integrals of one body in the REFERENCE'S order -- block by block, row by row, one serial chain of
// additions per integral (main.cpp:6649-6680) -- on the device.  One wave: a block's 64 x 7 integrands are staged in LDS with
// coalesced loads (the next block's are in flight meanwhile), lanes 0..6 each add up one integral over the 64 cells in cell
// order.  The order, and with it every bit of the sums, is the reference's single-threaded loop's; seven doubles leave the
// device instead of seven per cell of every body block.
__global__ __launch_bounds__(64) void k_body_sums(const double *__restrict__ terms, int nblk, double *__restrict__ sums) {
  __shared__ double t[BC * 7];
  const int lane = threadIdx.x;
  double acc = 0.0, nxt[7];
#pragma unroll
  for (int q = 0; q < 7; q++) nxt[q] = nblk > 0 ? terms[q * BC + lane] : 0.0;
  for (int k = 0; k < nblk; k++) {
#pragma unroll
    for (int q = 0; q < 7; q++) t[q * BC + lane] = nxt[q];  // t[j] = integrand (7 i + quantity) of cell i = j / 7
    wave_lds_sync();
    const int kn = k + 1 < nblk ? k + 1 : k;
#pragma unroll
    for (int q = 0; q < 7; q++) nxt[q] = terms[(size_t)kn * BC * 7 + q * BC + lane];
    if (lane < 7)
      for (int i = 0; i < BC; i++) acc += t[7 * i + lane];
    wave_lds_sync();
  }
  if (lane < 7) sums[lane] = acc;
}

// (2) main.cpp:6944-6978 for one body: V = alpha V + (1 - alpha) (u_s - omega p_y + udef_x, v_s + omega p_x + udef_y)
__global__ __launch_bounds__(WG) void k_body_blend(double2 *__restrict__ vel, const double *__restrict__ CHI,
                                                   const int32_t *__restrict__ blocks, const double *__restrict__ origin,
                                                   const double *__restrict__ chi, const double2 *__restrict__ udef, int nblk,
                                                   double cx, double cy, double us, double vs, double omega, double lambdt,
                                                   double h_uniform, double h0, const int32_t *__restrict__ level) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * WPG + (threadIdx.x >> 6); k < nblk; k += gridDim.x * WPG) {
    const int b = blocks[k];
    const double X = chi[(size_t)k * BC + lane];
    if (CHI[(size_t)b * BC + lane] > X) continue;
    if (X <= 0) continue;
    const double h = block_h(h_uniform, h0, level, b);
    const int ix = lane & 7, iy = lane >> 3;
    double p0 = origin[2 * k] + h * (ix + 0.5), p1 = origin[2 * k + 1] + h * (iy + 0.5);
    p0 -= cx;
    p1 -= cy;
    const double alpha = X > 0.5 ? 1 / (1 + lambdt) : 1;
    const double2 U = udef[(size_t)k * BC + lane];
    const double US = us - omega * p1 + U.x, VS = vs + omega * p0 + U.y;
    double2 V = vel[(size_t)b * BC + lane];
    V.x = alpha * V.x + (1 - alpha) * US;
    V.y = alpha * V.y + (1 - alpha) * VS;
    vel[(size_t)b * BC + lane] = V;
  }
}

// (3) main.cpp:6984-7006 for one body: tmpV += udef where the body's chi is not below the field's
__global__ __launch_bounds__(WG) void k_body_udef(double2 *__restrict__ tmpV, const double *__restrict__ CHI,
                                                  const int32_t *__restrict__ blocks, const double *__restrict__ chi,
                                                  const double2 *__restrict__ udef, int nblk) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * WPG + (threadIdx.x >> 6); k < nblk; k += gridDim.x * WPG) {
    const int b = blocks[k];
    if (chi[(size_t)k * BC + lane] < CHI[(size_t)b * BC + lane]) continue;
    double2 T = tmpV[(size_t)b * BC + lane];
    const double2 U = udef[(size_t)k * BC + lane];
    T.x += U.x;
    T.y += U.y;
    tmpV[(size_t)b * BC + lane] = T;
  }
}

// the reference's gsl_linalg_LU_decomp + gsl_linalg_LU_solve on its 3 x 3 system (main.cpp:6692-6703): Gaussian
// elimination with partial pivoting (first largest pivot), forward and back substitution, in that operation order
static void lu_solve3(double A[3][3], const double b[3], double x[3]) {
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < 3; k++) {
    int piv = k;
    double best = fabs(A[k][k]);
    for (int i = k + 1; i < 3; i++)
      if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 3; j++) std::swap(A[k][j], A[piv][j]);
      std::swap(perm[k], perm[piv]);
    }
    for (int i = k + 1; i < 3; i++) {
      A[i][k] /= A[k][k];
      for (int j = k + 1; j < 3; j++) A[i][j] -= A[i][k] * A[k][j];
    }
  }
  for (int i = 0; i < 3; i++) x[i] = b[perm[i]];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < i; j++) x[i] -= A[i][j] * x[j];
  for (int i = 2; i >= 0; i--) {
    for (int j = i + 1; j < 3; j++) x[i] -= A[i][j] * x[j];
    x[i] /= A[i][i];
  }
}

void bodies_release(cup2d_ctx *c) {
  if (!c->bodies) return;
  for (Body &B : c->bodies->list) B.release();
  delete c->bodies;
  c->bodies = nullptr;
}

static int body_grid(const cup2d_ctx *c, int nblk) {
  int g = (nblk + WPG - 1) / WPG;
  return g < 1 ? 1 : (g > c->grid ? c->grid : g);
}

}  // namespace cup2d

using namespace cup2d;

extern "C" {

int cup2d_body_clear(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  bodies_release(c);
  return CUP2D_OK;
}

int cup2d_body_set(cup2d_ctx *c, int body, int nblk, const int32_t *blocks, const double *origin, const double *chi,
                   const double *udef, double cx, double cy) {
  CUP2D_CHECK_CTX(c);
  if (body < 0 || body > 1023 || nblk < 0 || (nblk && (!blocks || !origin || !chi || !udef))) { set_error("body_set: bad argument"); return CUP2D_ERR_ARG; }
  for (int k = 0; k < nblk; k++)
    if (blocks[k] < 0 || blocks[k] >= c->nblocks || (k && blocks[k] <= blocks[k - 1])) {
      set_error("body_set: blocks[%d] = %d: owned block indices in ascending order expected (the order of the reference's block loop)", k, blocks[k]);
      return CUP2D_ERR_ARG;
    }
  if (!c->bodies) c->bodies = new Bodies;
  if ((size_t)body >= c->bodies->list.size()) c->bodies->list.resize(body + 1);
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  Body &B = c->bodies->list[body];
  B.release();
  B.nblk = nblk;
  B.cx = cx;
  B.cy = cy;
  if (nblk == 0) return CUP2D_OK;
  const size_t n = (size_t)nblk;
  CUP2D_HIP_CHECK(hipMalloc(&B.d_blocks, n * sizeof(int32_t)));
  CUP2D_HIP_CHECK(hipMalloc(&B.d_origin, n * 2 * sizeof(double)));
  CUP2D_HIP_CHECK(hipMalloc(&B.d_chi, n * BC * sizeof(double)));
  CUP2D_HIP_CHECK(hipMalloc(&B.d_udef, n * BC * 2 * sizeof(double)));
  CUP2D_HIP_CHECK(hipMalloc(&B.d_terms, n * BC * 7 * sizeof(double)));
  CUP2D_HIP_CHECK(hipMemcpy(B.d_blocks, blocks, n * sizeof(int32_t), hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(B.d_origin, origin, n * 2 * sizeof(double), hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(B.d_chi, chi, n * BC * sizeof(double), hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(B.d_udef, udef, n * BC * 2 * sizeof(double), hipMemcpyHostToDevice));
  return CUP2D_OK;
}

int cup2d_body_momentum(cup2d_ctx *c, int body, double lambda, double dt, double *uvw, double *integrals) {
  CUP2D_CHECK_CTX(c);
  if (!c->bodies || body < 0 || (size_t)body >= c->bodies->list.size() || !uvw) { set_error("body_momentum: no such body"); return CUP2D_ERR_ARG; }
  Body &B = c->bodies->list[body];
  double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (B.nblk > 0) {
    hipLaunchKernelGGL(k_body_terms, dim3(body_grid(c, B.nblk)), dim3(WG), 0, c->stream, (const double2 *)c->d_field[CUP2D_VEL], B.d_blocks,
                       B.d_origin, B.d_chi, (const double2 *)B.d_udef, B.d_terms, B.nblk, B.cx, B.cy, lambda * dt, c->h, c->amr.h0,
                       c->amr.active ? c->amr.d_level : nullptr);
    CUP2D_HIP_CHECK(hipGetLastError());
    // (the seven sums in the reference's order -- block by block, iy, ix, main.cpp:6649-6680 -- by one wave on the device)
    hipLaunchKernelGGL(k_body_sums, dim3(1), dim3(64), 0, c->stream, (const double *)B.d_terms, B.nblk, c->d_red);
    CUP2D_HIP_CHECK(hipGetLastError());
  } else {
    CUP2D_HIP_CHECK(hipMemsetAsync(c->d_red, 0, 7 * sizeof(double), c->stream));
  }
  // MPI_Allreduce of the seven sums (main.cpp:6682-6684): straight from the device buffer they were formed in
  if (c->allreduce && c->allreduce(c->comm_user, c->d_red, 7, 0, c->stream) != 0) return CUP2D_ERR_COMM;
  CUP2D_HIP_CHECK(hipMemcpyAsync(q, c->d_red, 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  const double PM = q[0], PJ = q[1], PX = q[2], PY = q[3];
  double A[3][3] = {{PM, 0, -PY}, {0, PM, PX}, {-PY, PX, PJ}};
  const double b[3] = {q[4], q[5], q[6]};
  lu_solve3(A, b, uvw);
  if (integrals)
    for (int k = 0; k < 7; k++) integrals[k] = q[k];
  return CUP2D_OK;
}

int cup2d_penalize(cup2d_ctx *c, double lambda, double dt, const double *uvw) {
  CUP2D_CHECK_CTX(c);
  const size_t nbodies = c->bodies ? c->bodies->list.size() : 0;
  if (nbodies && !uvw) { set_error("penalize: body velocities expected"); return CUP2D_ERR_ARG; }
  const int32_t *level = c->amr.active ? c->amr.d_level : nullptr;
  for (size_t s = 0; s < nbodies; s++) {
    const Body &B = c->bodies->list[s];
    if (B.nblk == 0) continue;
    hipLaunchKernelGGL(k_body_blend, dim3(body_grid(c, B.nblk)), dim3(WG), 0, c->stream, (double2 *)c->d_field[CUP2D_VEL],
                       c->d_field[CUP2D_CHI], B.d_blocks, B.d_origin, B.d_chi, (const double2 *)B.d_udef, B.nblk, B.cx, B.cy, uvw[3 * s],
                       uvw[3 * s + 1], uvw[3 * s + 2], lambda * dt, c->h, c->amr.h0, level);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  CUP2D_TRY(launch_zero(c, c->d_field[CUP2D_TMPV], (size_t)c->ntotal * BC * 2));  // main.cpp:6980-6983
  for (size_t s = 0; s < nbodies; s++) {
    const Body &B = c->bodies->list[s];
    if (B.nblk == 0) continue;
    hipLaunchKernelGGL(k_body_udef, dim3(body_grid(c, B.nblk)), dim3(WG), 0, c->stream, (double2 *)c->d_field[CUP2D_TMPV],
                       c->d_field[CUP2D_CHI], B.d_blocks, B.d_chi, (const double2 *)B.d_udef, B.nblk);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  return CUP2D_OK;
}

}  // extern "C"
