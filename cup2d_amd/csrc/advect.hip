// advect.hip -- WENO5 advection + 5-point diffusion on the block grid
// and the vorticity functor (a19).
//
// Reference: KernelAdvectDiffuse::operator() main.cpp:5441-5503 driven by computeA<VectorLab>
// (main.cpp:3024-3061) and the RK2 glue main.cpp:6607-6642.  Here one wavefront owns one 8x8
// block (64 cells = 64 lanes), stages the 14x14 ghosted tile in LDS straight from the
// device-resident slabs through the neighbour table (no host-side BlockLab, no hash maps),
// and -- in the fused modes -- applies the Runge-Kutta update in the same pass, so the tmpV
// field and the separate axpy sweep (48 B/cell) of the reference disappear.
#include "block.h"
#include "weno.h"

namespace cup2d {

// MODE 0: out = rhs                      (the functor alone: tmpV)
// MODE 1: out = vold + coef * rhs        (RK stage: coef = 0.5/h^2 or 1/h^2, main.cpp:6623, 6639)
template <class W, int MODE>
__global__ __launch_bounds__(WG) void k_advect_diffuse(const double2 *__restrict__ vel,
                                                       const double2 *__restrict__ vold,
                                                       double2 *__restrict__ out, const int *__restrict__ nbr,
                                                       int first, int count, double afac, double dfac, double coef) {
  __shared__ double2 labs[WPG][LAB3 * LAB3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *lab = labs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_vector_lab3(vel, nbr, b, lane, lab);
      wave_lds_sync();
      double2 xs[7], ys[7];
#pragma unroll
      for (int k = 0; k < 7; k++) {
        xs[k] = lab[(iy + 3) * LAB3 + ix + k];
        ys[k] = lab[(iy + k) * LAB3 + ix + 3];
      }
      const double u = xs[3].x, v = xs[3].y;
      // main.cpp:5493-5496: d/dx upwinds on u, d/dy upwinds on v
      const double dudx = W::derivative(u, xs[0].x, xs[1].x, xs[2].x, u, xs[4].x, xs[5].x, xs[6].x);
      const double dudy = W::derivative(v, ys[0].x, ys[1].x, ys[2].x, u, ys[4].x, ys[5].x, ys[6].x);
      const double dvdx = W::derivative(u, xs[0].y, xs[1].y, xs[2].y, v, xs[4].y, xs[5].y, xs[6].y);
      const double dvdy = W::derivative(v, ys[0].y, ys[1].y, ys[2].y, v, ys[4].y, ys[5].y, ys[6].y);
      // main.cpp:5497-5502, same operand order
      double2 r;
      r.x = afac * (u * dudx + v * dudy) + dfac * (xs[4].x + xs[2].x + ys[4].x + ys[2].x - 4 * u);
      r.y = afac * (u * dvdx + v * dvdy) + dfac * (xs[4].y + xs[2].y + ys[4].y + ys[2].y - 4 * v);
      if (MODE == 1) {
        const double2 o = vold[(size_t)b * BC + lane];
        r.x = o.x + r.x * coef;
        r.y = o.y + r.y * coef;
      }
      out[(size_t)b * BC + lane] = r;
      wave_lds_sync();  // tile is overwritten by the next group
    }
  }
}

int launch_advect(cup2d_ctx *c, const double *vel, const double *vold, double *out, int mode, double nu, double dt,
                  double coef, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  const double afac = -dt * c->h, dfac = nu * dt;  // main.cpp:5446-5447
  const int grid = grid_for(c, count);
  ProfScope prof(c, CUP2D_T_ADVECT_STAGE);
  const double2 *v = (const double2 *)vel, *vo = (const double2 *)vold;
  double2 *o = (double2 *)out;
#define LAUNCH(Wt, M)                                                                                         \
  hipLaunchKernelGGL((k_advect_diffuse<Wt, M>), dim3(grid), dim3(WG), 0, c->stream, v, vo, o, c->d_nbr, first, \
                     count, afac, dfac, coef)
  if (c->math == CUP2D_MATH_STRICT) {
    if (mode == 0) LAUNCH(WenoStrict, 0); else LAUNCH(WenoStrict, 1);
  } else {
    if (mode == 0) LAUNCH(WenoFast, 0); else LAUNCH(WenoFast, 1);
  }
#undef LAUNCH
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// KernelVorticity main.cpp:3343-3366: tmp = (0.5/h) * (u_S - u_N + v_E - v_W)
__global__ __launch_bounds__(WG) void k_vorticity(const double2 *__restrict__ vel, double *__restrict__ out,
                                                  const int *__restrict__ nbr, int first, int count, double i2h) {
  __shared__ double2 labs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *lab = labs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_vector_lab1(vel, nbr, b, lane, lab);
      wave_lds_sync();
      const int c0 = (iy + 1) * LAB1 + ix + 1;
      const double e0 = lab[c0 - LAB1].x, e1 = lab[c0 + LAB1].x, e2 = lab[c0 + 1].y, e3 = lab[c0 - 1].y;
      out[(size_t)b * BC + lane] = i2h * (e0 - e1 + e2 - e3);
      wave_lds_sync();
    }
  }
}

int launch_vorticity(cup2d_ctx *c, const double *vel, double *out, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  hipLaunchKernelGGL(k_vorticity, dim3(grid_for(c, count)), dim3(WG), 0, c->stream, (const double2 *)vel, out,
                     c->d_nbr, first, count, 0.5 / c->h);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

}  // namespace cup2d
