// pressure.hip -- the 5-point (halo 1) block operators around the Poisson solve
//  and the field-wide scalar reductions.
//
// Reference functors: pressure_rhs main.cpp:6105-6139, pressure_rhs1 main.cpp:6209-6230,
// pressureCorrectionKernel main.cpp:6021-6043, glue main.cpp:7016-7021, 7120-7187,
// dt main.cpp:6579-6595.  One wavefront per 8x8 block, 10x10 ghosted tile in LDS.
#include "block.h"

namespace cup2d {

// tmp = facDiv*(div vel) - facDiv*chi*(div udef)   [BODIES]
//       - Lap5(pold)                               [SUBLAP: pressure_rhs1 fused, main.cpp:7026]
//       pold_copy (SUBLAP): the block's own cells of pold are written there as well -- cup2d_step hands PRES in as
//       pold and POLD as the copy: `pold = pres` (main.cpp:7016-7021) without a pass of its own
template <bool BODIES, bool SUBLAP>
__global__ __launch_bounds__(WG) void k_pressure_rhs(const double2 *__restrict__ vel, const double2 *__restrict__ udef,
                                                     const double *__restrict__ chi, const double *__restrict__ pold,
                                                     double *__restrict__ out, const int *__restrict__ nbr, int first,
                                                     int count, double facDiv, double *__restrict__ pold_copy) {
  __shared__ double2 vlabs[WPG][LAB1 * LAB1];
  __shared__ double slabs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *vlab = vlabs[wave];
  double *slab = slabs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_vector_lab1(vel, nbr, b, lane, vlab);
      if (SUBLAP) load_scalar_lab1(pold, nbr, b, lane, slab);
      wave_lds_sync();
      // main.cpp:6133-6135
      double r = facDiv * (vlab[c0 + 1].x - vlab[c0 - 1].x + vlab[c0 + LAB1].y - vlab[c0 - LAB1].y);
      if (SUBLAP) {  // main.cpp:6228
        const double l0 = slab[c0], l1 = slab[c0 - 1], l2 = slab[c0 + 1], l3 = slab[c0 - LAB1], l4 = slab[c0 + LAB1];
        if (pold_copy) pold_copy[(size_t)b * BC + lane] = l0;
        if (!BODIES) r -= l1 + l2 + l3 + l4 - 4 * l0;
        else {
          // with bodies the chi term must be subtracted before the Laplacian (statement order
          // of main.cpp:6133-6138 then 6228)
          wave_lds_sync();
          load_vector_lab1(udef, nbr, b, lane, vlab);
          wave_lds_sync();
          r = r - facDiv * chi[(size_t)b * BC + lane] *
                      (vlab[c0 + 1].x - vlab[c0 - 1].x + vlab[c0 + LAB1].y - vlab[c0 - LAB1].y);
          r -= l1 + l2 + l3 + l4 - 4 * l0;
        }
      } else if (BODIES) {
        wave_lds_sync();
        load_vector_lab1(udef, nbr, b, lane, vlab);
        wave_lds_sync();
        r = r - facDiv * chi[(size_t)b * BC + lane] *
                    (vlab[c0 + 1].x - vlab[c0 - 1].x + vlab[c0 + LAB1].y - vlab[c0 - LAB1].y);
      }
      out[(size_t)b * BC + lane] = r;
      wave_lds_sync();
    }
  }
}

int launch_pressure_rhs(cup2d_ctx *c, const double *vel, const double *udef, const double *chi, const double *pold,
                        double *out, double dt, int first, int count, double *pold_copy) {
  if (count <= 0) return CUP2D_OK;
  const double facDiv = 0.5 * c->h / dt;  // main.cpp:6117
  const int grid = grid_for(c, count);
  const bool bodies = udef != nullptr, sub = pold != nullptr;
  ProfScope prof(c, CUP2D_T_POISSON_RHS);
#define LAUNCH(B, S)                                                                                              \
  hipLaunchKernelGGL((k_pressure_rhs<B, S>), dim3(grid), dim3(WG), 0, c->stream, (const double2 *)vel,             \
                     (const double2 *)udef, chi, pold, out, c->d_nbr, first, count, facDiv, sub ? pold_copy : nullptr)
  if (bodies && sub) LAUNCH(true, true);
  else if (bodies) LAUNCH(true, false);
  else if (sub) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// SUB = false: y  = Lap5(x)   (the assembled matrix of main.cpp:7034-7112 on a same-level grid)
// SUB = true : y -= Lap5(x)   (pressure_rhs1, main.cpp:6209-6230)
template <bool SUB>
__global__ __launch_bounds__(WG) void k_laplacian(const double *__restrict__ x, double *__restrict__ y,
                                                  const int *__restrict__ nbr, int first, int count) {
  __shared__ double slabs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *slab = slabs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_scalar_lab1(x, nbr, b, lane, slab);
      wave_lds_sync();
      const double l0 = slab[c0], l1 = slab[c0 - 1], l2 = slab[c0 + 1], l3 = slab[c0 - LAB1], l4 = slab[c0 + LAB1];
      const double lap = l1 + l2 + l3 + l4 - 4 * l0;
      const size_t o = (size_t)b * BC + lane;
      if (SUB) y[o] -= lap; else y[o] = lap;
      wave_lds_sync();
    }
  }
}

int launch_laplacian(cup2d_ctx *c, const double *x, double *y, int subtract, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  const int grid = grid_for(c, count);
  if (subtract)
    hipLaunchKernelGGL(k_laplacian<true>, dim3(grid), dim3(WG), 0, c->stream, x, y, c->d_nbr, first, count);
  else
    hipLaunchKernelGGL(k_laplacian<false>, dim3(grid), dim3(WG), 0, c->stream, x, y, c->d_nbr, first, count);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// UPDATE = false: tmpV = pFac * grad(pres)                         (main.cpp:6038-6041)
// UPDATE = true : vel += (pFac * grad(pres)) * ih2                 (+ main.cpp:7180-7187), tmpV untouched
// SHIFT: the tile is read as pres + pold - shift (mean removal main.cpp:7165-7172 fused in)
template <bool UPDATE>
__global__ __launch_bounds__(WG) void k_pressure_correction(const double *__restrict__ pres, double2 *__restrict__ tmpV,
                                                            double2 *__restrict__ vel, const int *__restrict__ nbr,
                                                            int first, int count, double pFac, double ih2,
                                                            double *__restrict__ umax_partials) {
  __shared__ double slabs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *slab = slabs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 1) * LAB1 + ix + 1;
  double m[1] = {0.0};  // UPDATE: max|u| of the velocity written (the next step's dt, ctx.h umax_partials)
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_scalar_lab1(pres, nbr, b, lane, slab);
      wave_lds_sync();
      double2 t;
      t.x = pFac * (slab[c0 + 1] - slab[c0 - 1]);
      t.y = pFac * (slab[c0 + LAB1] - slab[c0 - LAB1]);
      const size_t o = (size_t)b * BC + lane;
      if (UPDATE) {
        double2 v = vel[o];
        v.x += t.x * ih2;
        v.y += t.y * ih2;
        vel[o] = v;
        m[0] = fmax(m[0], fmax(fabs(v.x), fabs(v.y)));
      } else {
        tmpV[o] = t;
      }
      wave_lds_sync();
    }
  }
  if (UPDATE && umax_partials) workgroup_reduce_store<1, true>(m, umax_partials, 3);
}

int launch_pressure_correction(cup2d_ctx *c, const double *pres, double *tmpV, double *vel, double dt, int fused_update,
                               int first, int count, double *umax_partials) {
  if (count <= 0) return CUP2D_OK;
  const double pFac = -0.5 * dt * c->h;  // main.cpp:6027
  const double ih2 = 1.0 / c->h / c->h;  // main.cpp:7182
  const int grid = grid_for(c, count);
  if (fused_update)
    hipLaunchKernelGGL(k_pressure_correction<true>, dim3(grid), dim3(WG), 0, c->stream, pres, (double2 *)tmpV,
                       (double2 *)vel, c->d_nbr, first, count, pFac, ih2, umax_partials);
  else
    hipLaunchKernelGGL(k_pressure_correction<false>, dim3(grid), dim3(WG), 0, c->stream, pres, (double2 *)tmpV,
                       (double2 *)vel, c->d_nbr, first, count, pFac, ih2, nullptr);
  CUP2D_HIP_CHECK(hipGetLastError());
  if (fused_update && umax_partials) c->umax_partials = grid;
  return CUP2D_OK;
}

// y += a * x over n doubles (main.cpp:7180-7187 with a = 1/h/h)
__global__ __launch_bounds__(WG) void k_axpy(double *__restrict__ y, const double *__restrict__ x, double a, size_t n2) {
  double2 *y2 = (double2 *)y;
  const double2 *x2 = (const double2 *)x;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    double2 v = y2[i];
    const double2 w = x2[i];
    v.x += w.x * a;
    v.y += w.y * a;
    y2[i] = v;
  }
}
int launch_axpy_field(cup2d_ctx *c, double *y, const double *x, double a, size_t n) {
  if (n == 0) return CUP2D_OK;
  size_t n2 = n / 2;  // slabs are multiples of 64 doubles
  int grid = (int)((n2 + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  hipLaunchKernelGGL(k_axpy, dim3(grid), dim3(WG), 0, c->stream, y, x, a, n2);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// v = 0 over n doubles.  hipMemsetAsync moves 134 MB in ~73 us (1.8 TB/s, several fill kernels per call); a vector
// is zeroed four times per step (pres, p, nu, y).  16-byte stores, non-temporal: nothing reads the zeros soon.
__global__ __launch_bounds__(WG) void k_zero(double *__restrict__ v, size_t n2) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d *v2 = (v2d *)v;
  const v2d z = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) __builtin_nontemporal_store(z, v2 + i);
}
int launch_zero(cup2d_ctx *c, double *v, size_t n) {
  if (n == 0) return CUP2D_OK;
  size_t n2 = n / 2;  // slabs are multiples of 64 doubles
  int grid = (int)((n2 + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  hipLaunchKernelGGL(k_zero, dim3(grid), dim3(WG), 0, c->stream, v, n2);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// ---- field-wide reductions: partial per workgroup, finished by one workgroup ----------------
// op 0: sum(v)  1: max|v|
template <int OP>
__global__ __launch_bounds__(WG) void k_reduce_partial(const double *__restrict__ v, size_t n2,
                                                       double *__restrict__ partials) {
  const double2 *v2 = (const double2 *)v;
  double acc[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    const double2 w = v2[i];
    if (OP == 0) acc[0] += w.x + w.y;
    else acc[0] = fmax(acc[0], fmax(fabs(w.x), fabs(w.y)));
  }
  workgroup_reduce_store<1, OP == 1>(acc, partials, 0);
}
template <int OP>
__global__ __launch_bounds__(WG) void k_reduce_final(const double *__restrict__ partials, int n, double *__restrict__ out) {
  __shared__ double red[WG];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += WG) a = OP == 0 ? a + partials[i] : fmax(a, partials[i]);
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = WG / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = OP == 0 ? red[threadIdx.x] + red[threadIdx.x + s]
                                                         : fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
static int reduce_field(cup2d_ctx *c, const double *v, size_t n, int op, double *d_out) {
  size_t n2 = n / 2;
  int grid = (int)((n2 + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  if (grid < 1) grid = 1;
  ProfScope prof(c, CUP2D_T_REDUCE);
  if (op == 0) {
    hipLaunchKernelGGL(k_reduce_partial<0>, dim3(grid), dim3(WG), 0, c->stream, v, n2, c->d_partials);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(WG), 0, c->stream, c->d_partials, grid, d_out);
  } else {
    hipLaunchKernelGGL(k_reduce_partial<1>, dim3(grid), dim3(WG), 0, c->stream, v, n2, c->d_partials);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(WG), 0, c->stream, c->d_partials, grid, d_out);
  }
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
int launch_max_abs(cup2d_ctx *c, const double *v, size_t n, double *d_out) { return reduce_field(c, v, n, 1, d_out); }
// max over n per-workgroup maxima (slot 3 of d_partials after the projection of a cup2d_step)
int launch_max_from_partials(cup2d_ctx *c, const double *partials, int n, double *d_out) {
  ProfScope prof(c, CUP2D_T_REDUCE);
  hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(WG), 0, c->stream, partials, n, d_out);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// out[b] = max |f| over the 64 cells of block b (the per-block Linf norm the reference tags blocks by, main.cpp:4671-4690)
__global__ __launch_bounds__(WG) void k_block_linf(const double *__restrict__ f, double *__restrict__ out, int nblocks) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = blockIdx.x * WPG + wave; b < nblocks; b += gridDim.x * WPG) {
    const double m = wave_max(fabs(f[(size_t)b * BC + lane]));
    if (lane == 0) out[b] = m;
  }
}
int launch_block_linf(cup2d_ctx *c, const double *f, double *d_out) {
  hipLaunchKernelGGL(k_block_linf, dim3(grid_for(c, c->nblocks)), dim3(WG), 0, c->stream, f, d_out, c->nblocks);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// out = (a - shift[0]*s0) (+ b - shift2...) helpers for the mean removal, main.cpp:7120-7173
// MODE 0: p = p - mean0                    (7143-7148)   mean0 = red[0] / ncells_total
// MODE 1: p = p + pold - mean1             (7166-7172)
// MODE 0 also leaves the per-workgroup partial sums of the shifted field -- the second mean (7148-7158) -- in the loop
// structure of k_reduce_partial<0> (same grid, same order: the sum is the one that kernel would compute, bit for bit)
template <int MODE>
__global__ __launch_bounds__(WG) void k_shift(double *__restrict__ p, const double *__restrict__ pold,
                                              const double *__restrict__ red, double inv_cells, size_t n2,
                                              double *__restrict__ partials) {
  double2 *p2 = (double2 *)p;
  const double2 *q2 = (const double2 *)pold;
  const double avg = red[0] * inv_cells;
  double acc[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    double2 v = p2[i];
    if (MODE == 0) {
      v.x += -avg;
      v.y += -avg;
      acc[0] += v.x + v.y;
    } else {
      const double2 q = q2[i];
      v.x += q.x - avg;
      v.y += q.y - avg;
    }
    p2[i] = v;
  }
  if (MODE == 0) workgroup_reduce_store<1, false>(acc, partials, 0);
}

// main.cpp:7120-7187.  With uniform h the h^2 weights cancel: avg = sum(p)/ncells.  The global cell
// count comes from the allreduce of {sum, count} exactly like quantities[2] (main.cpp:7136-7141).
int project_impl(cup2d_ctx *c, double dt) {
  const size_t n = (size_t)c->nblocks * BC;
  const size_t n2 = n / 2;
  int grid = (int)((n2 + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  double *pres = c->d_field[CUP2D_PRES];
  double *pold = c->d_field[CUP2D_POLD];
  double cells = (double)n;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 0) {
      CUP2D_TRY(reduce_field(c, pres, n, 0, c->d_red));
    } else {  // the partial sums of the shifted field are there (k_shift<0>, same grid as reduce_field's)
      ProfScope prof(c, CUP2D_T_REDUCE);
      hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(WG), 0, c->stream, c->d_partials, grid, c->d_red);
      CUP2D_HIP_CHECK(hipGetLastError());
    }
    if (c->allreduce) {
      // second entry carries the cell count (main.cpp:7137: quantities[1] = avg1)
      CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_red + 1, &cells, sizeof(double), hipMemcpyHostToDevice, c->stream));
      if (c->allreduce(c->comm_user, c->d_red, 2, 0, c->stream) != 0) {
        set_error("allreduce callback failed");
        return CUP2D_ERR_COMM;
      }
      CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_red, c->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
      cells = c->h_red[1];
    }
    const double inv = 1.0 / cells;
    if (pass == 0)
      hipLaunchKernelGGL(k_shift<0>, dim3(grid), dim3(WG), 0, c->stream, pres, pold, c->d_red, inv, n2, c->d_partials);
    else
      hipLaunchKernelGGL(k_shift<1>, dim3(grid), dim3(WG), 0, c->stream, pres, pold, c->d_red, inv, n2, c->d_partials);
    CUP2D_HIP_CHECK(hipGetLastError());
    cells = (double)n;
  }
  CUP2D_TRY(exchange_halo(c, pres, 1, 1));
  return launch_pressure_correction(c, pres, nullptr, c->d_field[CUP2D_VEL], dt, 1, 0, c->nblocks, c->d_partials);
}

}  // namespace cup2d
