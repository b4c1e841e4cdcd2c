// advect.hip -- WENO5 advection + 5-point diffusion on the block grid
// and the vorticity functor (a19).
//
// Reference: KernelAdvectDiffuse::operator() main.cpp:5441-5503 driven by computeA<VectorLab>
// (main.cpp:3024-3061) and the RK2 glue main.cpp:6607-6642.
//
// Mapping.  One wavefront owns one 8x8 block (64 cells = 64 lanes).  The 14x14 ghosted tile is
// staged in LDS straight from the device-resident slabs through the neighbour table (no host-side
// BlockLab, no hash maps); the loads of the NEXT block are issued before the current one is
// computed.  The kernel is FP64-issue bound, not HBM bound, so the work is cut,
// not the bytes:
//   * a WENO face value is a function of five cells only (weno.h), so each lane evaluates the
//     `plus` and `minus` face values about ITS OWN cell once per direction and component (shared
//     smoothness indicators), the 64 rim centres of the block (centre -1 / centre 8 of every row and
//     column, both components) are spread over the 64 lanes as one extra reconstruction each, and
//     the upwind neighbour's value is picked up through LDS: 9 reconstructions per cell, no
//     divergent upwind branch, where the reference's expression costs 8 (uniform sign) to 16;
//   * in the fused modes the Runge-Kutta update is applied in the same pass, so the tmpV field and
//     the separate axpy sweep (48 B/cell) of the reference disappear.
#include <stdlib.h>

#include "advect_tile.h"
#include "block.h"
#include "weno.h"

namespace cup2d {

// registers holding one block's tile while it is in flight from HBM/L2.  Nothing here may touch
// a loaded value (not even a copy): the first use would put the s_waitcnt right behind the loads and
// turn the prefetch back into a blocking load.  The wall sign flips are applied in lab3_store.
struct LabRegs {
  double2 own, we, sn, old;
  bool flip_we, flip_sn;
};

// Branch-free on purpose: lanes 48..63 (no strip cell of their own) re-read their own cell, and a
// wave past its last block re-reads its current one, so that the loaded registers are never merged
// with other values at a control-flow join (the compiler would wait for the loads there).
template <int MODE>
static __device__ __forceinline__ void lab3_fetch(LabRegs &R, const double2 *__restrict__ f,
                                                  const double2 *__restrict__ vold, const int4 *__restrict__ nbr4,
                                                  int b, int lane) {
  const int4 nb4 = nbr4[b];  // b is wave-uniform: one scalar load
  const double2 *own = f + (size_t)b * BC;
  R.own = own[lane];
  if (MODE == 1) R.old = vold[(size_t)b * BC + lane];
  const bool strip = lane < 48;
  const int side = lane >= 24 && strip, t = strip ? lane - 24 * side : 0;
  {  // W / E strips: 8 rows x 3 columns
    const int r = t / 3, k = t - 3 * r;
    const int nb = side ? nb4.y : nb4.x;
    const int cell_nb = r * BS + (side ? k : 5 + k), cell_own = r * BS + (side ? 7 : 0);
    const double2 *src = nb >= 0 ? f + (size_t)nb * BC + cell_nb : own + cell_own;
    R.we = *src;
    R.flip_we = nb < 0;  // VectorLab::applyBCface, main.cpp:3131-3204: wall-normal component negated
  }
  {  // S / N strips: 3 rows x 8 columns
    const int j = t >> 3, x = t & 7;
    const int nb = side ? nb4.w : nb4.z;
    const int cell_nb = (side ? j : 5 + j) * BS + x, cell_own = (side ? 7 : 0) * BS + x;
    const double2 *src = nb >= 0 ? f + (size_t)nb * BC + cell_nb : own + cell_own;
    R.sn = *src;
    R.flip_sn = nb < 0;
  }
}
static __device__ __forceinline__ void lab3_store(const LabRegs &R, int lane, double2 *lab) {
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 3) * LABS + ix + 3] = R.own;
  if (lane < 48) {
    const int side = lane >= 24, t = lane - 24 * side;
    const int r = t / 3, k = t - 3 * r;
    double2 we = R.we, sn = R.sn;
    if (R.flip_we) we.x = -we.x;
    if (R.flip_sn) sn.y = -sn.y;
    lab[(r + 3) * LABS + (side ? 11 + k : k)] = we;
    const int j = t >> 3, x = t & 7;
    lab[(side ? 11 + j : j) * LABS + x + 3] = sn;
  }
}

// MODE 0: out = rhs                      (the functor alone: tmpV)
// MODE 1: out = vold + coef * rhs        (RK stage: coef = 0.5/h^2 or 1/h^2, main.cpp:6623, 6639)
template <class W, int MODE>
__global__ __launch_bounds__(WG, 4) void k_advect_diffuse(const double2 *__restrict__ vel,
                                                       const double2 *__restrict__ vold,
                                                       double2 *__restrict__ out, const int *__restrict__ nbr,
                                                       int first, int count, int chunk, double afac, double dfac, double coef) {
  __shared__ AdvectLds lds[WPG];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  AdvectLds &L = lds[wave];
  const int4 *nbr4 = (const int4 *)nbr;
  RimSlot rim;
  rim.init(L, lane);

  const GroupRange gr = chunk > 0 ? group_range_chunked(count, chunk) : group_range(count);
  int g = gr.begin;
  bool have = g < gr.end && g * WPG + wave < count;
  if (!have) return;
  LabRegs R;
  lab3_fetch<MODE>(R, vel, vold, nbr4, first + g * WPG + wave, lane);
  while (have) {
    const int b = first + g * WPG + wave;
    lab3_store(R, lane, L.lab);
    const double2 old = R.old;
    wave_lds_sync();
    // prefetch the next block of this wave (its last block is simply fetched twice)
    g += gr.stride;
    have = g < gr.end && g * WPG + wave < count;
    lab3_fetch<MODE>(R, vel, vold, nbr4, have ? first + g * WPG + wave : b, lane);
    double2 r = advect_cell<W>(L, rim, lane, afac, dfac);
    if (MODE == 1) {
      r.x = old.x + r.x * coef;
      r.y = old.y + r.y * coef;
    }
    out[(size_t)b * BC + lane] = r;
    wave_lds_sync();  // tile and face values are overwritten by the next block
  }
}

int launch_advect(cup2d_ctx *c, const double *vel, const double *vold, double *out, int mode, double nu, double dt,
                  double coef, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  const double afac = -dt * c->h, dfac = nu * dt;  // main.cpp:5446-5447
  ProfScope prof(c, CUP2D_T_ADVECT_STAGE);
  const double2 *v = (const double2 *)vel, *vo = (const double2 *)vold;
  double2 *o = (double2 *)out;
  // groups (of 4 blocks) per workgroup; CUP2D_ADVECT_CHUNK=0 selects the persistent grid
  static const int chunk = [] { const char *e = getenv("CUP2D_ADVECT_CHUNK"); return e ? atoi(e) : 16; }();
#define LAUNCH(Wt, M)                                                                                              \
  hipLaunchKernelGGL((k_advect_diffuse<Wt, M>),                                                                    \
                     dim3(chunk > 0 ? chunked_grid(count, chunk)                                                   \
                                    : resident_grid(c, reinterpret_cast<const void *>(&k_advect_diffuse<Wt, M>), count)), \
                     dim3(WG), 0, c->stream, v, vo, o, c->d_nbr, first, count, chunk, afac, dfac, coef)
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
