// block.h -- device-side helpers: work distribution over 8x8 blocks (one block = one wave64),
// ghosted tiles ("labs") staged in LDS, wave/workgroup reductions.
#pragma once
#include <hip/hip_runtime.h>

#include "ctx.h"
#include "ranges.h"

namespace cup2d {

// ---- work distribution ---------------------------------------------------------------------
// `count` blocks are processed in groups of WPG (one per wave).  The persistent grid hands each
// XCD (workgroup id % 8, MI355X_MICROARCH.md "block b runs on XCD b % 8") a contiguous range of
// groups, so the blocks whose ghost cells a wave reads were fetched by a neighbour on the SAME
// L2.  Purely a speed choice: any placement computes the same result.
// (ranges.h: host+device, replayed on the CPU by tests/walk_emul.cpp)
static __device__ __forceinline__ GroupRange group_range(int count) {
  return group_range_of(count, WPG, gridDim.x, blockIdx.x);
}
// Chunked variant for the FP64-issue-bound kernels: the grid has MORE workgroups than fit on the chip
// and each one walks `chunk` consecutive groups (stride 1) of its XCD's contiguous range.  VALU issue
// is arbitrated oldest-wave-first, so with one long-lived wave per slot the oldest wave of a SIMD races
// ahead and the youngest finishes alone at a third of the issue rate (measured: average wave lifetime
// 72 % of the kernel time); short-lived workgroups are replaced as they retire, which keeps four waves
// per SIMD until the end.  launch with chunked_grid().
static __device__ __forceinline__ GroupRange group_range_chunked(int count, int chunk) {
  return group_range_chunked_of(count, WPG, chunk, blockIdx.x);
}
static __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic between lanes of ONE wave needs no s_barrier: a wave's DS operations execute in
// order.  This keeps the compiler from moving accesses across the hand-off.
static __device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- ghosted tiles --------------------------------------------------------------------------
// Vector lab of halo width 3 (KernelAdvectDiffuse's Stencil{-3,-3,4,4}, main.cpp:5442): the
// 14x14 tile BlockLab::load assembles (main.cpp:2270-2687), cross only (the functor never reads
// corners).  Wall ghosts follow VectorLab::applyBCface (main.cpp:3131-3204): every ghost layer
// repeats the edge cell with the wall-normal component negated.
constexpr int LAB3 = BS + 6;
static __device__ __forceinline__ void load_vector_lab3(const double2 *__restrict__ f, const int *__restrict__ nbr,
                                                        int b, int lane, double2 *lab) {
  const int nW = uniform(nbr[4 * b + 0]), nE = uniform(nbr[4 * b + 1]);
  const int nS = uniform(nbr[4 * b + 2]), nN = uniform(nbr[4 * b + 3]);
  const double2 *own = f + (size_t)b * BC;
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 3) * LAB3 + ix + 3] = own[lane];
  if (lane < 48) {
    const int side = lane >= 24, t = lane - 24 * side;
    {  // W / E strips: 8 rows x 3 columns
      const int r = t / 3, k = t - 3 * r;
      const int nb = side ? nE : nW;
      double2 v;
      if (nb >= 0) {
        v = f[(size_t)nb * BC + r * BS + (side ? k : 5 + k)];
      } else {
        v = own[r * BS + (side ? 7 : 0)];
        v.x = -v.x;
      }
      lab[(r + 3) * LAB3 + (side ? 11 + k : k)] = v;
    }
    {  // S / N strips: 3 rows x 8 columns
      const int j = t >> 3, x = t & 7;
      const int nb = side ? nN : nS;
      double2 v;
      if (nb >= 0) {
        v = f[(size_t)nb * BC + (side ? j : 5 + j) * BS + x];
      } else {
        v = own[(side ? 7 : 0) * BS + x];
        v.y = -v.y;
      }
      lab[(side ? 11 + j : j) * LAB3 + x + 3] = v;
    }
  }
}

// Halo-1 labs (Stencil{-1,-1,2,2}): 10x10 tile, cross only.
constexpr int LAB1 = BS + 2;
// ghost source for lane 0..31: side 0..3 = W,E,S,N, position 0..7 along the face
static __device__ __forceinline__ void lab1_slot(int lane, int &side, int &src_cell, int &edge_cell, int &lab_idx) {
  // branch-free (selects, no switch): the callers keep the result in registers across their block loop
  side = lane >> 3;
  const int q = lane & 7;
  const bool horiz = side < 2, hi = side & 1;
  src_cell = horiz ? q * BS + (hi ? 0 : 7) : (hi ? q : 7 * BS + q);
  edge_cell = horiz ? q * BS + (hi ? 7 : 0) : (hi ? 7 * BS + q : q);
  lab_idx = horiz ? (q + 1) * LAB1 + (hi ? 9 : 0) : (hi ? 9 * LAB1 : 0) + q + 1;
}
// scalar, Neumann wall (ScalarLab::Neumann2D, main.cpp:3210-3255): ghost = edge cell
static __device__ __forceinline__ void load_scalar_lab1(const double *__restrict__ f, const int *__restrict__ nbr,
                                                        int b, int lane, double *lab) {
  const double *own = f + (size_t)b * BC;
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 1) * LAB1 + ix + 1] = own[lane];
  if (lane < 32) {
    int side, src, edge, li;
    lab1_slot(lane, side, src, edge, li);
    const int nb = nbr[4 * b + side];
    lab[li] = nb >= 0 ? f[(size_t)nb * BC + src] : own[edge];
  }
}
// the same tile split into "issue the global loads" and "write them to LDS", so that a persistent
// wave can have its NEXT block in flight while it works on the current one
struct ScalarLab1Regs {
  double own, ghost;
};
static __device__ __forceinline__ ScalarLab1Regs fetch_scalar_lab1(const double *__restrict__ f,
                                                                   const int4 *__restrict__ nbr4, int b, int lane) {
  // branch-free (lanes 32..63 re-read their own cell): a loaded register that is merged with another
  // value at a control-flow join makes the compiler wait for the load right there
  ScalarLab1Regs R;
  const int4 nb4 = nbr4[b];  // b is wave-uniform: one scalar load
  const double *own = f + (size_t)b * BC;
  R.own = own[lane];
  int side, src, edge, li;
  lab1_slot(lane & 31, side, src, edge, li);
  const int nb = side == 0 ? nb4.x : side == 1 ? nb4.y : side == 2 ? nb4.z : nb4.w;
  const double *g = lane < 32 ? (nb >= 0 ? f + (size_t)nb * BC + src : own + edge) : own + lane;
  R.ghost = *g;
  return R;
}
static __device__ __forceinline__ void store_scalar_lab1(const ScalarLab1Regs &R, int lane, double *lab) {
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 1) * LAB1 + ix + 1] = R.own;
  if (lane < 32) {
    int side, src, edge, li;
    lab1_slot(lane, side, src, edge, li);
    lab[li] = R.ghost;
  }
}
// vector, free-slip wall
static __device__ __forceinline__ void load_vector_lab1(const double2 *__restrict__ f, const int *__restrict__ nbr,
                                                        int b, int lane, double2 *lab) {
  const double2 *own = f + (size_t)b * BC;
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 1) * LAB1 + ix + 1] = own[lane];
  if (lane < 32) {
    int side, src, edge, li;
    lab1_slot(lane, side, src, edge, li);
    const int nb = nbr[4 * b + side];
    double2 v;
    if (nb >= 0) {
      v = f[(size_t)nb * BC + src];
    } else {
      v = own[edge];
      if (side < 2) v.x = -v.x; else v.y = -v.y;
    }
    lab[li] = v;
  }
}

// ---- reductions -----------------------------------------------------------------------------
static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}
static __device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// one partial per workgroup and slot, written to partials[slot*PSTRIDE + poff + blockIdx.x]; order
// of accumulation is fixed by the launch geometry, so results are run-to-run reproducible
// COHERENT: agent-scope (write-through) stores, for partials that a workgroup of the SAME launch reads
// back (krylov_common.h arrive_last)
template <int N, bool IS_MAX, bool COHERENT = false>
static __device__ __forceinline__ void workgroup_reduce_store(double (&v)[N], double *__restrict__ partials,
                                                              int slot0, int poff = 0) {
  __shared__ double red[N][WPG];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double w = IS_MAX ? wave_max(v[i]) : wave_sum(v[i]);
    if (lane == 0) red[i][wave] = w;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double a = red[threadIdx.x][0];
    for (int k = 1; k < WPG; k++) a = IS_MAX ? fmax(a, red[threadIdx.x][k]) : a + red[threadIdx.x][k];
    double *dst = partials + (size_t)(slot0 + threadIdx.x) * PSTRIDE + poff + blockIdx.x;
    if (COHERENT) __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = a;
  }
}

}  // namespace cup2d
