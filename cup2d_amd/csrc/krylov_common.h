// krylov_common.h -- device-side pieces shared by the five-sweep solver (krylov.hip) and the tile-fused
// solver (krylov_fused.hip): the scalar recurrences of BiCGSTABSolver::main (cuda.cu:303-330, 440-545)
// and the finish of a fused reduction.
#pragma once
#include <hip/hip_runtime.h>

#include "block.h"
#include "krylov_scalars.h"

namespace cup2d {

constexpr int RED_REC = 8;  // doubles per rank in a gathered reduction record (= the size of d_red)
// {sum, sum, max} (or up to RED_REC sums) of every rank's record, in RANK ORDER: a summation order that does not depend on
// the transport, the same on every rank (comm.hip k_gather_scalars; krylov_edge.h MERGE 3)
static __device__ __forceinline__ void sum_records(const double *__restrict__ g, int nranks, int nsum, int with_max, double (&v)[RED_REC]) {
  for (int k = 0; k < RED_REC; k++) v[k] = 0.0;
  for (int r = 0; r < nranks; r++) {
    for (int k = 0; k < nsum && k < RED_REC; k++)
      if (!(with_max && k == 2)) v[k] += g[RED_REC * r + k];
    if (with_max) v[2] = fmax(v[2], g[RED_REC * r + 2]);
  }
}

// One row of the hybrid sliced-ELL operator (ctx.h SellMatrix): slice s (= block; wave-uniform), lane = row.  Shared by
// k_sell (krylov.hip) and k_hyb_rows (krylov_fused.hip).
// (sell_row_at: the slice's entry range and neighbour record given -- k_hyb_rows reads them from one record per list entry)
static __device__ __forceinline__ double sell_row_at(const double *__restrict__ x, int s, int lane, long long base, int width, int4 rg,
                                                     const int32_t *__restrict__ col, const double *__restrict__ val) {
  const int32_t *cp = col + base + lane;
  const double *vp = val + base + lane;
  double a = 0.0;
  int k = 0;
  if (rg.x != SELL_STORED) {
    // a slice of plain same-level rows (ctx.h SellMatrix::d_reg): the 5-point sum straight from x, ghost = own
    // cell at a wall (the row has no entry there and one neighbour less on the diagonal: the same number)
    const int ix = lane & 7, iy = lane >> 3;
    const double *own = x + (size_t)s * BC;
    const double l0 = own[lane];
    const double l1 = ix > 0 ? own[lane - 1] : rg.x >= 0 ? x[(size_t)rg.x * BC + iy * BS + (BS - 1)] : l0;
    const double l2 = ix < BS - 1 ? own[lane + 1] : rg.y >= 0 ? x[(size_t)rg.y * BC + iy * BS] : l0;
    const double l3 = iy > 0 ? own[lane - BS] : rg.z >= 0 ? x[(size_t)rg.z * BC + (BS - 1) * BS + ix] : l0;
    const double l4 = iy < BS - 1 ? own[lane + BS] : rg.w >= 0 ? x[(size_t)rg.w * BC + ix] : l0;
    a = l1 + l2 + l3 + l4 - 4 * l0;
  }
  for (; k + 4 <= width; k += 4) {  // four independent gathers in flight
    const int c0 = cp[(k + 0) * 64], c1 = cp[(k + 1) * 64], c2 = cp[(k + 2) * 64], c3 = cp[(k + 3) * 64];
    const double v0 = vp[(k + 0) * 64], v1 = vp[(k + 1) * 64], v2 = vp[(k + 2) * 64], v3 = vp[(k + 3) * 64];
    const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
    a = __builtin_fma(v0, x0, a);
    a = __builtin_fma(v1, x1, a);
    a = __builtin_fma(v2, x2, a);
    a = __builtin_fma(v3, x3, a);
  }
  for (; k < width; k++) a = __builtin_fma(vp[k * 64], x[cp[k * 64]], a);
  return a;
}
static __device__ __forceinline__ double sell_row(const double *__restrict__ x, int s, int lane,
                                                  const long long *__restrict__ sptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const int4 *__restrict__ reg4) {
  const long long base = sptr[s];
  const int width = (int)((sptr[s + 1] - base) >> 6);
  return sell_row_at(x, s, lane, base, width, reg4[s], col, val);  // (reg4[s]: wave-uniform)
}

// Finish of a fused reduction by ONE workgroup: sums the per-workgroup partials of slots [0,nsum) and
// takes the max of slot 2, in a fixed order (thread t takes partials t, t+256, ...; then a binary tree),
// so the result does not depend on which workgroup runs it.  fused_stage >= 0: also runs the scalar
// update of that stage.  COHERENT: read the partials with agent-scope loads (the caller is a workgroup of
// the SAME launch that produced them, see arrive_last).
// EXT: the 3 x WG doubles of scratch are the caller's (k_fused: its dynamic LDS is dead by then and the launch has no
// 6 KB of static LDS to spare next to 158 KB of tiles and fragments).
template <bool COHERENT, bool EXT = false>
static __device__ __forceinline__ void finish_reduce(const double *partials, int G, int nsum, int with_max,
                                                     double *red, KrylovScalars *sc, int fused_stage,
                                                     int *host_status, double *ext = nullptr) {
  __shared__ double sm_own[EXT ? 1 : 3][EXT ? 1 : WG];
  double (*sm)[WG] = EXT ? reinterpret_cast<double (*)[WG]>(ext) : reinterpret_cast<double (*)[WG]>(&sm_own[0][0]);
  const auto ld = [&](const double *p) -> double {
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
  };
  if (threadIdx.x < WG) {  // the first 256 threads (callers may run wider workgroups): the order is fixed
    double a0 = 0, a1 = 0, mx = 0;
    for (int i = threadIdx.x; i < G; i += WG) {
      a0 += ld(partials + i);
      if (nsum > 1) a1 += ld(partials + PSTRIDE + i);
      if (with_max) mx = fmax(mx, ld(partials + 2 * PSTRIDE + i));
    }
    sm[0][threadIdx.x] = a0; sm[1][threadIdx.x] = a1; sm[2][threadIdx.x] = mx;
  }
  __syncthreads();
  for (int s = WG / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
      sm[2][threadIdx.x] = fmax(sm[2][threadIdx.x], sm[2][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    red[0] = sm[0][0]; red[1] = sm[1][0]; red[2] = sm[2][0];
    if (fused_stage >= 0) {
      const double loc[3] = {sm[0][0], sm[1][0], sm[2][0]};
      scalars_update(sc, loc, fused_stage);
      // end of an iteration: tell the host (pinned, device-visible word) whether the loop is over
      if ((fused_stage == 3 || fused_stage == 4) && host_status)
        __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same for NS sums and no maximum (k_edge MODE 3: five), scratch = the caller's dead dynamic LDS; partials read with
// agent-scope loads.
template <int NS>
static __device__ __forceinline__ void finish_reduce_n(const double *partials, int G, double *red, KrylovScalars *sc,
                                                       int fused_stage, double *ext) {
  double (*sm)[WG] = reinterpret_cast<double (*)[WG]>(ext);
  if (threadIdx.x < WG) {
    double a[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) a[k] = 0.0;
    for (int i = threadIdx.x; i < G; i += WG)
#pragma unroll
      for (int k = 0; k < NS; k++) a[k] += __hip_atomic_load(partials + (size_t)k * PSTRIDE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < NS; k++) sm[k][threadIdx.x] = a[k];
  }
  __syncthreads();
  for (int s = WG / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < NS; k++) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double loc[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) loc[k] = sm[k][0];
    for (int k = 0; k < NS; k++) red[k] = loc[k];  // (red holds eight doubles)
    if (fused_stage >= 0) scalars_update(sc, loc, fused_stage);
  }
}

// "The last workgroup to arrive finishes the reduction" -- removes the single-workgroup finish kernel
// (8.6 us + a kernel boundary, three times per BiCGSTAB iteration).  Every workgroup of the launch calls
// this once after it has stored its partials (workgroup_reduce_store with COHERENT = true: agent-scope
// stores); exactly one call per launch returns true, in all threads of that workgroup, and only after
// every other workgroup's partials are visible to agent-scope loads.  The counter is re-armed to zero by
// the last arriver, so consecutive launches on one stream can share it.
// Ordering (MI355X_MICROARCH.md, "handoff-flag", drained sc1 form): the partials are agent-scope
// (write-through, sc1) stores issued by wave 0 -> __syncthreads -> lane 0 drains its wave's stores with an
// explicit vmcnt(0) -> relaxed agent atomic ticket; the last arriver reads the partials with agent-scope
// (sc1) loads after an agent acquire.  NO agent-scope RELEASE fence: it lowers to buffer_wbl2, a write-back
// of the XCD's whole L2, and 2048 workgroups issuing one each at the end of a streaming sweep doubled the
// sweep's duration (measured: sweep B 103 -> 205 us at 4096^2).
static __device__ __forceinline__ bool arrive_last(unsigned *counter) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == gridDim.x - 1;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

}  // namespace cup2d
