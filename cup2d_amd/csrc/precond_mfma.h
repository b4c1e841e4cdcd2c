// precond_mfma.h -- block-Jacobi preconditioner z_b = P_inv p_b on the FP64 matrix cores.
//
// The reference applies P_inv (64x64, symmetric) to all blocks as ONE cuBLAS DGEMM
// (64x64)*(64xNblocks) (cuda.cu:484-486, 503-505).  It is the only dense contraction on the hot path:
// 8192 flops per 1 KiB block moved (8 flop/B, just under the FP64 ridge of MI355X), so it belongs on
// v_mfma_f64_16x16x4_f64 and not on LDS-fed scalar FMAs (the first version of this file's caller was
// LDS-bandwidth bound at ~200 us per application at 4096^2).
//
// Tiling.  One wave owns a tile of 16 blocks and computes
//     Z[blk][n] = sum_k X[blk][k] * P[k][n],   blk = 0..15, n, k = 0..63
// as 4 (n-tiles) x 16 (k-steps) MFMAs D(16x16) += A(16x4) B(4x16) with
//     A[i = blk][kk]  = X[blk][4*ks + kk]          lane l holds  i = l % 16, kk = l / 16
//     B[kk][j = n]    = P[4*ks + kk][16*nt + j]    lane l holds kk = l / 16,  j = l % 16
//     D[i][j]                                      lane l, v = 0..3 holds i = (l/16) + 4*v, j = l % 16
//       (the f64 MFMA's own C/D map; NOT the 4*(l/16)+v map of the f32/bf16 16x16 shapes)
// The B operands (all of P_inv: 64 doubles per lane, 128 VGPRs) are loaded once per wave and stay in
// registers while the wave streams tiles; the A operands come straight from global memory in fragment
// layout (per k-step each lane reads one double; a tile's 8 KiB are consumed completely).
#pragma once
#include <hip/hip_runtime.h>

#include "ctx.h"

namespace cup2d {

typedef double v4f64 __attribute__((ext_vector_type(4)));

struct PinvFragments {
  double b[16][4];  // [k-step][n-tile]
  __device__ __forceinline__ void load(const double *__restrict__ Pinv, int lane) {
    const int kk = lane >> 4, j = lane & 15;
#pragma unroll
    for (int ks = 0; ks < 16; ks++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) b[ks][nt] = Pinv[(4 * ks + kk) * BC + 16 * nt + j];
  }
};

// element offset inside a tile of 16 blocks of the A operand lane `lane` supplies at k-step ks
static __device__ __forceinline__ int a_offset(int lane, int ks) { return (lane & 15) * BC + 4 * ks + (lane >> 4); }

// acc[nt] (+)= X * P for one tile; xa[ks] = this lane's A operands
static __device__ __forceinline__ void precond_tile(const double (&xa)[16], const PinvFragments &P, v4f64 (&acc)[4]) {
#pragma unroll
  for (int nt = 0; nt < 4; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 16; ks++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], P.b[ks][nt], acc[nt], 0, 0, 0);
}

// store the D fragments of a tile: lane l, v -> block (l/16)+4*v, cell 16*nt + l%16
static __device__ __forceinline__ void store_tile(double *__restrict__ z, size_t tile_base, int nvalid, int lane,
                                                  const v4f64 (&acc)[4]) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int blk = q + 4 * v;
    if (blk < nvalid) {
#pragma unroll
      for (int nt = 0; nt < 4; nt++) z[tile_base + (size_t)blk * BC + 16 * nt + j] = acc[nt][v];
    }
  }
}

// tiles of 16 blocks distributed over the waves of a persistent grid, contiguous per XCD
struct TileRange {
  int begin, end, stride;
};
static __device__ __forceinline__ TileRange tile_range(int ntiles) {
  const int G = gridDim.x, w = blockIdx.x, wave = threadIdx.x >> 6;
  TileRange r;
  if (G >= 8 && (G % 8) == 0) {
    const int xcd = w & 7, slot = w >> 3, per = G >> 3;
    const long long lo = (long long)ntiles * xcd / 8, hi = (long long)ntiles * (xcd + 1) / 8;
    r.begin = (int)lo + slot * WPG + wave;
    r.end = (int)hi;
    r.stride = per * WPG;
  } else {
    r.begin = w * WPG + wave;
    r.end = ntiles;
    r.stride = G * WPG;
  }
  return r;
}

}  // namespace cup2d
