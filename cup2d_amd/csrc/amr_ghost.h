// amr_ghost.h -- the ghost cell across one side of a block of a block-AMR grid (halo 1), shared by the HIP kernels
// (amr.hip) and the regrid-time host code (amr_host.hip), so that both produce the same bits.
//
// Reference: BlockLab::load / post_load (main.cpp:2270-2687, 2689-2933) for a halo-1 stencil, LI (2203-2210).
#pragma once
#include "ctx.h"

namespace cup2d {

enum { AMR_WALL = 0, AMR_SAME = 1, AMR_COARSE = 2, AMR_FINE = 3 };

// main.cpp:2203-2210
static __host__ __device__ __forceinline__ double amr_LI(double a, double b, double c) {
  const double kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b;
  const double lambda = (b - c) - kappa;
  return (4.0 * kappa + 2.0 * lambda) + c;
}

// ghost value at position q of side s of a block; get(block, cell) reads one component of the field,
// e1 / e2 = the block's own edge cell and the next cell inwards at that position
template <class Get>
static __host__ __device__ __forceinline__ double amr_ghost(Get get, int kind, int n0, int n1, int half, int s, int q, double e1,
                                                   double e2, double wall_sign) {
  if (kind == AMR_WALL) return wall_sign * e1;
  if (kind == AMR_SAME) {
    const int cell = s == 0 ? q * BS + 7 : s == 1 ? q * BS : s == 2 ? 7 * BS + q : q;
    return get(n0, cell);
  }
  if (kind == AMR_FINE) {
    const int a = q >> 2, ee = q & 3;
    const int fb = a ? n1 : n0;
    if (s >= 2) {  // rows y, y+1 of the fine block, columns 2ee, 2ee+1 (main.cpp:2548-2564)
      const int r0 = s == 2 ? 6 : 0;
      return (get(fb, r0 * BS + 2 * ee) + get(fb, (r0 + 1) * BS + 2 * ee) + get(fb, r0 * BS + 2 * ee + 1) +
              get(fb, (r0 + 1) * BS + 2 * ee + 1)) / 4;
    }
    const int x = s == 0 ? 6 : 0;
    const int y0 = 2 * ee, y1 = ee == 0 ? 2 : 2 * ee + 1;  // ee == 0: rows 0 and 2 (main.cpp:2528-2529)
    return (get(fb, y0 * BS + x) + get(fb, y1 * BS + x) + get(fb, y0 * BS + x + 1) + get(fb, y1 * BS + x + 1)) / 4;
  }
  // coarser neighbour: its four cells along the face that span this block
  double cc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int j = 4 * half + k;
    cc[k] = get(n0, s == 0 ? j * BS + 7 : s == 1 ? j * BS : s == 2 ? 7 * BS + j : j);
  }
  const int qq = q >> 1;
  const double c1 = cc[qq];
  double d1, d2;
  if (qq == 0) {
    d1 = (-0.5 * cc[2] - 1.5 * cc[0]) + 2.0 * cc[1];
    d2 = (cc[2] + cc[0]) - 2.0 * cc[1];
  } else if (qq == 3) {
    d1 = (0.5 * cc[1] + 1.5 * cc[3]) - 2.0 * cc[2];
    d2 = (cc[1] + cc[3]) - 2.0 * cc[2];
  } else {
    d1 = 0.5 * (cc[qq + 1] - cc[qq - 1]);
    d2 = (cc[qq + 1] + cc[qq - 1]) - 2.0 * cc[qq];
  }
  const double dy = -0.25;
  const double t = (q & 1) ? c1 - dy * d1 + (0.5 * dy * dy) * d2 : c1 + dy * d1 + (0.5 * dy * dy) * d2;
  return amr_LI(t, e1, e2);
}

}  // namespace cup2d
