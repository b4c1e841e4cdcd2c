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

// the topology tables of cup2d_set_amr on the device (host pointers for the host-side uses of these forms)
struct AmrDev {
  const int32_t *kind, *nbr2, *half, *level;
  double *faces;
  double h0;
};

// ---- halo 3: the tile of KernelAdvectDiffuse (Stencil{-3,-3,4,4,true}, main.cpp:5442) -----------------------------
// Closed forms of the CROSS ghosts of BlockLab::load/post_load for this stencil (use_averages = true), derived from
// and pinned against a literal transcription of the reference (kept with the tests) and the reference's own tiles:
//   wall / same / finer as for halo 1, three layers; on W/E faces the first of every four rows of a finer neighbour
//   pairs fine rows 0 and 2 (main.cpp:2528-2531)
//   coarser, layers 1-2: the halo-1 tangential quadratic, then LI (layer 1) / LE (layer 2) with the two interior cells
//   coarser, layer 3   : TestInterp (main.cpp:2219-2230) on the 3x3 coarse cells around the ghost cell -- read from
//                        COMPONENT 0 for both components (main.cpp:2753-2763 passes `Test` without the component
//                        offset; parity is with the reference as it is).  The coarse cell one step beyond the outer end
//                        of the block's span lies across the coarse neighbour's tangential side: a wall (mirrored;
//                        x-walls negate component 0), a block of the coarse level (its cell) or of this level (2x2 mean).
static __host__ __device__ __forceinline__ double amr_LE(double a, double b, double c) {  // main.cpp:2211-2218
  const double kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b;
  const double lambda = (b - c) - kappa;
  return (9.0 * kappa + 3.0 * lambda) + c;
}

// tangential quadratic through the four coarse cells cc[] of the span at fine position q (main.cpp:2797-2846)
static __host__ __device__ __forceinline__ double amr_tangential(const double (&cc)[4], int q) {
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
  return (q & 1) ? c1 - dy * d1 + (0.5 * dy * dy) * d2 : c1 + dy * d1 + (0.5 * dy * dy) * d2;
}

// ghost (side s, layer k = 0 nearest, position q) of block b of a vector field; F(block, cell) reads one double2 of it --
// from memory in the kernel (amr.hip AmrField2), into a cell mask on the host (amr_host.hip cup2d_amr_trace_reads: which
// cells of which other blocks the tile of a block is made of = what an N-rank exchange has to deliver)
template <class F2>
static __host__ __device__ double2 amr_ghost3(F2 F, const AmrDev &T, int b, int s, int k, int q, double2 e1, double2 e2) {
  const int kind = T.kind[4 * b + s];
  const int n0 = T.nbr2[(4 * b + s) * 2], n1 = T.nbr2[(4 * b + s) * 2 + 1];
  double2 r;
  if (kind == AMR_WALL) {  // VectorLab::applyBCface main.cpp:3131-3204
    r.x = s < 2 ? -e1.x : e1.x;
    r.y = s < 2 ? e1.y : -e1.y;
    return r;
  }
  if (kind == AMR_SAME) {
    const int cell = s == 0 ? q * BS + 7 - k : s == 1 ? q * BS + k : s == 2 ? (7 - k) * BS + q : k * BS + q;
    return F(n0, cell);
  }
  if (kind == AMR_FINE) {
    const int a = q >> 2, t = q & 3;
    const int fb = a ? n1 : n0;
    double2 q00, q10, q01, q11;
    if (s < 2) {
      const int c0 = s == 1 ? 2 * k : 6 - 2 * k;
      const int y0 = 2 * t, y1 = t == 0 ? 2 : 2 * t + 1;
      q00 = F(fb, y0 * BS + c0); q10 = F(fb, y1 * BS + c0); q01 = F(fb, y0 * BS + c0 + 1); q11 = F(fb, y1 * BS + c0 + 1);
    } else {
      const int y = s == 3 ? 2 * k : 6 - 2 * k;
      q00 = F(fb, y * BS + 2 * t); q10 = F(fb, (y + 1) * BS + 2 * t); q01 = F(fb, y * BS + 2 * t + 1); q11 = F(fb, (y + 1) * BS + 2 * t + 1);
    }
    r.x = (q00.x + q10.x + q01.x + q11.x) / 4;
    r.y = (q00.y + q10.y + q01.y + q11.y) / 4;
    return r;
  }
  // ---- coarser neighbour ----
  const int half = T.half[4 * b + s];
  if (k < 2) {
    double ccx[4], ccy[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int jj = 4 * half + j;
      const double2 v = F(n0, s == 0 ? jj * BS + 7 : s == 1 ? jj * BS : s == 2 ? 7 * BS + jj : jj);
      ccx[j] = v.x;
      ccy[j] = v.y;
    }
    const double tx = amr_tangential(ccx, q), ty = amr_tangential(ccy, q);
    if (k == 0) { r.x = amr_LI(tx, e1.x, e2.x); r.y = amr_LI(ty, e1.y, e2.y); }
    else { r.x = amr_LE(tx, e1.x, e2.x); r.y = amr_LE(ty, e1.y, e2.y); }
    return r;
  }
  // layer 3: 3x3 coarse cells (component 0) around (Xc, Yc) in the coarse neighbour's cell coordinates
  const int tside = s < 2 ? (half == 0 ? 2 : 3) : (half == 0 ? 0 : 1);  // the outer end of the span
  const int ekind = T.kind[4 * n0 + tside];
  const int en0 = T.nbr2[(4 * n0 + tside) * 2], en1 = T.nbr2[(4 * n0 + tside) * 2 + 1];
  const auto coarse0 = [&](int X, int Y) -> double {
    if (X >= 0 && X < BS && Y >= 0 && Y < BS) return F(n0, Y * BS + X).x;
    if (ekind == AMR_WALL) {
      if (s < 2) return F(n0, (Y < 0 ? 0 : Y > 7 ? 7 : Y) * BS + X).x;  // y-wall: component 0 copied
      return -F(n0, Y * BS + (X < 0 ? 0 : X > 7 ? 7 : X)).x;            // x-wall: component 0 negated
    }
    if (ekind == AMR_SAME) return F(en0, ((Y + BS) % BS) * BS + ((X + BS) % BS)).x;
    if (ekind == AMR_FINE) {  // blocks of this block's level: FillCoarseVersion's 2x2 mean (main.cpp:2958-2996)
      if (s < 2) {
        const int eb = s == 0 ? en1 : en0;
        const int Xc = X - (s == 0 ? 4 : 0), y0 = Y < 0 ? 6 : 0;
        return (F(eb, y0 * BS + 2 * Xc).x + F(eb, (y0 + 1) * BS + 2 * Xc).x + F(eb, y0 * BS + 2 * Xc + 1).x + F(eb, (y0 + 1) * BS + 2 * Xc + 1).x) / 4;
      }
      const int eb = s == 2 ? en1 : en0;
      const int Yc = Y - (s == 2 ? 4 : 0), x0 = X < 0 ? 6 : 0;
      return (F(eb, 2 * Yc * BS + x0).x + F(eb, (2 * Yc + 1) * BS + x0).x + F(eb, 2 * Yc * BS + x0 + 1).x + F(eb, (2 * Yc + 1) * BS + x0 + 1).x) / 4;
    }
    return 0.0;  // two levels coarser across the corner: the reference reads an unset cell there
  };
  int Xc, Yc;
  double dx, dy;
  if (s < 2) {
    Xc = s == 1 ? 1 : 6;
    Yc = 4 * half + (q >> 1);
    dx = s == 1 ? -0.25 : 0.25;
    dy = 0.25 * (2 * (q & 1) - 1);
  } else {
    Yc = s == 3 ? 1 : 6;
    Xc = 4 * half + (q >> 1);
    dy = s == 3 ? -0.25 : 0.25;
    dx = 0.25 * (2 * (q & 1) - 1);
  }
  double C[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i][j] = coarse0(Xc - 1 + i, Yc - 1 + j);
  const double dudx = 0.5 * (C[2][1] - C[0][1]);
  const double dudy = 0.5 * (C[1][2] - C[1][0]);
  const double dudxdy = 0.25 * ((C[0][0] + C[2][2]) - (C[2][0] + C[0][2]));
  const double dudx2 = (C[0][1] + C[2][1]) - 2.0 * C[1][1];
  const double dudy2 = (C[1][0] + C[1][2]) - 2.0 * C[1][1];
  r.x = (C[1][1] + (dx * dudx + dy * dudy)) + (((0.5 * dx * dx) * dudx2 + (0.5 * dy * dy) * dudy2) + (dx * dy) * dudxdy);
  r.y = r.x;
  return r;
}

}  // namespace cup2d
