// advect_tile.h -- the WENO5 advect-diffuse functor on ONE ghosted tile held in LDS: shared by the uniform-grid
// kernel (advect.hip, tile assembled from the neighbour table) and the block-AMR kernel (amr.hip, tile assembled
// with the reference's coarse-fine interpolation).  Reference: KernelAdvectDiffuse::operator() main.cpp:5441-5503.
#pragma once
#include "block.h"
#include "weno.h"

namespace cup2d {

// per-wave LDS: the ghosted tile + the face values handed between lanes
//   Fx[c][iy][0..8]: index k holds plus about centre k-1 (k = 0..8)   [x direction, component c]
//   Gx[c][iy][0..8]: index k holds minus about centre k   (k = 0..8)
//   Fy/Gy[c][0..8][ix]: the same along y
// The tile rows are LABS = 24 double2 apart, not 14: a ds_read_b128 is serviced in four fixed groups of
// 16 lanes (MI355X_MICROARCH.md, LDS) and with a 14-slot row stride three lanes of a group share a
// 16-byte slot (3 LDS cycles per group instead of 1); with 24 every group covers all 16 slots once.
constexpr int LABS = 24;
constexpr int FROW = 9;
struct AdvectLds {
  double2 lab[LAB3 * LABS];
  double Fx[2][BS * FROW], Gx[2][BS * FROW];
  double Fy[2][FROW * BS], Gy[2][FROW * BS];
};

// rim assignment of a lane: direction, component, side (0: plus about centre -1, 1: minus about centre 8),
// position along the face -- one extra reconstruction per lane covers the block's 64 rim centres
struct RimSlot {
  int rdir, rcomp, rside, rcen, rstep;
  double *rdst;
  __device__ __forceinline__ void init(AdvectLds &L, int lane) {
    rdir = lane >> 5; rcomp = (lane >> 4) & 1; rside = (lane >> 3) & 1;
    const int rpos = lane & 7;
    const int rc = rside ? 11 : 2;                       // tile coordinate of the rim centre
    rstep = (rside ? -1 : 1) * (rdir ? LABS : 1);        // walk so that s[] is fed mirrored on the high side
    rcen = rdir ? rc * LABS + rpos + 3 : (rpos + 3) * LABS + rc;
    rdst = rdir ? (rside ? &L.Gy[rcomp][8 * BS + rpos] : &L.Fy[rcomp][rpos])
                : (rside ? &L.Gx[rcomp][rpos * FROW + 8] : &L.Fx[rcomp][rpos * FROW]);
  }
};

// KernelAdvectDiffuse for the cell of this lane, from the ghosted tile in L.lab (complete and visible to the
// wave): afac (u.D)u + dfac Lap5 u, main.cpp:5493-5502.  Uses L.F*/L.G* as scratch; ends with the face values
// still in LDS (the caller synchronises before it overwrites the tile).
template <class W>
static __device__ __forceinline__ double2 advect_cell(AdvectLds &L, const RimSlot &R, int lane, double afac, double dfac) {
  const int ix = lane & 7, iy = lane >> 3;
  const int c0 = (iy + 3) * LABS + ix + 3;
  // ---- the lane's own centre: cross of half-width 2 ----
  double2 xs[5], ys[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    xs[k] = L.lab[c0 + (k - 2)];
    ys[k] = L.lab[c0 + (k - 2) * LABS];
  }
  const double u = xs[2].x, v = xs[2].y;
  // which face values anybody in this block upwinds on (wave-uniform): x derivatives follow the
  // sign of u, y derivatives the sign of v (main.cpp:5493-5496)
  const bool up = u > 0, vp = v > 0;
  const unsigned long long bu = __ballot(up), bv = __ballot(vp);
  const bool nPx = bu != 0ull, nMx = bu != ~0ull, nPy = bv != 0ull, nMy = bv != ~0ull;
  double Pxu, Mxu, Pxv, Mxv, Pyu, Myu, Pyv, Myv;
  {
    const double s[5] = {xs[0].x, xs[1].x, xs[2].x, xs[3].x, xs[4].x};
    W::fluxes(s, nPx, nMx, Pxu, Mxu);
  }
  {
    const double s[5] = {xs[0].y, xs[1].y, xs[2].y, xs[3].y, xs[4].y};
    W::fluxes(s, nPx, nMx, Pxv, Mxv);
  }
  {
    const double s[5] = {ys[0].x, ys[1].x, ys[2].x, ys[3].x, ys[4].x};
    W::fluxes(s, nPy, nMy, Pyu, Myu);
  }
  {
    const double s[5] = {ys[0].y, ys[1].y, ys[2].y, ys[3].y, ys[4].y};
    W::fluxes(s, nPy, nMy, Pyv, Myv);
  }
  L.Fx[0][iy * FROW + ix + 1] = Pxu;
  L.Gx[0][iy * FROW + ix] = Mxu;
  L.Fx[1][iy * FROW + ix + 1] = Pxv;
  L.Gx[1][iy * FROW + ix] = Mxv;
  L.Fy[0][(iy + 1) * BS + ix] = Pyu;
  L.Gy[0][iy * BS + ix] = Myu;
  L.Fy[1][(iy + 1) * BS + ix] = Pyv;
  L.Gy[1][iy * BS + ix] = Myv;
  // ---- one rim centre per lane ----
  if (R.rdir ? (R.rside ? nMy : nPy) : (R.rside ? nMx : nPx)) {
    const double *labd = (const double *)L.lab;
    const double s0 = labd[2 * (R.rcen - 2 * R.rstep) + R.rcomp], s1 = labd[2 * (R.rcen - R.rstep) + R.rcomp];
    const double s2 = labd[2 * R.rcen + R.rcomp];
    const double s3 = labd[2 * (R.rcen + R.rstep) + R.rcomp], s4 = labd[2 * (R.rcen + 2 * R.rstep) + R.rcomp];
    *R.rdst = W::plus(s0, s1, s2, s3, s4);
  }
  wave_lds_sync();
  // ---- upwind differences (derivative(), main.cpp:202-208) ----
  // U > 0: plus(c) - plus(c-1)   else: minus(c+1) - minus(c)
  // a block whose lanes all upwind to the same side (the common case) takes a wave-uniform branch: no
  // per-lane selects; the values are the same either way
  double dudx, dvdx, dudy, dvdy;
  if (!nMx) {
    dudx = Pxu - L.Fx[0][iy * FROW + ix];
    dvdx = Pxv - L.Fx[1][iy * FROW + ix];
  } else if (!nPx) {
    dudx = L.Gx[0][iy * FROW + ix + 1] - Mxu;
    dvdx = L.Gx[1][iy * FROW + ix + 1] - Mxv;
  } else {
    const double nxu = up ? L.Fx[0][iy * FROW + ix] : L.Gx[0][iy * FROW + ix + 1];
    const double nxv = up ? L.Fx[1][iy * FROW + ix] : L.Gx[1][iy * FROW + ix + 1];
    dudx = up ? Pxu - nxu : nxu - Mxu;
    dvdx = up ? Pxv - nxv : nxv - Mxv;
  }
  if (!nMy) {
    dudy = Pyu - L.Fy[0][iy * BS + ix];
    dvdy = Pyv - L.Fy[1][iy * BS + ix];
  } else if (!nPy) {
    dudy = L.Gy[0][(iy + 1) * BS + ix] - Myu;
    dvdy = L.Gy[1][(iy + 1) * BS + ix] - Myv;
  } else {
    const double nyu = vp ? L.Fy[0][iy * BS + ix] : L.Gy[0][(iy + 1) * BS + ix];
    const double nyv = vp ? L.Fy[1][iy * BS + ix] : L.Gy[1][(iy + 1) * BS + ix];
    dudy = vp ? Pyu - nyu : nyu - Myu;
    dvdy = vp ? Pyv - nyv : nyv - Myv;
  }
  // main.cpp:5497-5502, same operand order
  double2 r;
  r.x = afac * (u * dudx + v * dudy) + dfac * (xs[3].x + xs[1].x + ys[3].x + ys[1].x - 4 * u);
  r.y = afac * (u * dvdx + v * dvdy) + dfac * (xs[3].y + xs[1].y + ys[3].y + ys[1].y - 4 * v);
  return r;
}

}  // namespace cup2d
