// amr_host.hip -- regrid-time HOST code for block-AMR grids (no kernels): the Poisson matrix of an adapted grid.
//
// Reference: the serial row-assembly loop main.cpp:7034-7112 with Solver::makeFlux / interpolate / D1 / D2
// (main.cpp:5915-5997) and SpRowInfo::mapColVal (cuda.h:1-24).  In the reference this is host C++ too; here it works
// on the dense topology tables of cup2d_set_amr instead of the tree / Info hash maps.  (cup2d_amd/amr.py keeps the
// same algorithm in Python as its readable statement; tests require the two to agree bit for bit.)
#include <algorithm>
#include <vector>

#include "ctx.h"

namespace {

constexpr int BS = CUP2D_BS;

struct Row {  // a row under construction: duplicate columns are summed in the order they arrive (mapColVal)
  int n = 0;
  long long col[24];
  double val[24];
  void add(long long c, double v) {
    for (int k = 0; k < n; k++)
      if (col[k] == c) { val[k] += v; return; }
    col[n] = c;
    val[n] = v;
    n++;
  }
};
inline long long cell(int b, int ix, int iy) { return 64LL * b + 8 * iy + ix; }

// first / second derivative stencils ALONG the face at coarse cell (ix, iy) of block b (D1, D2 main.cpp:5915-5942)
void taylor(int b, int s, int ix, int iy, bool second, long long (&c)[3], double (&w)[3]) {
  const int t = s < 2 ? iy : ix;
  const auto nei = [&](int d) { return s < 2 ? cell(b, ix, iy + d) : cell(b, ix + d, iy); };
  if (t == 7 || t == 3) {
    c[0] = nei(-2); c[1] = nei(-1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = -1. / 16.; w[2] = 1. / 32.; } else { w[0] = 1. / 8.; w[1] = -1. / 2.; w[2] = 3. / 8.; }
  } else if (t == 0 || t == 4) {
    c[0] = nei(2); c[1] = nei(1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = -1. / 16.; w[2] = 1. / 32.; } else { w[0] = -1. / 8.; w[1] = 1. / 2.; w[2] = -3. / 8.; }
  } else {
    c[0] = nei(-1); c[1] = nei(1); c[2] = cell(b, ix, iy);
    if (second) { w[0] = 1. / 32.; w[1] = 1. / 32.; w[2] = -1. / 16.; } else { w[0] = -1. / 8.; w[1] = 1. / 8.; w[2] = 0.; }
  }
}
// Solver::interpolate main.cpp:5943-5960
void interpolate(Row &r, int bc, int s, int ixc, int iyc, long long fine_close, long long fine_far, double sign_int,
                 double sign_taylor) {
  r.add(fine_close, sign_int * 2. / 3.);
  r.add(fine_far, -sign_int * 1. / 5.);
  const double tf = sign_int * 8. / 15.;
  r.add(cell(bc, ixc, iyc), tf);
  long long c[3];
  double w[3];
  taylor(bc, s, ixc, iyc, false, c, w);
  for (int i = 0; i < 3; i++) r.add(c[i], sign_taylor * tf * w[i]);
  taylor(bc, s, ixc, iyc, true, c, w);
  for (int i = 0; i < 3; i++) r.add(c[i], tf * w[i]);
}

}  // namespace

extern "C" long long cup2d_amr_poisson_coo(int nblocks, const int32_t *kind, const int32_t *nbr2, const int32_t *half,
                                           long long cap, int32_t *row, int32_t *col, double *val) {
  if (nblocks <= 0 || !kind || !nbr2 || !half) {
    cup2d::set_error("amr_poisson_coo: bad argument");
    return CUP2D_ERR_ARG;
  }
  if ((long long)nblocks * 64 > 2147483647LL) {
    cup2d::set_error("amr_poisson_coo: %d blocks exceed int32 row indices", nblocks);
    return CUP2D_ERR_ARG;
  }
  const bool fill = row && col && val;
  long long nnz = 0;
  std::vector<std::pair<long long, double>> sorted;
  for (int b = 0; b < nblocks; b++)
    for (int iy = 0; iy < BS; iy++)
      for (int ix = 0; ix < BS; ix++) {
        Row r;
        const long long me = cell(b, ix, iy);
        if (ix > 0 && ix < BS - 1 && iy > 0 && iy < BS - 1) {  // main.cpp:7075-7087
          r.add(cell(b, ix, iy - 1), 1.);
          r.add(cell(b, ix - 1, iy), 1.);
          r.add(me, -4.);
          r.add(cell(b, ix + 1, iy), 1.);
          r.add(cell(b, ix, iy + 1), 1.);
        } else {
          const bool inblock[4] = {ix > 0, ix < BS - 1, iy > 0, iy < BS - 1};
          const long long inner[4] = {cell(b, ix - 1, iy), cell(b, ix + 1, iy), cell(b, ix, iy - 1), cell(b, ix, iy + 1)};
          for (int s = 0; s < 4; s++) {
            if (inblock[s]) {
              r.add(inner[s], 1.);
              r.add(me, -1.);
              continue;
            }
            const int k = kind[4 * b + s];
            if (k == CUP2D_AMR_WALL) continue;
            const int n0 = nbr2[(4 * b + s) * 2], n1 = nbr2[(4 * b + s) * 2 + 1];
            if (k == CUP2D_AMR_SAME) {  // makeFlux, same level
              r.add(s == 0 ? cell(n0, 7, iy) : s == 1 ? cell(n0, 0, iy) : s == 2 ? cell(n0, ix, 7) : cell(n0, ix, 0), 1.);
              r.add(me, -1.);
            } else if (k == CUP2D_AMR_COARSER) {
              const int h = half[4 * b + s];
              const int ixc = s == 0 ? 7 : s == 1 ? 0 : ix / 2 + 4 * h;
              const int iyc = s == 2 ? 7 : s == 3 ? 0 : iy / 2 + 4 * h;
              const long long inward = s == 0 ? cell(b, ix + 1, iy) : s == 1 ? cell(b, ix - 1, iy) : s == 2 ? cell(b, ix, iy + 1)
                                                                                                    : cell(b, ix, iy - 1);
              const int t = s < 2 ? iy : ix;
              interpolate(r, n0, s, ixc, iyc, me, inward, 1., t % 2 == 0 ? -1. : 1.);
              r.add(me, -1.);
            } else {  // the two finer cells across the face, in the child that covers this cell
              const int t = s < 2 ? iy : ix;
              const int fb = t >= 4 ? n1 : n0, f = (t % 4) * 2;
              for (int j = 0; j < 2; j++) {
                long long close, far;
                if (s == 0) { close = cell(fb, 7, f + j); far = cell(fb, 6, f + j); }
                else if (s == 1) { close = cell(fb, 0, f + j); far = cell(fb, 1, f + j); }
                else if (s == 2) { close = cell(fb, f + j, 7); far = cell(fb, f + j, 6); }
                else { close = cell(fb, f + j, 0); far = cell(fb, f + j, 1); }
                r.add(close, 1.);
                interpolate(r, b, s, ix, iy, close, far, -1., j == 0 ? -1. : 1.);
              }
            }
          }
        }
        if (fill) {
          if (nnz + r.n > cap) {
            cup2d::set_error("amr_poisson_coo: capacity %lld too small", cap);
            return CUP2D_ERR_ARG;
          }
          sorted.clear();
          for (int k = 0; k < r.n; k++) sorted.emplace_back(r.col[k], r.val[k]);
          std::sort(sorted.begin(), sorted.end(), [](const auto &a, const auto &c2) { return a.first < c2.first; });
          for (const auto &e : sorted) {
            row[nnz] = (int32_t)me;
            col[nnz] = (int32_t)e.first;
            val[nnz] = e.second;
            nnz++;
          }
        } else {
          nnz += r.n;
        }
      }
  return nnz;
}
