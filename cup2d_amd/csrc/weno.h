// weno.h -- WENO5 face reconstructions of CUP2D's advection operator for gfx950.
//
// The reference (main.cpp:162-208) evaluates, for every cell and every one of the four
// derivatives du/dx, dv/dx, du/dy, dv/dy, TWO 5-point reconstructions chosen by the sign of the
// advecting velocity:
//     U > 0 :  plus(c)  - plus(c-1)          plus(c)  = weno5_plus  about centre c
//     else  :  minus(c+1) - minus(c)         minus(c) = weno5_minus about centre c
// plus(c) and minus(c) read the same five values s[c-2..c+2], share the three smoothness
// indicators, and minus(a,b,c,d,e) == plus(e,d,c,b,a) BIT FOR BIT (every difference between the two
// reference functions is an operand swap of a commutative + or a sign flip inside a square).  A
// face value is therefore a pure function of its five inputs and can be computed ONCE per centre
// and handed to the neighbouring cell, instead of twice per cell per sign: 8 + 1 reconstructions
// per cell (4 centres x {plus, minus} + the block's 64 rim centres spread over the 64 lanes)
// instead of 8..16, with no change to any rounded intermediate.
//
// Two arithmetic policies:
//   WenoStrict : operation-for-operation the reference expressions, IEEE division, no FMA
//                contraction (the translation unit is built with -ffp-contract=off), so the
//                kernel is bit-identical to the reference CPU functor.
//   WenoFast   : same algorithm on first differences D_j = s[j+1]-s[j]; the nonlinear weights
//                w_k = (g_k/b_k^2) / sum_j (g_j/b_j^2), b_k = beta_k + 1e-6, are multiplied through
//                by b_1^2 b_2^2 b_3^2 (ONE division per reconstruction instead of four), the
//                candidate stencils are written as u + (combination of differences) so that
//                sum_k w_k = 1 is used exactly, products are explicit FMAs and the division is
//                v_rcp_f64 + one Newton step.  b_k >= 1e-6 keeps every product far inside the
//                FP64 range.  Differs from Strict by round-off only (tolerance in
//                tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>

namespace cup2d {

struct WenoStrict {
  // main.cpp:162-181
  static __device__ __forceinline__ double plus(double um2, double um1, double u, double up1, double up2) {
    const double e = 1e-6;
    double t1 = (um2 + u) - 2 * um1, t2 = (um2 + 3 * u) - 4 * um1;
    double b1 = 13.0 / 12.0 * (t1 * t1) + 0.25 * (t2 * t2);
    double t3 = (um1 + up1) - 2 * u, t4 = um1 - up1;
    double b2 = 13.0 / 12.0 * (t3 * t3) + 0.25 * (t4 * t4);
    double t5 = (u + up2) - 2 * up1, t6 = (3 * u + up2) - 4 * up1;
    double b3 = 13.0 / 12.0 * (t5 * t5) + 0.25 * (t6 * t6);
    double d1 = b1 + e, d2 = b2 + e, d3 = b3 + e;
    double what1 = 0.1 / (d1 * d1);
    double what2 = 0.6 / (d2 * d2);
    double what3 = 0.3 / (d3 * d3);
    double aux = 1.0 / ((what1 + what3) + what2);
    double w1 = what1 * aux, w2 = what2 * aux, w3 = what3 * aux;
    double f1 = (11.0 / 6.0) * u + ((1.0 / 3.0) * um2 - (7.0 / 6.0) * um1);
    double f2 = (5.0 / 6.0) * u + ((-1.0 / 6.0) * um1 + (1.0 / 3.0) * up1);
    double f3 = (1.0 / 3.0) * u + ((+5.0 / 6.0) * up1 - (1.0 / 6.0) * up2);
    return (w1 * f1 + w3 * f3) + w2 * f2;
  }
  // both face values about one centre; s[0..4] = values at c-2..c+2.
  // weno5_minus(a,b,c,d,e) (main.cpp:182-201) == weno5_plus(e,d,c,b,a) bit for bit, see above.
  // needP / needM are wave-uniform: a side nobody upwinds on is not evaluated (the reference
  // evaluates only the upwind pair too)
  static __device__ __forceinline__ void fluxes(const double (&s)[5], bool needP, bool needM, double &P, double &M) {
    P = M = 0.0;
    if (needP) P = plus(s[0], s[1], s[2], s[3], s[4]);
    if (needM) M = plus(s[4], s[3], s[2], s[1], s[0]);
  }
};

// 1/d: v_rcp_f64 seed + ONE Newton step.  Measured on MI355X (tools/fp64_peak.hip): the seed has a
// relative error of 4.4e-8, one step leaves 2.0e-15, two steps < 1e-16.  The quotient it multiplies is
// the O(difference) correction to the centre value, so 2e-15 of it is far below the stated tolerance.
static __device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}

struct WenoFast {
  // s[0..4] = values at c-2..c+2.  With D0..D3 = successive differences:
  //   smoothness (x4, the factor cancels in the weights):
  //     4*beta1 = 13/3 (D1-D0)^2 + (3 D1 - D0)^2
  //     4*beta2 = 13/3 (D2-D1)^2 + (D1 + D2)^2
  //     4*beta3 = 13/3 (D3-D2)^2 + (D3 - 3 D2)^2
  //   candidates minus the centre value u = s[2]:
  //     plus : 5/6 D1 - 1/3 D0 | 1/6 D1 + 1/3 D2 | 2/3 D2 - 1/6 D3     gammas .1 .6 .3
  //     minus: 1/6 D0 - 2/3 D1 | -1/3 D1 - 1/6 D2 | 1/3 D3 - 5/6 D2    gammas .3 .6 .1
  static __device__ __forceinline__ void fluxes(const double (&s)[5], bool needP, bool needM, double &P, double &M) {
    const double e4 = 4e-6, k = 13.0 / 3.0;
    P = M = 0.0;
    const double D0 = s[1] - s[0], D1 = s[2] - s[1], D2 = s[3] - s[2], D3 = s[4] - s[3];
    const double t1 = D1 - D0, t2 = __builtin_fma(3.0, D1, -D0);
    const double t3 = D2 - D1, t4 = D1 + D2;
    const double t5 = D3 - D2, t6 = __builtin_fma(-3.0, D2, D3);
    const double b1 = __builtin_fma(k * t1, t1, __builtin_fma(t2, t2, e4));
    const double b2 = __builtin_fma(k * t3, t3, __builtin_fma(t4, t4, e4));
    const double b3 = __builtin_fma(k * t5, t5, __builtin_fma(t6, t6, e4));
    const double q1 = b1 * b1, q2 = b2 * b2, q3 = b3 * b3;
    const double W1 = q2 * q3, W2 = q1 * q3, W3 = q1 * q2;  // ~ 1/b_k^2 up to the common factor
    if (needP) {
      const double g1 = __builtin_fma(0.1 * 5.0 / 6.0, D1, (-0.1 / 3.0) * D0);
      const double g2 = __builtin_fma(0.6 / 6.0, D1, (0.6 / 3.0) * D2);
      const double g3 = __builtin_fma(0.3 * 2.0 / 3.0, D2, (-0.3 / 6.0) * D3);
      const double num = __builtin_fma(W2, g2, __builtin_fma(W3, g3, W1 * g1));
      const double den = __builtin_fma(0.6, W2, __builtin_fma(0.3, W3, 0.1 * W1));
      P = __builtin_fma(num, fast_rcp(den), s[2]);
    }
    if (needM) {
      const double g1 = __builtin_fma(0.3 / 6.0, D0, (-0.3 * 2.0 / 3.0) * D1);
      const double g2 = __builtin_fma(-0.6 / 3.0, D1, (-0.6 / 6.0) * D2);
      const double g3 = __builtin_fma(0.1 / 3.0, D3, (-0.1 * 5.0 / 6.0) * D2);
      const double num = __builtin_fma(W2, g2, __builtin_fma(W3, g3, W1 * g1));
      const double den = __builtin_fma(0.6, W2, __builtin_fma(0.1, W3, 0.3 * W1));
      M = __builtin_fma(num, fast_rcp(den), s[2]);
    }
  }
  static __device__ __forceinline__ double plus(double um2, double um1, double u, double up1, double up2) {
    const double s[5] = {um2, um1, u, up1, up2};
    double P, M;
    fluxes(s, true, false, P, M);
    return P;
  }
};

}  // namespace cup2d
