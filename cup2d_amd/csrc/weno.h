// weno.h -- WENO5 face reconstructions of CUP2D's advection operator for gfx950.
//
// Two arithmetic policies for the same algorithm (main.cpp:162-208):
//   WenoStrict : operation-for-operation the reference expressions, IEEE division, no FMA
//                contraction (the translation unit is built with -ffp-contract=off), so the
//                kernel is bit-identical to the reference CPU functor.
//   WenoFast   : the nonlinear weights w_k = (g_k/d_k^2) / sum_j (g_j/d_j^2), d_k = beta_k + 1e-6,
//                are multiplied through by d_1^2 d_2^2 d_3^2, which leaves ONE division per
//                reconstruction instead of four; products are explicit FMAs.  d_k >= 1e-6 so the
//                products stay far inside the FP64 range.  Differs from Strict by round-off only.
#pragma once
#include <hip/hip_runtime.h>

namespace cup2d {

// smoothness indicators of the 5-point stencil (a,b,c,d,e) = (u[-2..+2]); identical for the
// "plus" and "minus" reconstructions about the same centre (main.cpp:164-169, 186-191)
struct WenoStrict {
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
  static __device__ __forceinline__ double minus(double um2, double um1, double u, double up1, double up2) {
    const double e = 1e-6;
    double t1 = (um2 + u) - 2 * um1, t2 = (um2 + 3 * u) - 4 * um1;
    double b1 = 13.0 / 12.0 * (t1 * t1) + 0.25 * (t2 * t2);
    double t3 = (um1 + up1) - 2 * u, t4 = um1 - up1;
    double b2 = 13.0 / 12.0 * (t3 * t3) + 0.25 * (t4 * t4);
    double t5 = (u + up2) - 2 * up1, t6 = (3 * u + up2) - 4 * up1;
    double b3 = 13.0 / 12.0 * (t5 * t5) + 0.25 * (t6 * t6);
    double d1 = b1 + e, d2 = b2 + e, d3 = b3 + e;
    double what1 = 0.3 / (d1 * d1);
    double what2 = 0.6 / (d2 * d2);
    double what3 = 0.1 / (d3 * d3);
    double aux = 1.0 / ((what1 + what3) + what2);
    double w1 = what1 * aux, w2 = what2 * aux, w3 = what3 * aux;
    double f1 = (1.0 / 3.0) * u + ((-1.0 / 6.0) * um2 + (5.0 / 6.0) * um1);
    double f2 = (5.0 / 6.0) * u + ((1.0 / 3.0) * um1 - (1.0 / 6.0) * up1);
    double f3 = (11.0 / 6.0) * u + ((-7.0 / 6.0) * up1 + (1.0 / 3.0) * up2);
    return (w1 * f1 + w3 * f3) + w2 * f2;
  }
  // main.cpp:202-208
  static __device__ __forceinline__ double derivative(double U, double um3, double um2, double um1, double u,
                                                      double up1, double up2, double up3) {
    return U > 0 ? plus(um2, um1, u, up1, up2) - plus(um3, um2, um1, u, up1)
                 : minus(um1, u, up1, up2, up3) - minus(um2, um1, u, up1, up2);
  }
};

// reciprocal with two Newton steps on v_rcp_f64 (relative error ~1e-16), then one residual
// correction of the quotient: a division in ~8 FP64 instructions instead of the IEEE sequence
static __device__ __forceinline__ double fast_div(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

struct WenoFast {
  // D_k = prod_{j != k} (beta_j + eps)^2, shared by plus and minus about one centre
  static __device__ __forceinline__ void betas(double um2, double um1, double u, double up1, double up2,
                                               double &D1, double &D2, double &D3) {
    const double e = 1e-6, k13 = 13.0 / 12.0;
    double t1 = __builtin_fma(-2.0, um1, um2 + u);
    double t2 = __builtin_fma(-4.0, um1, __builtin_fma(3.0, u, um2));
    double t3 = __builtin_fma(-2.0, u, um1 + up1);
    double t4 = um1 - up1;
    double t5 = __builtin_fma(-2.0, up1, u + up2);
    double t6 = __builtin_fma(-4.0, up1, __builtin_fma(3.0, u, up2));
    double d1 = __builtin_fma(k13, t1 * t1, __builtin_fma(0.25 * t2, t2, e));
    double d2 = __builtin_fma(k13, t3 * t3, __builtin_fma(0.25 * t4, t4, e));
    double d3 = __builtin_fma(k13, t5 * t5, __builtin_fma(0.25 * t6, t6, e));
    double q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    D1 = q2 * q3;
    D2 = q1 * q3;
    D3 = q1 * q2;
  }
  static __device__ __forceinline__ double plus_w(double um2, double um1, double u, double up1, double up2,
                                                  double D1, double D2, double D3) {
    double n1 = 0.1 * D1, n2 = 0.6 * D2, n3 = 0.3 * D3;
    double f1 = __builtin_fma(11.0 / 6.0, u, __builtin_fma(1.0 / 3.0, um2, (-7.0 / 6.0) * um1));
    double f2 = __builtin_fma(5.0 / 6.0, u, __builtin_fma(-1.0 / 6.0, um1, (1.0 / 3.0) * up1));
    double f3 = __builtin_fma(1.0 / 3.0, u, __builtin_fma(5.0 / 6.0, up1, (-1.0 / 6.0) * up2));
    double num = __builtin_fma(n2, f2, __builtin_fma(n3, f3, n1 * f1));
    double den = (n1 + n3) + n2;
    return fast_div(num, den);
  }
  static __device__ __forceinline__ double minus_w(double um2, double um1, double u, double up1, double up2,
                                                   double D1, double D2, double D3) {
    double n1 = 0.3 * D1, n2 = 0.6 * D2, n3 = 0.1 * D3;
    double f1 = __builtin_fma(1.0 / 3.0, u, __builtin_fma(-1.0 / 6.0, um2, (5.0 / 6.0) * um1));
    double f2 = __builtin_fma(5.0 / 6.0, u, __builtin_fma(1.0 / 3.0, um1, (-1.0 / 6.0) * up1));
    double f3 = __builtin_fma(11.0 / 6.0, u, __builtin_fma(-7.0 / 6.0, up1, (1.0 / 3.0) * up2));
    double num = __builtin_fma(n2, f2, __builtin_fma(n3, f3, n1 * f1));
    double den = (n1 + n3) + n2;
    return fast_div(num, den);
  }
  static __device__ __forceinline__ double plus(double um2, double um1, double u, double up1, double up2) {
    double D1, D2, D3;
    betas(um2, um1, u, up1, up2, D1, D2, D3);
    return plus_w(um2, um1, u, up1, up2, D1, D2, D3);
  }
  static __device__ __forceinline__ double minus(double um2, double um1, double u, double up1, double up2) {
    double D1, D2, D3;
    betas(um2, um1, u, up1, up2, D1, D2, D3);
    return minus_w(um2, um1, u, up1, up2, D1, D2, D3);
  }
  static __device__ __forceinline__ double derivative(double U, double um3, double um2, double um1, double u,
                                                      double up1, double up2, double up3) {
    return U > 0 ? plus(um2, um1, u, up1, up2) - plus(um3, um2, um1, u, up1)
                 : minus(um1, u, up1, up2, up3) - minus(um2, um1, u, up1, up2);
  }
};

}  // namespace cup2d
