/* oracle/cup2d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-threaded-order restatement of CUP2D's per-block stencil hot
 * path on a uniform, wall-bounded nx x ny cell grid (SURVEY.md section 8a rows
 * a1-a19).  Every function cites the /root/reference lines it follows.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it;
 * the product (cup2d_amd/csrc) never links or imports anything in oracle/.
 *
 * Pinning: the reference ships no tests or golden vectors.
 * This restatement is pinned instead against the reference's OWN functors,
 * compiled from /root/reference/main.cpp by oracle/ref_harness.cpp
 * (oracle/_ref/ref_harness): tests/test_oracle_vs_reference.py demands
 * bit-identical output for every block functor (a2, a4, a9, a10, a11, a19) and
 * committed fixtures of those outputs live in tests/golden/.  The BiCGSTAB
 * solver (a17) restates cuda.cu, whose cuBLAS/cuSPARSE reduction order is not
 * specified: that part is "parity unpinned" beyond round-off.
 *
 * Layout: global row-major, cell (ix,iy) at iy*nx+ix; vector fields interleaved
 * (u,v).  The reference stores the same cells in 8x8 blocks (main.cpp:510,
 * 5497-5502); block order only changes reduction order.
 *
 * Arithmetic: compiled with -ffp-contract=off so that every operation rounds
 * once, in the operand order of the reference expression it restates.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BS 8

/* ---- ghost access: VectorLab::applyBCface (main.cpp:3131-3204): free-slip wall,
 * every ghost layer = edge cell with the wall-normal component negated;
 * ScalarLab::Neumann2D (main.cpp:3210-3255): ghost = edge cell.  Only the
 * cross (no corner) is ever read by the functors below. ---- */
static inline double vget(const double *v, int nx, int ny, int ix, int iy, int c) {
  double s = 1.0;
  if (ix < 0) { ix = 0; if (c == 0) s = -s; }
  else if (ix >= nx) { ix = nx - 1; if (c == 0) s = -s; }
  if (iy < 0) { iy = 0; if (c == 1) s = -s; }
  else if (iy >= ny) { iy = ny - 1; if (c == 1) s = -s; }
  return s * v[2 * ((size_t)iy * nx + ix) + c];
}
static inline double sget(const double *p, int nx, int ny, int ix, int iy) {
  if (ix < 0) ix = 0; else if (ix >= nx) ix = nx - 1;
  if (iy < 0) iy = 0; else if (iy >= ny) iy = ny - 1;
  return p[(size_t)iy * nx + ix];
}

/* ---- a1: weno5_plus / weno5_minus / derivative (main.cpp:162-208) ---- */
static double weno5_plus(double um2, double um1, double u, double up1, double up2) {
  const double e = 1e-6;
  double t1 = (um2 + u) - 2 * um1, t2 = (um2 + 3 * u) - 4 * um1;
  double b1 = 13.0 / 12.0 * (t1 * t1) + 0.25 * (t2 * t2);
  double t3 = (um1 + up1) - 2 * u, t4 = um1 - up1;
  double b2 = 13.0 / 12.0 * (t3 * t3) + 0.25 * (t4 * t4);
  double t5 = (u + up2) - 2 * up1, t6 = (3 * u + up2) - 4 * up1;
  double b3 = 13.0 / 12.0 * (t5 * t5) + 0.25 * (t6 * t6);
  double g1 = 0.1, g2 = 0.6, g3 = 0.3;
  double d1 = b1 + e, d2 = b2 + e, d3 = b3 + e;
  double what1 = g1 / (d1 * d1);
  double what2 = g2 / (d2 * d2);
  double what3 = g3 / (d3 * d3);
  double aux = 1.0 / ((what1 + what3) + what2);
  double w1 = what1 * aux, w2 = what2 * aux, w3 = what3 * aux;
  double f1 = (11.0 / 6.0) * u + ((1.0 / 3.0) * um2 - (7.0 / 6.0) * um1);
  double f2 = (5.0 / 6.0) * u + ((-1.0 / 6.0) * um1 + (1.0 / 3.0) * up1);
  double f3 = (1.0 / 3.0) * u + ((+5.0 / 6.0) * up1 - (1.0 / 6.0) * up2);
  return (w1 * f1 + w3 * f3) + w2 * f2;
}
static double weno5_minus(double um2, double um1, double u, double up1, double up2) {
  const double e = 1e-6;
  double t1 = (um2 + u) - 2 * um1, t2 = (um2 + 3 * u) - 4 * um1;
  double b1 = 13.0 / 12.0 * (t1 * t1) + 0.25 * (t2 * t2);
  double t3 = (um1 + up1) - 2 * u, t4 = um1 - up1;
  double b2 = 13.0 / 12.0 * (t3 * t3) + 0.25 * (t4 * t4);
  double t5 = (u + up2) - 2 * up1, t6 = (3 * u + up2) - 4 * up1;
  double b3 = 13.0 / 12.0 * (t5 * t5) + 0.25 * (t6 * t6);
  double g1 = 0.3, g2 = 0.6, g3 = 0.1;
  double d1 = b1 + e, d2 = b2 + e, d3 = b3 + e;
  double what1 = g1 / (d1 * d1);
  double what2 = g2 / (d2 * d2);
  double what3 = g3 / (d3 * d3);
  double aux = 1.0 / ((what1 + what3) + what2);
  double w1 = what1 * aux, w2 = what2 * aux, w3 = what3 * aux;
  double f1 = (1.0 / 3.0) * u + ((-1.0 / 6.0) * um2 + (5.0 / 6.0) * um1);
  double f2 = (5.0 / 6.0) * u + ((1.0 / 3.0) * um1 - (1.0 / 6.0) * up1);
  double f3 = (11.0 / 6.0) * u + ((-7.0 / 6.0) * up1 + (1.0 / 3.0) * up2);
  return (w1 * f1 + w3 * f3) + w2 * f2;
}
static double derivative(double U, double um3, double um2, double um1, double u, double up1, double up2,
                         double up3) {
  return U > 0 ? weno5_plus(um2, um1, u, up1, up2) - weno5_plus(um3, um2, um1, u, up1)
               : weno5_minus(um1, u, up1, up2, up3) - weno5_minus(um2, um1, u, up1, up2);
}
/* exported for unit tests of a1 */
double oracle_weno5_plus(const double *s) { return weno5_plus(s[0], s[1], s[2], s[3], s[4]); }
double oracle_weno5_minus(const double *s) { return weno5_minus(s[0], s[1], s[2], s[3], s[4]); }
double oracle_derivative(double U, const double *s) {
  return derivative(U, s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
}

/* ---- a2: KernelAdvectDiffuse::operator() (main.cpp:5441-5503)
 * tmpV = afac*(u.D)u + dfac*Lap5(u), afac = -dt*h, dfac = nu*dt ---- */
void oracle_advect_diffuse_rhs(int nx, int ny, double h, double nu, double dt, const double *vel,
                               double *tmpV) {
  const double dfac = nu * dt, afac = -dt * h;
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      double sx[2][7], sy[2][7];
      for (int c = 0; c < 2; c++)
        for (int k = -3; k <= 3; k++) {
          sx[c][k + 3] = vget(vel, nx, ny, ix + k, iy, c);
          sy[c][k + 3] = vget(vel, nx, ny, ix, iy + k, c);
        }
      const double u = sx[0][3], v = sx[1][3];
      double dudx = derivative(u, sx[0][0], sx[0][1], sx[0][2], u, sx[0][4], sx[0][5], sx[0][6]);
      double dudy = derivative(v, sy[0][0], sy[0][1], sy[0][2], u, sy[0][4], sy[0][5], sy[0][6]);
      double dvdx = derivative(u, sx[1][0], sx[1][1], sx[1][2], v, sx[1][4], sx[1][5], sx[1][6]);
      double dvdy = derivative(v, sy[1][0], sy[1][1], sy[1][2], v, sy[1][4], sy[1][5], sy[1][6]);
      size_t o = 2 * ((size_t)iy * nx + ix);
      tmpV[o] = afac * (u * dudx + v * dudy) + dfac * (sx[0][4] + sx[0][2] + sy[0][4] + sy[0][2] - 4 * u);
      tmpV[o + 1] = afac * (u * dvdx + v * dvdy) + dfac * (sx[1][4] + sx[1][2] + sy[1][4] + sy[1][2] - 4 * v);
    }
}

/* the same functor on ONE ghosted tile (14 x 14 x 2, cross cells only are read): the form KernelAdvectDiffuse has in
 * the reference, used for adapted grids where the tile comes from BlockLab's coarse-fine interpolation (oracle/amr.py) */
void oracle_advect_diffuse_lab(const double *lab, double h, double nu, double dt, double *out) {
  const double dfac = nu * dt, afac = -dt * h;
  const int nm = 14;
  for (int iy = 0; iy < 8; iy++)
    for (int ix = 0; ix < 8; ix++) {
      double sx[2][7], sy[2][7];
      for (int c = 0; c < 2; c++)
        for (int k = -3; k <= 3; k++) {
          sx[c][k + 3] = lab[2 * (nm * (iy + 3) + ix + 3 + k) + c];
          sy[c][k + 3] = lab[2 * (nm * (iy + 3 + k) + ix + 3) + c];
        }
      const double u = sx[0][3], v = sx[1][3];
      double dudx = derivative(u, sx[0][0], sx[0][1], sx[0][2], u, sx[0][4], sx[0][5], sx[0][6]);
      double dudy = derivative(v, sy[0][0], sy[0][1], sy[0][2], u, sy[0][4], sy[0][5], sy[0][6]);
      double dvdx = derivative(u, sx[1][0], sx[1][1], sx[1][2], v, sx[1][4], sx[1][5], sx[1][6]);
      double dvdy = derivative(v, sy[1][0], sy[1][1], sy[1][2], v, sy[1][4], sy[1][5], sy[1][6]);
      size_t o = 2 * ((size_t)iy * 8 + ix);
      out[o] = afac * (u * dudx + v * dudy) + dfac * (sx[0][4] + sx[0][2] + sy[0][4] + sy[0][2] - 4 * u);
      out[o + 1] = afac * (u * dvdx + v * dvdy) + dfac * (sx[1][4] + sx[1][2] + sy[1][4] + sy[1][2] - 4 * v);
    }
}

/* ---- a4: RK2 glue (main.cpp:6607-6642): vold = vel; V = Vold + 0.5*tmpV/h^2; V = Vold + tmpV/h^2.
 * stage1 (optional) receives the mid-point velocity. ---- */
void oracle_rk2_advect_diffuse(int nx, int ny, double h, double nu, double dt, double *vel, double *stage1) {
  const size_t n2 = 2 * (size_t)nx * ny;
  double *vold = (double *)malloc(n2 * sizeof(double));
  double *tmpV = (double *)malloc(n2 * sizeof(double));
  memcpy(vold, vel, n2 * sizeof(double));
  for (int stage = 0; stage < 2; stage++) {
    oracle_advect_diffuse_rhs(nx, ny, h, nu, dt, vel, tmpV);
    const double ih2 = (stage == 0 ? 0.5 : 1.0) / (h * h);
    for (size_t j = 0; j < n2; j++)
      vel[j] = vold[j] + tmpV[j] * ih2;
    if (stage == 0 && stage1)
      memcpy(stage1, vel, n2 * sizeof(double));
  }
  free(vold);
  free(tmpV);
}

/* ---- a19: KernelVorticity (main.cpp:3343-3366) ---- */
void oracle_vorticity(int nx, int ny, double h, const double *vel, double *out) {
  const double i2h = 0.5 / h;
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      double e0 = vget(vel, nx, ny, ix, iy - 1, 0), e1 = vget(vel, nx, ny, ix, iy + 1, 0);
      double e2 = vget(vel, nx, ny, ix + 1, iy, 1), e3 = vget(vel, nx, ny, ix - 1, iy, 1);
      out[(size_t)iy * nx + ix] = i2h * (e0 - e1 + e2 - e3);
    }
}

/* ---- a9: pressure_rhs::operator() (main.cpp:6105-6139); udef/chi may be NULL (= 0) ---- */
void oracle_pressure_rhs(int nx, int ny, double h, double dt, const double *vel, const double *udef,
                         const double *chi, double *out) {
  const double facDiv = 0.5 * h / dt;
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      double v0 = vget(vel, nx, ny, ix + 1, iy, 0), v1 = vget(vel, nx, ny, ix - 1, iy, 0);
      double v2 = vget(vel, nx, ny, ix, iy + 1, 1), v3 = vget(vel, nx, ny, ix, iy - 1, 1);
      double dudef = 0, X = 0;
      if (udef) {
        double u0 = vget(udef, nx, ny, ix + 1, iy, 0), u1 = vget(udef, nx, ny, ix - 1, iy, 0);
        double u2 = vget(udef, nx, ny, ix, iy + 1, 1), u3 = vget(udef, nx, ny, ix, iy - 1, 1);
        dudef = u0 - u1 + u2 - u3;
      }
      if (chi) X = chi[(size_t)iy * nx + ix];
      out[(size_t)iy * nx + ix] = facDiv * (v0 - v1 + v2 - v3) - facDiv * X * dudef;
    }
}

/* ---- a10: pressure_rhs1::operator() (main.cpp:6209-6230): tmp -= (pW+pE+pS+pN-4p) ---- */
void oracle_laplacian_sub(int nx, int ny, const double *p, double *tmp) {
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      double l0 = sget(p, nx, ny, ix, iy), l1 = sget(p, nx, ny, ix - 1, iy), l2 = sget(p, nx, ny, ix + 1, iy);
      double l3 = sget(p, nx, ny, ix, iy - 1), l4 = sget(p, nx, ny, ix, iy + 1);
      tmp[(size_t)iy * nx + ix] -= l1 + l2 + l3 + l4 - 4 * l0;
    }
}

/* ---- a12: the assembled Poisson matrix on a uniform grid (main.cpp:7034-7112):
 * interior rows [1,1,-4,1,1]; block-edge rows add (col,+1),(self,-1) per existing
 * neighbour; walls contribute nothing => graph Laplacian.  y = A x. ---- */
void oracle_apply_A(int nx, int ny, const double *x, double *y) {
  /* evaluated with the expression of the reference's matrix-free form of the same
   * operator, pressure_rhs1 (main.cpp:6228): W+E+S+N-4C with Neumann (clamped) ghosts */
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      double l0 = sget(x, nx, ny, ix, iy), l1 = sget(x, nx, ny, ix - 1, iy), l2 = sget(x, nx, ny, ix + 1, iy);
      double l3 = sget(x, nx, ny, ix, iy - 1), l4 = sget(x, nx, ny, ix, iy + 1);
      y[(size_t)iy * nx + ix] = l1 + l2 + l3 + l4 - 4 * l0;
    }
}

/* ---- a11: pressureCorrectionKernel (main.cpp:6021-6043): tmpV = -0.5*dt*h*grad p ---- */
void oracle_pressure_correction(int nx, int ny, double h, double dt, const double *pres, double *tmpV) {
  const double pFac = -0.5 * dt * h;
#pragma omp parallel for schedule(static)
  for (int iy = 0; iy < ny; iy++)
    for (int ix = 0; ix < nx; ix++) {
      size_t o = 2 * ((size_t)iy * nx + ix);
      tmpV[o] = pFac * (sget(pres, nx, ny, ix + 1, iy) - sget(pres, nx, ny, ix - 1, iy));
      tmpV[o + 1] = pFac * (sget(pres, nx, ny, ix, iy + 1) - sget(pres, nx, ny, ix, iy - 1));
    }
}
/* projection update (main.cpp:7180-7187): V += tmpV/h/h */
void oracle_add_scaled(int nx, int ny, double h, const double *tmpV, double *vel) {
  const double ih2 = 1.0 / h / h;
  for (size_t j = 0; j < 2 * (size_t)nx * ny; j++)
    vel[j] += tmpV[j] * ih2;
}

/* ---- a18: dt (main.cpp:6579-6595) ---- */
double oracle_max_abs(size_t count, const double *v) {
  double m = 0;
  for (size_t j = 0; j < count; j++) { double a = fabs(v[j]); if (a > m) m = a; }
  return m;
}
double oracle_compute_dt(double h, double nu, double cfl, double umax) {
  double dtDiffusion = 0.25 * h * h / (nu + 0.25 * h * umax);
  double dtAdvection = h / (umax + 1e-8);
  double a = cfl * dtAdvection;
  return dtDiffusion < a ? dtDiffusion : a;
}

/* ---- a13: getA_local + Cholesky + inverse (main.cpp:46-57, 6451-6488):
 * P_inv = -(L L^T)^{-1} of the 64x64 in-block Dirichlet Laplacian ---- */
static double getA_local(int I1, int I2) {
  int j1 = I1 / BS, i1 = I1 % BS, j2 = I2 / BS, i2 = I2 % BS;
  if (i1 == i2 && j1 == j2) return 4.0;
  if (abs(i1 - i2) + abs(j1 - j2) == 1) return -1.0;
  return 0.0;
}
void oracle_P_inv(double *P_inv /* 64*64 */) {
  enum { N = BS * BS };
  static double L[N][N], Li[N][N];
  memset(L, 0, sizeof L);
  memset(Li, 0, sizeof Li);
  for (int i = 0; i < N; i++) Li[i][i] = 1.0;
  for (int i = 0; i < N; i++) {
    double s1 = 0;
    for (int k = 0; k <= i - 1; k++) s1 += L[i][k] * L[i][k];
    L[i][i] = sqrt(getA_local(i, i) - s1);
    for (int j = i + 1; j < N; j++) {
      double s2 = 0;
      for (int k = 0; k <= i - 1; k++) s2 += L[i][k] * L[j][k];
      L[j][i] = (getA_local(j, i) - s2) / L[i][i];
    }
  }
  for (int br = 0; br < N; br++) {
    double bsf = 1. / L[br][br];
    for (int c = 0; c <= br; c++) Li[br][c] *= bsf;
    for (int wr = br + 1; wr < N; wr++) {
      double wsf = L[wr][br];
      for (int c = 0; c <= br; c++) Li[wr][c] -= wsf * Li[br][c];
    }
  }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      double aux = 0.;
      for (int k = 0; k < N; k++) aux += (i <= k && j <= k) ? Li[k][i] * Li[k][j] : 0.;
      P_inv[i * N + j] = -aux;
    }
}

/* ---- block-Jacobi preconditioner apply (cuda.cu:484-486: Dgemm(T,N) with P_inv):
 * per 8x8 block z_b = P_inv^T p_b (P_inv symmetric) ---- */
void oracle_precond(int nx, int ny, const double *P_inv, const double *in, double *out) {
  const int nbx = nx / BS, nby = ny / BS;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < nbx * nby; b++) {
    const int bx = b % nbx, by = b / nbx;
    double loc[BS * BS];
    for (int j = 0; j < BS * BS; j++)
      loc[j] = in[(size_t)(by * BS + j / BS) * nx + bx * BS + j % BS];
    for (int i = 0; i < BS * BS; i++) {
      double s = 0;
      for (int k = 0; k < BS * BS; k++) s += P_inv[i * BS * BS + k] * loc[k];
      out[(size_t)(by * BS + i / BS) * nx + bx * BS + i % BS] = s;
    }
  }
}

/* ---- a17: BiCGSTABSolver::main (cuda.cu:403-548) with the matrix-free operator.
 * x: in = initial guess, out = x_opt.  info[0]=iterations, info[1]=restarts,
 * info[2]=error_opt (Linf of r), info[3]=error_init.  max_iter: reference hard-codes 1000. ---- */
static double dotp(size_t m, const double *a, const double *b) {
  double s = 0;
  for (size_t i = 0; i < m; i++) s += a[i] * b[i];
  return s;
}
void oracle_bicgstab(int nx, int ny, const double *P_inv, const double *b, double *x, double max_error,
                     double max_rel_error, int max_restarts, int max_iter, double *info) {
  const size_t m = (size_t)nx * ny;
  double *r = malloc(m * 8), *rhat = malloc(m * 8), *p = calloc(m, 8), *nu = calloc(m, 8);
  double *t = malloc(m * 8), *z = malloc(m * 8), *x_opt = malloc(m * 8);
  double alpha = 1, beta = 1, omega = 1, rho_prev = 1, rho_curr = 1, b1;
  const double eps = 1e-21;
  int restarts = 0, k = 0;
  memcpy(z, x, m * 8);
  oracle_apply_A(nx, ny, z, nu);
  for (size_t i = 0; i < m; i++) r[i] = b[i] - nu[i];
  double error = oracle_max_abs(m, r), error_init = error, error_opt = error;
  memcpy(x_opt, x, m * 8);
  memcpy(rhat, r, m * 8);
  memset(nu, 0, m * 8);
  for (k = 0; k < max_iter; k++) {
    rho_curr = dotp(m, rhat, r);
    double n1 = sqrt(dotp(m, r, r)), n2 = sqrt(dotp(m, rhat, rhat));
    n1 *= n1;
    n2 *= n2;
    const int serious_breakdown = rho_curr * rho_curr < 1e-16 * n1 * n2;
    beta = (rho_curr / (rho_prev + eps)) * (alpha / (omega + eps));
    if (serious_breakdown && max_restarts > 0) {
      restarts++;
      if (restarts >= max_restarts) break;
      memcpy(rhat, r, m * 8);
      double nr = sqrt(dotp(m, rhat, rhat));
      rho_curr = nr * nr;
      memset(nu, 0, m * 8);
      memset(p, 0, m * 8);
      rho_prev = 1.; alpha = 1.; omega = 1.;
      beta = (rho_curr / (rho_prev + eps)) * (alpha / (omega + eps));
    }
    b1 = -omega;
    for (size_t i = 0; i < m; i++) p[i] += b1 * nu[i];
    for (size_t i = 0; i < m; i++) p[i] *= beta;
    for (size_t i = 0; i < m; i++) p[i] += r[i];
    oracle_precond(nx, ny, P_inv, p, z);
    oracle_apply_A(nx, ny, z, nu);
    b1 = dotp(m, rhat, nu);
    alpha = rho_curr / (b1 + eps);
    for (size_t i = 0; i < m; i++) x[i] += alpha * z[i];
    b1 = -alpha;
    for (size_t i = 0; i < m; i++) r[i] += b1 * nu[i];
    oracle_precond(nx, ny, P_inv, r, z);
    oracle_apply_A(nx, ny, z, t);
    b1 = dotp(m, t, r);
    double tt = sqrt(dotp(m, t, t));
    tt *= tt;
    omega = b1 / (tt + eps);
    for (size_t i = 0; i < m; i++) x[i] += omega * z[i];
    b1 = -omega;
    for (size_t i = 0; i < m; i++) r[i] += b1 * t[i];
    error = oracle_max_abs(m, r);
    if (error < error_opt) {
      error_opt = error;
      memcpy(x_opt, x, m * 8);
      if ((error <= max_error) || (error / error_init <= max_rel_error)) { k++; break; }
    }
    rho_prev = rho_curr;
  }
  memcpy(x, x_opt, m * 8);
  if (info) { info[0] = k; info[1] = restarts; info[2] = error_opt; info[3] = error_init; }
  free(r); free(rhat); free(p); free(nu); free(t); free(z); free(x_opt);
}

/* ---- mean removal + pold add (main.cpp:7120-7173), uniform h: pres = x - avg(x);
 * pres += pold - avg(pres) ---- */
void oracle_pressure_update(int nx, int ny, double h, const double *x, const double *pold, double *pres) {
  const size_t m = (size_t)nx * ny;
  const double vv = h * h;
  double avg = 0, avg1 = 0;
  for (size_t j = 0; j < m; j++) { pres[j] = x[j]; avg += pres[j] * vv; avg1 += vv; }
  avg = avg / avg1;
  for (size_t j = 0; j < m; j++) pres[j] += -avg;
  avg = 0; avg1 = 0;
  for (size_t j = 0; j < m; j++) { avg += pres[j] * vv; avg1 += vv; }
  avg = avg / avg1;
  for (size_t j = 0; j < m; j++) pres[j] += pold[j] - avg;
}

/* ---- one full body-free time step, main.cpp:6576-7187 in order.
 * vel, pres: in/out.  Returns dt.  info as oracle_bicgstab. ---- */
double oracle_step(int nx, int ny, double h, double nu, double cfl, const double *P_inv, double *vel,
                   double *pres, double tol, double rel_tol, int max_restarts, int max_iter, double *info) {
  const size_t m = (size_t)nx * ny;
  const double umax = oracle_max_abs(2 * m, vel);
  const double dt = oracle_compute_dt(h, nu, cfl, umax);
  oracle_rk2_advect_diffuse(nx, ny, h, nu, dt, vel, NULL);
  double *b = malloc(m * 8), *pold = malloc(m * 8), *x = calloc(m, 8), *tmpV = malloc(2 * m * 8);
  oracle_pressure_rhs(nx, ny, h, dt, vel, NULL, NULL, b); /* tmpV (udef) = 0, chi = 0 */
  memcpy(pold, pres, m * 8);                              /* pold = pres; pres = 0 */
  oracle_laplacian_sub(nx, ny, pold, b);
  oracle_bicgstab(nx, ny, P_inv, b, x, tol, rel_tol, max_restarts, max_iter, info);
  oracle_pressure_update(nx, ny, h, x, pold, pres);
  oracle_pressure_correction(nx, ny, h, dt, pres, tmpV);
  oracle_add_scaled(nx, ny, h, tmpV, vel);
  free(b); free(pold); free(x); free(tmpV);
  return dt;
}
