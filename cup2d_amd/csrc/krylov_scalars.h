// krylov_scalars.h -- the device-resident scalar state of BiCGSTAB and its recurrences (BiCGSTABSolver::main,
// cuda.cu:303-330, 440-545), stage by stage as the sweeps' reductions deliver their sums.  Host+device and free of HIP
// built-ins: the kernels (krylov.hip, krylov_fused.hip, krylov_edge.h, comm.hip) run scalars_update on one lane;
// tests/scalars_host.cpp compiles the very same functions with g++ and tests/test_two_launch_recurrence.py drives them
// through whole solves on the CPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KRYLOV_HD static __host__ __device__ inline
#else
#define KRYLOV_HD static inline
#endif

namespace cup2d {

// device-resident BiCGSTAB state: the reference's BiCGSTABScalars (cuda.cu:24-34) plus the
// control flow its host loop keeps in local variables (cuda.cu:404-545)
struct KrylovScalars {
  double alpha, beta, omega, eps, rho_prev, rho_curr;
  double rr;         // ||r||^2
  double rhat2;      // ||rhat||^2 (changes only at a restart)
  double err, err_init, err_opt;
  double max_error, max_rel_error;
  int max_restarts, max_iter;
  int iter, restarts;
  int status;        // 0 running, 1 converged, 2 restart limit, 3 iteration cap
  int restart_flag;  // next p-update must do rhat = r, p = r (cuda.cu:461-476)
  int x_is_best;     // the iterate held in x is the best so far (cuda.cu:535-538)
  int ycur, ybest;   // fused solver: which of its three y buffers holds the current / the best iterate
  int best_is_x0;    // fused solver: no iterate has beaten the initial guess yet (y_best = 0: its buffer is never written or read)
  double omega_r;    // the omega sweep E formed r = s - omega t with (a restart resets omega, not this): stored-edge ring
  double rho_next;   // k_edge MODE 2 / 3: rhat . r' from the sums of sweep D (krylov_common.h stage 5)
};

// beginning of an iteration (cuda.cu:440-477): consumes rho = rhat.r and ||r||^2
KRYLOV_HD void begin_iteration(KrylovScalars *sc) {
  if (sc->iter >= sc->max_iter) { sc->status = 3; return; }
  const bool serious_breakdown = sc->rho_curr * sc->rho_curr < 1e-16 * sc->rr * sc->rhat2;
  sc->beta = (sc->rho_curr / (sc->rho_prev + sc->eps)) * (sc->alpha / (sc->omega + sc->eps));  // set_beta
  sc->restart_flag = 0;
  if (serious_breakdown && sc->max_restarts > 0) {
    sc->restarts++;
    if (sc->restarts >= sc->max_restarts) { sc->status = 2; return; }
    sc->restart_flag = 1;
    sc->rhat2 = sc->rr;     // rhat = r
    sc->rho_curr = sc->rr;  // Dnrm2(rhat)^2
    sc->rho_prev = 1.; sc->alpha = 1.; sc->omega = 1.;  // breakdown_update
    sc->beta = (sc->rho_curr / (sc->rho_prev + sc->eps)) * (sc->alpha / (sc->omega + sc->eps));
  }
}
// The fused solver keeps the accumulated correction y in THREE buffers and never copies the best iterate
// (cuda.cu:535-538 copies x to x_opt): sweep E reads buffer ycur and writes the buffer that is neither ycur
// nor ybest; stage 3 below then rotates.  (The five-sweep solver updates x in place and copies.)
KRYLOV_HD int y_out_buffer(int cur, int best) {
  return (cur != 0 && best != 0) ? 0 : ((cur != 1 && best != 1) ? 1 : 2);
}
// STAGE 0: after k_init_residual  red = {r.r, -, max|r|}
// STAGE 1: after sweep B          red = {rhat.nu}              -> alpha (set_alpha)
// STAGE 2: after sweep D          red = {t.r, t.t}             -> omega (set_omega)
// STAGE 3: after sweep E          red = {rhat.r, r.r, max|r|}  -> error bookkeeping, next beta
// The organisation with sweep E and the next A+B in ONE launch (k_edge MODE 2 / 3; one GPU):
// STAGE 5: after sweep D (MODE 3)  red = {t.s, t.t, rhat.s, rhat.t, s.s} -> omega, and -- BEFORE r' = s - omega t exists --
//          what the beginning of the next iteration needs of it: rho' = rhat.r' = rhat.s - omega rhat.t (the same sum in
//          another order of rounding), ||r'||^2 = s.s - 2 omega t.s + omega^2 t.t (plus its own error margin, below) for the
//          breakdown test, hence beta and the restart decision the launch is going to use.  No bookkeeping: the iteration is
//          not over.
// STAGE 4: after MODE 2            red = {rhat.nu'', r'.r', max|r'|} -> stage 3's bookkeeping with rho' of stage 5 and the
//          decision taken there (a restart takes rho = ||rhat||^2 = r'.r' as summed HERE, cell by cell), then stage 1's alpha
KRYLOV_HD void scalars_update(KrylovScalars *sc, const double *red, int stage) {
  switch (stage) {
  case 0:
    sc->err = sc->err_init = sc->err_opt = red[2];
    sc->x_is_best = 1;
    sc->ycur = sc->ybest = 0;
    sc->best_is_x0 = 1;
    sc->rr = red[0]; sc->rhat2 = red[0]; sc->rho_curr = red[0];
    begin_iteration(sc);
    break;
  case 1:
    sc->alpha = sc->rho_curr / (red[0] + sc->eps);
    break;
  case 2:
    sc->omega = red[0] / (red[1] + sc->eps);
    sc->omega_r = sc->omega;
    break;
  case 3:
    sc->iter++;
    sc->err = red[2];
    sc->ycur = y_out_buffer(sc->ycur, sc->ybest);  // where sweep E put the new iterate
    if (sc->err < sc->err_opt) {
      sc->err_opt = sc->err;
      sc->x_is_best = 1;
      sc->ybest = sc->ycur;
      sc->best_is_x0 = 0;
      if (sc->err <= sc->max_error || sc->err / sc->err_init <= sc->max_rel_error) { sc->status = 1; return; }
    } else {
      sc->x_is_best = 0;
    }
    sc->rho_prev = sc->rho_curr;  // set_rho
    sc->rho_curr = red[0];
    sc->rr = red[1];
    begin_iteration(sc);
    break;
  case 5: {
    sc->omega = red[0] / (red[1] + sc->eps);
    sc->omega_r = sc->omega;
    const double w = sc->omega;
    sc->rho_next = red[2] - w * red[3];
    double rr = red[4] - 2.0 * w * red[0];
    rr = rr + (w * w) * red[1];
    rr = rr > 0.0 ? rr : 0.0;
    // ||r'||^2 as a difference of three sums of size ||s||^2 carries an absolute error of a few ulp of those sums: when
    // r' << s (s nearly an eigenvector of A P_inv) the value is round-off, and a spuriously SMALL one could hide a breakdown the
    // reference's directly summed norm (cuda.cu:440-447) would see.  The test therefore runs on the upper end of the error
    // interval: it restarts whenever the exact norm would, and in addition only where |rho'| is below the round-off of its
    // own two sums (~1e-16 ||rhat|| ||s||), i.e. where rho' carries no information and a restart -- which takes rho from
    // r'.r' summed cell by cell in the next launch -- is the only sound continuation.  For ||r'|| >= 1e-5 ||s|| the margin is
    // below 1e-3 of rr: the decision of the directly summed form.  tests/test_two_launch_recurrence.py pins both regimes.
    const double ww = w * w;
    const double wts = w * red[0];
    const double margin = 1e-14 * (red[4] + 2.0 * (wts < 0.0 ? -wts : wts) + ww * red[1]);  // ~45 ulp of the terms: tree sums of 1e7 cells
    const double rr_hi = rr + margin;
    const bool serious_breakdown = sc->rho_next * sc->rho_next < 1e-16 * rr_hi * sc->rhat2;
    sc->beta = (sc->rho_next / (sc->rho_curr + sc->eps)) * (sc->alpha / (sc->omega + sc->eps));  // set_beta, rho_prev = rho_curr by then
    sc->restart_flag = serious_breakdown && sc->max_restarts > 0 ? 1 : 0;
    break;
  }
  case 4:
    sc->iter++;
    sc->err = red[2];
    sc->ycur = y_out_buffer(sc->ycur, sc->ybest);
    if (sc->err < sc->err_opt) {
      sc->err_opt = sc->err;
      sc->x_is_best = 1;
      sc->ybest = sc->ycur;
      sc->best_is_x0 = 0;
      if (sc->err <= sc->max_error || sc->err / sc->err_init <= sc->max_rel_error) { sc->status = 1; return; }
    } else {
      sc->x_is_best = 0;
    }
    sc->rho_prev = sc->rho_curr;  // set_rho
    sc->rho_curr = sc->rho_next;
    sc->rr = red[1];
    // begin_iteration with the decision of stage 5 (beta is set)
    if (sc->iter >= sc->max_iter) { sc->status = 3; return; }
    if (sc->restart_flag) {
      sc->restarts++;
      if (sc->restarts >= sc->max_restarts) { sc->status = 2; return; }
      sc->rhat2 = sc->rr;
      sc->rho_curr = sc->rr;
      sc->rho_prev = 1.; sc->alpha = 1.; sc->omega = 1.;
      sc->beta = (sc->rho_curr / (sc->rho_prev + sc->eps)) * (sc->alpha / (sc->omega + sc->eps));
    }
    sc->alpha = sc->rho_curr / (red[0] + sc->eps);  // stage 1
    break;
  }
}

}  // namespace cup2d
