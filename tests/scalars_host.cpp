// scalars_host.cpp -- TEST INFRASTRUCTURE.  The scalar recurrences of the BiCGSTAB organisations exactly as the kernels run
// them (cup2d_amd/csrc/krylov_scalars.h is host+device), behind a C interface for tests/test_two_launch_recurrence.py.
// Built on demand: g++ -O1 -ffp-contract=off -shared -fPIC.
#include <string.h>

#include "../cup2d_amd/csrc/krylov_scalars.h"

using cup2d::KrylovScalars;

extern "C" {
int sc_size() { return (int)sizeof(KrylovScalars); }
// solve_fused_impl's initial state (krylov_fused.hip)
void sc_init(void *p, double max_error, double max_rel_error, int max_restarts, int max_iter) {
  KrylovScalars init;
  memset(&init, 0, sizeof init);
  init.alpha = init.beta = init.omega = init.omega_r = init.rho_prev = init.rho_curr = 1.0;
  init.eps = 1e-21;
  init.err = init.err_init = init.err_opt = 1e50;
  init.max_error = max_error; init.max_rel_error = max_rel_error;
  init.max_restarts = max_restarts; init.max_iter = max_iter;
  memcpy(p, &init, sizeof init);
}
void sc_update(void *p, const double *red, int stage) { cup2d::scalars_update(static_cast<KrylovScalars *>(p), red, stage); }
int sc_y_out_buffer(int cur, int best) { return cup2d::y_out_buffer(cur, best); }
// out: alpha, omega, beta, rho_curr, rho_prev, rr, rhat2, err, err_init, err_opt | iout: status, iter, restarts, restart_flag,
// ycur, ybest, best_is_x0
void sc_get(const void *p, double *out, int *iout) {
  const KrylovScalars *s = static_cast<const KrylovScalars *>(p);
  const double d[10] = {s->alpha, s->omega, s->beta, s->rho_curr, s->rho_prev, s->rr, s->rhat2, s->err, s->err_init, s->err_opt};
  const int i[7] = {s->status, s->iter, s->restarts, s->restart_flag, s->ycur, s->ybest, s->best_is_x0};
  memcpy(out, d, sizeof d);
  memcpy(iout, i, sizeof i);
}
}
