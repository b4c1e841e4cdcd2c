/* Minimal stand-in for the GNU Scientific Library symbols that
 * /root/reference/main.cpp names (main.cpp:6692-6703: a 3x3 LU solve in the
 * rigid-body penalisation, reached only when a body exists).  GSL is not
 * installed in this image; the oracle harness never runs with bodies, but the
 * translation unit must compile.  This is TEST INFRASTRUCTURE, not product.
 * The functions are nevertheless functional (plain partial-pivot LU). */
#ifndef CUP2D_ORACLE_GSL_STUB_H
#define CUP2D_ORACLE_GSL_STUB_H
#include <cmath>
#include <cstdlib>
struct gsl_matrix { size_t n1, n2; double *data; };
struct gsl_vector { size_t n; double *data; bool owner; };
struct gsl_permutation { size_t n; size_t *p; };
struct gsl_matrix_view { gsl_matrix matrix; };
struct gsl_vector_view { gsl_vector vector; };
static inline gsl_matrix_view gsl_matrix_view_array(double *a, size_t n1, size_t n2) {
  gsl_matrix_view v; v.matrix.n1 = n1; v.matrix.n2 = n2; v.matrix.data = a; return v;
}
static inline gsl_vector_view gsl_vector_view_array(double *a, size_t n) {
  gsl_vector_view v; v.vector.n = n; v.vector.data = a; v.vector.owner = false; return v;
}
static inline gsl_vector *gsl_vector_alloc(size_t n) {
  gsl_vector *v = (gsl_vector *)malloc(sizeof *v);
  v->n = n; v->data = (double *)calloc(n, sizeof(double)); v->owner = true; return v;
}
static inline void gsl_vector_free(gsl_vector *v) { if (v) { if (v->owner) free(v->data); free(v); } }
static inline double gsl_vector_get(const gsl_vector *v, size_t i) { return v->data[i]; }
static inline gsl_permutation *gsl_permutation_alloc(size_t n) {
  gsl_permutation *p = (gsl_permutation *)malloc(sizeof *p);
  p->n = n; p->p = (size_t *)malloc(n * sizeof(size_t));
  for (size_t i = 0; i < n; i++) p->p[i] = i;
  return p;
}
static inline void gsl_permutation_free(gsl_permutation *p) { if (p) { free(p->p); free(p); } }
static inline int gsl_linalg_LU_decomp(gsl_matrix *A, gsl_permutation *p, int *signum) {
  const size_t n = A->n1; double *a = A->data; *signum = 1;
  for (size_t k = 0; k < n; k++) {
    size_t piv = k; double best = std::fabs(a[k * n + k]);
    for (size_t i = k + 1; i < n; i++)
      if (std::fabs(a[i * n + k]) > best) { best = std::fabs(a[i * n + k]); piv = i; }
    if (piv != k) {
      for (size_t j = 0; j < n; j++) { double t = a[k * n + j]; a[k * n + j] = a[piv * n + j]; a[piv * n + j] = t; }
      size_t t = p->p[k]; p->p[k] = p->p[piv]; p->p[piv] = t; *signum = -*signum;
    }
    for (size_t i = k + 1; i < n; i++) {
      a[i * n + k] /= a[k * n + k];
      for (size_t j = k + 1; j < n; j++) a[i * n + j] -= a[i * n + k] * a[k * n + j];
    }
  }
  return 0;
}
static inline int gsl_linalg_LU_solve(const gsl_matrix *LU, const gsl_permutation *p,
                                      const gsl_vector *b, gsl_vector *x) {
  const size_t n = LU->n1; const double *a = LU->data;
  for (size_t i = 0; i < n; i++) x->data[i] = b->data[p->p[i]];
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < i; j++) x->data[i] -= a[i * n + j] * x->data[j];
  for (size_t ii = n; ii-- > 0;) {
    for (size_t j = ii + 1; j < n; j++) x->data[ii] -= a[ii * n + j] * x->data[j];
    x->data[ii] /= a[ii * n + ii];
  }
  return 0;
}
#endif
