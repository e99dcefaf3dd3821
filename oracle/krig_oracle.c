/*
 * krig_oracle.c -- plain C restatement of the reference's global-kriging CPU path.
 * TEST INFRASTRUCTURE ONLY (checker + bench.py's cpu_baseline leg); never linked into the product.
 *
 * Follows /root/reference/src/estimation/krig.jl:166-186 (exactsolve):
 *   :176  krig = fit(estimator, pdata)            -> build LHS = [C F; F' 0], factorise once
 *   :180  predictprob(krig, var, pdomain[ind])    -> per point: RHS = [c0; f0], two triangular
 *                                                    solves against the factor (2 (n+nc)^2 flop),
 *                                                    mu = lambda.z, sigma^2 = max(0, sill - RHS.[lambda;nu])
 * [DEP] GeoStatsModels uses Cholesky (SK) / Bunch-Kaufman (OK); any stable factorisation gives the
 * same weights, here LU with partial pivoting.  The point loop is single threaded exactly like the
 * reference's comprehension at krig.jl:180; `nthreads > 1` optionally splits it with OpenMP.
 * PARITY: pinned only by tests/test_oracle_*.py (reference assertions + KATs); otherwise unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { VG_GAUSSIAN = 0, VG_EXPONENTIAL = 1, VG_SPHERICAL = 2, VG_MATERN = 3 };

typedef struct {
  int kind, dim;
  double sill, nugget, range, nu;
} vg_t;

static double cov_h(const vg_t* v, double h) {
  /* C(h) = sill - gamma(h), gamma = (sill - nugget) f(h / range) + nugget (h > 0)   */
  double x = h / v->range, f;
  switch (v->kind) {
    case VG_GAUSSIAN: f = 1.0 - exp(-3.0 * x * x); break;
    case VG_EXPONENTIAL: f = 1.0 - exp(-3.0 * x); break;
    case VG_SPHERICAL: f = x < 1.0 ? 1.5 * x - 0.5 * x * x * x : 1.0; break;
    default: {
      double d = sqrt(2.0 * v->nu) * 3.0 * x;
      if (v->nu == 0.5) f = 1.0 - exp(-d);
      else if (v->nu == 1.5) f = 1.0 - (1.0 + d) * exp(-d);
      else f = 1.0 - (1.0 + d + d * d / 3.0) * exp(-d);
    }
  }
  return v->sill - ((v->sill - v->nugget) * f + (h > 0.0 ? v->nugget : 0.0));
}

static double dist(const double* a, const double* b, int dim) {
  double s = 0.0;
  for (int k = 0; k < dim; ++k) s += (a[k] - b[k]) * (a[k] - b[k]);
  return sqrt(s);
}

/* variant: 0 = simple (mean given), 1 = ordinary.  Returns 0 on success. */
int krig_oracle_global(int kind, int dim, double sill, double nugget, double range, double nu, int variant,
                       double sk_mean, const double* x, const double* z, int64_t n, const double* x0, int64_t m,
                       double* mean_out, double* var_out, int nthreads) {
  vg_t v = {kind, dim, sill, nugget, range, nu};
  const int64_t nc = variant == 1 ? 1 : 0, N = n + nc;
  double* A = (double*)calloc((size_t)(N * N), sizeof(double)); /* row-major LU in place */
  int64_t* piv = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
  if (!A || !piv) return 1;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) A[i * N + j] = cov_h(&v, dist(x + i * dim, x + j * dim, dim));
  if (nc)
    for (int64_t i = 0; i < n; ++i) A[i * N + n] = A[n * N + i] = 1.0;
  /* fit: LU with partial pivoting (krig.jl:176) */
  for (int64_t k = 0; k < N; ++k) {
    int64_t p = k;
    for (int64_t i = k + 1; i < N; ++i)
      if (fabs(A[i * N + k]) > fabs(A[p * N + k])) p = i;
    piv[k] = p;
    if (A[p * N + k] == 0.0) { free(A); free(piv); return 2; }
    if (p != k)
      for (int64_t j = 0; j < N; ++j) { double t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t; }
    for (int64_t i = k + 1; i < N; ++i) {
      const double l = A[i * N + k] /= A[k * N + k];
      for (int64_t j = k + 1; j < N; ++j) A[i * N + j] -= l * A[k * N + j];
    }
  }
  /* predict: one RHS + two triangular solves per point (krig.jl:180) */
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    double* rhs = (double*)malloc(sizeof(double) * (size_t)N);
    double* w = (double*)malloc(sizeof(double) * (size_t)N);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t p = 0; p < m; ++p) {
      for (int64_t i = 0; i < n; ++i) rhs[i] = cov_h(&v, dist(x + i * dim, x0 + p * dim, dim));
      if (nc) rhs[n] = 1.0;
      memcpy(w, rhs, sizeof(double) * (size_t)N);
      for (int64_t k = 0; k < N; ++k)
        if (piv[k] != k) { double t = w[k]; w[k] = w[piv[k]]; w[piv[k]] = t; }
      for (int64_t i = 1; i < N; ++i) {
        double s = w[i];
        const double* row = A + i * N;
        for (int64_t j = 0; j < i; ++j) s -= row[j] * w[j];
        w[i] = s;
      }
      for (int64_t i = N - 1; i >= 0; --i) {
        double s = w[i];
        const double* row = A + i * N;
        for (int64_t j = i + 1; j < N; ++j) s -= row[j] * w[j];
        w[i] = s / row[i];
      }
      double mu = variant == 0 ? sk_mean : 0.0, c = 0.0;
      for (int64_t i = 0; i < n; ++i) mu += w[i] * (variant == 0 ? z[i] - sk_mean : z[i]);
      for (int64_t i = 0; i < N; ++i) c += rhs[i] * w[i];
      double s2 = sill - c;
      mean_out[p] = mu;
      var_out[p] = s2 > 0.0 ? s2 : 0.0;
    }
    free(rhs);
    free(w);
  }
  free(A);
  free(piv);
  return 0;
}
