/*
 * aberth.c — oracle restatement of the third-party crate `aberth` 0.4.1
 * (Cargo.lock), as called at src/initial_orbit_determination/gauss.rs:655,971:
 *     aberth::aberth(&[f64; 9], max_iterations, epsilon) -> Roots{roots, stop_reason}
 * The crate's source is not under /root/reference; this follows its published
 * algorithm (Aberth–Ehrlich simultaneous iteration with Aberth's initial
 * guesses, Jacobi-style sweeps, complex arithmetic as in num-complex 0.4) and
 * is pinned by the reference's KAT gauss.rs:1543-1570 (root values AND order).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include "oracle.h"

/* 1 = Horner steps use num-complex's MulAdd (fused), 0 = plain acc*x + c. */
int orc_aberth_horner_fma = 1;

typedef struct { double re, im; } cplx;

static inline cplx c_add(cplx a, cplx b) { return (cplx){a.re + b.re, a.im + b.im}; }
static inline cplx c_sub(cplx a, cplx b) { return (cplx){a.re - b.re, a.im - b.im}; }
static inline cplx c_mul(cplx a, cplx b) {
  return (cplx){a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
/* num-complex Div: (a * conj(b)) / |b|^2 */
static inline cplx c_div(cplx a, cplx b) {
  double n = b.re * b.re + b.im * b.im;
  double re = a.re * b.re + a.im * b.im;
  double im = a.im * b.re - a.re * b.im;
  return (cplx){re / n, im / n};
}
/* num-complex MulAdd for Complex<T> */
static inline cplx c_mul_add(cplx a, cplx b, cplx add) {
  double re = fma(a.re, b.re, add.re) - (a.im * b.im);
  double im = fma(a.re, b.im, fma(a.im, b.re, add.im));
  return (cplx){re, im};
}

static cplx horner(const double *coef, int n_terms, cplx x) {
  cplx r = {coef[n_terms - 1], 0.0};
  for (int i = n_terms - 2; i >= 0; --i) {
    cplx c = {coef[i], 0.0};
    r = orc_aberth_horner_fma ? c_mul_add(r, x, c) : c_add(c_mul(r, x), c);
  }
  return r;
}

int orc_aberth9(const double poly[9], uint32_t max_iter, double eps, double re[8], double im[8],
                uint32_t *iterations) {
  enum { N = 8, TERMS = 9 };
  /* --- initial guesses (Aberth 1973): monic form, centre a = -c_{n-1}/n,
   *     shifted polynomial via Pascal rows, S(w) = w^n - sum |c_i| w^i,
   *     r0 = first positive integer with S(r0) > 0,
   *     z_k = a + r0 * exp(i (2 pi k / n + pi / (2n))). */
  double monic[TERMS];
  for (int i = 0; i < TERMS; ++i) monic[i] = poly[i] / poly[N];
  double a = -monic[N - 1] / (double)N;
  for (int ci = 0; ci <= N; ++ci) {
    double c = monic[ci];
    monic[ci] = 0.0;
    /* Pascal row ci: binom(ci, j), paired with index j and power ci - j */
    double binom = 1.0;
    for (int j = 0; j <= ci; ++j) {
      int power = ci - j;
      double ap = 1.0; /* a.powi(power) */
      for (int t = 0; t < power; ++t) ap *= a;
      monic[j] = fma(c, binom * ap, monic[j]);
      binom = binom * (double)(ci - j) / (double)(j + 1);
    }
  }
  double s_of_w[TERMS];
  for (int i = 0; i < N; ++i) s_of_w[i] = -fabs(monic[i]);
  s_of_w[N] = monic[N];
  double r0 = 1.0;
  for (int guard = 0; guard < 4096; ++guard) { /* the crate loops without a bound; 4096 (r0 up to 4097: |c6| < 1.6e7) here and on the device alike */
    cplx s = horner(s_of_w, TERMS, (cplx){r0, 0.0});
    if (s.re > 0.0) break;
    r0 += 1.0;
  }
  cplx zs[N], nz[N];
  const double frac_2pi_n = ORC_DPI / (double)N;
  const double frac_pi_2n = (ORC_PI / 2.0) / (double)N;
  for (int k = 0; k < N; ++k) {
    double theta = fma(frac_2pi_n, (double)k, frac_pi_2n);
    double ct = orc_aberth_cos_sin[k][0], st = orc_aberth_cos_sin[k][1];
    if (orc_use_libm || (orc_libm_mask & 32)) sincos(theta, &st, &ct); /* both results used: stays one sincos call */ /* theta.cos(), theta.sin(): one sincos call */
    zs[k].re = fma(r0, ct, a);
    zs[k].im = r0 * st;
  }
  double dydx[N];
  for (int i = 1; i <= N; ++i) dydx[i - 1] = (double)i * poly[i];

  for (uint32_t it = 1; it <= max_iter; ++it) {
    int converged = 1;
    for (int i = 0; i < N; ++i) {
      cplx p = horner(poly, TERMS, zs[i]);
      cplx dp = horner(dydx, N, zs[i]);
      cplx sum = {0.0, 0.0};
      for (int k = 0; k < N; ++k) {
        if (k == i) continue;
        sum = c_add(sum, c_div((cplx){1.0, 0.0}, c_sub(zs[i], zs[k])));
      }
      cplx den = c_sub(c_mul(p, sum), dp);
      cplx z = c_add(zs[i], c_div(p, den));
      nz[i] = z;
      if (isnan(z.re) || isnan(z.im) || isinf(z.re) || isinf(z.im)) {
        for (int k = 0; k < N; ++k) { re[k] = nz[k].re; im[k] = nz[k].im; }
        if (iterations) *iterations = it;
        return ORC_ABERTH_FAILED;
      }
      if (!(fabs(z.re - zs[i].re) < eps && fabs(z.im - zs[i].im) < eps)) converged = 0;
    }
    if (converged) {
      for (int k = 0; k < N; ++k) { re[k] = nz[k].re; im[k] = nz[k].im; }
      if (iterations) *iterations = it;
      return ORC_ABERTH_CONVERGED;
    }
    for (int k = 0; k < N; ++k) zs[k] = nz[k];
  }
  for (int k = 0; k < N; ++k) { re[k] = zs[k].re; im[k] = zs[k].im; }
  if (iterations) *iterations = max_iter;
  return ORC_ABERTH_MAXITER;
}
