/*
 * kepler.c — oracle restatement of src/kepler/ (universal-variable two-body).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

static inline double dot3(const double a[3], const double b[3]) {
  /* nalgebra dot, U3xU1 special case: (a0*b0 + a1*b1) + a2*b2 */
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static inline double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
static inline void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double clampd(double x, double lo, double hi) {
  /* f64::clamp */
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
static inline double powi2(double x) { return x * x; }
static inline double powi3(double x) { return x * x * x; }

/* kepler/angles.rs:10-12 — f64::rem_euclid(DPI) */
double orc_principal_angle(double a) {
  double r = fmod(a, ORC_DPI);
  return (r < 0.0) ? r + ORC_DPI : r;
}

/* kepler/stumpff.rs:135-191 */
static void stumpff_series(double psi, double psi2, double beta, double alpha, double s[4]) {
  const double tol = 100.0 * ORC_EPS, ovf = 1.0 / ORC_EPS;
  double s2 = 0.5 * psi2, t2 = s2;
  double s3 = (s2 * psi) / 3.0, t3 = s3;
  double d2l = 3.0, d2h = 4.0, d3l = 4.0, d3h = 5.0;
  for (int j = 1; j <= 70; ++j) {
    t2 *= beta / (d2l * d2h);
    s2 += t2;
    t3 *= beta / (d3l * d3h);
    s3 += t3;
    int n2 = fabs(t2) < tol, n3 = fabs(t3) < tol;
    int o2 = fabs(t2) > ovf, o3 = fabs(t3) > ovf;
    if ((n2 && n3) || o2 || o3) break;
    d2l += 2.0; d2h += 2.0; d3l += 2.0; d3h += 2.0;
  }
  s[1] = psi + alpha * s3;
  s[0] = 1.0 + alpha * s2;
  s[2] = s2;
  s[3] = s3;
}

/* kepler/stumpff.rs:201-297 */
static void stumpff_halving(double psi, double beta, double alpha, double s[4]) {
  const double tol = 100.0 * ORC_EPS, ovf = 1.0 / ORC_EPS;
  double rp = psi, rb = beta;
  int nh = 0;
  while (fabs(rb) >= 100.0 && nh < 30) { rp *= 0.5; rb *= 0.25; ++nh; }
  double s0 = 1.0, s1 = rp, t0 = 1.0, t1 = rp;
  for (int j = 1; j <= 70; ++j) {
    t0 *= rb / ((double)(2 * j - 1) * (double)(2 * j));
    s0 += t0;
    if (fabs(t0) < tol || fabs(t0) > ovf) break;
  }
  for (int j = 1; j <= 70; ++j) {
    t1 *= rb / ((double)(2 * j) * (double)(2 * j + 1));
    s1 += t1;
    if (fabs(t1) < tol || fabs(t1) > ovf) break;
  }
  for (int i = 0; i < nh; ++i) {
    double c = s0, sn = s1;
    s0 = 2.0 * c * c - 1.0;
    s1 = 2.0 * c * sn;
  }
  s[0] = s0; s[1] = s1;
  s[3] = (s1 - psi) / alpha;
  s[2] = (s0 - 1.0) / alpha;
}

/* kepler/stumpff.rs:78-126 */
void orc_s_funct(double psi, double alpha, double s[4]) {
  if (psi == 0.0) { s[0] = 1.0; s[1] = s[2] = s[3] = 0.0; return; }
  double psi2 = psi * psi, beta = alpha * psi2;
  if (fabs(beta) < 100.0) stumpff_series(psi, psi2, beta, alpha, s);
  else stumpff_halving(psi, beta, alpha, s);
}

/* kepler/prelim_kepler/prelim_elliptic.rs:9-31 */
static double initial_ecc_anomaly(double r0, double a0, double e0, double sig0) {
  double c = (1.0 - r0 / a0) / e0;
  double u;
  if (fabs(c) <= 1.0) u = orc_acos(c);
  else if (c >= 1.0) u = 0.0;
  else u = ORC_PI;
  if (sig0 < 0.0) u = -u;
  return orc_principal_angle(u);
}

/* kepler/prelim_kepler/prelim_elliptic.rs:72-134 */
double orc_prelim_elliptic(const orc_kepler_params *p) {
  double contr = p->convergency;
  double a0 = -1.0 / p->alpha;
  double n = sqrt(p->mu) * sqrt(-powi3(p->alpha));
  if (p->e0 < contr) return n * p->dt / sqrt(-p->alpha);
  double u0 = initial_ecc_anomaly(p->r0, a0, p->e0, p->sig0);
  double m0 = orc_principal_angle(u0 - p->e0 * orc_sin1(u0)); /* isolated sin: prelim_elliptic.rs:97 */
  double m = m0 + n * p->dt;
  double u = m;
  for (int i = 0; i < p->max_iter_prelim_kepuni; ++i) {
    double su, cu;
    orc_sincos(u, &su, &cu);
    double res = u - p->e0 * su - m;
    double der = 1.0 - p->e0 * cu;
    double step = -res / der;
    u += step;
    if (fabs(step) < contr * 1e3) break;
  }
  return (u - u0) / sqrt(-p->alpha);
}

/* kepler/prelim_kepler/prelim_hyperbolic.rs:45-141 */
double orc_prelim_hyperbolic(const orc_kepler_params *p) {
  double a0 = -1.0 / p->alpha;
  double n = sqrt(p->mu) * sqrt(powi3(p->alpha));
  double c = (1.0 - p->r0 / a0) / p->e0;
  double f0 = (c > 1.0) ? orc_log(c + sqrt(powi2(c) - 1.0)) : 0.0; /* log/sinh/cosh correctly rounded, crmath.c */
  if (p->sig0 < 0.0) f0 = -f0;
  double m = (p->e0 * orc_sinh(f0) - f0) + n * p->dt;
  double f = 0.0;
  for (int i = 0; i < p->max_iter_prelim_kepuni; ++i) {
    if (fabs(f) < 15.0) {
      double res = p->e0 * orc_sinh(f) - f - m;
      double der = p->e0 * orc_cosh(f) - 1.0;
      double step = -res / der;
      double cand = f + step;
      f = (f * cand < 0.0) ? f / 2.0 : cand;
    } else {
      f /= 2.0;
    }
    if (fabs(f) < p->convergency * 1e3) break; /* quirk Q5: tests F, not the step */
  }
  return (f - f0) / sqrt(p->alpha);
}

/* kepler/prelim_kepler/prelim_parabolic.rs:149-165 */
static void cubic_res_der(double psi, double r0, double sig0, double smdt, double *res, double *der) {
  *res = powi3(psi) / 6.0 + sig0 / 2.0 * powi2(psi) + r0 * psi - smdt;
  *der = powi2(psi) / 2.0 + sig0 * psi + r0;
}

static double prelim_parabolic_cardano(const orc_kepler_params *p);

/* kepler/prelim_kepler/prelim_parabolic.rs:198-246 (ParabolicPrelimMethod::Newton): safeguarded Newton on the cubic,
 * only when it is monotonic (:171-173), else Cardano */
static double prelim_parabolic_newton(const orc_kepler_params *p) {
  double smdt = sqrt(p->mu) * p->dt;
  if (p->dt == 0.0) return 0.0;
  if (p->sig0 * p->sig0 > 2.0 * p->r0) return prelim_parabolic_cardano(p);
  double psi = smdt / p->r0;
  for (int i = 0; i < p->max_iter_prelim_kepuni; ++i) {
    double res, der;
    cubic_res_der(psi, p->r0, p->sig0, smdt, &res, &der);
    if (!isfinite(der) || fabs(der) < 10.0 * ORC_EPS) { psi *= 0.5; continue; }
    double raw = -res / der, mx = 2.0 * (1.0 + fabs(psi));
    double step = clampd(raw, -mx, mx);
    psi += step;
    if (fabs(step) < p->convergency) break;
  }
  return psi;
}

/* kepler/prelim_kepler/prelim_parabolic.rs:120-130 */
double orc_prelim_parabolic(const orc_kepler_params *p) {
  return p->parabolic_method == 1 ? prelim_parabolic_newton(p) : prelim_parabolic_cardano(p);
}

/* kepler/prelim_kepler/prelim_parabolic.rs:264-477 (Cardano, the default method) */
static double prelim_parabolic_cardano(const orc_kepler_params *p) {
  double r0 = p->r0, sig0 = p->sig0, dt = p->dt;
  double smdt = sqrt(p->mu) * dt;
  if (dt == 0.0) return 0.0;
  double lead = 1.0 / 6.0;
  double qc = (sig0 / 2.0) / lead;
  double lc = r0 / lead;
  double cc = -smdt / lead;
  double shift = qc / 3.0;
  double dl = lc - qc * shift;
  double dc = 2.0 * powi3(shift) - lc * shift + cc;
  double half = dc / 2.0;
  double disc = half * half + powi3(dl / 3.0);
  double cand[3];
  int nc;
  if (disc > 0.0) {
    double sq = sqrt(disc);
    cand[0] = (cbrt(-half + sq) + cbrt(-half - sq)) - shift;
    nc = 1;
  } else {
    double arg = clampd((3.0 * dc) / (2.0 * dl) * sqrt(-3.0 / dl), -1.0, 1.0);
    double base = acos(arg) / 3.0;
    double amp = 2.0 * sqrt(-dl / 3.0);
    cand[0] = amp * cos(base) - shift;
    cand[1] = amp * cos(base - 2.0 * ORC_PI / 3.0) - shift;
    cand[2] = amp * cos(base - 4.0 * ORC_PI / 3.0) - shift;
    nc = 3;
  }
  /* select_physical_root :438-477 — min_by keeps the FIRST minimum */
  double lin = smdt / r0;
  double mono[3];
  int nm = 0;
  for (int i = 0; i < nc; ++i) {
    double res, der;
    cubic_res_der(cand[i], r0, sig0, smdt, &res, &der);
    if (der >= 0.0) mono[nm++] = cand[i];
  }
  const double *set = nm ? mono : cand;
  int ns = nm ? nm : nc;
  double sel = set[0];
  for (int i = 1; i < ns; ++i)
    if (fabs(set[i] - lin) < fabs(sel - lin)) sel = set[i];
  /* polish_root_by_newton :371-393 */
  double psi = sel;
  for (int i = 0; i < 2; ++i) {
    double res, der;
    cubic_res_der(psi, r0, sig0, smdt, &res, &der);
    if (der == 0.0 || !isfinite(der)) break;
    psi -= res / der;
  }
  return psi;
}

/* kepler/params.rs:185-191 + kepler/orbit_type.rs:32-40 */
double orc_prelim_kepuni(const orc_kepler_params *p) {
  if (p->alpha < 0.0) return orc_prelim_elliptic(p);
  if (p->alpha > 0.0) return orc_prelim_hyperbolic(p);
  return orc_prelim_parabolic(p);
}

/* kepler/newton_solver.rs:151-161,240-352 */
int orc_solve_kepuni_with_guess(const orc_kepler_params *p, orc_kepler_solution *sol) {
  double smdt = sqrt(p->mu) * p->dt;
  double res_tol = 10.0 * ORC_EPS * (1.0 + fabs(smdt));
  double psi = p->has_guess ? p->psi_guess : orc_prelim_kepuni(p);
  for (int it = 0; it < 50; ++it) {
    if (!isfinite(psi)) { psi = 0.5; continue; }
    double s[4];
    orc_s_funct(psi, p->alpha, s);
    double residual = p->r0 * s[1] + p->sig0 * s[2] + s[3] - sqrt(p->mu) * p->dt;
    double derivative = p->r0 * s[0] + p->sig0 * s[1] + s[2];
    if (fabs(residual) <= res_tol) {
      sol->psi = psi; sol->s0 = s[0]; sol->s1 = s[1]; sol->s2 = s[2]; sol->s3 = s[3];
      return 1;
    }
    if (!isfinite(derivative) || fabs(derivative) < 10.0 * ORC_EPS) { psi *= 0.5; continue; }
    double raw = -residual / derivative;
    double mx = 2.0 * (1.0 + fabs(psi));
    double step = clampd(raw, -mx, mx);
    double cand = psi + step;
    psi = (cand * psi < 0.0) ? 0.5 * psi : cand;
    double sa = fabs(step);
    if (sa <= p->convergency) { /* quirk Q3: new psi, previous Stumpff values */
      sol->psi = psi; sol->s0 = s[0]; sol->s1 = s[1]; sol->s2 = s[2]; sol->s3 = s[3];
      return 1;
    }
    if (sa <= p->convergency * (1.0 + fabs(psi))) {
      orc_s_funct(psi, p->alpha, s);
      sol->psi = psi; sol->s0 = s[0]; sol->s1 = s[1]; sol->s2 = s[2]; sol->s3 = s[3];
      return 1;
    }
  }
  return 0;
}

/* kepler/brent_dekker_solver.rs:59-62 */
static double kep_residual(double psi, const orc_kepler_params *p) {
  double s[4];
  orc_s_funct(psi, p->alpha, s);
  return p->r0 * s[1] + p->sig0 * s[2] + s[3] - sqrt(p->mu) * p->dt;
}

/* kepler/brent_dekker_solver.rs:150-177,469-526 */
int orc_solve_kepuni_brent_dekker(const orc_kepler_params *p, orc_kepler_solution *sol) {
  const double PHI = 1.618033988749895;
  double psi0 = p->has_guess ? p->psi_guess : orc_prelim_kepuni(p);
  double hw = (fabs(psi0) > 1.0) ? fabs(psi0) : 1.0;
  double lo = psi0 - hw, hi = psi0 + hw;
  double flo = kep_residual(lo, p), fhi = kep_residual(hi, p);
  int ok = 0;
  for (int i = 0; i < 60; ++i) {
    if (flo * fhi <= 0.0) { ok = 1; break; }
    double w = hi - lo;
    if (fabs(flo) < fabs(fhi)) { lo = lo - PHI * w; flo = kep_residual(lo, p); }
    else { hi = hi + PHI * w; fhi = kep_residual(hi, p); }
  }
  if (!ok) return 0;
  /* BrentState::from_bracket :214-233 */
  double a = lo, fa = kep_residual(lo, p), b = hi, fb = kep_residual(hi, p);
  if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  double c = a, fc = fa, prev_len = fabs(hi - lo);
  int prev_bis = 1;
  double tol = p->convergency;
  for (int it = 0; it < 100; ++it) {
    if (fabs(fb) <= tol || 0.5 * fabs(b - a) <= tol) {
      double s[4];
      orc_s_funct(b, p->alpha, s);
      sol->psi = b; sol->s0 = s[0]; sol->s1 = s[1]; sol->s2 = s[2]; sol->s3 = s[3];
      return 1;
    }
    double interp;
    if (fabs(fa - fc) > ORC_EPS && fabs(fb - fc) > ORC_EPS) {
      double ta = a * fb * fc / ((fa - fb) * (fa - fc));
      double tb = b * fa * fc / ((fb - fa) * (fb - fc));
      double tc = c * fa * fb / ((fc - fa) * (fc - fb));
      interp = ta + tb + tc;
    } else {
      interp = b - fb * (b - a) / (fb - fa);
    }
    double ref_len = prev_bis ? fabs(b - c) : prev_len;
    double q3 = (3.0 * a + b) / 4.0;
    int inside = (q3 < b) ? (interp > q3 && interp < b) : (interp > b && interp < q3);
    int progress = fabs(interp - b) < 0.5 * ref_len;
    double next;
    int was_bis;
    if (inside && progress) { next = interp; was_bis = 0; }
    else { next = 0.5 * (a + b); was_bis = 1; }
    double fnext = kep_residual(next, p);
    prev_len = fabs(b - c);
    prev_bis = was_bis;
    c = b; fc = fb;
    if (fa * fnext < 0.0) { b = next; fb = fnext; }
    else { a = next; fa = fnext; }
    if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  }
  return 0;
}

/* kepler/params.rs:130-142 */
int orc_kepler_solve(const orc_kepler_params *p, orc_kepler_solution *sol) {
  if (p->kind == 0) return orc_solve_kepuni_with_guess(p, sol);
  if (p->kind == 1) return orc_solve_kepuni_brent_dekker(p, sol);
  if (orc_solve_kepuni_with_guess(p, sol)) return 1;
  return orc_solve_kepuni_brent_dekker(p, sol);
}

/* orb_elem.rs:257-301 */
int orc_eccentricity_control(const double r[3], const double v[3], double peri_max, double ecc_max,
                             int *accepted, double *ecc, double *peri, double *energy) {
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double v2 = dot3(v, v);
  double dist = norm3(r);
  double h[3];
  cross3(r, v, h);
  double h2 = dot3(h, h);
  if (sqrt(h2) == 0.0) return 0;
  double vxh[3];
  cross3(v, h, vxh);
  double inv_mu = 1.0 / mu, inv_d = 1.0 / dist;
  double lenz[3];
  for (int i = 0; i < 3; ++i) lenz[i] = vxh[i] * inv_mu - r[i] * inv_d;
  double e = norm3(lenz);
  double q = h2 / (mu * (1.0 + e));
  double en = v2 / 2.0 - mu / dist;
  *accepted = (e < ecc_max) && (q < peri_max);
  *ecc = e; *peri = q; *energy = en;
  return 1;
}

/* kepler/velocity.rs:94-211 */
int orc_velocity_correction_with_guess(const double x1[3], const double x2[3], const double v2[3],
                                       double dt, double peri_max, double ecc_max, int has_guess,
                                       double chi_guess, double eps, double v_out[3], double *f,
                                       double *g, double *chi) {
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double r2 = norm3(x2);
  double sig0 = dot3(x2, v2) / sqrt(mu);
  double h[3];
  cross3(x2, v2, h);
  double hn = norm3(h);
  if (!isfinite(hn) || hn <= 1e6 * ORC_EPS) return 1;
  int acc;
  double ecc, peri, energy;
  if (!orc_eccentricity_control(x2, v2, peri_max, ecc_max, &acc, &ecc, &peri, &energy)) return 2;
  orc_kepler_params p;
  memset(&p, 0, sizeof p);
  p.dt = dt; p.r0 = r2; p.sig0 = sig0; p.mu = mu;
  p.alpha = 2.0 * energy / mu;
  p.e0 = ecc;
  p.convergency = eps;
  p.has_guess = has_guess; p.psi_guess = chi_guess;
  p.max_iter_prelim_kepuni = 20;
  p.kind = 0;
  orc_kepler_solution sol;
  if (!orc_kepler_solve(&p, &sol)) return 3;
  double fc = 1.0 - sol.s2 / r2;
  double gc = dt - sol.s3 / sqrt(mu);
  double ga = fabs(gc);
  double gmin = 100.0 * ORC_EPS * (1.0 + fabs(dt));
  if (!isfinite(ga) || ga < gmin) return 4;
  for (int i = 0; i < 3; ++i) {
    /* axpy(-f, x2, 1.0): (-f * x2) * 1 + 1 * x1 ; then unscale by g */
    double t = (-fc) * x2[i] + x1[i];
    v_out[i] = t / gc;
  }
  *f = fc; *g = gc; *chi = sol.psi;
  return 0;
}

/* kepler/propagation.rs:114-207 */
int orc_propagate_universal(const double r[3], const double v[3], double t0, double t1,
                            double convergency, int kind, double out[11]) {
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double r0 = norm3(r);
  if (r0 < ORC_EPS) return 1;
  double v2 = dot3(v, v);
  double sig0 = dot3(r, v) / sqrt(mu);
  double alpha = (v2 - 2.0 * mu / r0) / mu;
  double h[3];
  cross3(r, v, h);
  double h2 = dot3(h, h);
  double e = sqrt(1.0 + alpha * h2 / mu);
  if (!(e >= 0.0)) e = 0.0; /* f64::max(NaN, 0) = 0 */
  orc_kepler_params p;
  memset(&p, 0, sizeof p);
  p.r0 = r0; p.sig0 = sig0; p.mu = mu; p.alpha = alpha; p.dt = t1 - t0; p.e0 = e;
  p.convergency = convergency; p.has_guess = 0; p.max_iter_prelim_kepuni = 20; p.kind = kind;
  return orc_propagate_with_params(r, v, &p, out);
}

/* the same with the full SolverType (kepler/params.rs:24-73): kind, convergency, psi_guess, max_iter_prelim_kepuni,
 * parabolic_solving_method are read from *st; status 2 = NewtonRaphsonKeplerConvergence, 4 =
 * BrentDekkerKeplerConvergence (params.rs:130-142) */
int orc_propagate_universal_ex(const double r[3], const double v[3], double t0, double t1,
                               const outfit_solver_type_t *st, double out[11]) {
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double r0 = norm3(r);
  if (r0 < ORC_EPS) return 1;
  double v2 = dot3(v, v);
  double h[3];
  cross3(r, v, h);
  double alpha = (v2 - 2.0 * mu / r0) / mu;
  double e = sqrt(1.0 + alpha * dot3(h, h) / mu);
  if (!(e >= 0.0)) e = 0.0;
  orc_kepler_params p;
  memset(&p, 0, sizeof p);
  p.r0 = r0; p.sig0 = dot3(r, v) / sqrt(mu); p.mu = mu; p.alpha = alpha; p.dt = t1 - t0; p.e0 = e;
  p.convergency = st->convergency; p.has_guess = st->has_psi_guess; p.psi_guess = st->psi_guess;
  p.max_iter_prelim_kepuni = st->max_iter_prelim_kepuni > 0x7fffffff ? 0x7fffffff : (int)st->max_iter_prelim_kepuni;
  p.kind = st->kind; p.parabolic_method = st->parabolic_solving_method;
  return orc_propagate_with_params(r, v, &p, out);
}

int orc_propagate_with_params(const double r[3], const double v[3], const orc_kepler_params *pp, double out[11]) {
  const orc_kepler_params p = *pp;
  const double mu = p.mu, r0 = p.r0, sig0 = p.sig0;
  orc_kepler_solution s;
  if (!orc_kepler_solve(&p, &s)) return p.kind == 0 ? 2 : 4;
  double sqrt_mu = sqrt(mu);
  double r1 = r0 * s.s0 + sig0 * s.s1 + s.s2;
  if (r1 < ORC_EPS) return 3;
  double f = 1.0 - s.s2 / r0;
  double g = (r0 * s.s1 + sig0 * s.s2) / sqrt_mu;
  double fd = -(sqrt_mu / (r0 * r1)) * s.s1;
  double gd = 1.0 - s.s2 / r1;
  for (int i = 0; i < 3; ++i) {
    out[i] = f * r[i] + g * v[i];
    out[3 + i] = fd * r[i] + gd * v[i];
  }
  out[6] = f; out[7] = g; out[8] = fd; out[9] = gd; out[10] = s.psi;
  return 0;
}
