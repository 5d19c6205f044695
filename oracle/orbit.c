/*
 * orbit.c — oracle restatement of src/orb_elem.rs (ccek1) and the subset of
 * src/orbit_type/ on the IOD path (to_equinoctial, propagate_twobody).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include "oracle.h"

static inline double wrap_0_2pi(double x) { return orc_principal_angle(x); }

/* orb_elem.rs:58-226. rotmt (ref_system.rs:453-462) = nalgebra
 * Rotation3::from_axis_angle; for a coordinate axis every off-pattern entry is
 * an exact zero, so Rx(i)^T * Rz(Om)^T has single-product entries. */
void orc_ccek1(const double pos[3], const double vel[3], double epoch, orc_elements *out) {
  const double EPS_EQ = 1e-15, EPS_PARAB = 1e-12, EPS_E = 5e-15;
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double h[3] = {pos[1] * vel[2] - pos[2] * vel[1], pos[2] * vel[0] - pos[0] * vel[2],
                 pos[0] * vel[1] - pos[1] * vel[0]};
  double h2 = (h[0] * h[0] + h[1] * h[1]) + h[2] * h[2];
  double hn = sqrt(h2);
  double hu[3] = {h[0] / hn, h[1] / hn, h[2] / hn};
  double sin_i = sqrt(hu[0] * hu[0] + hu[1] * hu[1]);
  double inc = wrap_0_2pi(orc_atan2(sin_i, hu[2]));
  double node;
  if (sin_i <= EPS_EQ) { inc = 0.0; node = 0.0; }
  else node = wrap_0_2pi(orc_atan2(hu[0], -hu[1]));

  double ci = orc_cos(inc), si = orc_sin(inc), cn = orc_cos(node), sn = orc_sin(node); /* correctly rounded, crmath.c */
  /* M = Rx(i)^T * Rz(Om)^T ; Rx^T = [1 0 0;0 c s;0 -s c], Rz^T = [c s 0;-s c 0;0 0 1] */
  double M[3][3] = {{cn, sn, 0.0}, {ci * (-sn), ci * cn, si}, {(-si) * (-sn), (-si) * cn, ci}};
  double po[3], vo[3];
  for (int r = 0; r < 3; ++r) {
    po[r] = (M[r][0] * pos[0] + M[r][1] * pos[1]) + M[r][2] * pos[2];
    vo[r] = (M[r][0] * vel[0] + M[r][1] * vel[1]) + M[r][2] * vel[2];
  }
  double rv = po[0] * vo[0] + po[1] * vo[1];
  double rad = sqrt(po[0] * po[0] + po[1] * po[1]);
  double vsq = vo[0] * vo[0] + vo[1] * vo[1];
  double inv_a = 2.0 / rad - vsq / mu;

  out->epoch = epoch;
  int parabolic = 0;
  if (inv_a > EPS_PARAB) {
    double a = 1.0 / inv_a;
    double n = sqrt(mu / (a * a * a));
    double esin = rv / (n * a * a);
    double ecos = vsq * rad / mu - 1.0;
    double e = sqrt(esin * esin + ecos * ecos);
    if (fabs(e - 1.0) < EPS_E) {
      parabolic = 1;
    } else {
      double ea = orc_atan2(esin, ecos);
      double ma = wrap_0_2pi(ea - e * orc_sin(ea));
      double x1 = orc_cos(ea) - e;
      double rd = sqrt(1.0 - e * e);
      double x2 = rd * orc_sin(ea);
      double nm = sqrt(x1 * x1 + x2 * x2);
      double x1n = x1 / nm, x2n = x2 / nm;
      double sp = x1n * po[1] - x2n * po[0];
      double cp = x1n * po[0] + x2n * po[1];
      double argp = wrap_0_2pi(orc_atan2(sp, cp));
      out->kind = OUTFIT_KIND_KEPLERIAN;
      out->e[0] = a; out->e[1] = e; out->e[2] = inc; out->e[3] = node; out->e[4] = argp;
      out->e[5] = ma;
      return;
    }
  } else if (fabs(inv_a) <= EPS_PARAB) {
    parabolic = 1;
  } else {
    double p = h2 / mu;
    double ecos = p / rad - 1.0;
    double esin = rv * p / (hn * rad);
    double nu = orc_atan2(esin, ecos);
    double e = sqrt(ecos * ecos + esin * esin);
    if (fabs(e - 1.0) < EPS_E) {
      parabolic = 1;
    } else {
      out->kind = OUTFIT_KIND_COMETARY;
      out->e[0] = p / (1.0 + e); out->e[1] = e; out->e[2] = inc; out->e[3] = node;
      out->e[4] = wrap_0_2pi(orc_atan2(po[1], po[0]) - nu);
      out->e[5] = nu;
      return;
    }
  }
  if (parabolic) {
    double p = h2 / mu;
    double cnu = p / rad - 1.0;
    double snu = rv * p / (rad * hn);
    double nu = orc_atan2(snu, cnu);
    out->kind = OUTFIT_KIND_COMETARY;
    out->e[0] = p / 2.0; out->e[1] = 1.0; out->e[2] = inc; out->e[3] = node;
    out->e[4] = wrap_0_2pi(orc_atan2(po[1], po[0]) - nu);
    out->e[5] = nu;
  }
}

/* equinoctial_element.rs:285-313 */
static void from_kepler_internal(double epoch, double a, double e, double inc, double node,
                                 double argp, double ma, orc_equinoctial *eq) {
  double dig = node + argp;
  eq->h = e * orc_sin_s(dig);
  eq->k = e * orc_cos_s(dig);
  double t = orc_tan_s(inc / 2.0);
  eq->p = t * orc_sin_s(node);
  eq->q = t * orc_cos_s(node);
  eq->lambda = orc_principal_angle(dig + ma);
  eq->epoch = epoch;
  eq->a = a;
}

/* orbit_type/mod.rs:399-443 ; cometary_element.rs:224-289,483-503 */
int orc_to_equinoctial(const orc_elements *el, orc_equinoctial *eq) {
  if (el->kind == OUTFIT_KIND_KEPLERIAN) {
    from_kepler_internal(el->epoch, el->e[0], el->e[1], el->e[2], el->e[3], el->e[4], el->e[5], eq);
    return 0;
  }
  double e = el->e[1];
  if (fabs(e - 1.0) < 1e-12) return OUTFIT_TRAJ_INVALID_CONVERSION;
  double p = el->e[0] * (1.0 + e);
  double a = -p / (e * e - 1.0);
  if (e <= 1.0) return OUTFIT_TRAJ_INVALID_ORBIT;
  const double EPSC = 1e-15;
  double s = sqrt((e - 1.0) / (e + 1.0));
  double t = orc_tan_s(0.5 * el->e[5]);
  double x = s * t;
  if (x < -1.0 + EPSC) x = -1.0 + EPSC;
  if (x > 1.0 - EPSC) x = 1.0 - EPSC;
  double hh = 2.0 * atanh(x);
  double m = e * sinh(hh) - hh;
  from_kepler_internal(el->epoch, a, e, el->e[2], el->e[3], el->e[4], m, eq);
  return 0;
}

/* equinoctial_element.rs:326-348 with roots 0.0.8 find_root_newton_raphson and
 * SimpleConvergency{eps: 100*EPSILON, max_iter: 25} restated. */
uint64_t orc_dbg_kepler_evals[32]; /* histogram of f evaluations per solve (work-distribution study) */
static int solve_kepler_equation_counted(const orc_equinoctial *eq, double lambda1, double varpi, double *F, int *evals);
int orc_solve_kepler_equation(const orc_equinoctial *eq, double lambda1, double varpi, double *F) {
  int evals = 0;
  int rc = solve_kepler_equation_counted(eq, lambda1, varpi, F, &evals);
  __atomic_fetch_add(&orc_dbg_kepler_evals[evals < 31 ? evals : 31], 1, __ATOMIC_RELAXED);
  return rc;
}
static int solve_kepler_equation_counted(const orc_equinoctial *eq, double lambda1, double varpi, double *F, int *evals) {
  const double eps = ORC_EPS * 1e2;
  double x = ORC_PI + varpi;
  int iter = 0;
  for (;;) {
    *evals += 1;
    double f = x - eq->k * orc_sin_s(x) + eq->h * orc_cos_s(x) - lambda1;
    double d = 1.0 - eq->k * orc_cos_s(x) - eq->h * orc_sin_s(x);
    if (fabs(f) < eps) { *F = x; return 0; }
    if (fabs(d) < eps) {
      if (iter == 0) { x = x + 1.0; iter = iter + 1; continue; }
      return 1; /* SearchError::ZeroDerivative */
    }
    double x1 = x - f / d;
    if (fabs(x - x1) < eps) { *F = x1; return 0; }
    x = x1;
    iter = iter + 1;
    if (iter >= 25) return 1; /* SearchError::NoConvergency */
  }
}

/* equinoctial_element.rs:639-759,809-867 */
int orc_propagate_twobody(const orc_equinoctial *eq, double t0, double t1, double pos[3],
                          double vel[3]) {
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  double n = sqrt(mu / (eq->a * eq->a * eq->a));
  double lam = eq->lambda + n * (t1 - t0);
  double e2 = eq->h * eq->h + eq->k * eq->k;
  double varpi = 0.0;
  if (e2 > ORC_EPS * 1e2) varpi = orc_principal_angle(orc_atan2_s(eq->h, eq->k));
  lam = orc_principal_angle(lam);
  if (lam < varpi) lam += ORC_DPI;
  double F;
  if (orc_solve_kepler_equation(eq, lam, varpi, &F)) return 1;

  double beta = 1.0 / (1.0 + sqrt(1.0 - e2));
  double bet = beta * eq->h * eq->k;
  double sf = orc_sin_s(F), cf = orc_cos_s(F);
  double xe = eq->a * ((1.0 - beta * (eq->h * eq->h)) * cf + bet * sf - eq->k);
  double ye = eq->a * ((1.0 - beta * (eq->k * eq->k)) * sf + bet * cf - eq->h);
  double u = 1.0 + eq->p * eq->p + eq->q * eq->q;
  double inv_u = 1.0 / u;
  double cc = 2.0 * eq->p * eq->q * inv_u;
  double fv[3] = {(1.0 - eq->p * eq->p + eq->q * eq->q) * inv_u, cc, -2.0 * eq->p * inv_u};
  double gv[3] = {cc, (1.0 + eq->p * eq->p - eq->q * eq->q) * inv_u, 2.0 * eq->q * inv_u};
  for (int i = 0; i < 3; ++i) pos[i] = xe * fv[i] + ye * gv[i];
  double vc = n * (eq->a * eq->a) / sqrt(xe * xe + ye * ye);
  double vxe = vc * (bet * cf - (1.0 - beta * (eq->h * eq->h)) * sf);
  double vye = vc * ((1.0 - beta * (eq->k * eq->k)) * cf - bet * sf);
  for (int i = 0; i < 3; ++i) vel[i] = vxe * fv[i] + vye * gv[i];
  return 0;
}
