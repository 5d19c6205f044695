/*
 * diffcor.c — oracle restatement of the differential orbit correction
 * (src/differential_orbit_correction/{mod,diff_cor,single_iteration,least_square,
 * outlier_rejection,obs_fit_data,obs_dataset_api}.rs) with the two-body partials of
 * src/orbit_type/equinoctial_element.rs:442-582 and
 * src/ephemeris/observation_ephemeris.rs:222-275,347-365,418-450.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Third-party arithmetic restated here (nalgebra 0.34.2, sources absent from
 * /root/reference): `Cholesky::new` + `inverse` (column-oriented factorisation,
 * forward then adjoint substitution on the identity), `Matrix6 * Vector6` (gemv:
 * column axpy), `dot` (sequential for 6-vectors). The Householder `QR::new` +
 * `try_inverse` fallback (least_square.rs:305-309) is restated from the crate's
 * published algorithm and is NOT pinned by any reference test: it is only reached
 * when the normal matrix is not positive definite.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define VLIGHT_AU (ORC_VLIGHT / ORC_AU * ORC_SECONDS_PER_DAY)

static const double ROT_ECL_TO_EQU[3][3] = {{1.0, 0.0, 0.0},
                                            {0.0, 9.174820620691818e-1, -3.977771559319137e-1},
                                            {0.0, 3.977771559319137e-1, 9.174820620691818e-1}};

static inline void gemv3(const double A[3][3], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = (A[r][0] * x[0] + A[r][1] * x[1]) + A[r][2] * x[2];
}
static inline double dot3(const double a[3], const double b[3]) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

/* diff_cor.rs:161-178, outlier_rejection.rs:48-55, equinoctial_element.rs:169-179 */
void orc_diffcor_config_default(outfit_diffcor_config_t *c) {
  memset(c, 0, sizeof *c);
  c->max_newton_iterations = 30;
  c->max_outlier_rejection_passes = 10;
  c->convergence_threshold = 1e-4;
  c->convergence_before_rejection_threshold = 2.0;
  c->rms_stagnation_ratio = 0.98;
  c->rms_divergence_ratio = 1.5;
  c->max_stagnation_iterations = 3;
  c->enable_outlier_rejection = 1;
  c->propagator = OUTFIT_PROPAGATOR_TWOBODY;
  c->chi_squared_rejection_threshold = 25.0;
  c->chi_squared_recovery_threshold = 9.0;
  c->eccentricity_limit = 1.2;
  c->min_semi_major_axis = 1e-6;
  c->max_semi_major_axis = 1e4;
  c->min_periapsis_distance = 1e-6;
  c->max_apoapsis_distance = 1e4;
  for (int j = 0; j < 6; ++j) c->free_elements[j] = 1;
}

/* equinoctial_element.rs:258-268 */
int orc_is_bizarre(const orc_equinoctial *eq, const outfit_diffcor_config_t *c) {
  double e = sqrt(eq->h * eq->h + eq->k * eq->k);
  double peri = eq->a * (1.0 - e);
  double apo = eq->a * (1.0 + e);
  return e > c->eccentricity_limit || eq->a < c->min_semi_major_axis ||
         eq->a > c->max_semi_major_axis || peri < c->min_periapsis_distance ||
         apo > c->max_apoapsis_distance;
}

/* equinoctial_element.rs:442-582 ; dpos/dvel are [element j][component] (Matrix6x3) */
void orc_compute_derivative(const orc_equinoctial *eq, double t0, double t1, double n, double lam1,
                            double F, double inv_u, double beta, double sF, double cF, double xe,
                            double ye, double vxe, double vye, const double f[3], const double g[3],
                            const double pos[3], const double vel[3], double dpos[6][3],
                            double dvel[6][3]) {
  const double a = eq->a, h = eq->h, k = eq->k, p = eq->p, q = eq->q;
  const double mu = ORC_GAUSS_GRAV * ORC_GAUSS_GRAV;
  /* compute_w_vector :376-383 */
  double w[3] = {2.0 * p * inv_u, -2.0 * q * inv_u, (1.0 - p * p - q * q) * inv_u};
  double r = sqrt(xe * xe + ye * ye);
  double inv_r = 1.0 / r;
  double inv_1_beta = 1.0 / (1.0 - beta);
  double beta3 = beta * beta * beta;
  double a2 = a * a;
  double inv_r2 = inv_r * inv_r;

  double tmp1 = lam1 - F;
  double tmp2 = beta + (h * h) * beta3 * inv_1_beta;
  double tmp3 = h * k * beta3 * inv_1_beta;
  double tmp4 = beta * h - sF;
  double tmp5 = beta * k - cF;
  double tmp6 = beta + (k * k) * beta3 * inv_1_beta;
  double tmp7 = 1.0 - r / a;
  double tmp8 = sF - h;
  double tmp9 = cF - k;
  double tmp10 = a * cF * inv_r;
  double tmp11 = a * sF * inv_r;
  double tmp12 = n * a2 * inv_r;
  double dt = t1 - t0;

  for (int i = 0; i < 3; ++i) dpos[0][i] = (pos[i] - 3.0 * vel[i] * dt / 2.0) / a;

  double dx1de2 = -a * (tmp1 * tmp2 + a * cF * tmp4 * inv_r);
  double dx2de2 = a * (tmp1 * tmp3 - 1.0 + a * cF * tmp5 * inv_r);
  for (int i = 0; i < 3; ++i) dpos[1][i] = dx1de2 * f[i] + dx2de2 * g[i];

  double dx1de3 = -a * (tmp1 * tmp3 + 1.0 - a * sF * tmp4 * inv_r);
  double dx2de3 = a * (tmp1 * tmp6 - a * sF * tmp5 * inv_r);
  for (int i = 0; i < 3; ++i) dpos[2][i] = dx1de3 * f[i] + dx2de3 * g[i];

  for (int i = 0; i < 3; ++i) dpos[3][i] = 2.0 * (q * (ye * f[i] - xe * g[i]) - xe * w[i]) * inv_u;
  for (int i = 0; i < 3; ++i) dpos[4][i] = 2.0 * (p * (-ye * f[i] + xe * g[i]) + ye * w[i]) * inv_u;
  for (int i = 0; i < 3; ++i) dpos[5][i] = vel[i] / n;

  double r3 = r * r * r;
  double three_mu = 3.0 * mu;
  for (int i = 0; i < 3; ++i) dvel[0][i] = -(vel[i] - three_mu * pos[i] * dt / r3) / (2.0 * a);

  double dx4de2 = tmp12 * (tmp7 * tmp2 + a2 * tmp8 * tmp4 * inv_r2 + tmp10 * cF);
  double dx5de2 = -tmp12 * (tmp7 * tmp3 + a2 * tmp8 * tmp5 * inv_r2 - tmp10 * sF);
  for (int i = 0; i < 3; ++i) dvel[1][i] = dx4de2 * f[i] + dx5de2 * g[i];

  double dx4de3 = tmp12 * (tmp7 * tmp3 + a2 * tmp9 * tmp4 * inv_r2 - tmp11 * cF);
  double dx5de3 = -tmp12 * (tmp7 * tmp6 + a2 * tmp9 * tmp5 * inv_r2 + tmp11 * sF);
  for (int i = 0; i < 3; ++i) dvel[2][i] = dx4de3 * f[i] + dx5de3 * g[i];

  for (int i = 0; i < 3; ++i) dvel[3][i] = 2.0 * (q * (vye * f[i] - vxe * g[i]) - vxe * w[i]) * inv_u;
  for (int i = 0; i < 3; ++i) dvel[4][i] = 2.0 * (p * (-vye * f[i] + vxe * g[i]) + vye * w[i]) * inv_u;
  double a3 = a * a * a, inv_r3 = inv_r * inv_r * inv_r;
  double c6 = -n * a3;
  for (int i = 0; i < 3; ++i) dvel[5][i] = c6 * pos[i] * inv_r3;
}

/* equinoctial_element.rs:809-845 + :639-759 with compute_derivatives = true */
int orc_propagate_twobody_partials(const orc_equinoctial *eq, double t0, double t1, double pos[3],
                                   double vel[3], double dpos[6][3], double dvel[6][3]) {
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
  orc_compute_derivative(eq, t0, t1, n, lam, F, inv_u, beta, sf, cf, xe, ye, vxe, vye, fv, gv, pos,
                         vel, dpos, dvel);
  return 0;
}

/* observation_ephemeris.rs:222-275 (all three vectors equatorial despite the `_ecl` names) */
void orc_topocentric_radec_and_partials(const double ast_pos[3], const double ast_vel[3],
                                        const double obs_pos[3], double *ra, double *dec,
                                        double d_ra[3], double d_dec[3]) {
  double rel[3], cor[3];
  for (int i = 0; i < 3; ++i) rel[i] = ast_pos[i] - obs_pos[i];
  double rel_norm = sqrt(dot3(rel, rel));
  double lt = rel_norm / VLIGHT_AU; /* aberration.rs:139-145 */
  for (int i = 0; i < 3; ++i) cor[i] = rel[i] - lt * ast_vel[i];
  double x = cor[0], y = cor[1], z = cor[2];
  double rho = sqrt(dot3(cor, cor));
  double rho_xy = orc_hypot_s(x, y);
  double rho_xy_sq = rho_xy * rho_xy;
  *dec = orc_atan2_s(z, rho_xy);
  double a = fmod(orc_atan2_s(y, x), ORC_DPI); /* rem_euclid(TAU) */
  if (a < 0.0) a += ORC_DPI;
  *ra = a;
  double rho_sq = rho * rho;
  double grad_ra[3] = {-y / rho_xy_sq, x / rho_xy_sq, 0.0};
  double grad_dec[3] = {-z * x / (rho_xy * rho_sq), -z * y / (rho_xy * rho_sq), rho_xy / rho_sq};
  double aberr = 1.0 / (rel_norm * VLIGHT_AU);
  double ca = dot3(grad_ra, ast_vel) * aberr;
  double cd = dot3(grad_dec, ast_vel) * aberr;
  for (int i = 0; i < 3; ++i) {
    d_ra[i] = grad_ra[i] - ca * rel[i];
    d_dec[i] = grad_dec[i] - cd * rel[i];
  }
}

/* observation_ephemeris.rs:418-450 (+ :347-365). 0 ok, 1 Err (skipped by the caller) */
int orc_obs_and_partials_2body(const orc_equinoctial *eq, double mjd, const double obs_pos_equ[3],
                               double *ra, double *dec, double d_ra_d_elem[6],
                               double d_dec_d_elem[6]) {
  if (sqrt(eq->h * eq->h + eq->k * eq->k) >= 1.0) return 1; /* check_elliptical_orbit :288-296 */
  double dt = mjd - eq->epoch;
  double pe[3], ve[3], dpos[6][3], dvel[6][3];
  if (orc_propagate_twobody_partials(eq, 0.0, dt, pe, ve, dpos, dvel)) return 1;
  double pq[3], vq[3], d_ra[3], d_dec[3];
  gemv3(ROT_ECL_TO_EQU, pe, pq);
  gemv3(ROT_ECL_TO_EQU, ve, vq);
  orc_topocentric_radec_and_partials(pq, vq, obs_pos_equ, ra, dec, d_ra, d_dec);
  for (int j = 0; j < 6; ++j) {
    double dq[3];
    gemv3(ROT_ECL_TO_EQU, dpos[j], dq);
    d_ra_d_elem[j] = dot3(d_ra, dq);
    d_dec_d_elem[j] = dot3(d_dec, dq);
  }
  return 0;
}

/* least_square.rs:188-198. The reference's two `while` loops do not terminate when a - b is infinite or so large
 * that subtracting 2 pi no longer changes it (an observed angle of 1e300): here — and identically on the device — they
 * stop after 1024 steps (|a - b| < 6400 rad is reduced exactly as the reference does) and the result is then NaN. */
double orc_angular_diff(double a, double b) {
  double d = a - b;
  for (int i = 0; i < 1024 && d > ORC_PI; ++i) d -= ORC_DPI;
  for (int i = 0; i < 1024 && d < -ORC_PI; ++i) d += ORC_DPI;
  if (d > ORC_PI || d < -ORC_PI) return NAN;
  return d;
}

/* ---- nalgebra restatements (m[i][j] = matrix[(i,j)]) ------------------------------ */

/* nalgebra linalg/cholesky.rs `Cholesky::new` then `inverse()`; returns 0 if not PD */
static int cholesky_inverse6(const double min[6][6], double inv[6][6]) {
  double m[6][6];
  memcpy(m, min, sizeof m);
  for (int j = 0; j < 6; ++j) {
    for (int k = 0; k < j; ++k) {
      double factor = -m[j][k];
      for (int i = j; i < 6; ++i) m[i][j] = factor * m[i][k] + m[i][j]; /* axpy, b = 1 */
    }
    double diag = m[j][j];
    if (diag == 0.0 || !(diag >= 0.0)) return 0; /* is_zero / try_sqrt */
    double denom = sqrt(diag);
    m[j][j] = denom;
    for (int i = j + 1; i < 6; ++i) m[i][j] /= denom;
  }
  for (int c = 0; c < 6; ++c) {
    double b[6] = {0, 0, 0, 0, 0, 0};
    b[c] = 1.0;
    /* solve_lower_triangular_unchecked_mut */
    for (int i = 0; i < 6; ++i) {
      double coeff = b[i] / m[i][i];
      b[i] = coeff;
      for (int r = i + 1; r < 6; ++r) b[r] = (-coeff) * m[r][i] + b[r];
    }
    /* ad_solve_lower_triangular_unchecked_mut */
    for (int i = 5; i >= 0; --i) {
      double dot = 0.0;
      for (int r = i + 1; r < 6; ++r) dot += m[r][i] * b[r];
      b[i] = (b[i] - dot) / m[i][i];
    }
    for (int r = 0; r < 6; ++r) inv[r][c] = b[r];
  }
  return 1;
}

/* nalgebra linalg/{qr,householder}.rs + geometry/reflection.rs: `QR::new(m).try_inverse()` */
static int qr_inverse6(const double min[6][6], double inv[6][6]) {
  double m[6][6], diag[6];
  memcpy(m, min, sizeof m);
  for (int c = 0; c < 6; ++c) {
    /* reflection_axis_mut on rows c.. of column c */
    double sq = 0.0;
    for (int r = c; r < 6; ++r) sq += m[r][c] * m[r][c];
    double nrm = sqrt(sq);
    double x0 = m[c][c];
    double modulus = x0 >= 0.0 ? x0 : -x0, sign = x0 >= 0.0 ? 1.0 : -1.0;
    double signed_norm = sign * nrm;
    double factor = (sq + modulus * nrm) * 2.0;
    m[c][c] += signed_norm;
    if (factor != 0.0) {
      double sf = sqrt(factor);
      for (int r = c; r < 6; ++r) m[r][c] /= sf;
      double n2 = 0.0;
      for (int r = c; r < 6; ++r) n2 += m[r][c] * m[r][c];
      double nn = sqrt(n2);
      for (int r = c; r < 6; ++r) m[r][c] /= nn;
      diag[c] = -signed_norm;
      double sg = (diag[c] > 0.0 || (diag[c] == 0.0 && !signbit(diag[c]))) ? 1.0 : -1.0; /* signum */
      double m_two = sg * -2.0;
      for (int j = c + 1; j < 6; ++j) { /* reflect_with_sign on the trailing columns */
        double dot = 0.0;
        for (int r = c; r < 6; ++r) dot += m[r][c] * m[r][j];
        double fct = (dot - 0.0) * m_two;
        for (int r = c; r < 6; ++r) m[r][j] = fct * m[r][c] + sg * m[r][j];
      }
    } else {
      diag[c] = signed_norm;
    }
  }
  for (int col = 0; col < 6; ++col) {
    double b[6] = {0, 0, 0, 0, 0, 0};
    b[col] = 1.0;
    for (int c = 0; c < 6; ++c) { /* q_tr_mul: reflect_with_sign(rows c.., signum(diag[c])) */
      double sg = (diag[c] > 0.0 || (diag[c] == 0.0 && !signbit(diag[c]))) ? 1.0 : -1.0;
      double dot = 0.0;
      for (int r = c; r < 6; ++r) dot += m[r][c] * b[r];
      double fct = (dot - 0.0) * (sg * -2.0);
      for (int r = c; r < 6; ++r) b[r] = fct * m[r][c] + sg * b[r];
    }
    for (int i = 5; i >= 0; --i) { /* solve_upper_triangular_mut */
      double d = fabs(diag[i]);
      if (d == 0.0) return 0;
      double coeff = b[i] / d;
      b[i] = coeff;
      for (int r = 0; r < i; ++r) b[r] = (-coeff) * m[r][i] + b[r];
    }
    for (int r = 0; r < 6; ++r) inv[r][col] = b[r];
  }
  return 1;
}

static void gemv6(const double A[6][6], const double x[6], double y[6]) {
  for (int r = 0; r < 6; ++r) {
    double acc = A[r][0] * x[0];
    for (int j = 1; j < 6; ++j) acc = A[r][j] * x[j] + acc;
    y[r] = acc;
  }
}
static double dot6(const double a[6], const double b[6]) {
  double res = 0.0;
  for (int i = 0; i < 6; ++i) res += a[i] * b[i];
  return res;
}

/* least_square.rs:211-303 (weight_cross = 0: ObservationEquation::uncorrelated :113-133) */
void orc_solve_weighted_least_squares(const orc_obs_equation *eqs, size_t n,
                                      const uint8_t free_elements[6], orc_lsq_solution *out) {
  size_t active = 0;
  for (size_t i = 0; i < n; ++i) active += eqs[i].active ? 1 : 0;
  out->num_measurements = (uint32_t)(active * 2);
  double (*N)[6] = out->normal;
  double rhs[6] = {0, 0, 0, 0, 0, 0};
  memset(out->normal, 0, sizeof out->normal);
  double wss = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const orc_obs_equation *e = &eqs[i];
    if (!e->active) continue;
    const double *ga = e->d_ra, *gd = e->d_dec;
    double wa = e->weight_ra, wd = e->weight_dec, wc = e->weight_cross;
    double ra = e->residual_ra, rd = e->residual_dec;
    for (int j = 0; j < 6; ++j) {
      for (int k = 0; k < 6; ++k)
        N[j][k] += ga[j] * wa * ga[k] + gd[j] * wd * gd[k] + wc * (gd[j] * ga[k] + ga[j] * gd[k]);
      rhs[j] += (ga[j] * wa + gd[j] * wc) * ra + (ga[j] * wc + gd[j] * wd) * rd;
    }
    wss += wa * ra * ra + wd * rd * rd + 2.0 * wc * ra * rd;
  }
  for (int j = 0; j < 6; ++j) {
    if (!free_elements[j]) {
      for (int k = 0; k < 6; ++k) { N[j][k] = 0.0; N[k][j] = 0.0; }
      N[j][j] = 1.0;
      rhs[j] = 0.0;
    }
  }
  out->used_qr = 0;
  out->inversion_succeeded = cholesky_inverse6(N, out->covariance);
  if (!out->inversion_succeeded) {
    out->inversion_succeeded = qr_inverse6(N, out->covariance);
    out->used_qr = out->inversion_succeeded;
    if (!out->inversion_succeeded) memset(out->covariance, 0, sizeof out->covariance);
  }
  if (out->inversion_succeeded) gemv6(out->covariance, rhs, out->dx);
  else memset(out->dx, 0, sizeof out->dx);
  for (int j = 0; j < 6; ++j)
    if (!free_elements[j]) out->dx[j] = 0.0;
  out->normalised_rms = out->num_measurements > 0 ? sqrt(wss / (double)out->num_measurements) : 0.0;
}

/* least_square.rs:341-362 */
void orc_rescale_covariance(double covariance[6][6], double normal[6][6], size_t num_free,
                            size_t num_measurements, double normalised_rms) {
  double mu = 1.0;
  if (num_free < num_measurements) {
    double factor = sqrt((double)num_measurements / (double)(num_measurements - num_free));
    mu = normalised_rms > 1.0 ? normalised_rms * factor : factor;
  }
  double mu2 = mu * mu;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      covariance[i][j] *= mu2;
      normal[i][j] /= mu2;
    }
}

/* outlier_rejection.rs:132-235 ; returns the number of selection changes */
size_t orc_update_observation_selection(orc_obs_fit *fit, const orc_obs_equation *eqs, size_t n,
                                        const double covariance[6][6], double chi2_reject,
                                        double chi2_recover) {
  size_t changes = 0;
  for (size_t i = 0; i < n; ++i) {
    orc_obs_fit *fd = &fit[i];
    const orc_obs_equation *e = &eqs[i];
    if (fd->selection == OUTFIT_OBS_FORCED_OUT) continue;
    double var_ra = fd->sigma_ra * fd->sigma_ra;
    double var_dec = fd->sigma_dec * fd->sigma_dec;
    double cov_cross = -fd->sigma_ra * fd->sigma_dec * e->weight_cross / (e->weight_ra * e->weight_dec);
    double gga[6], ggd[6];
    gemv6(covariance, e->d_ra, gga);
    gemv6(covariance, e->d_dec, ggd);
    double paa = dot6(e->d_ra, gga), pdd = dot6(e->d_dec, ggd), pad = dot6(e->d_ra, ggd);
    double p00 = var_ra - paa, p01 = cov_cross - pad, p10 = cov_cross - pad, p11 = var_dec - pdd;
    double det = p00 * p11 - p01 * p01;
    double scale = fmax(fabs(p00), fabs(p11));
    double thr = ORC_EPS * scale * scale;
    if (fabs(det) < thr || scale == 0.0) continue;
    double i00 = p11 / det, i01 = -p01 / det, i10 = -p10 / det, i11 = p00 / det;
    double r0 = fd->residual_ra, r1 = fd->residual_dec;
    double v0 = i00 * r0 + i01 * r1, v1 = i10 * r0 + i11 * r1;
    double chi2 = r0 * v0 + r1 * v1;
    if (fd->selection == OUTFIT_OBS_ACTIVE && chi2 > chi2_reject) {
      fd->selection = OUTFIT_OBS_REJECTED;
      changes += 1;
    } else if (fd->selection == OUTFIT_OBS_REJECTED && chi2 <= chi2_recover) {
      fd->selection = OUTFIT_OBS_ACTIVE;
      changes += 1;
    }
  }
  return changes;
}

/* single_iteration.rs:158-328 with apply_correction = true, PropagatorKind::TwoBody */
static void single_iteration(const orc_equinoctial *el, const double *mjd, const double *ra,
                             const double *dec, const double *obs_pos /* 3/obs */, size_t n,
                             const orc_obs_fit *fit_in, const uint8_t free_elements[6],
                             orc_obs_equation *eqs, orc_obs_fit *fit_out, orc_lsq_solution *sol,
                             orc_equinoctial *corrected, double *correction_norm) {
  for (size_t i = 0; i < n; ++i) {
    orc_obs_equation *e = &eqs[i];
    fit_out[i] = fit_in[i];
    memset(e, 0, sizeof *e);
    e->weight_ra = 1.0 / (1.0 * 1.0); /* dummy sigma = 1 */
    e->weight_dec = 1.0 / (1.0 * 1.0);
    if (fit_in[i].selection != OUTFIT_OBS_ACTIVE) continue;
    double cra, cdec, ga[6], gd[6];
    if (orc_obs_and_partials_2body(el, mjd[i], obs_pos + 3 * i, &cra, &cdec, ga, gd)) continue;
    double rra = orc_angular_diff(ra[i] - fit_in[i].bias_ra, cra);
    double rdec = (dec[i] - fit_in[i].bias_dec) - cdec;
    double ta = rra / fit_in[i].sigma_ra, td = rdec / fit_in[i].sigma_dec;
    double chi = sqrt(ta * ta + td * td);
    memcpy(e->d_ra, ga, sizeof ga);
    memcpy(e->d_dec, gd, sizeof gd);
    e->residual_ra = rra;
    e->residual_dec = rdec;
    e->weight_ra = 1.0 / (fit_in[i].sigma_ra * fit_in[i].sigma_ra);
    e->weight_dec = 1.0 / (fit_in[i].sigma_dec * fit_in[i].sigma_dec);
    e->weight_cross = 0.0;
    e->active = 1;
    fit_out[i].residual_ra = rra;
    fit_out[i].residual_dec = rdec;
    fit_out[i].chi = chi;
  }
  orc_solve_weighted_least_squares(eqs, n, free_elements, sol);
  double cdx[6];
  gemv6(sol->normal, sol->dx, cdx);
  *correction_norm = sqrt(dot6(sol->dx, cdx));
  double coord[6] = {el->a, el->h, el->k, el->p, el->q, el->lambda};
  for (int j = 0; j < 6; ++j)
    if (free_elements[j]) coord[j] += sol->dx[j];
  corrected->epoch = el->epoch;
  corrected->a = coord[0]; corrected->h = coord[1]; corrected->k = coord[2];
  corrected->p = coord[3]; corrected->q = coord[4]; corrected->lambda = coord[5];
}

/* diff_cor.rs:282-436. Returns OUTFIT_DC_*; on OUTFIT_DC_NONE `out`/`fit` hold the output. */
int orc_run_differential_correction(const orc_equinoctial *initial, const double *mjd,
                                    const double *ra, const double *dec, const double *obs_pos,
                                    size_t n, orc_obs_fit *fit /* in: initial, out: final */,
                                    const outfit_diffcor_config_t *cfg, orc_diffcor_output *out) {
  size_t num_free = 0;
  for (int j = 0; j < 6; ++j) num_free += cfg->free_elements[j] ? 1 : 0;
  orc_equinoctial elements = *initial;
  orc_obs_equation *eqs = (orc_obs_equation *)calloc(n ? n : 1, sizeof *eqs);
  orc_obs_equation *last_eqs = (orc_obs_equation *)calloc(n ? n : 1, sizeof *eqs);
  orc_obs_fit *fit_new = (orc_obs_fit *)calloc(n ? n : 1, sizeof *fit_new);
  size_t n_last_eqs = 0; /* `last_equations = vec![]` per outer pass */
  uint32_t total_iters = 0, outer_started = 0;
  double last_normal[6][6], last_cov[6][6];
  memset(last_normal, 0, sizeof last_normal);
  memset(last_cov, 0, sizeof last_cov);
  double last_rms = 1.7976931348623157e308; /* f64::MAX */
  uint32_t last_num_meas = 0;
  int rc = OUTFIT_DC_NONE;

  for (uint64_t outer = 0; outer <= cfg->max_outlier_rejection_passes; ++outer) {
    outer_started += 1;
    double prev_rms = 1.7976931348623157e308;
    uint64_t stagnation = 0;
    n_last_eqs = 0;
    int inner_converged = 0;
    for (uint64_t inner = 0; inner < cfg->max_newton_iterations; ++inner) {
      total_iters += 1;
      orc_lsq_solution sol;
      orc_equinoctial corrected;
      double cnorm;
      single_iteration(&elements, mjd, ra, dec, obs_pos, n, fit, cfg->free_elements, eqs, fit_new,
                       &sol, &corrected, &cnorm);
      if (!sol.inversion_succeeded) { rc = OUTFIT_DC_FAILED; goto done; }
      if (orc_is_bizarre(&corrected, cfg)) { rc = OUTFIT_DC_BIZARRE; goto done; }
      double new_rms = sol.normalised_rms;
      if (prev_rms < 1.7976931348623157e308 && new_rms / prev_rms >= cfg->rms_divergence_ratio) {
        rc = OUTFIT_DC_DIVERGED;
        goto done;
      }
      int stagnated = prev_rms < 1.7976931348623157e308 && new_rms / prev_rms >= cfg->rms_stagnation_ratio;
      if (stagnated) {
        stagnation += 1;
        if (stagnation >= cfg->max_stagnation_iterations) break;
      } else {
        stagnation = 0;
      }
      memcpy(last_eqs, eqs, n * sizeof *eqs);
      n_last_eqs = n;
      memcpy(last_normal, sol.normal, sizeof last_normal);
      memcpy(last_cov, sol.covariance, sizeof last_cov);
      last_rms = new_rms;
      last_num_meas = sol.num_measurements;
      elements = corrected;
      memcpy(fit, fit_new, n * sizeof *fit);
      prev_rms = new_rms;
      if (cnorm < cfg->convergence_threshold) { inner_converged = 1; break; }
    }
    if (!cfg->enable_outlier_rejection) break;
    if (outer == 0 && last_rms < cfg->convergence_before_rejection_threshold) break;
    if (!inner_converged) break;
    /* update_observation_selection asserts equal lengths (outlier_rejection.rs:138-142):
     * unreachable with n_last_eqs != n because inner_converged implies an accepted iteration */
    (void)n_last_eqs;
    size_t changes = orc_update_observation_selection(fit, last_eqs, n, last_cov,
                                                      cfg->chi_squared_rejection_threshold,
                                                      cfg->chi_squared_recovery_threshold);
    if (changes == 0) break;
  }
  orc_rescale_covariance(last_cov, last_normal, num_free, last_num_meas, last_rms);
  out->elements = elements;
  memcpy(out->covariance, last_cov, sizeof last_cov);
  memcpy(out->normal, last_normal, sizeof last_normal);
  out->normalised_rms = last_rms;
  out->num_measurements = last_num_meas;
done:
  out->total_newton_iterations = total_iters;
  out->outer_passes = outer_started;
  free(eqs);
  free(last_eqs);
  free(fit_new);
  return rc;
}

/* mod.rs:66-112 for one trajectory of a batch (+ obs_dataset_api.rs:101-130) */
void orc_differential_correction(const outfit_iod_params_t *ip, const outfit_diffcor_config_t *cfg,
                                 const outfit_obs_batch_t *b, const outfit_ephem_t *eph, uint64_t traj,
                                 const outfit_gauss_result_t *initial /* nullable */,
                                 outfit_lsq_result_t *out, outfit_obs_fit_t *obs_fit /* nullable */) {
  uint64_t base = b->traj_offset[traj];
  size_t n = (size_t)(b->traj_offset[traj + 1] - base);
  memset(out, 0, sizeof *out);
  if (obs_fit) memset(obs_fit + base, 0, n * sizeof *obs_fit);
  if (initial && initial->status == OUTFIT_TRAJ_OK) out->iod = *initial;
  else orc_estimate_best_orbit(ip, b, eph, traj, 1, &out->iod);
  if (out->iod.status != OUTFIT_TRAJ_OK) { out->status = OUTFIT_LSQ_IOD_ERROR; return; }

  orc_elements el;
  el.kind = out->iod.kind;
  el.epoch = out->iod.epoch;
  memcpy(el.e, out->iod.elem, sizeof el.e);
  orc_equinoctial eq;
  /* Correctly-rounded mode: from here on every elementary function is correctly rounded (crmath.c) — the corrector's
   * ill-conditioned normal equations amplify a last place, so that mode pins every function. Glibc-faithful mode
   * (orc_math_glibc): no scope — the corrector runs on glibc's bits like the rest of the path, i.e. as the reference does
   * on a glibc host (round 3: every function it calls is restated). */
  const int scope = orc_math_glibc ? 0 : 1;
  orc_cr_scope += scope;
  int ec = orc_to_equinoctial(&el, &eq);
  orc_cr_scope -= scope;
  if (ec) { out->status = OUTFIT_LSQ_CONVERSION_ERROR; out->conv_error = ec; return; }

  orc_obs_fit *fit = (orc_obs_fit *)calloc(n ? n : 1, sizeof *fit);
  double *obs_pos = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    fit[i].sigma_ra = b->sigma_ra[base + i];
    fit[i].sigma_dec = b->sigma_dec[base + i];
    fit[i].selection = OUTFIT_OBS_ACTIVE;
    if (eph && b->obs_geo_ecl) {
      int err;
      orc_observer_position_equ(eph, b->mjd_tt[base + i], b->obs_geo_ecl + 3 * (base + i),
                                obs_pos + 3 * i, &err);
    } else {
      memcpy(obs_pos + 3 * i, b->obs_helio_equ + 3 * (base + i), 3 * sizeof(double));
    }
  }
  orc_diffcor_output dc;
  memset(&dc, 0, sizeof dc);
  orc_cr_scope += scope;
  int rc = orc_run_differential_correction(&eq, b->mjd_tt + base, b->ra + base, b->dec + base, obs_pos,
                                           n, fit, cfg, &dc);
  orc_cr_scope -= scope;
  out->total_newton_iterations = dc.total_newton_iterations;
  out->outer_passes = dc.outer_passes;
  out->epoch = eq.epoch;
  if (rc != OUTFIT_DC_NONE) {
    out->status = OUTFIT_LSQ_IOD_KEPT;
    out->dc_error = rc;
  } else {
    out->status = OUTFIT_LSQ_DIFFCOR;
    out->elem[0] = dc.elements.a; out->elem[1] = dc.elements.h; out->elem[2] = dc.elements.k;
    out->elem[3] = dc.elements.p; out->elem[4] = dc.elements.q; out->elem[5] = dc.elements.lambda;
    out->normalised_rms = dc.normalised_rms;
    out->num_measurements = dc.num_measurements;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) out->covariance[6 * i + j] = dc.covariance[i][j];
    uint32_t rej = 0;
    for (size_t i = 0; i < n; ++i) {
      rej += fit[i].selection != OUTFIT_OBS_ACTIVE;
      if (obs_fit) {
        obs_fit[base + i].residual_ra = fit[i].residual_ra;
        obs_fit[base + i].residual_dec = fit[i].residual_dec;
        obs_fit[base + i].chi = fit[i].chi;
        obs_fit[base + i].selection = fit[i].selection;
      }
    }
    out->n_rejected = rej;
  }
  free(fit);
  free(obs_pos);
}

typedef struct {
  const outfit_iod_params_t *ip;
  const outfit_diffcor_config_t *cfg;
  const outfit_obs_batch_t *b;
  const outfit_ephem_t *eph;
  const outfit_gauss_result_t *initial;
  volatile uint64_t *next;
  outfit_lsq_result_t *out;
  outfit_obs_fit_t *obs_fit;
} lsq_work_t;

static void *lsq_worker(void *arg) {
  lsq_work_t *w = (lsq_work_t *)arg;
  for (;;) {
    uint64_t t = __atomic_fetch_add(w->next, 16, __ATOMIC_RELAXED);
    if (t >= w->b->n_traj) break;
    uint64_t hi = t + 16 < w->b->n_traj ? t + 16 : w->b->n_traj;
    for (uint64_t i = t; i < hi; ++i)
      orc_differential_correction(w->ip, w->cfg, w->b, w->eph, i, w->initial ? &w->initial[i] : NULL,
                                  &w->out[i], w->obs_fit);
  }
  return NULL;
}

/* obs_dataset_api.rs:57-99 (the reference loops sequentially; threads here only speed the checker) */
int orc_fit_lsq(const outfit_iod_params_t *ip, const outfit_diffcor_config_t *cfg,
                const outfit_obs_batch_t *batch, const outfit_ephem_t *ephem,
                const outfit_gauss_result_t *initial, int n_threads, outfit_lsq_result_t *out,
                outfit_obs_fit_t *obs_fit) {
  int rc = orc_iod_params_validate(ip);
  if (rc) return rc;
  if (cfg->propagator != OUTFIT_PROPAGATOR_TWOBODY) return OUTFIT_ERR_INVALID_PARAM;
  if (!batch || !batch->traj_offset || !out) return OUTFIT_ERR_INVALID_BATCH;
  if (ephem && batch->obs_geo_ecl) {
    uint64_t n = batch->traj_offset[batch->n_traj];
    for (uint64_t o = 0; o < n; ++o) {
      double p[3];
      if (orc_earth_ephemeris(ephem, batch->mjd_tt[o], p)) return OUTFIT_ERR_EPHEM_RANGE;
    }
  }
  volatile uint64_t next = 0;
  lsq_work_t w = {ip, cfg, batch, ephem, initial, &next, out, obs_fit};
  if (n_threads <= 1) { lsq_worker(&w); return 0; }
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, lsq_worker, &w);
  for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
  free(th);
  return 0;
}
