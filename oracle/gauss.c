/*
 * gauss.c — oracle restatement of src/initial_orbit_determination/gauss.rs.
 * Matrices are m[row][col]; for the observer / position matrices the columns
 * are the three epochs (gauss.rs:156, triplet_generation/mod.rs:423-427).
 * nalgebra operation order (gemv by columns, 3-vector dot = (a+b)+c, 3x3
 * adjugate inverse) is reproduced; see SURVEY.md Appendix C.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

uint64_t orc_dbg_aberth_sweeps[64]; /* histogram of Aberth sweeps per candidate (debug counter) */

#define VLIGHT_AU (ORC_VLIGHT / ORC_AU * ORC_SECONDS_PER_DAY) /* constants.rs:79 */

static inline double dot3(const double a[3], const double b[3]) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* nalgebra gemv: y = ((A[:,0]*x0) + A[:,1]*x1) + A[:,2]*x2 */
static inline void gemv3(const double A[3][3], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = (A[r][0] * x[0] + A[r][1] * x[1]) + A[r][2] * x[2];
}
static inline double colnorm(const double A[3][3], int c) {
  return sqrt((A[0][c] * A[0][c] + A[1][c] * A[1][c]) + A[2][c] * A[2][c]);
}

/* debug statistics (single-threaded use only): iterations run by the last correction loop */
int orc_dbg_corr_iters = 0;

/* nalgebra 0.34 try_inverse, 3x3 branch (adjugate / determinant) — pinned
 * bit-exactly by gauss.rs:1470-1487. */
static int inv3(const double m[3][3], double o[3][3]) {
  double m11 = m[0][0], m12 = m[0][1], m13 = m[0][2];
  double m21 = m[1][0], m22 = m[1][1], m23 = m[1][2];
  double m31 = m[2][0], m32 = m[2][1], m33 = m[2][2];
  double minor_m12_m23 = m22 * m33 - m32 * m23;
  double minor_m11_m23 = m21 * m33 - m31 * m23;
  double minor_m11_m22 = m21 * m32 - m31 * m22;
  double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
  if (det == 0.0) return 0;
  o[0][0] = minor_m12_m23 / det;
  o[0][1] = (m13 * m32 - m33 * m12) / det;
  o[0][2] = (m12 * m23 - m22 * m13) / det;
  o[1][0] = -minor_m11_m23 / det;
  o[1][1] = (m11 * m33 - m31 * m13) / det;
  o[1][2] = (m13 * m21 - m23 * m11) / det;
  o[2][0] = minor_m11_m22 / det;
  o[2][1] = (m12 * m31 - m32 * m11) / det;
  o[2][2] = (m11 * m22 - m21 * m12) / det;
  return 1;
}

/* gauss.rs:532-549 (+ unit_vector/unit_matrix :488-502) */
int orc_gauss_prelim_fn(const orc_gauss_obs *g, orc_gauss_prelim *o) {
  double tau1 = ORC_GAUSS_GRAV * (g->t[0] - g->t[1]);
  double tau3 = ORC_GAUSS_GRAV * (g->t[2] - g->t[1]);
  double tau13 = tau3 - tau1;
  o->tau1 = tau1; o->tau3 = tau3;
  o->a[0] = tau3 / tau13; o->a[1] = -1.0; o->a[2] = -(tau1 / tau13);
  o->b[0] = o->a[0] * (tau13 * tau13 - tau3 * tau3) / 6.0;
  o->b[1] = 0.0;
  o->b[2] = o->a[2] * (tau13 * tau13 - tau1 * tau1) / 6.0;
  for (int c = 0; c < 3; ++c) {
    double sr, cr, sd, cd;
    orc_sincos(g->ra[c], &sr, &cr);
    orc_sincos(g->dec[c], &sd, &cd);
    o->S[0][c] = cr * cd;
    o->S[1][c] = sr * cd;
    o->S[2][c] = sd;
  }
  return inv3(o->S, o->Sinv) ? 0 : 1;
}

/* gauss.rs:585-614 */
void orc_coeff_eight_poly(const orc_gauss_obs *g, const orc_gauss_prelim *p, double *c6, double *c3,
                          double *c0) {
  double ra[3], rb[3];
  gemv3(g->obs, p->a, ra);
  gemv3(g->obs, p->b, rb);
  double row1[3] = {p->Sinv[1][0], p->Sinv[1][1], p->Sinv[1][2]};
  double a2 = dot3(row1, ra), b2 = dot3(row1, rb);
  double R2[3] = {g->obs[0][1], g->obs[1][1], g->obs[2][1]};
  double r22 = ((0.0 + R2[0] * R2[0]) + R2[1] * R2[1]) + R2[2] * R2[2];
  double s2r2 = ((0.0 + p->S[0][1] * R2[0]) + p->S[1][1] * R2[1]) + p->S[2][1] * R2[2];
  *c6 = -(a2 * a2) - r22 - (2.0 * a2 * s2r2);
  *c3 = -(2.0 * b2 * (a2 + s2r2));
  *c0 = -(b2 * b2);
}

/* gauss.rs:214-240 ; f64::is_sign_positive tests the sign bit (quirk Q11) */
uint32_t orc_descartes_upper_bound(double c0, double c3, double c6, double zero_eps) {
  double v[3] = {c6, c3, c0};
  int last = 1, count = 0;
  for (int i = 0; i < 3; ++i) {
    int s;
    if (fabs(v[i]) <= zero_eps) s = 0;
    else s = (signbit(v[i]) || isnan(v[i])) ? -1 : 1; /* NaN: the sign bit is platform-dependent; negative by convention (device too) */
    if (s == 0) continue;
    if (last != 0 && s != last) ++count;
    last = s;
  }
  return (uint32_t)count;
}

/* gauss.rs:648-667 */
int orc_solve_8poly(const double poly[9], uint32_t max_iter, double aberth_eps, double root_eps,
                    double roots[8], int *n_roots) {
  double re[8], im[8];
  int st = orc_aberth9(poly, max_iter, aberth_eps, re, im, NULL);
  *n_roots = 0;
  if (st == ORC_ABERTH_FAILED) return 1;
  for (int i = 0; i < 8; ++i)
    if (re[i] > 0.0 && fabs(im[i]) < root_eps) roots[(*n_roots)++] = re[i];
  return 0;
}

/* gauss.rs:702-724 */
int orc_position_vector_and_reference_epoch(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                                            const double S[3][3], const double Sinv[3][3],
                                            const double c[3], double pos[3][3], double *epoch) {
  double gcap[3], crhom[3], rho[3];
  gemv3(g->obs, c, gcap);
  gemv3(Sinv, gcap, crhom);
  for (int i = 0; i < 3; ++i) rho[i] = -(crhom[i] / c[i]);
  if (rho[1] < ip->min_rho2_au) return 1; /* SpuriousRootDetected */
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) pos[r][k] = g->obs[r][k] + rho[k] * S[r][k];
  *epoch = g->t[1] - rho[1] / VLIGHT_AU;
  return 0;
}

/* gauss.rs:754-781 */
void orc_gibbs_correction(const double pos[3][3], double tau1, double tau3, double vel[3]) {
  double tau13 = tau3 - tau1;
  double n1 = colnorm(pos, 0), n2 = colnorm(pos, 1), n3 = colnorm(pos, 2);
  double r1m3 = 1.0 / (n1 * n1 * n1);
  double r2m3 = 1.0 / (n2 * n2 * n2);
  double r3m3 = 1.0 / (n3 * n3 * n3);
  double d1 = tau3 * (r1m3 / 12.0 - 1.0 / (tau1 * tau13));
  double d2 = (tau1 + tau3) * (r2m3 / 12.0 - 1.0 / (tau1 * tau3));
  double d3 = -tau1 * (r3m3 / 12.0 + 1.0 / (tau3 * tau13));
  double d[3] = {-d1, d2, d3};
  for (int r = 0; r < 3; ++r)
    vel[r] = ORC_GAUSS_GRAV * (((0.0 + pos[r][0] * d[0]) + pos[r][1] * d[1]) + pos[r][2] * d[2]);
}

/* gauss.rs:816-870 */
int orc_accept_root(const orc_gauss_obs *g, const outfit_iod_params_t *ip, double root,
                    const orc_gauss_prelim *p, double pos[3][3], double vel[3], double *epoch) {
  double r2m3 = 1.0 / (root * root * root);
  double c[3] = {p->a[0] + p->b[0] * r2m3, -1.0, p->a[2] + p->b[2] * r2m3};
  if (orc_position_vector_and_reference_epoch(g, ip, p->S, p->Sinv, c, pos, epoch)) return 0;
  double r2[3] = {pos[0][1], pos[1][1], pos[2][1]};
  orc_gibbs_correction(pos, p->tau1, p->tau3, vel);
  int acc;
  double e, q, en;
  if (!orc_eccentricity_control(r2, vel, ip->max_perihelion_au, ip->max_ecc, &acc, &e, &q, &en))
    return 0;
  return acc ? 1 : 0;
}

/* gauss.rs:1284-1418 ; returns 1 = Some, 0 = None */
int orc_pos_and_vel_correction(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                               const double pos_in[3][3], const double vel_in[3],
                               const double S[3][3], const double Sinv[3][3], double peri_max,
                               double ecc_max, double err_max, uint64_t itmax, double pos[3][3],
                               double vel[3], double *epoch_out) {
  memcpy(pos, pos_in, sizeof(double) * 9);
  memcpy(vel, vel_in, sizeof(double) * 3);
  double epoch = 0.0;
  int has01 = 0, has21 = 0;
  double chi01 = 0.0, chi21 = 0.0;
  double dt01 = g->t[0] - g->t[1];
  double dt21 = g->t[2] - g->t[1];
  if (fabs(dt01) <= ORC_EPS || fabs(dt21) <= ORC_EPS) return 0;

  for (uint64_t it = 0; it < itmax; ++it) {
    orc_dbg_corr_iters = (int)it + 1;
    double r1[3] = {pos[0][0], pos[1][0], pos[2][0]};
    double r2[3] = {pos[0][1], pos[1][1], pos[2][1]};
    double r3[3] = {pos[0][2], pos[1][2], pos[2][2]};
    double v1[3], v2[3], f1, g1, c1, f2, g2, c2;
    int e1 = orc_velocity_correction_with_guess(r1, r2, vel, dt01, peri_max, ecc_max, has01, chi01,
                                                ip->kepler_eps, v1, &f1, &g1, &c1);
    int e2 = orc_velocity_correction_with_guess(r3, r2, vel, dt21, peri_max, ecc_max, has21, chi21,
                                                ip->kepler_eps, v2, &f2, &g2, &c2);
    if (e1 || e2) continue;
    has01 = 1; chi01 = c1;
    has21 = 1; chi21 = c2;
    if (!isfinite(g1) || !isfinite(g2)) continue;
    double nv[3];
    for (int i = 0; i < 3; ++i) nv[i] = (v1[i] + v2[i]) * 0.5;
    double fl = f1 * g2 - f2 * g1;
    if (!isfinite(fl) || fabs(fl) < ORC_EPS) continue;
    double inv_f = 1.0 / fl;
    double c[3] = {g2 * inv_f, -1.0, -g1 * inv_f};
    double np[3][3], ne;
    if (orc_position_vector_and_reference_epoch(g, ip, S, Sinv, c, np, &ne)) continue;
    double rmid[3] = {np[0][1], np[1][1], np[2][1]};
    int acc;
    double e, q, en;
    if (!orc_eccentricity_control(rmid, nv, peri_max, ecc_max, &acc, &e, &q, &en)) return 0;
    if (!acc) return 0;
    /* Matrix3::norm: sum over columns of (x^2 + y^2) + z^2, accumulated from 0 */
    double s = 0.0, sd = 0.0;
    for (int k = 0; k < 3; ++k) {
      s += (np[0][k] * np[0][k] + np[1][k] * np[1][k]) + np[2][k] * np[2][k];
      double dx = np[0][k] - pos[0][k], dy = np[1][k] - pos[1][k], dz = np[2][k] - pos[2][k];
      sd += (dx * dx + dy * dy) + dz * dz;
    }
    double denom = sqrt(s);
    if (!isfinite(denom) || denom <= ORC_EPS) continue;
    double rel = sqrt(sd) / denom;
    memcpy(pos, np, sizeof np);
    memcpy(vel, nv, sizeof nv);
    epoch = ne;
    if (rel <= err_max) break;
  }
  *epoch_out = epoch;
  return 1;
}


/* gauss.rs:906-923 (ROT_EQUMJ2000_TO_ECLMJ2000, constants.rs:93-103) + orbit_type/mod.rs:281-287 */
static void orbit_from_state(const double pos[3][3], const double vel[3], double epoch,
                             orc_elements *el) {
  static const double ROT[3][3] = {{1.0, 0.0, 0.0},
                                   {0.0, 9.174820620691818e-1, 3.977771559319137e-1},
                                   {0.0, -3.977771559319137e-1, 9.174820620691818e-1}};
  double r[3] = {pos[0][1], pos[1][1], pos[2][1]};
  double re[3], ve[3];
  gemv3(ROT, r, re);
  gemv3(ROT, vel, ve);
  orc_ccek1(re, ve, epoch, el);
}

/* gauss.rs:1119-1206 */
int orc_prelim_orbit_all(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                         orc_gauss_result out[8], int *n_out) {
  *n_out = 0;
  orc_gauss_prelim p;
  if (orc_gauss_prelim_fn(g, &p)) return OUTFIT_CAUSE_SINGULAR_DIRECTION_MATRIX;
  double c6, c3, c0;
  orc_coeff_eight_poly(g, &p, &c6, &c3, &c0);
  double poly[9] = {c0, 0.0, 0.0, c3, 0.0, 0.0, c6, 0.0, 1.0};
  if (orc_descartes_upper_bound(c0, c3, c6, 0.0) == 0) return OUTFIT_CAUSE_GAUSS_NO_ROOTS_FOUND;

  double re[8], im[8];
  uint32_t sweeps = 0;
  int st = orc_aberth9(poly, ip->aberth_max_iter, ip->aberth_eps, re, im, &sweeps);
  __atomic_fetch_add(&orc_dbg_aberth_sweeps[sweeps < 63 ? sweeps : 63], 1, __ATOMIC_RELAXED); /* work-distribution study */
  if (st == ORC_ABERTH_FAILED) return OUTFIT_CAUSE_POLYNOMIAL_ROOT_FINDING_FAILED;
  uint64_t max_sol = ip->max_tested_solutions;
  int n = 0;
  orc_gauss_result sol[8];
  for (int i = 0; i < 8; ++i) {
    if (!(re[i] > 0.0 && fabs(im[i]) < ip->root_imag_eps)) continue;
    double r2 = re[i];
    if (!(r2 >= ip->r2_min_au && r2 <= ip->r2_max_au)) continue;
    double pos[3][3], vel[3], epoch;
    if (!orc_accept_root(g, ip, r2, &p, pos, vel, &epoch)) continue;
    double pc[3][3], vc[3], ec;
    if (orc_pos_and_vel_correction(g, ip, pos, vel, p.S, p.Sinv, ip->max_perihelion_au,
                                   ip->max_ecc, ip->newton_eps, ip->newton_max_it, pc, vc, &ec)) {
      if ((uint64_t)n < max_sol && n < 8) {
        sol[n].stage = OUTFIT_STAGE_CORRECTED;
        orbit_from_state(pc, vc, ec, &sol[n].el);
        ++n;
      }
    } else {
      if ((uint64_t)n < max_sol && n < 8) {
        sol[n].stage = OUTFIT_STAGE_PRELIM;
        orbit_from_state(pos, vel, epoch, &sol[n].el);
        ++n;
      }
    }
    if ((uint64_t)n >= max_sol) break;
  }
  if (n == 0) return OUTFIT_CAUSE_GAUSS_NO_ROOTS_FOUND;
  memcpy(out, sol, sizeof(orc_gauss_result) * (size_t)n);
  *n_out = n;
  return 0;
}

/* gauss.rs:1238-1247 */
int orc_prelim_orbit(const orc_gauss_obs *g, const outfit_iod_params_t *ip, orc_gauss_result *out) {
  orc_gauss_result all[8];
  int n;
  int rc = orc_prelim_orbit_all(g, ip, all, &n);
  if (rc) return rc;
  for (int i = 0; i < n; ++i)
    if (all[i].stage == OUTFIT_STAGE_CORRECTED) { *out = all[i]; return 0; }
  *out = all[0];
  return 0;
}
