/*
 * ephem.c — oracle restatement of the legacy-DE Chebyshev interpolation
 * (src/jpl_ephem/horizon/{horizon_data,horizon_records}.rs) and of the
 * per-observation residual (src/ephemeris/{observation_ephemeris,aberration}.rs).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include "oracle.h"

#define VLIGHT_AU (ORC_VLIGHT / ORC_AU * ORC_SECONDS_PER_DAY)

static const double ROT_EQU_TO_ECL[3][3] = {{1.0, 0.0, 0.0},
                                            {0.0, 9.174820620691818e-1, 3.977771559319137e-1},
                                            {0.0, -3.977771559319137e-1, 9.174820620691818e-1}};
static const double ROT_ECL_TO_EQU[3][3] = {{1.0, 0.0, 0.0},
                                            {0.0, 9.174820620691818e-1, -3.977771559319137e-1},
                                            {0.0, 3.977771559319137e-1, 9.174820620691818e-1}};

static inline void gemv3(const double A[3][3], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = (A[r][0] * x[0] + A[r][1] * x[1]) + A[r][2] * x[2];
}

/* horizon_records.rs:204-298 (position only: compute_velocity = false) */
static void cheb_position(const double *c /* x[nc] y[nc] z[nc] */, uint32_t nc, double tau,
                          uint32_t n_sub, double out[3]) {
  double dt1 = (double)(int64_t)tau; /* `tau as i64` */
  double temp = (double)n_sub * tau;
  double fr = fmod(temp, 1.0);       /* rem_euclid(1.0) */
  if (fr < 0.0) fr += 1.0;
  double tc = 2.0 * (fr + dt1) - 1.0;
  double tcheb[32] = {0.0}; /* vec![0.0; n_coeff]; ncoef <= 32 is checked by the caller */
  double twot = 0.0;
  tcheb[0] = 1.0;
  if (tc != tcheb[1]) { tcheb[1] = tc; twot = tc + tc; }
  for (uint32_t i = 2; i < nc; ++i) tcheb[i] = twot * tcheb[i - 1] - tcheb[i - 2];
  for (int ax = 0; ax < 3; ++ax) {
    double s = 0.0;
    for (uint32_t i = 0; i < nc; ++i) s += c[(size_t)ax * nc + i] * tcheb[i];
    out[ax] = s;
  }
}

/* horizon_data.rs:711-777,810-849 + jpl_ephem/mod.rs:145-162 + interpolation_result.rs:82-84 */
int orc_earth_ephemeris(const outfit_ephem_t *eph, double mjd_tt, double pos[3]) {
  for (int b = 0; b < 3; ++b)
    if (eph->ncoef[b] < 2 || eph->ncoef[b] > 32 || eph->nsub[b] < 1) return OUTFIT_ERR_INVALID_BATCH;
  double et_trunc = trunc(mjd_tt);
  double et_jd = 2400000.5 + et_trunc;
  if (et_jd < eph->jd_start || et_jd > eph->jd_end) return OUTFIT_ERR_EPHEM_RANGE;
  double nrf = floor((et_jd - eph->jd_start) / eph->block_days);
  uint64_t nr = (uint64_t)nrf;
  if (fabs(et_jd - eph->jd_end) < 1e-10) nr -= 1;
  if (nr >= eph->n_blocks) return OUTFIT_ERR_EPHEM_RANGE;
  double interval_start = (double)nr * eph->block_days + eph->jd_start;
  double tau = ((et_jd - interval_start) + (mjd_tt - et_trunc)) / eph->block_days;

  size_t per_block = 0, off[3];
  for (int b = 0; b < 3; ++b) { off[b] = per_block; per_block += (size_t)3 * eph->ncoef[b] * eph->nsub[b]; }
  const double *blk = eph->coef + (size_t)nr * per_block;
  double body[3][3];
  for (int b = 0; b < 3; ++b) {
    double ns = (double)eph->nsub[b];
    double sif = floor(tau * ns);
    if (sif > ns - 1.0) sif = ns - 1.0; /* f64::min */
    size_t si = (size_t)sif;
    cheb_position(blk + off[b] + si * 3 * eph->ncoef[b], eph->ncoef[b], tau, eph->nsub[b], body[b]);
  }
  for (int i = 0; i < 3; ++i) {
    double earth = body[0][i] - body[1][i] / (1.0 + eph->emrat);
    pos[i] = (earth - body[2][i]) / ORC_AU;
  }
  return 0;
}

/* observation_ephemeris.rs:303-318 */
void orc_observer_position_equ(const outfit_ephem_t *eph, double mjd_tt, const double geo_ecl[3],
                               double out[3], int *err) {
  double earth_equ[3], earth_ecl[3], obs_ecl[3];
  *err = orc_earth_ephemeris(eph, mjd_tt, earth_equ);
  if (*err) { out[0] = out[1] = out[2] = NAN; return; }
  gemv3(ROT_EQU_TO_ECL, earth_equ, earth_ecl);
  for (int i = 0; i < 3; ++i) obs_ecl[i] = geo_ecl[i] + earth_ecl[i];
  gemv3(ROT_ECL_TO_EQU, obs_ecl, out);
}

/* observation_ephemeris.rs:222-275 (ra, dec only) + aberration.rs:139-145 */
void orc_topocentric_radec(const double ast_pos[3], const double ast_vel[3], const double obs_pos[3],
                           double *ra, double *dec) {
  double rel[3], cor[3];
  for (int i = 0; i < 3; ++i) rel[i] = ast_pos[i] - obs_pos[i];
  double dt = sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2]) / VLIGHT_AU;
  for (int i = 0; i < 3; ++i) cor[i] = rel[i] - dt * ast_vel[i];
  double rho_xy = orc_hypot_s(cor[0], cor[1]); /* platform libm, or glibc restated in that mode (crmath.c) */
  *dec = orc_atan2_s(cor[2], rho_xy);
  double a = fmod(orc_atan2_s(cor[1], cor[0]), ORC_DPI); /* rem_euclid(TAU) */
  if (a < 0.0) a += ORC_DPI;
  *ra = a;
}

/* observation_ephemeris.rs:288-296,322-339,369-416 */
int orc_ephemeris_error(const orc_equinoctial *eq, double mjd, double ra, double dec, double sra,
                        double sdec, const double obs_pos_equ[3], double *term) {
  if (sqrt(eq->h * eq->h + eq->k * eq->k) >= 1.0) return 1; /* check_elliptical_orbit */
  double dt = mjd - eq->epoch;
  double pe[3], ve[3];
  if (orc_propagate_twobody(eq, 0.0, dt, pe, ve)) return 1;
  double pq[3], vq[3];
  gemv3(ROT_ECL_TO_EQU, pe, pq);
  gemv3(ROT_ECL_TO_EQU, ve, vq);
  double alpha, delta;
  orc_topocentric_radec(pq, vq, obs_pos_equ, &alpha, &delta);
  double da = fmod(ra - alpha, ORC_DPI); /* Rust `%` keeps the sign of the dividend (quirk Q6) */
  if (da > ORC_PI) da -= ORC_DPI;
  double dd = dec - delta;
  double t1 = orc_cos1_s(dec) * (da / sra); /* isolated cos: observation_ephemeris.rs:412 */
  double t2 = dd / sdec;
  *term = t1 * t1 + t2 * t2;
  return 0;
}
