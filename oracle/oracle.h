/*
 * oracle.h — CPU restatement of Outfit's Gauss-IOD hot path (plain C, fp64).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboutfit_oracle.so, and only as the checker / CPU baseline. The product
 * path (outfit_amd/csrc, liboutfit_hip.so) neither links nor calls this code.
 *
 * Parity pin: the reference is pure Rust and cannot be built here (no
 * cargo/rustc, crates not vendored) — see DESIGN.md. This restatement follows
 * the reference source function by function (file:line cited at each
 * function) and is pinned by the reference's own self-contained known-answer
 * tests, committed as data under tests/golden/.
 * Arithmetic that lives in third-party crates absent from /root/reference is
 * restated from their published algorithm:
 *   aberth 0.4.1 (Cargo.lock) — Aberth–Ehrlich iteration, pinned by
 *                               gauss.rs:1543-1570 (values AND order);
 *   roots 0.0.8               — find_root_newton_raphson + SimpleConvergency,
 *                               pinned by equinoctial_element.rs:1267-1315;
 *   nalgebra 0.34.2           — 3x3 try_inverse / dot / norm / gemv operation
 *                               order, pinned by gauss.rs:1470-1487.
 *
 * Build: -O2 -ffp-contract=off -fno-fast-math (Rust never contracts a*b+c).
 */
#ifndef OUTFIT_ORACLE_H
#define OUTFIT_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/outfit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* constants.rs */
#define ORC_DPI 6.283185307179586476925286766559 /* constants.rs:31  2*PI (rounds to 2.0*f64 PI) */
#define ORC_PI 3.14159265358979323846
#define ORC_EPS 2.220446049250313e-16 /* f64::EPSILON */
#define ORC_GAUSS_GRAV 0.01720209895  /* constants.rs:70 */
#define ORC_AU 149597870.7            /* constants.rs:37 */
#define ORC_VLIGHT 2.99792458e5       /* constants.rs:76 */
#define ORC_SECONDS_PER_DAY 86400.0   /* constants.rs:34 */

/* ---- crmath.c: correctly-rounded sin/cos/acos (see the header of that file) ---- */
extern int orc_libm_mask; /* per-function platform-libm selection (study aid, crmath.c) */
extern int orc_use_libm; /* 1 = platform libm instead (results then depend on the libm) */
extern int orc_tie_order;  /* triplets.c: 1 = reverse runs of equal triplet weights (test aid) */
extern int orc_math_glibc; /* 1 = sin/cos/sinh/cosh restated from glibc 2.35 (glibcmath.c), the rest correctly rounded */
double orc_glibc_sin(double x);   /* glibcmath.c */
double orc_glibc_cos(double x);
double orc_glibc_expm1(double x);
double orc_glibc_exp(double x);
double orc_glibc_sinh(double x);
double orc_glibc_cosh(double x);
void   orc_glibc_compare_with_host(uint64_t n, uint64_t seed, uint64_t *counts /* [6] */);
extern const double orc_aberth_cos_sin[8][2];
void   orc_sincos(double x, double *s, double *c);
double orc_glibc_sin1(double x);   /* glibcmath.c: the standalone sin / cos entry points (glibc's FMA variants) */
double orc_glibc_cos1(double x);
double orc_sin1(double x);         /* crmath.c: an ISOLATED sin / cos of the reference (no partner on the same operand) */
double orc_cos1(double x);
double orc_cos1_s(double x);
double orc_glibc_acos(double x);              /* glibcmath3.c: glibc 2.35's acos and log, restated */
double orc_glibc_log(double x);
double orc_glibc_atan2(double y, double x);   /* glibcmath2.c: glibc 2.35's atan2 / tan / hypot, restated */
double orc_glibc_tan(double x);
double orc_glibc_hypot(double x, double y);
void   orc_sincos_prepass(double x, double *s, double *c); /* earthrot.c: correctly rounded in both math modes */
double orc_sin(double x);
double orc_cos(double x);
double orc_acos(double x);
double orc_sinh(double x); /* binary128 -> double: correctly rounded (hyperbolic cold start) */
double orc_cosh(double x);
double orc_log(double x);
double orc_atan2(double y, double x);
extern __thread int orc_cr_scope; /* > 0 inside the differential corrector: the *_s functions are correctly rounded */
double orc_sin_s(double x);
double orc_cos_s(double x);
double orc_tan_s(double x);
double orc_atan2_s(double y, double x);
double orc_hypot_s(double x, double y);

/* ---- kepler/ --------------------------------------------------------------- */
typedef struct {
  double dt, r0, sig0, mu, alpha, e0;
  double convergency;          /* SolverParams.convergency   (params.rs:26)  */
  int has_guess;               /* SolverParams.psi_guess     (params.rs:28)  */
  double psi_guess;
  int max_iter_prelim_kepuni;  /* params.rs:30, default 20                    */
  int kind;                    /* 0 NewtonRaphson, 1 BrentDecker, 2 Auto      */
  int parabolic_method;        /* ParabolicPrelimMethod: 0 Cardano (default), 1 Newton (prelim_parabolic.rs:77-93) */
} orc_kepler_params;

typedef struct { double psi, s0, s1, s2, s3; } orc_kepler_solution;

void   orc_s_funct(double psi, double alpha, double s[4]);
double orc_principal_angle(double a);
double orc_prelim_elliptic(const orc_kepler_params *p);
double orc_prelim_hyperbolic(const orc_kepler_params *p);
double orc_prelim_parabolic(const orc_kepler_params *p);
double orc_prelim_kepuni(const orc_kepler_params *p);
int    orc_solve_kepuni_with_guess(const orc_kepler_params *p, orc_kepler_solution *sol);
int    orc_solve_kepuni_brent_dekker(const orc_kepler_params *p, orc_kepler_solution *sol);
int    orc_kepler_solve(const orc_kepler_params *p, orc_kepler_solution *sol);
/* returns 0 ok, nonzero Err(VelocityCorrectionError / Kepler convergence) */
int    orc_velocity_correction_with_guess(const double x1[3], const double x2[3], const double v2[3],
                                          double dt, double peri_max, double ecc_max, int has_guess,
                                          double chi_guess, double eps, double v_out[3], double *f,
                                          double *g, double *chi);
/* out: r1[3] v1[3] f g fdot gdot psi ; returns status as outfit_hip_propagate_universal */
int    orc_propagate_with_params(const double r[3], const double v[3], const orc_kepler_params *p, double out[11]);
int    orc_propagate_universal_ex(const double r[3], const double v[3], double t0, double t1,
                                  const outfit_solver_type_t *st, double out[11]);
int    orc_propagate_universal(const double r[3], const double v[3], double t0, double t1,
                               double convergency, int kind, double out[11]);

/* ---- orb_elem.rs / orbit_type/ ----------------------------------------------- */
/* returns 0 if None (zero angular momentum) else 1; out = accepted, ecc, peri, energy */
int orc_eccentricity_control(const double r[3], const double v[3], double peri_max, double ecc_max,
                             int *accepted, double *ecc, double *peri, double *energy);
typedef struct {
  int kind;        /* OUTFIT_KIND_* */
  double epoch;
  double e[6];     /* Keplerian (a,e,i,Om,om,M) or Cometary (q,e,i,Om,om,nu) */
} orc_elements;
void orc_ccek1(const double pos[3], const double vel[3], double epoch, orc_elements *out);
typedef struct { double epoch, a, h, k, p, q, lambda; } orc_equinoctial;
/* 0 ok, OUTFIT_TRAJ_INVALID_CONVERSION, OUTFIT_TRAJ_INVALID_ORBIT */
int orc_to_equinoctial(const orc_elements *el, orc_equinoctial *eq);
/* roots::find_root_newton_raphson restatement; 0 ok, 1 Err */
int orc_solve_kepler_equation(const orc_equinoctial *eq, double lambda1, double varpi, double *F);
int orc_propagate_twobody(const orc_equinoctial *eq, double t0, double t1, double pos[3],
                          double vel[3]);

/* ---- aberth crate ------------------------------------------------------------ */
#define ORC_ABERTH_CONVERGED 0
#define ORC_ABERTH_MAXITER 1
#define ORC_ABERTH_FAILED 2
int orc_aberth9(const double poly[9], uint32_t max_iter, double eps, double re[8], double im[8],
                uint32_t *iterations);

/* ---- gauss.rs ---------------------------------------------------------------- */
typedef struct {
  uint32_t idx[3];
  double ra[3], dec[3], t[3];
  double obs[3][3]; /* obs[row][col]: row = x,y,z ; col = epoch (Matrix3 columns = epochs) */
} orc_gauss_obs;

typedef struct {
  double tau1, tau3;
  double S[3][3], Sinv[3][3]; /* [row][col] */
  double a[3], b[3];
} orc_gauss_prelim;

int  orc_gauss_prelim_fn(const orc_gauss_obs *g, orc_gauss_prelim *out); /* 0 ok, 1 singular */
void orc_coeff_eight_poly(const orc_gauss_obs *g, const orc_gauss_prelim *p, double *c6, double *c3,
                          double *c0);
uint32_t orc_descartes_upper_bound(double c0, double c3, double c6, double zero_eps);
int  orc_solve_8poly(const double poly[9], uint32_t max_iter, double aberth_eps, double root_eps,
                     double roots[8], int *n_roots);
int  orc_position_vector_and_reference_epoch(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                                             const double S[3][3], const double Sinv[3][3],
                                             const double c[3], double pos[3][3], double *epoch);
void orc_gibbs_correction(const double pos[3][3], double tau1, double tau3, double vel[3]);
int  orc_accept_root(const orc_gauss_obs *g, const outfit_iod_params_t *ip, double root,
                     const orc_gauss_prelim *p, double pos[3][3], double vel[3], double *epoch);
int  orc_pos_and_vel_correction(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                                const double pos_in[3][3], const double vel_in[3],
                                const double S[3][3], const double Sinv[3][3], double peri_max,
                                double ecc_max, double err_max, uint64_t itmax, double pos[3][3],
                                double vel[3], double *epoch);
typedef struct {
  int stage; /* OUTFIT_STAGE_* */
  orc_elements el;
} orc_gauss_result;
/* returns 0 ok else OUTFIT_CAUSE_* ; n_out <= 8 (at most 8 roots) */
int orc_prelim_orbit_all(const orc_gauss_obs *g, const outfit_iod_params_t *ip,
                         orc_gauss_result out[8], int *n_out);
int orc_prelim_orbit(const orc_gauss_obs *g, const outfit_iod_params_t *ip, orc_gauss_result *out);

/* ---- triplet_generation/ ------------------------------------------------------- */
/* returns number of triplets written (<= max_triplets); idx = 3 per triplet; w = weights */
size_t orc_generate_triplet_indices(const double *mjd, size_t n_obs, const outfit_iod_params_t *ip,
                                    uint32_t *idx, double *w);
size_t orc_downsample_uniform_with_edges(size_t n, size_t max_keep, size_t *keep);

/* ---- jpl_ephem/horizon + ephemeris/ ---------------------------------------------- */
/* Earth wrt Sun, equatorial J2000, AU. 0 ok / OUTFIT_ERR_EPHEM_RANGE */
int  orc_earth_ephemeris(const outfit_ephem_t *eph, double mjd_tt, double pos[3]);
void orc_observer_position_equ(const outfit_ephem_t *eph, double mjd_tt, const double geo_ecl[3],
                               double out[3], int *err);
void orc_topocentric_radec(const double ast_pos[3], const double ast_vel[3], const double obs_pos[3],
                           double *ra, double *dec);
/* ephemeris_error ; 0 ok, 1 Err */
int orc_ephemeris_error(const orc_equinoctial *eq, double mjd, double ra, double dec, double sra,
                        double sdec, const double obs_pos_equ[3], double *term);

/* ---- trajectory.rs / obs_dataset_api.rs ------------------------------------------- */
void orc_select_rms_interval(const double *mjd, size_t n_obs, const uint32_t idx[3],
                             const outfit_iod_params_t *ip, size_t *i_start, size_t *i_end);
int  orc_fit_full_iod(const outfit_iod_params_t *params, const outfit_obs_batch_t *batch,
                      const outfit_ephem_t *ephem, int n_threads, int hoist_ephem,
                      outfit_gauss_result_t *out);
void orc_estimate_best_orbit(const outfit_iod_params_t *ip, const outfit_obs_batch_t *b,
                             const outfit_ephem_t *eph, uint64_t traj, int hoist_ephem,
                             outfit_gauss_result_t *out);
void orc_iod_params_default(outfit_iod_params_t *p);
int  orc_iod_params_validate(const outfit_iod_params_t *p);

/* ---- differential_orbit_correction/  --------------------------------- */
typedef struct { /* ObservationEquation, least_square.rs:66-95 */
  double d_ra[6], d_dec[6];
  double residual_ra, residual_dec;
  double weight_ra, weight_dec, weight_cross;
  int active;
} orc_obs_equation;
typedef struct { /* ObsFitData, obs_fit_data.rs:63-100 */
  double sigma_ra, sigma_dec, bias_ra, bias_dec, residual_ra, residual_dec, chi;
  int selection; /* OUTFIT_OBS_* */
} orc_obs_fit;
typedef struct { /* DifferentialCorrectionResult + OrbitalUncertainty, least_square.rs:23-60,135-160 */
  double dx[6];
  double normal[6][6], covariance[6][6]; /* [i][j] = matrix[(i,j)] */
  int inversion_succeeded, used_qr;
  double normalised_rms;
  uint32_t num_measurements;
} orc_lsq_solution;
typedef struct { /* DifferentialCorrectionOutput, diff_cor.rs:185-209 */
  orc_equinoctial elements;
  double covariance[6][6], normal[6][6];
  double normalised_rms;
  uint32_t total_newton_iterations, num_measurements, outer_passes;
} orc_diffcor_output;

void   orc_diffcor_config_default(outfit_diffcor_config_t *c);
int    orc_is_bizarre(const orc_equinoctial *eq, const outfit_diffcor_config_t *c);
void   orc_compute_derivative(const orc_equinoctial *eq, double t0, double t1, double n, double lam1,
                              double F, double inv_u, double beta, double sF, double cF, double xe,
                              double ye, double vxe, double vye, const double f[3], const double g[3],
                              const double pos[3], const double vel[3], double dpos[6][3],
                              double dvel[6][3]);
int    orc_propagate_twobody_partials(const orc_equinoctial *eq, double t0, double t1, double pos[3],
                                      double vel[3], double dpos[6][3], double dvel[6][3]);
void   orc_topocentric_radec_and_partials(const double ast_pos[3], const double ast_vel[3],
                                          const double obs_pos[3], double *ra, double *dec,
                                          double d_ra[3], double d_dec[3]);
int    orc_obs_and_partials_2body(const orc_equinoctial *eq, double mjd, const double obs_pos_equ[3],
                                  double *ra, double *dec, double d_ra_d_elem[6],
                                  double d_dec_d_elem[6]);
double orc_angular_diff(double a, double b);
void   orc_solve_weighted_least_squares(const orc_obs_equation *eqs, size_t n,
                                        const uint8_t free_elements[6], orc_lsq_solution *out);
void   orc_rescale_covariance(double covariance[6][6], double normal[6][6], size_t num_free,
                              size_t num_measurements, double normalised_rms);
size_t orc_update_observation_selection(orc_obs_fit *fit, const orc_obs_equation *eqs, size_t n,
                                        const double covariance[6][6], double chi2_reject,
                                        double chi2_recover);
int    orc_run_differential_correction(const orc_equinoctial *initial, const double *mjd,
                                       const double *ra, const double *dec, const double *obs_pos,
                                       size_t n, orc_obs_fit *fit, const outfit_diffcor_config_t *cfg,
                                       orc_diffcor_output *out);
void   orc_differential_correction(const outfit_iod_params_t *ip, const outfit_diffcor_config_t *cfg,
                                   const outfit_obs_batch_t *b, const outfit_ephem_t *eph, uint64_t traj,
                                   const outfit_gauss_result_t *initial, outfit_lsq_result_t *out,
                                   outfit_obs_fit_t *obs_fit);
int    orc_fit_lsq(const outfit_iod_params_t *ip, const outfit_diffcor_config_t *cfg,
                   const outfit_obs_batch_t *batch, const outfit_ephem_t *ephem,
                   const outfit_gauss_result_t *initial, int n_threads, outfit_lsq_result_t *out,
                   outfit_obs_fit_t *obs_fit);

/* ---- observer-position pre-pass: earthrot.c ------------------------------ */
typedef struct { int kind; /* 0 Equm, 1 Equt, 2 Eclm */ int j2000; double date; } orc_ref_system;
double orc_obleq(double tjm);
void   orc_nutn80(double tjm, double *dpsi_arcsec, double *deps_arcsec);
void   orc_rotmt(double alpha, int k, double m[3][3]);
void   orc_rnut80(double tjm, double m[3][3]);
double orc_equequ(double tjm);
void   orc_prec(double tjm, double m[3][3]);
int    orc_rotpn(const orc_ref_system *src, const orc_ref_system *dst, double rot[3][3]);
double orc_gmst(double tjm);
void   orc_pvobs(double tt, double tut, const double dxbf[3], double out[3]);
void   orc_helio_position(const double earth_equ[3], const double geo_ecl[3], double out[3]);
int    orc_observer_prepass(const outfit_ephem_t *eph, double tt, double tut, const double dxbf[3],
                            double geo_ecl[3], double helio_equ[3]);
int    orc_observer_positions(const outfit_ephem_t *eph, uint64_t n, const double *tt, const double *tut,
                              const double *dxbf, double *geo_ecl, double *helio_equ);

#ifdef __cplusplus
}
#endif
#endif
