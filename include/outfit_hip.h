/*
 * outfit_hip.h — C ABI of the MI355X-native batched Gauss IOD engine.
 *
 * This is the drop-in boundary for the ONE hot path of FusRoman/Outfit:
 * `FitIOD::fit_full_iod{,_parallel}`
 *   (reference: src/initial_orbit_determination/obs_dataset_api.rs:145-207).
 *
 * The reference keeps `prepare_iod` (photom error model + OutfitCache,
 * obs_dataset_api.rs:254-275) on its side; what crosses this boundary is what
 * `process_traj`/`fit_single_traj`/`estimate_best_orbit` consume
 * (obs_dataset_api.rs:210-226,277-296; trajectory.rs:429-545):
 * per-observation (mjd_tt, ra, dec, sigma_ra, sigma_dec) after the error
 * model, time-sorted per trajectory, plus the two cached observer vectors
 * (cache/mod.rs:144-174) and, optionally, a legacy-DE Chebyshev table
 * (jpl_ephem/horizon/horizon_data.rs:810-849).
 *
 * Conventions
 *   - all pointers are caller-owned; inputs are read-only; the library never
 *     retains them after a call returns (sessions copy what they need);
 *   - plain `double` / fixed-width integers only, structure-of-arrays;
 *   - return value 0 = success, negative = whole-call failure
 *     (outfit_hip_last_error() has the text). Per-trajectory failures are
 *     DATA (status/cause in outfit_gauss_result_t), never a call failure
 *     (mirrors obs_dataset_api.rs:63-68,159-171);
 *   - blocking entry points are re-entrant and may run concurrently from several host
 *     threads (each call owns its session, streams and scratch); a session is owned by
 *     one host thread at a time.
 */
#ifndef OUTFIT_HIP_H
#define OUTFIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OUTFIT_HIP_ABI_VERSION 7

/* ---- whole-call return codes --------------------------------------------- */
#define OUTFIT_OK                 0
#define OUTFIT_ERR_INVALID_PARAM -1 /* IODParamsBuilder::build rules, mod.rs:544-624 (OutfitError::InvalidIODParameter) */
#define OUTFIT_ERR_INVALID_BATCH -2 /* malformed offsets / null pointers                                          */
#define OUTFIT_ERR_NO_DEVICE     -3 /* HIP runtime reports no usable gfx950 device                                */
#define OUTFIT_ERR_HIP           -4 /* a HIP runtime call failed                                                  */
#define OUTFIT_ERR_EPHEM_RANGE   -5 /* an epoch lies outside the table (reference panics: horizon_data.rs:722-724)*/
#define OUTFIT_ERR_BAD_HANDLE    -6
#define OUTFIT_ERR_NO_MEMORY     -8 /* a host allocation failed inside the library (std::bad_alloc): entry points never
                                       let an exception cross the C ABI (-7: OUTFIT_ERR_EPHEM_FILE, below) */
#define OUTFIT_ERR_INTERNAL      -9 /* v7: any other C++ exception inside the library (a bug, not a resource limit);
                                       outfit_hip_last_error() carries what() */

/* ---- per-trajectory status (outfit_gauss_result_t.status) ------------------
 * mirrors the Err variants `estimate_best_orbit` can return
 * (trajectory.rs:439-451,491-497,531-544; outfit_errors.rs:195-250). */
#define OUTFIT_TRAJ_OK                    0 /* FitOrbitResult::IODGauss((GaussResult, rms))            */
#define OUTFIT_TRAJ_NO_FEASIBLE_TRIPLETS  1 /* OutfitError::NoFeasibleTriplets{span,n_obs,dt_min,dt_max}*/
#define OUTFIT_TRAJ_NO_VIABLE_ORBIT       2 /* OutfitError::NoViableOrbit{cause,attempts}               */
#define OUTFIT_TRAJ_INVALID_CONVERSION    3 /* to_equinoctial()? aborted: parabolic band (cometary_element.rs:265-269) */
#define OUTFIT_TRAJ_INVALID_ORBIT         4 /* to_equinoctial()? aborted: hyperbolic_mean_anomaly e<=1 (cometary_element.rs:225-229) */

/* ---- root cause carried by NoViableOrbit (outfit_gauss_result_t.cause) ----- */
#define OUTFIT_CAUSE_NONE                         0
#define OUTFIT_CAUSE_GAUSS_NO_ROOTS_FOUND         1 /* gauss.rs:1133-1135,1201-1205 */
#define OUTFIT_CAUSE_SINGULAR_DIRECTION_MATRIX    2 /* gauss.rs:544-546             */
#define OUTFIT_CAUSE_POLYNOMIAL_ROOT_FINDING_FAILED 3 /* gauss.rs:984             */
#define OUTFIT_CAUSE_NON_FINITE_SCORE             4 /* trajectory.rs:509-511        */

/* ---- GaussResult variant / OrbitalElements variant ------------------------- */
#define OUTFIT_STAGE_PRELIM    0 /* GaussResult::PrelimOrbit    (gauss_result.rs:99-102) */
#define OUTFIT_STAGE_CORRECTED 1 /* GaussResult::CorrectedOrbit                          */
#define OUTFIT_KIND_KEPLERIAN  0 /* (a, e, i, Omega, omega, M)  keplerian_element.rs:145 */
#define OUTFIT_KIND_COMETARY   1 /* (q, e, i, Omega, omega, nu) cometary_element.rs:180-202 */

/* Mirrors `IODParams` (initial_orbit_determination/mod.rs:225-266), field for
 * field and in declaration order. Defaults: mod.rs:308-344. */
typedef struct outfit_iod_params {
  uint64_t n_noise_realizations;
  double   noise_scale;
  double   extf;
  double   dtmax;
  double   dt_min;
  double   dt_max_triplet;
  double   optimal_interval_time;
  uint64_t max_obs_for_triplets;
  uint32_t max_triplets;
  uint32_t _pad0;
  double   gap_max;
  double   max_ecc;
  double   max_perihelion_au;
  double   min_rho2_au;
  uint32_t aberth_max_iter;
  uint32_t _pad1;
  double   aberth_eps;
  double   kepler_eps;
  uint64_t max_tested_solutions;
  double   r2_min_au;
  double   r2_max_au;
  double   newton_eps;
  uint64_t newton_max_it;
  double   root_imag_eps;
} outfit_iod_params_t;

/* One batch = all trajectories of a prepared dataset (what `process_traj`
 * iterates over, obs_dataset_api.rs:159-171,191-205). Observations of
 * trajectory t are rows traj_offset[t] .. traj_offset[t+1]-1. The path works on each
 * trajectory in epoch order: rows that are not already ascending in mjd_tt are put in that
 * order inside the boundary exactly as fit_single_traj does (stable sort by f64::total_cmp,
 * obs_dataset_api.rs:222-223) — `best_triplet` and the RMS window then refer to the sorted
 * order, as in the reference; per-observation outputs (obs_fit) come back in the caller's
 * row order. Already-sorted input (the usual case) costs one pass over mjd_tt. */
typedef struct outfit_obs_batch {
  uint64_t        n_traj;
  const uint64_t *traj_offset;   /* n_traj + 1 entries                                            */
  const double   *mjd_tt;        /* Observation::mjd_tt()                                          */
  const double   *ra;            /* equ_coord().ra        [rad]                                    */
  const double   *dec;           /* equ_coord().dec       [rad]                                    */
  const double   *sigma_ra;      /* equ_coord().ra_error  [rad] (after error model)                */
  const double   *sigma_dec;     /* equ_coord().dec_error [rad]                                    */
  const double   *obs_helio_equ; /* 3/obs, AU, equatorial J2000 = cache.get_helio_position         */
  const double   *obs_geo_ecl;   /* 3/obs, AU, ecliptic J2000 = get_observer_geocentric_position;
                                    NULL => the RMS stage uses obs_helio_equ directly              */
  const double   *noise_z;       /* NULL or, per trajectory, max_triplets*n_noise*6 standard
                                    normals in (triplet, realisation, ra0 ra1 ra2 dec0 dec1 dec2)
                                    order (gauss.rs:341-386); required when n_noise_realizations>0
                                    and noise_source == OUTFIT_NOISE_HOST                          */
  /* -- device-side draws (statistical parity only, see device/noise.hpp) -- */
  uint32_t        noise_source;  /* OUTFIT_NOISE_HOST (0) or OUTFIT_NOISE_DEVICE                   */
  uint32_t        _pad0;
  uint64_t        noise_seed;    /* the `base_seed` of obs_dataset_api.rs:272 (Philox key)         */
  const uint64_t *traj_key;      /* NULL or one key per trajectory (TrajId::stable_hash(),
                                    obs_dataset_api.rs:285); NULL => traj_key_base + index          */
  uint64_t        traj_key_base;
  /* -- observer positions on the device (needs `ephem`) --
   * When both are given, obs_helio_equ and obs_geo_ecl are ignored (may be NULL) and what
   * OutfitCache::build would have put there (Observer::pvobs + helio_position,
   * observer_extension.rs:180-237) is evaluated on the device. */
  const double   *obs_body_fixed; /* 3/obs, AU: Observer::earth_fixed_position (observer_extension.rs:156-170)   */
  const double   *mjd_ut1;        /* per obs: `epoch.to_ut1(provider).to_mjd_tai_days()` (observer_extension.rs:191-192) */
} outfit_obs_batch_t;
#define OUTFIT_NOISE_HOST   0
#define OUTFIT_NOISE_DEVICE 1

/* Legacy-DE ("horizon") Chebyshev table, restricted to what
 * `HorizonData::ephemeris(Earth, Sun, ..)` reads (horizon_data.rs:810-849):
 * body 0 = Earth-Moon barycentre (IPT row 2), 1 = Moon (row 9), 2 = Sun (row 10).
 * Per block the coefficients are packed as
 *   for body b: for sub-interval s: for axis (x,y,z): ncoef[b] doubles
 * i.e. the order of the DE record itself (horizon_records.rs:132-171),
 * in km (the reference divides by AU afterwards, interpolation_result.rs:82). */
typedef struct outfit_ephem {
  double   jd_start;     /* header.start_period   */
  double   jd_end;       /* header.end_period     */
  double   block_days;   /* header.period_lenght (32 for DE440) */
  double   emrat;        /* header.earth_moon_mass_ratio */
  uint32_t ncoef[3];     /* DE440: 13, 13, 11 (horizon_data.rs:869-885) */
  uint32_t nsub[3];      /* DE440:  2,  8,  2 */
  uint64_t n_blocks;
  const double *coef;    /* n_blocks * sum_b(3*ncoef[b]*nsub[b]) doubles */
} outfit_ephem_t;

/* What the host needs to rebuild
 *   Result<FitOrbitResult::IODGauss((GaussResult, IODRMS)), OutfitError>
 * for one trajectory (constants.rs:134-137,195). 96 bytes.
 *
 * KNOWN DEVIATION — the order among triplets of EQUAL weight when max_triplets > 20. The reference sorts the kept
 * triplets with `sort_unstable_by` (triplet_generation/mod.rs:404): an insertion sort up to 20 elements, which this
 * library restates, and Rust's ipnsort beyond, whose order among equal keys is unspecified and is NOT restated — here
 * equal weights keep the order in which the bounded heap released them (a stable order). Exactly equal weights are
 * common on regular cadences (exposures a fixed interval apart). What that order can reach, and nothing else:
 *   - `best_triplet`: an index of THIS ABI into the sorted list (the reference returns no triplet index);
 *   - which tied triplet receives which noise draws (noise_z is indexed by triplet rank; the reference draws from one
 *     RNG stream in triplet order): a different but equally distributed assignment;
 *   - `cause` of OUTFIT_TRAJ_NO_VIABLE_ORBIT, which is the LAST error in triplet order (trajectory.rs:458-544): when
 *     every triplet fails and the final run of kept triplets is a tie whose members fail differently
 *     (tests/test_oracle_pipeline.py::test_order_of_equal_weights_can_decide_the_cause_of_no_viable_orbit);
 *   - the winner between two candidates whose RMS is the same double (the fold keeps the first strict minimum).
 * Without noise the selected orbit, its RMS, `status`, `attempts` and `n_triplets` do not depend on it
 * (test_order_of_equal_weights_decides_nothing_the_reference_returns). For max_triplets <= 20 there is no deviation. */
typedef struct outfit_gauss_result {
  int32_t  status;           /* OUTFIT_TRAJ_*                                   */
  int32_t  cause;            /* OUTFIT_CAUSE_* (NoViableOrbit only; tie order: see above) */
  int32_t  stage;            /* OUTFIT_STAGE_*                                  */
  int32_t  kind;             /* OUTFIT_KIND_*                                   */
  uint32_t attempts;         /* n_attempts (trajectory.rs:463,478)              */
  uint32_t best_triplet;     /* index into the weight-sorted triplet list (tie order: see above) */
  uint32_t best_realization; /* 0 = noise-free original (gauss.rs:342-346)      */
  uint32_t n_triplets;       /* triplets generated for this trajectory          */
  double   epoch;            /* reference_epoch (MJD TT, light-time corrected)  */
  double   elem[6];          /* see OUTFIT_KIND_*                               */
  double   rms;              /* IODRMS                                          */
} outfit_gauss_result_t;

/* ---- parameters ------------------------------------------------------------ */
/* IODParams::default() (mod.rs:308-344). */
void outfit_hip_iod_params_default(outfit_iod_params_t *p);
/* IODParamsBuilder::build validation (mod.rs:544-624); 0 or OUTFIT_ERR_INVALID_PARAM. */
int  outfit_hip_iod_params_validate(const outfit_iod_params_t *p);

/* ---- math mode ----------------------------------------------------------------
 * Which bits the elementary functions return that feed the correction loop and the returned elements
 * (direction cosines gauss.rs:488-502, cold starts prelim_elliptic.rs / prelim_hyperbolic.rs, the rotations of
 * orb_elem.rs:58-226). The reference calls the platform libm there, so its own results depend on that libm; the
 * loop amplifies a last-place difference to ~1e-10.
 *   OUTFIT_MATH_GLIBC (default): sin, cos (the sincos pair AND the standalone entry points, which differ in glibc),
 *       sinh, cosh, atan2, tan, hypot, acos, log return the bits of ONE libm build — Ubuntu GLIBC 2.35-0ubuntu3.12,
 *       x86-64, the FMA / AVX2 ifunc variants (libm.so.6 sha256 e5141752c850ea45...; restated in
 *       device/glibcmath.hpp, glibcmath2.hpp, glibcmath3.hpp; outfit_amd/libm_pin.py tells whether a host runs that
 *       build) — what the reference computes on the platform its known answers were recorded on. The RMS of every candidate is screened in a cheap arithmetic with an error bound and
 *       the contenders (those that can still be the minimum) are re-evaluated with glibc's functions (kernel K3b).
 *       Measured against the reference's arithmetic on glibc (the CPU restatement with the platform libm): 300 000 of
 *       300 000 (bench workload) and 200 000 of 200 000 (topocentric) records byte-identical, `rms` included
 *       (profiles/round03g_parity_glibc_bench_300000_final.json, round03e_parity_glibc_topo_200000.json).
 *   OUTFIT_MATH_CORRECTLY_ROUNDED: every such function correctly rounded — platform independent (what a reference
 *       built against another libm would be compared with); against a glibc run of the reference ~0.8 % of the
 *       orbits move by more than 1e-12 rad.
 * The mode is a property of a CALL (outfit_hip_call_options_t below) or of a session (fixed at creation); the
 * process-wide value set here is only the default for calls that do not name one. Its initial value comes from
 * the environment variable OUTFIT_MATH_MODE ("glibc" / "cr"), else OUTFIT_MATH_GLIBC.
 * The differential corrector (fit_lsq) follows the mode of its call or session as the IOD does: glibc's bits in
 * OUTFIT_MATH_GLIBC (its LSQ records and fit rows are then byte-identical to the CPU restatement run with that libm),
 * correctly rounded in OUTFIT_MATH_CORRECTLY_ROUNDED. */
#define OUTFIT_MATH_DEFAULT           (-1) /* in call options: the process-wide default at the time of the call */
#define OUTFIT_MATH_CORRECTLY_ROUNDED 0
#define OUTFIT_MATH_GLIBC             1
int outfit_hip_set_math_mode(int mode); /* 0 or OUTFIT_ERR_INVALID_PARAM */
int outfit_hip_get_math_mode(void);

/* Per-call options (ABI v6, extended in v7). Everything a call depends on besides its arguments lives here, so that
 * concurrent callers with different settings do not meet in process-wide state. `struct_size` = sizeof the caller's
 * struct: later versions append fields. NULL options = all defaults.
 *
 * ABI v7 gives two of v6's reserved words a meaning (the struct keeps its 32 bytes; a v6 caller, whose reserved words
 * had to be zero, gets the v6 behaviour):
 *   n_obs_capacity    rows the caller's per-observation columns hold (mjd_tt, ra, dec, sigma_*, mjd_ut1: that many
 *                     doubles; obs_helio_equ, obs_geo_ecl, obs_body_fixed: three times that). 0 = not stated. When
 *                     stated, a batch whose traj_offset[n_traj] exceeds it is OUTFIT_ERR_INVALID_BATCH instead of an
 *                     out-of-bounds read: the columns are indexed by the offsets themselves, and a table can be
 *                     monotone and still run past them. (The reference materialises each trajectory from data it owns,
 *                     obs_dataset_api.rs:210-226, and cannot overrun; across a C ABI only the caller knows the lengths.)
 *   noise_z_capacity  doubles in noise_z; 0 = not stated. When stated with host-drawn noise, fewer than
 *                     n_traj x max_triplets x n_noise_realizations x 6 is OUTFIT_ERR_INVALID_BATCH.
 * Contract since the round-4 build of v6, restated for v7: options with a struct_size that does not cover math_mode,
 * with a non-zero reserved word, or with an unknown math_mode are OUTFIT_ERR_INVALID_PARAM (initialise the struct
 * with outfit_hip_call_options_default). */
typedef struct outfit_hip_call_options {
  uint32_t struct_size;
  int32_t  math_mode;        /* OUTFIT_MATH_DEFAULT, OUTFIT_MATH_CORRECTLY_ROUNDED or OUTFIT_MATH_GLIBC */
  uint64_t n_obs_capacity;   /* v7; 0 = unchecked */
  uint64_t noise_z_capacity; /* v7; 0 = unchecked */
  uint64_t reserved[1];      /* zero */
} outfit_hip_call_options_t;
void outfit_hip_call_options_default(outfit_hip_call_options_t *o);

/* ---- blocking drop-in entry point ------------------------------------------
 * Replaces the body of fit_full_iod / fit_full_iod_parallel after prepare_iod
 * (obs_dataset_api.rs:159-171,191-205): trajectories are sharded over
 * `n_devices` GPUs of this process (0 = all visible), results are written to
 * out[0..n_traj) in trajectory order. */
int outfit_hip_fit_full_iod(const outfit_iod_params_t *params,
                            const outfit_obs_batch_t  *batch,
                            const outfit_ephem_t      *ephem /* nullable */,
                            int n_devices,
                            outfit_gauss_result_t     *out);
/* The same with explicit options (nullable). Safe to call from several host threads at once, with different
 * parameters and modes: every call owns its session, streams and scratch (tests/test_gpu_reentrancy.py; the
 * reference's analogue is parallel == sequential, tests/test_gauss_iod.rs:170-190). */
int outfit_hip_fit_full_iod_ex(const outfit_iod_params_t *params,
                               const outfit_obs_batch_t  *batch,
                               const outfit_ephem_t      *ephem /* nullable */,
                               int n_devices,
                               const outfit_hip_call_options_t *options /* nullable */,
                               outfit_gauss_result_t     *out);

/* ---- resident-session API (one device) --------------------------------------
 * Same computation with inputs kept in HBM between runs; what the blocking
 * call uses internally per device and what bench.py / a long-lived host uses.
 * `stream` is a hipStream_t; NULL = the session's own (non-blocking) stream, NOT the legacy default stream: a
 * caller that orders other work (a collective, a copy) after a run must pass the explicit stream that work is on. */
typedef struct outfit_hip_session outfit_hip_session_t;

int outfit_hip_session_create(const outfit_iod_params_t *params,
                              const outfit_obs_batch_t  *batch,
                              const outfit_ephem_t      *ephem /* nullable */,
                              int device,
                              outfit_hip_session_t     **out_session);
int outfit_hip_session_create_ex(const outfit_iod_params_t *params,
                                 const outfit_obs_batch_t  *batch,
                                 const outfit_ephem_t      *ephem /* nullable */,
                                 int device,
                                 const outfit_hip_call_options_t *options /* nullable */,
                                 outfit_hip_session_t     **out_session);
/* Enqueue every kernel of the path; d_out (device memory, n_traj records) may be
 * NULL to use the session's own result buffer. Asynchronous w.r.t. the host. */
int outfit_hip_session_run(outfit_hip_session_t *s, void *stream, void *d_out);
/* Wait for the last run and copy the session's result buffer to host memory. */
int outfit_hip_session_results(outfit_hip_session_t *s, outfit_gauss_result_t *out);
/* Wait for the last run and report what the result calls would: OUTFIT_ERR_EPHEM_RANGE when an observation epoch lies
 * outside the ephemeris table (the reference panics, horizon_data.rs:722-724), else 0. For callers that consume
 * `d_out` on the device (a gather) and never fetch results through the session. */
int outfit_hip_session_check(outfit_hip_session_t *s);
/* Device pointer of the session's own result buffer (n_traj records). */
void *outfit_hip_session_device_results(outfit_hip_session_t *s);
/* Diagnostic: the per-candidate scores K2c left for the LAST chunk of the last run (candidate c of the chunk's
 * trajectory t at t * cand_per_traj + c): the sum of squared normalised residuals over the RMS window, the window's
 * observation count and the candidate's flag word (0x10 = a preliminary orbit exists, 0x80 = a residual
 * evaluation failed; see kernels.h). In
 * OUTFIT_MATH_GLIBC these are the SCREENING values (device/screen.hpp) that k_refine_best starts from; with
 * OUTFIT_RMS_EXACT_ALL=1 the exact ones — tests hold one against the other. Any of sum / n_win / flags may be NULL;
 * *n_out = number of candidates of that chunk (also when capacity is too small: OUTFIT_ERR_INVALID_BATCH). */
int outfit_hip_session_candidate_scores(outfit_hip_session_t *s, uint64_t capacity, double *sum, uint32_t *n_win,
                                        uint32_t *flags, uint64_t *n_out);
/* Per-kernel totals since profiling was switched on: summed duration (ms, HIP events
 * recorded on the stream each run was enqueued on) and number of launches. names[i]
 * are static strings; returns the number of kernels reported. Synchronises. */
int outfit_hip_session_kernel_times(outfit_hip_session_t *s, int max_n,
                                    const char **names, float *ms, int *launches);
/* Enable/disable the per-kernel HIP-event timing above (default off); enabling resets the totals. */
int outfit_hip_session_set_profiling(outfit_hip_session_t *s, int enabled);
void outfit_hip_session_destroy(outfit_hip_session_t *s);

/* ---- kepler::propagate_universal (kepler/propagation.rs:114-174) -------------
 * Batched over n states; solver kind NewtonRaphson (kepler/params.rs:66-73)
 * with `convergency`; out is 11 doubles per state:
 *   r1[3], v1[3], f, g, f_dot, g_dot, psi.
 * status[i]: 0 ok, 1 DegenerateState (initial radius), 2 solver did not
 * converge (NewtonRaphsonKeplerConvergence), 3 DegenerateState (propagated). */
int outfit_hip_propagate_universal(uint64_t n, const double *r0v0 /* 6n */,
                                   const double *t0, const double *t1,
                                   double convergency,
                                   double *out /* 11n */, int32_t *status /* n */);

/* The full `SolverType { kind, params: SolverParams }` of the reference (kepler/params.rs:24-73). */
#define OUTFIT_SOLVER_NEWTON_RAPHSON 0 /* SolverKind::NewtonRaphson (default)                                   */
#define OUTFIT_SOLVER_BRENT_DECKER   1 /* SolverKind::BrentDecker   (kepler/brent_dekker_solver.rs)               */
#define OUTFIT_SOLVER_AUTO           2 /* SolverKind::Auto: Newton, then Brent-Dekker on failure (params.rs:138-140) */
#define OUTFIT_PARABOLIC_CARDANO     0 /* ParabolicPrelimMethod::Cardano (default; prelim_parabolic.rs:77-93)     */
#define OUTFIT_PARABOLIC_NEWTON      1 /* ParabolicPrelimMethod::Newton                                           */
typedef struct outfit_solver_type {
  int32_t  kind;                     /* OUTFIT_SOLVER_*                                   */
  int32_t  parabolic_solving_method; /* OUTFIT_PARABOLIC_*                                */
  double   convergency;              /* SolverParams.convergency, default 100 eps         */
  uint64_t max_iter_prelim_kepuni;   /* default 20                                        */
  int32_t  has_psi_guess;            /* SolverParams.psi_guess: Option<f64>               */
  int32_t  _pad;
  double   psi_guess;                /* applied to every state of the batch when present  */
} outfit_solver_type_t;
void outfit_hip_solver_type_default(outfit_solver_type_t *s); /* SolverType::default() (params.rs:35-44,66-73) */
/* status[i] as above, plus 4 = BrentDekkerKeplerConvergence (kinds BrentDecker and Auto, params.rs:134-141). */
int outfit_hip_propagate_universal_ex(uint64_t n, const double *r0v0 /* 6n */, const double *t0, const double *t1,
                                      const outfit_solver_type_t *solver,
                                      double *out /* 11n */, int32_t *status /* n */);

/* ==== next row: FitLSQ::fit_lsq — differential orbit correction =========
 * Replaces the per-trajectory loop of `FitLSQ::fit_lsq`
 * (differential_orbit_correction/obs_dataset_api.rs:57-130) after the dataset
 * preparation (:68-74, stays with the caller as for fit_full_iod): for every
 * trajectory `differential_correction` (mod.rs:66-112) = IOD seed ->
 * to_equinoctial -> `run_differential_correction` (diff_cor.rs:282-436: Newton
 * weighted least squares with two-body partials, chi-square outlier rejection,
 * covariance rescaling), falling back to the IOD orbit when the corrector errs. */

#define OUTFIT_PROPAGATOR_TWOBODY 0 /* PropagatorKind::TwoBody (propagator/mod.rs). NBody is not
                                       implemented here: any other value is OUTFIT_ERR_INVALID_PARAM */

/* Mirrors `DifferentialCorrectionConfig` (diff_cor.rs:39-159; defaults :161-178) with
 * `OutlierRejectionConfig` (outlier_rejection.rs:24-55) and `EquinoctialLimits`
 * (equinoctial_element.rs:161-179) flattened in declaration order. */
typedef struct outfit_diffcor_config {
  uint64_t max_newton_iterations;                  /* 30   */
  uint64_t max_outlier_rejection_passes;           /* 10   */
  double   convergence_threshold;                  /* 1e-4 */
  double   convergence_before_rejection_threshold; /* 2.0  */
  double   rms_stagnation_ratio;                   /* 0.98 */
  double   rms_divergence_ratio;                   /* 1.5  */
  uint64_t max_stagnation_iterations;              /* 3    */
  uint32_t enable_outlier_rejection;               /* 1    */
  uint32_t propagator;                             /* OUTFIT_PROPAGATOR_TWOBODY */
  double   chi_squared_rejection_threshold;        /* 25   */
  double   chi_squared_recovery_threshold;         /* 9    */
  double   eccentricity_limit;                     /* 1.2  */
  double   min_semi_major_axis;                    /* 1e-6 */
  double   max_semi_major_axis;                    /* 1e4  */
  double   min_periapsis_distance;                 /* 1e-6 */
  double   max_apoapsis_distance;                  /* 1e4  */
  uint8_t  free_elements[6];                       /* all 1: (a, h, k, p, q, lambda) */
  uint8_t  _pad[2];
} outfit_diffcor_config_t;

/* outfit_lsq_result_t.status */
#define OUTFIT_LSQ_DIFFCOR          0 /* Ok(FitOrbitResult::DifferentialCorrection((Equinoctial{..}, normalised_rms))) */
#define OUTFIT_LSQ_IOD_KEPT         1 /* corrector returned Err(_) => Ok(initial_orbit) (mod.rs:109): elements are in `iod` */
#define OUTFIT_LSQ_IOD_ERROR        2 /* no usable seed: run_iod_on_observations(..)? failed (mod.rs:86); see iod.status/cause */
#define OUTFIT_LSQ_CONVERSION_ERROR 3 /* seed.to_equinoctial()? failed (mod.rs:89-91); conv_error = OUTFIT_TRAJ_INVALID_* */
/* outfit_lsq_result_t.dc_error: why the corrector gave up (status IOD_KEPT) */
#define OUTFIT_DC_NONE      0
#define OUTFIT_DC_FAILED    1 /* OutfitError::DifferentialCorrectionFailed: normal-equation inversion failed (diff_cor.rs:335-341) */
#define OUTFIT_DC_BIZARRE   2 /* OutfitError::BizarreOrbit (diff_cor.rs:344-349)                                                  */
#define OUTFIT_DC_DIVERGED  3 /* OutfitError::DifferentialCorrectionDiverged (diff_cor.rs:354-356)                                */
/* ObsSelection (obs_fit_data.rs:22-45) */
#define OUTFIT_OBS_ACTIVE     0
#define OUTFIT_OBS_REJECTED   1
#define OUTFIT_OBS_FORCED_OUT 2

/* `DifferentialCorrectionOutput` (diff_cor.rs:185-209) as converted by
 * `From<DifferentialCorrectionOutput> for OrbitalElements` (:211-224), plus the seed. */
typedef struct outfit_lsq_result {
  int32_t  status;                  /* OUTFIT_LSQ_*                                              */
  int32_t  dc_error;                /* OUTFIT_DC_*                                               */
  int32_t  conv_error;              /* OUTFIT_TRAJ_INVALID_* when status == CONVERSION_ERROR     */
  uint32_t total_newton_iterations; /* over all outer passes                                      */
  uint32_t num_measurements;        /* 2 x active observations at the last accepted iteration     */
  uint32_t n_rejected;              /* observations whose final selection is not Active           */
  uint32_t outer_passes;            /* outlier-rejection passes started                           */
  uint32_t _pad;
  double   epoch;                   /* reference_epoch of the seed (unchanged by the corrector)   */
  double   elem[6];                 /* equinoctial (a, h, k, p, q, lambda)                        */
  double   normalised_rms;          /* the `Chi2` slot of FitOrbitResult::DifferentialCorrection  */
  double   covariance[36];          /* rescaled; covariance[6*i+j] = matrix[(i,j)]                */
  outfit_gauss_result_t iod;        /* the seed (given or computed here)                          */
} outfit_lsq_result_t;

/* `ObsFitData` (obs_fit_data.rs:63-100) after the last iteration, one per observation. */
typedef struct outfit_obs_fit {
  double  residual_ra;  /* O-C, wrapped to [-pi, pi] (least_square.rs:188-198) */
  double  residual_dec;
  double  chi;
  int32_t selection;    /* OUTFIT_OBS_* */
  int32_t _pad;
} outfit_obs_fit_t;

/* DifferentialCorrectionConfig::default() (diff_cor.rs:161-178). */
void outfit_hip_diffcor_config_default(outfit_diffcor_config_t *c);
/* 0, or OUTFIT_ERR_INVALID_PARAM for an unsupported propagator. */
int  outfit_hip_diffcor_config_validate(const outfit_diffcor_config_t *c);

/* Blocking drop-in for the trajectory loop of fit_lsq. `initial_orbits` (nullable) holds one
 * record per trajectory, the `initial_orbits: Option<&FullOrbitResult>` map flattened:
 * status OUTFIT_TRAJ_OK = use as the seed; any other status = absent/Err entry, for which the
 * reference runs the IOD (obs_dataset_api.rs:118-120, mod.rs:86). NULL = run the IOD for all.
 * `obs_fit` (nullable) receives one record per observation, indexed like the observation arrays
 * (row traj_offset[t] + i). (The session form rebases: its obs_fit row r is observation row traj_offset[0] + r
 * of the batch the session was created from.) */
int outfit_hip_fit_lsq(const outfit_iod_params_t     *iod_params,
                       const outfit_diffcor_config_t *config,
                       const outfit_obs_batch_t      *batch,
                       const outfit_ephem_t          *ephem /* nullable */,
                       const outfit_gauss_result_t   *initial_orbits /* nullable, n_traj */,
                       int n_devices,
                       outfit_lsq_result_t           *out /* n_traj */,
                       outfit_obs_fit_t              *obs_fit /* nullable, n_obs */);
int outfit_hip_fit_lsq_ex(const outfit_iod_params_t     *iod_params,
                          const outfit_diffcor_config_t *config,
                          const outfit_obs_batch_t      *batch,
                          const outfit_ephem_t          *ephem /* nullable */,
                          const outfit_gauss_result_t   *initial_orbits /* nullable, n_traj */,
                          int n_devices,
                          const outfit_hip_call_options_t *options /* nullable */,
                          outfit_lsq_result_t           *out /* n_traj */,
                          outfit_obs_fit_t              *obs_fit /* nullable, n_obs */);

/* Session form: the corrector runs on the session's resident observations. With
 * `initial_orbits` NULL the seeds are the records the last outfit_hip_session_run wrote (its
 * `d_out`, which must still be alive, or the session's own buffer; enqueued before on the same
 * stream); otherwise entries with status OUTFIT_TRAJ_OK replace them. Asynchronous unless seeds
 * are given; fetch with outfit_hip_session_lsq_results. */
int outfit_hip_session_run_lsq(outfit_hip_session_t *s, const outfit_diffcor_config_t *config,
                               const outfit_gauss_result_t *initial_orbits /* nullable, host */,
                               void *stream);
int outfit_hip_session_lsq_results(outfit_hip_session_t *s, outfit_lsq_result_t *out,
                                   outfit_obs_fit_t *obs_fit /* nullable */);

/* ==== next row: legacy JPL DE binary reader -> outfit_ephem_t =================
 * Host-side only (no GPU needed). Replaces HorizonData::read_horizon_file
 * (jpl_ephem/horizon/horizon_data.rs:598-684) for what the path consumes: the header and the
 * Chebyshev segments of EMB, Moon and Sun (IPT rows 2, 9, 10). Little-endian files
 * (the "Linux" distribution the reference downloads, horizon_version.rs). */
typedef struct outfit_de_header {       /* HorizonHeader (horizon_data.rs:130-160) + what sizes the file */
  double   start_period;                /* SS[0], JD */
  double   end_period;                  /* SS[1]     */
  double   period_length;               /* SS[2] (`period_lenght`), days per record */
  double   earth_moon_mass_ratio;       /* EMRAT     */
  double   au_km;                       /* AU constant of the header (the reference skips it) */
  uint32_t ipt[15][3];                  /* [offset (1-based), n_coeffs, n_subintervals]; rows 13/14 = 0 when absent */
  uint32_t numde;                       /* 440 -> jpl_version "DE440" */
  int32_t  ncon;
  uint64_t recsize;                     /* bytes per record (8144 for DE440) */
  uint64_t n_blocks;                    /* data records in the file */
} outfit_de_header_t;
typedef struct outfit_ephem_file outfit_ephem_file_t;
#define OUTFIT_ERR_EPHEM_FILE -7        /* cannot open / not a DE file / truncated (the reference panics) */

int  outfit_hip_ephem_open(const char *path, outfit_ephem_file_t **out);
int  outfit_hip_ephem_header(const outfit_ephem_file_t *e, outfit_de_header_t *header);
/* `view->coef` points into the handle and stays valid until outfit_hip_ephem_close. */
int  outfit_hip_ephem_view(const outfit_ephem_file_t *e, outfit_ephem_t *view);
/* `extract_body_records` (horizon_data.rs:468-491) for ANY IPT row (0..14) and one data block, read on demand: the
 * block's n_sub HorizonRecords (each x[n_coeffs], y[n_coeffs], z[n_coeffs]; 2 / 1 axes for rows 11 / 14) into `out`,
 * its [start_jd, end_jd] into jd_range (nullable). Returns the number of doubles written (> 0) or a negative code.
 * Not needed by the path (which consumes rows 2, 9, 10 only): it lets the reference's record known answers for
 * other bodies (horizon_data.rs:898-1003) be checked through this reader. */
int  outfit_hip_ephem_body_record(const outfit_ephem_file_t *e, uint32_t ipt_row, uint64_t block,
                                  double *jd_range /* [2], nullable */, double *out, uint64_t capacity);
void outfit_hip_ephem_close(outfit_ephem_file_t *e);

/* Observer::pvobs + helio_position for n observations: geocentric position in the ecliptic
 * J2000 frame and heliocentric position in the equatorial J2000 frame, AU, 3 doubles per observation each —
 * the two vectors OutfitCache::build stores (cache/observer_centric_cache.rs:80-110). */
int outfit_hip_observer_positions(const outfit_ephem_t *ephem, uint64_t n, const double *mjd_tt,
                                  const double *mjd_ut1, const double *obs_body_fixed /* 3n */,
                                  double *geo_ecl /* 3n */, double *helio_equ /* 3n */);

/* The standard normals the device generator yields for OUTFIT_NOISE_DEVICE, in the layout of `noise_z`
 * (n_traj * max_triplets * n_noise * 6 doubles): what a host needs to replay a device-noise run through
 * the host-normal mode (or through a CPU implementation). traj_key may be NULL (traj_key_base + index). */
int outfit_hip_device_normals(uint64_t noise_seed, const uint64_t *traj_key, uint64_t traj_key_base,
                              uint64_t n_traj, uint32_t max_triplets, uint64_t n_noise, double *out);

/* ---- diagnostics ------------------------------------------------------------ */
/* Device self-test: the kernels replace some fp64 divisions by cheaper sequences that
 * must give the SAME bits as IEEE division (outfit_amd/csrc/device/fastdiv.hpp); this
 * compares them with `/` on n_pairs pseudo-random operand pairs and returns the number
 * of differing results (must be 0). No counterpart in the reference. */
int outfit_hip_selftest_division(uint64_t n_pairs, uint64_t seed, uint64_t *mismatches);
/* Host self-test of the rule "no C++ exception crosses this ABI": raises std::bad_alloc (kind 0), std::runtime_error (1)
 * or a non-standard exception (2) inside an entry point; must return OUTFIT_ERR_NO_MEMORY with outfit_hip_last_error()
 * set. Needs no device. */
int outfit_hip_selftest_abi_guard(int kind);
/* Device self-test of the Aberth sweep that shares one refined reciprocal between the two components of
 * every complex division (outfit_amd/csrc/device/gauss.hpp) against the plain divisions, on n_polys random
 * sparse octics: counts[0] = polynomials whose return code or iterates differ in any bit (must be 0),
 * counts[1] = polynomials on which the range guard sent the solver back to the plain divisions. */
int outfit_hip_selftest_aberth(uint64_t n_polys, uint64_t seed, uint64_t *counts /* [2] */);
/* Device self-test of the correctly-rounded sin/cos/acos (outfit_amd/csrc/device/crmath.hpp): the fast
 * path (table + rounding test) against the full double-double evaluation on n_args pseudo-random
 * arguments. counts[0] = sin/cos arguments with differing bits, counts[1] = acos arguments with differing
 * bits (both must be 0), counts[2] = sin/cos arguments that needed the full evaluation, counts[3] = sinh/cosh/log
 * (the hyperbolic cold start, prelim_hyperbolic.rs:45-141) and atan2 (orb_elem.rs:58-226) arguments with differing
 * bits (must be 0). */
int outfit_hip_selftest_crmath(uint64_t n_args, uint64_t seed, uint64_t *counts /* [4] */);
/* The device's correctly-rounded functions on caller-given arguments (host arrays of n doubles), for checks
 * against high-precision known answers: fn & 15 = 0 sin, 1 cos, 2 acos, 3 sinh, 4 cosh, 5 log (y > 1), 6 atan2
 * (x then holds n pairs y, x), 7 / 8 = sin / cos as the residual loop of the RMS stage evaluates them (~1 ulp, not
 * correctly rounded); 9 / 10 / 11 / 12 = sin / cos / sinh / cosh of OUTFIT_MATH_GLIBC (device/glibcmath.hpp);
 * fn & 16 selects the full double-double evaluation instead of the table fast path + rounding test. */
int outfit_hip_crmath_eval(int fn, uint64_t n, const double *x, double *out);
/* The device's glibc-faithful functions of round 3 on caller-given arguments (host arrays of n doubles): fn 0 / 1 = sin /
 * cos as glibc's STANDALONE entry points return them (the FMA variants; they differ from sincos's bits for 0.07 % of
 * arguments — device/glibcmath.hpp), 2 = tan, 3 = atan2(x[i], y[i]), 4 = hypot(x[i], y[i]) (device/glibcmath2.hpp),
 * 5 = acos, 6 = log (device/glibcmath3.hpp). */
int outfit_hip_glibc_eval(int fn, uint64_t n, const double *x, const double *y /* fn 3, 4 */, double *out);
/* Profiling aid: streams n_doubles doubles in and out (8 bytes per lane, coalesced) `repeats` times. Run under
 * rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE it calibrates those counters for the 8-byte access width of this
 * library's kernels (tools/profile_gpu.sh); the true traffic is n_doubles * 8 bytes each way per repeat. */
int outfit_hip_calibration_stream(uint64_t n_doubles, int repeats);
/* `prelim_kepuni` (kepler/params.rs:185-191), the cold-start value of the universal Kepler solve, for n explicit
 * problems of 6 doubles (dt, r0, sig0, alpha, e0, convergency; mu = GAUSS_GRAV^2), in the current math mode —
 * the reference's libm-dependent KAT (kepler/params.rs:689-725) is checked through this. */
int outfit_hip_prelim_kepuni(uint64_t n, const double *problems /* 6n */, double *out /* n */);
/* Device memory freed by sessions and blocking calls is kept per device for the next call of a similar shape
 * (hipMalloc/hipFree of the candidate scratch cost ~10 ms per 100 k-object call); this returns it to the driver.
 * The environment variable OUTFIT_NO_CACHE=1 disables the cache; OUTFIT_CACHE_LIMIT_MB bounds what one device may hold
 * (default 16384 = 16 GiB; blocks beyond it are freed as before). Parked blocks are invisible to other allocators of the
 * process (torch, RCCL): call this before handing the device to them. OUTFIT_CACHE_POISON=1 (debugging) fills every
 * block handed out with 0xFF bytes. */
int outfit_hip_trim_cache(void);
const char *outfit_hip_last_error(void); /* thread-local, never NULL */
int         outfit_hip_abi_version(void);
int         outfit_hip_device_count(void);

/* ---- environment switches (none is needed in production; all are read by the host layer only) ------------------
 * Behaviour:    OUTFIT_MATH_MODE (glibc / cr: initial process default), OUTFIT_CACHE_LIMIT_MB, OUTFIT_NO_CACHE,
 *               OUTFIT_IOD_CHUNK (trajectories per chunk), OUTFIT_SHARD_DEVICES (logical shards, e.g. "0,0"),
 *               OUTFIT_TRACE (phase trace of the blocking call on stderr).
 * Forms of a kernel that must give the same bytes (testing / measurement aids; tests hold them against each other):
 *               OUTFIT_K0_WAVE_MIN (reduced epochs from which a trajectory takes the wave form of K0; 0 never, 3 all),
 *               OUTFIT_PAIR_MODE (0 / 1: throughput / latency forms of K2a, K2b, K3b whatever the batch size),
 *               OUTFIT_REFINE_SCAN_ALL (K3b sweeps every candidate instead of reading K3's contender list),
 *               OUTFIT_RMS_EXACT_ALL, OUTFIT_RMS_NO_SCREEN (K2c exact for every candidate), OUTFIT_RMS_WAVE,
 *               OUTFIT_K0_SLICES, OUTFIT_GENERAL_SERIAL, OUTFIT_NO_STREAMED_UPLOAD, OUTFIT_DIFFCOR_GROUP,
 *               OUTFIT_K3_THREADS (64 .. 1024 threads staging a tile of K3), OUTFIT_REFINE_GROUP (4 / 16 lanes per
 *               exact evaluation in K3b).
 * Fault injection for tests: OUTFIT_CACHE_POISON (NaN-fill device blocks), OUTFIT_TEST_EXACT_RMS_FAILS,
 *               OUTFIT_TEST_GUARD_REDO (m: every (candidate + iteration) divisible by m is reported as a guarded-division
 *               hit in the fast correction kernel and replayed by the general one). */

#ifdef __cplusplus
}
#endif
#endif /* OUTFIT_HIP_H */
