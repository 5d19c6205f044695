// kernels.h — argument blocks and launchers shared by kernels.hip and the C-ABI
// host layer (api.hip). Everything here is plain data passed by value to a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "device/ephem.hpp"
#include "device/gauss.hpp"
#include "device/triplets.hpp"
#include "device/diffcor.hpp"
#include "device/noise.hpp"
#include "device/screen.hpp"

namespace outfit {

constexpr unsigned kRootSlots = 8;       // a degree-8 polynomial has at most 8 roots
#ifndef OUTFIT_GENERAL_BLOCK
#define OUTFIT_GENERAL_BLOCK 256
#endif
constexpr unsigned kCorrectBlock = OUTFIT_GENERAL_BLOCK;  // lanes per workgroup of the correction kernel
constexpr unsigned kCorrectLdsDoubles = 34;  // per lane: S, S^-1, observer matrix (27) + prelim state (7)
constexpr unsigned kFastBlock = 64;          // lanes per workgroup of the fast correction kernel
#ifndef OUTFIT_FAST_WAVES
#define OUTFIT_FAST_WAVES 3
#endif
constexpr unsigned kFastWavesPerSimd = OUTFIT_FAST_WAVES;  // residency the fast kernel is compiled and launched for

// per-candidate flag word (phase A writes it, phase B/C update it, K3 replays it)
constexpr unsigned kFlagEmpty = 0x80000000u;  // slot beyond n_triplets (never read by K3)
constexpr unsigned kFlagOk = 0x10u;           // prelim_orbit returned Ok; low nibble = cause otherwise
constexpr unsigned kFlagCorrected = 0x20u;    // GaussResult::CorrectedOrbit
constexpr unsigned kFlagCometary = 0x40u;     // OrbitalElements::Cometary
constexpr unsigned kFlagRmsErr = 0x80u;       // an ephemeris_error returned Err
constexpr unsigned kFlagConvShift = 8;        // 3 bits: to_equinoctial error (3 / 4)
constexpr unsigned kFlagRootsShift = 12;      // 4 bits: number of filtered roots (phase A -> B)
constexpr unsigned kFlagFirstShift = 16;      // 4 bits: slot of the first accepted root (K2a' -> K2b)
constexpr unsigned kFlagSlow = 0x100000u;     // irregular candidate: handled by the general kernel
constexpr unsigned kFlagFirstNone = 0x200000u;  // the fast kernel ran the first accepted root's loop to its None:
                                                // the general kernel pushes that PrelimOrbit and goes on with the next root

// bits 23-27: K2c screen, how far its RMS may be from the exact one (device/screen.hpp, kFlagLevelShift)

// same 96-byte layout as outfit_gauss_result_t (include/outfit_hip.h)
struct GaussResultRec {
  int status, cause, stage, kind;
  unsigned attempts, best_triplet, best_realization, n_triplets;
  double epoch, elem[6], rms;
};
static_assert(sizeof(GaussResultRec) == 96, "result record must match outfit_gauss_result_t");

struct TripletArgs {
  unsigned long long n_traj, traj0;     // trajectories in this chunk, first trajectory
  const unsigned long long* traj_offset;
  const double* mjd;
  TripletParams P;
  double* heap_w;                       // [k_cap][heap_stride]
  unsigned long long* heap_id;
  unsigned long long heap_stride;
  unsigned* n_triplets;                 // [n_traj] (chunk-local)
  unsigned long long* triplets;         // [n_traj][k_cap] packed (i,j,k)
  // Two forms of the same selection (device/triplets.hpp): one LANE per trajectory, and one WAVE per trajectory for
  // those whose reduced length is >= wave_min_obs (0 = the wave form is not used in this session). The lane kernels
  // leave such trajectories alone; k_triplet_select_wave runs over `wave_list` (absolute trajectory ids, ascending).
  unsigned wave_min_obs;
  const unsigned* wave_list;
  unsigned n_wave_list;
};

// the two halves of K1 (they run on different streams: the direction cosines ahead of K2a, the observer
// positions and cos(dec) beside K2a, first needed by K2c)
constexpr unsigned kPrepassUnit = 1u, kPrepassObserver = 2u;

struct PrepassArgs {
  unsigned long long n_obs;
  unsigned parts;  // kPrepassUnit | kPrepassObserver
  const double* mjd;
  const double* ra;
  const double* dec;
  const double* geo_ecl;
  double* obs_pos;
  double* cos_dec;
  double* unit;    // [3][n_obs] direction cosines of the (noise-free) observations
  // the screening loop's per-observation record (kObsRecDoubles doubles per observation, see RmsArgs::obs_rec): written
  // with the observer half; `helio` = the observer positions when there is no table (the residuals use obs_helio_equ)
  double* obs_rec;
  const double* sra;
  const double* sdec;
  const double* helio;
  bool has_table;  // false: only cos(dec) is produced, the residuals use obs_helio_equ
  int math_mode;   // kMathCr / kMathGlibc for the direction cosines
  EphemTable T;
  int* error_flag;
};

struct GeometryArgs {
  unsigned long long n_obs;
  const double *mjd_tt, *mjd_ut1, *body_fixed;  // body_fixed: 3 per observation, AU, Earth-fixed
  EphemTable T;
  double *geo_ecl, *helio_equ;                  // 3 per observation each
  int* error_flag;
};

// Inputs every candidate-level kernel needs to find its triplet
struct CandidateIndex {
  unsigned long long n_cand, traj0;
  unsigned cand_per_traj, n_real, k_cap;
  const unsigned long long* traj_offset;
  const unsigned* n_triplets;
  const unsigned long long* triplets;
  const double *mjd, *ra, *dec, *sra, *sdec, *helio, *noise_z;
  double noise_scale;
  const double* unit;                 // [3][n_obs_total], written by K1
  unsigned long long n_obs_total;
  // device-side noise (noise.hpp): used instead of noise_z when noise_device is set
  bool noise_device;
  unsigned long long noise_seed, traj_key_base;
  const unsigned long long* traj_key;  // nullable
};

// candidate-level scratch, structure-of-arrays over the candidates of one chunk
struct CandidateScratch {
  unsigned* flags;   // [n_cand]
  double* S;         // [9][n_cand]   direction matrix, m[r][c] at (3r+c)
  double* roots;     // [kRootSlots][n_cand] filtered roots in solver order
  double* psi0;      // [2][n_cand] cold-start values of the two universal-Kepler solves
  unsigned long long* slow_list;   // [n_cand] candidates for the general kernel from the front; from the back, the
                                   // candidates the K2c screen hands to the exact evaluation (redo list)
  unsigned long long* slow_count;  // [2] lengths of the two
  double* state;     // [7][n_cand]   r2[3], v2[3], epoch (equatorial J2000)
  double* sum;       // [n_cand]
  unsigned* n_win;   // [n_cand]
};

struct RootsArgs {
  CandidateIndex I;
  CandidateScratch W;
  GaussParams P;
  unsigned long long cand_begin, cand_count;  // slice of the chunk's candidates handled by this launch
};

struct CorrectArgs {
  CandidateIndex I;
  CandidateScratch W;
  GaussParams P;
  unsigned long long* queue;  // work-queue head (zeroed before every launch)
  unsigned test_redo_mod;     // test aid (OUTFIT_TEST_GUARD_REDO=m): the fast kernel reports kCorrRedo — "a guarded division
                              // saw an operand outside its window" — for every (candidate + iteration) divisible by m
};

// What the screening loop of K2c reads per (candidate, observation), packed by K1 into ONE 96-byte record per
// observation (device/screen.hpp ScreenObs): mjd_tt, ra, cos ra, sin ra, cos dec, sin dec, cos(dec) / sigma_ra,
// 1 / sigma_dec, observer x, y, z, dec. Nine separate arrays cost nine 64-bit address computations and nine loads per
// pass and two reciprocals of values that depend on the observation only; the record is one address, six 16-byte loads,
// no reciprocal — and with the sines and cosines of the observed angles the residual needs no full atan2 (screen_term).
// The exact loops keep the reference's operands.
constexpr unsigned kObsRecDoubles = 12;

struct RmsArgs {
  CandidateIndex I;
  CandidateScratch W;
  const double* obs_pos;
  const double* cos_dec;
  const double* obs_rec;
  double extf, dtmax;
  SinCosConsts sc;  // polynomial constants as kernel arguments = scalar registers (device/orbit.hpp)
  int screen_direct;  // rms_mode 2 only: 1 = the screen takes the elements straight from the state (device/screen.hpp
                      // screen_prologue; short trajectories — the host decides, kernels.hip rms_candidate says why)
  int rms_mode;     // 0 exact with the platform functions (correctly-rounded mode) | 1 every candidate with the
                    // glibc-faithful functions (OUTFIT_RMS_EXACT_ALL=1) | 2 screening arithmetic, after which
                    // k_refine_best re-evaluates the contenders exactly (the glibc-faithful mode's default)
};

struct SelectArgs {
  unsigned long long n_traj, traj0, n_cand;
  unsigned cand_per_traj, n_real;
  const unsigned* n_triplets;
  CandidateScratch W;
  GaussResultRec* out;  // indexed by absolute trajectory
  int math_mode;        // kMathCr / kMathGlibc for the returned elements
  // set when k_refine_best follows (glibc-faithful mode): K3 leaves the winner's elements to it and stores, per
  // trajectory of the chunk, the two numbers of its contender rule: bound[2 lt] = min_c (screened_c + slack_c),
  // bound[2 lt + 1] = 1 / sigma_min (kernels.hip, K3b)
  double* bound;
  // ... and the trajectory's candidates that passed the contender test against the RUNNING bound of the fold, a
  // superset of those that pass it against the final one: list[lt][0] = count (kContenderOverflow: more than fit, K3b
  // scans the trajectory's candidates itself), list[lt][1..] = candidate indices within the trajectory
  unsigned* list;
  const unsigned long long* traj_offset;
  const double *sra, *sdec;
};
constexpr unsigned kContenderListCap = 7;            // entries per trajectory (+ the count: 32 B per trajectory)
constexpr unsigned kContenderOverflow = 0xFFFFFFFFu;

// same layouts as outfit_lsq_result_t (480 B) / outfit_obs_fit_t (32 B)
struct LsqResultRec {
  int status, dc_error, conv_error;
  unsigned total_newton_iterations, num_measurements, n_rejected, outer_passes, pad;
  double epoch, elem[6], normalised_rms, covariance[36];
  GaussResultRec iod;
};
static_assert(sizeof(LsqResultRec) == 480, "record must match outfit_lsq_result_t");
struct ObsFitRec {
  double residual_ra, residual_dec, chi;
  int selection, pad;
};
static_assert(sizeof(ObsFitRec) == 32, "record must match outfit_obs_fit_t");

struct DiffCorArgs {
  unsigned long long n_traj, n_obs;
  const unsigned long long* traj_offset;
  const double *mjd, *ra, *dec, *sra, *sdec, *obs_pos;
  const GaussResultRec* iod;       // the session's own IOD results
  const GaussResultRec* initial;   // nullable: caller-provided seeds (status 0 = use)
  DiffCorConfig cfg;
  double* res;                     // [2][3][n_obs]
  unsigned char* sel;              // [n_obs]
  double* cov;                     // [2][36][n_traj]
  double* qr_work;                 // [42][n_traj * group]
  LsqResultRec* out;               // [n_traj]
  ObsFitRec* obs_fit;              // nullable [n_obs]
  unsigned long long* queue;       // work-queue head (zeroed before every launch)
  int math_mode;                   // kMathCr: correctly-rounded functions throughout | kMathGlibc: glibc's bits throughout
};

hipError_t launch_triplet_select(const TripletArgs& a, bool lds_heap, hipStream_t s);
hipError_t launch_triplet_select_wave(const TripletArgs& a, size_t lds_bytes, hipStream_t s);
hipError_t launch_observer_prepass(const PrepassArgs& a, hipStream_t s);
hipError_t launch_gauss_roots(const RootsArgs& a, hipStream_t s);
hipError_t reset_slow_list(const CandidateScratch& w, hipStream_t s);
hipError_t launch_gauss_correct_fast(const CorrectArgs& a, unsigned n_cu, hipStream_t s);
hipError_t launch_gauss_correct(const CorrectArgs& a, unsigned n_cu, hipStream_t s);
hipError_t launch_orbit_rms(const RmsArgs& a, bool slow_list, hipStream_t s);
hipError_t launch_select_best(const SelectArgs& a, hipStream_t s);
hipError_t launch_refine_best(const SelectArgs& a, const RmsArgs& r, hipStream_t s);
constexpr unsigned kDiffcorMaxGroup = 8;  // lanes that may share one trajectory in k_diffcor
hipError_t launch_diffcor(const DiffCorArgs& a, unsigned group, unsigned n_cu, hipStream_t s);
hipError_t launch_observer_geometry(const GeometryArgs& a, hipStream_t s);
hipError_t launch_device_normals(unsigned long long seed, const unsigned long long* traj_key,
                                 unsigned long long key_base, unsigned long long n_traj, unsigned k_cap,
                                 unsigned n_noise, double* out, hipStream_t s);
hipError_t launch_selftest_aberth(unsigned long long n_threads, unsigned long long n_per_thread,
                                  unsigned long long seed, unsigned long long* d_out, hipStream_t s);
hipError_t launch_calibration_stream(unsigned long long n, const double* in, double* out, hipStream_t s);
hipError_t launch_prelim_kepuni(unsigned long long n, const double* prob, int math_mode, double* out, hipStream_t s);
hipError_t launch_glibc_eval(int fn, unsigned long long n, const double* x, const double* y, double* out, hipStream_t s);
hipError_t launch_crmath_eval(int fn, unsigned long long n, const double* x, double* out, hipStream_t s);
hipError_t launch_selftest_crmath(unsigned long long n_threads, unsigned long long n_per_thread,
                                  unsigned long long seed, unsigned long long* d_out, hipStream_t s);
hipError_t launch_selftest_division(unsigned long long n_threads, unsigned long long n_per_thread,
                                    unsigned long long seed, unsigned long long* d_mismatches, hipStream_t s);
hipError_t launch_propagate_universal(unsigned long long n, const double* rv, const double* t0,
                                      const double* t1, const SolverSpec& sv, int math_mode, double* out, int* status,
                                      hipStream_t s);

}  // namespace outfit
