// kernels.h — argument blocks and launchers shared by kernels.hip and the C-ABI
// host layer (api.hip). Everything here is plain data passed by value to a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "device/ephem.hpp"
#include "device/gauss.hpp"
#include "device/triplets.hpp"

namespace outfit {

constexpr unsigned kCandBlock = 256;  // lanes per workgroup of the candidate kernel

// per-candidate flag word written by K2 and replayed by K3
constexpr unsigned kFlagEmpty = 0x80000000u;  // slot beyond n_triplets (never read by K3)
constexpr unsigned kFlagOk = 0x10u;           // prelim_orbit returned Ok; low nibble = cause otherwise
constexpr unsigned kFlagCorrected = 0x20u;    // GaussResult::CorrectedOrbit
constexpr unsigned kFlagCometary = 0x40u;     // OrbitalElements::Cometary
constexpr unsigned kFlagRmsErr = 0x80u;       // an ephemeris_error returned Err
constexpr unsigned kFlagConvShift = 8;        // 3 bits: to_equinoctial error (3 / 4)

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
};

struct PrepassArgs {
  unsigned long long n_obs;
  const double* mjd;
  const double* geo_ecl;
  double* obs_pos;
  EphemTable T;
  int* error_flag;
};

struct CandidateArgs {
  unsigned long long n_cand, traj0;
  unsigned cand_per_traj, n_real, k_cap;
  const unsigned long long* traj_offset;
  const double *mjd, *ra, *dec, *sra, *sdec, *helio, *obs_pos, *noise_z;
  const unsigned* n_triplets;
  const unsigned long long* triplets;
  GaussParams P;
  double extf, dtmax, noise_scale;
  unsigned* flags;
  unsigned* n_win;
  double* sum;
  double* el_epoch;
  double* el;  // [6][n_cand]
};

struct SelectArgs {
  unsigned long long n_traj, traj0, n_cand;
  unsigned cand_per_traj, n_real;
  const unsigned* n_triplets;
  const unsigned* flags;
  const unsigned* n_win;
  const double* sum;
  const double* el_epoch;
  const double* el;
  GaussResultRec* out;  // indexed by absolute trajectory
};

hipError_t launch_triplet_select(const TripletArgs& a, hipStream_t s);
hipError_t launch_observer_prepass(const PrepassArgs& a, hipStream_t s);
hipError_t launch_gauss_candidate(const CandidateArgs& a, hipStream_t s);
hipError_t launch_select_best(const SelectArgs& a, hipStream_t s);
hipError_t launch_propagate_universal(unsigned long long n, const double* rv, const double* t0,
                                      const double* t1, double convergency, double* out, int* status,
                                      hipStream_t s);

}  // namespace outfit
