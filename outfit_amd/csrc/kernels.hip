// kernels.hip — the HIP kernels of the batched Gauss-IOD path for gfx950.
//
//   K0 triplet_select     one lane per trajectory           (generate_triplets,
//                          triplet_generation/mod.rs:328-408)
//   K1 observer_prepass   one lane per observation          (resolve_observer_geometry,
//                          observation_ephemeris.rs:303-318; candidate-independent)
//   K2 gauss_candidate    one lane per (trajectory, triplet, realisation)
//                          (prelim_orbit gauss.rs:1238 + to_equinoctial mod.rs:399 +
//                          rms_orbit_error trajectory.rs:352)
//   K3 select_best        one lane per trajectory           (the sequential fold of
//                          estimate_best_orbit, trajectory.rs:455-544)
//   propagate_universal   one lane per state                (kepler/propagation.rs:114)
//
// Compiled with -ffp-contract=off: the reference (Rust) never fuses a*b+c.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "device/ephem.hpp"
#include "device/gauss.hpp"
#include "device/orbit.hpp"
#include "device/triplets.hpp"

namespace outfit {

// ------------------------------------------------------------------------------
// K0
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_triplet_select(TripletArgs a) {
  const unsigned long long lt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lt >= a.n_traj) return;
  const unsigned long long traj = a.traj0 + lt;
  const unsigned long long base = a.traj_offset[traj];
  const unsigned long long n_obs = a.traj_offset[traj + 1] - base;
  // heap slot s of lane lt: scratch[(s * n_lanes_padded) + lt]
  HeapView h{a.heap_w + lt, a.heap_id + lt, a.heap_stride};
  const unsigned n = select_triplets(a.mjd + base, n_obs, a.P, h);
  a.n_triplets[lt] = n;
  for (unsigned s = 0; s < n; ++s) a.triplets[lt * a.P.k_cap + s] = h.I(s);
}

// ------------------------------------------------------------------------------
// K1: observer position used by the residuals. Blocks of coefficient data are
// staged through LDS when the workgroup's epochs span at most kStageBlocks
// consecutive table blocks (the usual case: epochs are grouped per trajectory).
// ------------------------------------------------------------------------------
constexpr unsigned kStageBlocks = 8;
constexpr unsigned kMaxPerBlock = 1024;  // doubles; DE440 subset needs 456

__global__ void __launch_bounds__(256)
k_observer_prepass(PrepassArgs a) {
  __shared__ double lds[kStageBlocks * kMaxPerBlock];
  __shared__ unsigned long long s_min, s_max;
  __shared__ int s_bad;
  const unsigned long long o = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = o < a.n_obs;
  if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_bad = 0; }
  __syncthreads();
  unsigned long long nr = 0;
  double tau = 0.0;
  bool ok = true;
  if (live) {
    ok = record_index(a.T, a.mjd[o], nr, tau);
    if (ok) { atomicMin(&s_min, nr); atomicMax(&s_max, nr); }
    else s_bad = 1;
  }
  __syncthreads();
  const unsigned long long bmin = s_min, bmax = s_max;
  const bool staged = (bmin <= bmax) && (bmax - bmin < kStageBlocks) && (a.T.per_block <= kMaxPerBlock);
  if (staged) {
    // coalesced copy of blocks [bmin, bmax] into LDS
    const unsigned long long n_d = (bmax - bmin + 1) * a.T.per_block;
    const double* src = a.T.coef + bmin * a.T.per_block;
    for (unsigned long long i = threadIdx.x; i < n_d; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  if (s_bad && threadIdx.x == 0) atomicExch(a.error_flag, 1);
  if (!live) return;
  Vec3 out = {nan(""), nan(""), nan("")};
  if (ok) {
    const double* blk = staged ? (lds + (nr - bmin) * a.T.per_block) : (a.T.coef + nr * a.T.per_block);
    const Vec3 earth = earth_from_block(a.T, blk, tau);
    out = observer_equ(earth, {a.geo_ecl[3 * o], a.geo_ecl[3 * o + 1], a.geo_ecl[3 * o + 2]});
  }
  a.obs_pos[3 * o] = out.x;
  a.obs_pos[3 * o + 1] = out.y;
  a.obs_pos[3 * o + 2] = out.z;
}

// ------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------
// trajectory.rs:294-350
__device__ __forceinline__ void select_rms_interval(const double* mjd, unsigned long long n_obs, unsigned i1,
                                           unsigned i3, double extf, double dtmax,
                                           unsigned long long& i_start, unsigned long long& i_end) {
  const double t1 = mjd[i1], t3 = mjd[i3];
  double dt = (extf >= 0.0) ? (t3 - t1) * extf : 10.0 * (mjd[n_obs - 1] - mjd[0]);
  if (dtmax >= 0.0) dt = fmax(dt, dtmax);
  unsigned long long s = 0;
  for (unsigned long long i = (unsigned long long)i1 + 1; i-- > 0;) {
    if (t1 - mjd[i] > dt) break;
    s = i;
  }
  unsigned long long e = n_obs - 1;
  for (unsigned long long i = i3; i < n_obs; ++i) {
    if (mjd[i] - t3 > dt) break;
    e = i;
  }
  i_start = s;
  i_end = e;
}

__global__ void __launch_bounds__(kCandBlock)
k_gauss_candidate(CandidateArgs a) {
  extern __shared__ double lds_roots[];  // [2 gen][8 roots][re|im][blockDim.x]
  const unsigned long long cid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= a.n_cand) return;
  const unsigned long long lt = cid / a.cand_per_traj;
  const unsigned within = (unsigned)(cid % a.cand_per_traj);
  const unsigned t = within / a.n_real, r = within % a.n_real;
  if (t >= a.n_triplets[lt]) { a.flags[cid] = kFlagEmpty; return; }
  const unsigned long long traj = a.traj0 + lt;
  const unsigned long long base = a.traj_offset[traj];
  const unsigned long long n_obs = a.traj_offset[traj + 1] - base;

  unsigned idx[3];
  unpack_triplet(a.triplets[lt * a.k_cap + t], idx[0], idx[1], idx[2]);
  GaussTriplet g;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long o = base + idx[k];
    g.ra[k] = a.ra[o];
    g.dec[k] = a.dec[o];
    g.t[k] = a.mjd[o];
    g.obs.m[0][k] = a.helio[3 * o];
    g.obs.m[1][k] = a.helio[3 * o + 1];
    g.obs.m[2][k] = a.helio[3 * o + 2];
  }
  if (r > 0) {  // noisy realisation (gauss.rs:341-386), normals drawn by the host
    const double* z = a.noise_z + ((traj * a.k_cap + t) * (unsigned long long)(a.n_real - 1) + (r - 1)) * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = base + idx[k];
      g.ra[k] = g.ra[k] + z[k] * (a.sra[o] * a.noise_scale);
      g.dec[k] = g.dec[k] + z[3 + k] * (a.sdec[o] * a.noise_scale);
    }
  }

  const RootStore st{lds_roots + threadIdx.x, blockDim.x};
  const StateSolution sol = gauss_prelim_state(g, a.P, st);
  if (sol.cause != 0) { a.flags[cid] = (unsigned)sol.cause; return; }

  // build_result -> compute_orbit_from_state -> ccek1 (gauss.rs:906-923,1063-1076)
  const Elements el = ccek1(equ_to_ecl(sol.r2), equ_to_ecl(sol.v2), sol.epoch);
  unsigned flags = kFlagOk | (sol.corrected ? kFlagCorrected : 0u) | (el.kind ? kFlagCometary : 0u);
  a.el_epoch[cid] = el.epoch;
#pragma unroll
  for (int k = 0; k < 6; ++k) a.el[(unsigned long long)k * a.n_cand + cid] = el.e[k];

  Equinoctial eq;
  const int conv = to_equinoctial(el, eq);
  if (conv != 0) {  // `?` at trajectory.rs:493 aborts the trajectory: K3 replays that
    a.flags[cid] = flags | ((unsigned)conv << kFlagConvShift);
    return;
  }

  // rms_orbit_error (trajectory.rs:352-427): full left-fold sum; pruning is replayed in K3
  unsigned long long i_start, i_end;
  select_rms_interval(a.mjd + base, n_obs, idx[0], idx[2], a.extf, a.dtmax, i_start, i_end);
  a.n_win[cid] = (unsigned)(i_end - i_start + 1);
  double acc = 0.0;
  bool err = false;
  if (sqrt(eq.h * eq.h + eq.k * eq.k) >= 1.0) {
    err = true;  // check_elliptical_orbit (observation_ephemeris.rs:288-296)
  } else {
    const TwoBodyFrame F = twobody_frame(eq);
    for (unsigned long long i = i_start; i <= i_end; ++i) {
      const unsigned long long o = base + i;
      const Vec3 op = {a.obs_pos[3 * o], a.obs_pos[3 * o + 1], a.obs_pos[3 * o + 2]};
      double term;
      if (!ephemeris_error(eq, F, a.mjd[o], a.ra[o], a.dec[o], a.sra[o], a.sdec[o], op, term)) {
        err = true;
        break;
      }
      acc = acc + term;
    }
  }
  a.sum[cid] = acc;
  a.flags[cid] = flags | (err ? kFlagRmsErr : 0u);
}

// ------------------------------------------------------------------------------
// K3: replay of the sequential loop of estimate_best_orbit over the candidates of
// one trajectory, in (triplet, realisation) order.
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_select_best(SelectArgs a) {
  const unsigned long long lt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lt >= a.n_traj) return;
  GaussResultRec out;
  out.status = 0; out.cause = 0; out.stage = 0; out.kind = 0;
  out.attempts = 0; out.best_triplet = 0; out.best_realization = 0;
  out.epoch = 0.0; out.rms = 0.0;
  for (int k = 0; k < 6; ++k) out.elem[k] = 0.0;
  const unsigned nt = a.n_triplets[lt];
  out.n_triplets = nt;
  if (nt == 0) {
    out.status = 1;  // NoFeasibleTriplets (trajectory.rs:439-451)
    a.out[a.traj0 + lt] = out;
    return;
  }
  double best = INFINITY;
  long long best_c = -1;
  int last_error = 0;
  unsigned attempts = 0;
  bool aborted = false;
  const unsigned long long c0 = lt * a.cand_per_traj;
  const unsigned n_c = nt * a.n_real;
  for (unsigned c = 0; c < n_c; ++c) {
    const unsigned long long cid = c0 + c;
    const unsigned f = a.flags[cid];
    attempts += 1;
    if (!(f & kFlagOk)) { last_error = (int)(f & 0xFu); continue; }  // prelim_orbit Err
    const unsigned conv = (f >> kFlagConvShift) & 0x7u;
    if (conv) { out.status = (int)conv; aborted = true; break; }      // to_equinoctial()? (quirk Q2)
    // rms_orbit_error with prune_if_rms_ge = Some(best): the partial sums are
    // non-decreasing, so "some partial sum >= cutoff" == "final sum >= cutoff".
    const double denom = 2.0 * (double)a.n_win[cid];
    const double cutoff = isfinite(best) ? best * best * denom : INFINITY;
    double rms;
    if (f & kFlagRmsErr) rms = best;
    else {
      const double s = a.sum[cid];
      rms = (s >= cutoff) ? best : sqrt(s / denom);
    }
    if (!isfinite(rms)) { last_error = 4; continue; }  // NonFiniteScore
    if (rms < best) { best = rms; best_c = (long long)c; }
  }
  out.attempts = attempts;
  if (!aborted) {
    if (best_c >= 0) {
      const unsigned long long cid = c0 + (unsigned long long)best_c;
      const unsigned f = a.flags[cid];
      out.status = 0;
      out.stage = (f & kFlagCorrected) ? 1 : 0;
      out.kind = (f & kFlagCometary) ? 1 : 0;
      out.best_triplet = (unsigned)best_c / a.n_real;
      out.best_realization = (unsigned)best_c % a.n_real;
      out.epoch = a.el_epoch[cid];
      for (int k = 0; k < 6; ++k) out.elem[k] = a.el[(unsigned long long)k * a.n_cand + cid];
      out.rms = best;
    } else {
      out.status = 2;  // NoViableOrbit{cause, attempts}
      out.cause = last_error;
    }
  }
  a.out[a.traj0 + lt] = out;
}

// ------------------------------------------------------------------------------
// kepler::propagate_universal, batched
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_propagate_universal(unsigned long long n, const double* rv, const double* t0, const double* t1,
                      double convergency, double* out, int* status) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double o[11];
  for (int k = 0; k < 11; ++k) o[k] = nan("");
  const Vec3 r = {rv[6 * i], rv[6 * i + 1], rv[6 * i + 2]};
  const Vec3 v = {rv[6 * i + 3], rv[6 * i + 4], rv[6 * i + 5]};
  status[i] = propagate_universal(r, v, t0[i], t1[i], convergency, o);
  for (int k = 0; k < 11; ++k) out[11 * i + k] = o[k];
}

// ------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------
hipError_t launch_triplet_select(const TripletArgs& a, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_traj + 63) / 64);
  hipLaunchKernelGGL(k_triplet_select, dim3(grid), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_observer_prepass(const PrepassArgs& a, hipStream_t s) {
  if (a.n_obs == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_obs + 255) / 256);
  hipLaunchKernelGGL(k_observer_prepass, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gauss_candidate(const CandidateArgs& a, hipStream_t s) {
  if (a.n_cand == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_cand + kCandBlock - 1) / kCandBlock);
  const size_t lds = (size_t)2 * 8 * 2 * kCandBlock * sizeof(double);
  hipLaunchKernelGGL(k_gauss_candidate, dim3(grid), dim3(kCandBlock), lds, s, a);
  return hipGetLastError();
}
hipError_t launch_select_best(const SelectArgs& a, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_traj + 63) / 64);
  hipLaunchKernelGGL(k_select_best, dim3(grid), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_propagate_universal(unsigned long long n, const double* rv, const double* t0,
                                      const double* t1, double convergency, double* out, int* status,
                                      hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_propagate_universal, dim3(grid), dim3(256), 0, s, n, rv, t0, t1, convergency, out,
                     status);
  return hipGetLastError();
}

}  // namespace outfit
