// kernels.hip — the HIP kernels of the batched Gauss-IOD path for gfx950.
//
//   K0  triplet_select     one lane per trajectory     generate_triplets
//                          (triplet_generation/mod.rs:328-408)
//   K1  observer_prepass   one lane per observation    resolve_observer_geometry
//                          (observation_ephemeris.rs:303-318; candidate-independent)
//   K2a gauss_roots        one lane per candidate      gauss_prelim, coeff_eight_poly, Descartes,
//                          aberth, root filter (gauss.rs:1119-1150)
//   K2b gauss_correct      persistent lanes + queue    accept_root, pos_and_vel_correction
//                          (gauss.rs:816-870,1284-1418)
//   K2c orbit_rms          one lane per candidate      ccek1, to_equinoctial, rms_orbit_error
//                          (orb_elem.rs:58, orbit_type/mod.rs:399, trajectory.rs:352)
//   K3  select_best        one lane per trajectory     the sequential fold of
//                          estimate_best_orbit (trajectory.rs:455-544)
//   propagate_universal    one lane per state          kepler/propagation.rs:114
//
// Why K2 is three kernels: 38 % of the candidates run the correction loop to its
// 50-iteration cap while the mean is 26 (the 1e-10 threshold sits on the noise floor of
// the f/g series), so a lane-per-candidate kernel makes every wave pay ~50 iterations.
// K2b instead keeps every lane busy: a lane that finishes pulls the next candidate from a
// global queue, and all lanes of a wave execute the same single-iteration body.
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
// candidate decoding shared by K2a/K2b/K2c
// ------------------------------------------------------------------------------
struct CandidateRef {
  unsigned long long lt, base, n_obs;
  unsigned t, r;
  unsigned idx[3];
  bool live;
};

__device__ __forceinline__ CandidateRef decode_candidate(const CandidateIndex& I, unsigned long long cid) {
  CandidateRef c;
  c.lt = cid / I.cand_per_traj;
  const unsigned within = (unsigned)(cid % I.cand_per_traj);
  c.t = within / I.n_real;
  c.r = within % I.n_real;
  c.live = c.t < I.n_triplets[c.lt];
  const unsigned long long traj = I.traj0 + c.lt;
  c.base = I.traj_offset[traj];
  c.n_obs = I.traj_offset[traj + 1] - c.base;
  c.idx[0] = c.idx[1] = c.idx[2] = 0;
  if (c.live) unpack_triplet(I.triplets[c.lt * I.k_cap + c.t], c.idx[0], c.idx[1], c.idx[2]);
  return c;
}

// GaussObs of the candidate: the three observations, noise applied for realisation r > 0
// (gauss.rs:341-386; the standard normals are drawn by the host, in the reference's order)
__device__ __forceinline__ void gather_angles(const CandidateIndex& I, const CandidateRef& c, double (&ra)[3],
                                              double (&dec)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long o = c.base + c.idx[k];
    ra[k] = I.ra[o];
    dec[k] = I.dec[o];
  }
  if (c.r > 0) {
    const unsigned long long traj = I.traj0 + c.lt;
    const double* z = I.noise_z + ((traj * I.k_cap + c.t) * (unsigned long long)(I.n_real - 1) + (c.r - 1)) * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = c.base + c.idx[k];
      ra[k] = ra[k] + z[k] * (I.sra[o] * I.noise_scale);
      dec[k] = dec[k] + z[3 + k] * (I.sdec[o] * I.noise_scale);
    }
  }
}
__device__ __forceinline__ void gather_times_observer(const CandidateIndex& I, const CandidateRef& c,
                                                      double (&t)[3], Mat3& obs) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long o = c.base + c.idx[k];
    t[k] = I.mjd[o];
    obs.m[0][k] = I.helio[3 * o];
    obs.m[1][k] = I.helio[3 * o + 1];
    obs.m[2][k] = I.helio[3 * o + 2];
  }
}

// ------------------------------------------------------------------------------
// K2a: direction matrix, 8th-degree polynomial, Aberth roots, root filter
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gauss_roots(RootsArgs a) {
  const unsigned long long cid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= a.I.n_cand) return;
  const CandidateRef c = decode_candidate(a.I, cid);
  if (!c.live) { a.W.flags[cid] = kFlagEmpty; return; }
  const unsigned long long n = a.I.n_cand;

  double ra[3], dec[3], t[3];
  Mat3 obs, Sinv;
  gather_angles(a.I, c, ra, dec);
  const Mat3 S = unit_matrix(ra, dec);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) a.W.S[(unsigned long long)(3 * r + k) * n + cid] = S.m[r][k];
  if (!try_inverse(S, Sinv)) { a.W.flags[cid] = (unsigned)kCauseSingular; return; }
  gather_times_observer(a.I, c, t, obs);
  const GaussGeometry q = gauss_times(t);
  double c6, c3, c0;
  coeff_eight_poly(obs, S, Sinv, q, c6, c3, c0);
  if (descartes_upper_bound(c0, c3, c6) == 0) { a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }

  Cx z[8];
  if (aberth_deg8(c0, c3, c6, a.P.aberth_max_iter, a.P.aberth_eps, z) == 2) {
    a.W.flags[cid] = (unsigned)kCauseRootFinding;
    return;
  }
  // visit_real_positive_roots + the r2 window (gauss.rs:964-986,1148), solver order kept
  unsigned nf = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double re = z[i].re;
    if (re > 0.0 && fabs(z[i].im) < a.P.root_imag_eps && re >= a.P.r2_min_au && re <= a.P.r2_max_au) {
      a.W.roots[(unsigned long long)nf * n + cid] = re;
      ++nf;
    }
  }
  a.W.flags[cid] = nf ? (kFlagOk | (nf << kFlagRootsShift)) : (unsigned)kCauseNoRoots;
}

// ------------------------------------------------------------------------------
// K2b: persistent lanes. Each lane owns one candidate at a time and walks its
// filtered roots in solver order exactly like the closure at gauss.rs:1147-1197:
// accept_root -> correction loop -> first CorrectedOrbit wins (prelim_orbit,
// gauss.rs:1238-1247), a root whose loop returns None contributes a PrelimOrbit,
// the walk stops after max_tested_solutions pushes.
// ------------------------------------------------------------------------------
constexpr unsigned kRefillThreshold = 16;  // idle lanes per wave before the queue is touched

__global__ void __launch_bounds__(kCorrectBlock, 2)
k_gauss_correct(CorrectArgs a) {
  extern __shared__ double lds[];  // [kCorrectLdsDoubles][blockDim.x]
  const unsigned tid = threadIdx.x, bs = blockDim.x;
  const unsigned long long n = a.I.n_cand;
  auto L = [&](unsigned k) -> double& { return lds[k * bs + tid]; };

  long long cand = -1;
  bool exhausted = false, need_init = false, have_prelim = false;
  unsigned slot = 0, n_filt = 0, pushes = 0;
  double t_mid = 0.0, dt01 = 0.0, dt21 = 0.0;
  GaussGeometry q{};
  CorrectionState st{};

  for (;;) {
    // ---- refill idle lanes from the global queue (wave-uniform decisions) ----------
    const bool needy = (cand < 0) && !exhausted;
    const unsigned long long need_mask = __ballot(needy);
    const unsigned long long active_mask = __ballot(cand >= 0);
    const unsigned n_need = (unsigned)__popcll(need_mask);
    if (n_need && (n_need >= kRefillThreshold || active_mask == 0ull)) {
      unsigned long long base = 0;
      const unsigned lane = __lane_id();
      const unsigned leader = (unsigned)__ffsll((long long)need_mask) - 1u;
      if (lane == leader) base = atomicAdd(a.queue, (unsigned long long)n_need);
      base = __shfl(base, (int)leader);
      if (needy) {
        const unsigned rank = (unsigned)__popcll(need_mask & ((1ull << lane) - 1ull));
        const unsigned long long my = base + rank;
        if (my >= n) {
          exhausted = true;
        } else {
          const unsigned f = a.W.flags[my];
          const unsigned nf = (f >> kFlagRootsShift) & 0xFu;
          if ((f & kFlagOk) && !(f & kFlagEmpty) && nf > 0) {
            cand = (long long)my;
            n_filt = nf; slot = 0; pushes = 0; have_prelim = false; need_init = true;
            const CandidateRef c = decode_candidate(a.I, my);
            double t[3];
            Mat3 obs, S, Sinv;
            gather_times_observer(a.I, c, t, obs);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int k = 0; k < 3; ++k) S.m[r][k] = a.W.S[(unsigned long long)(3 * r + k) * n + my];
            (void)try_inverse(S, Sinv);  // non-singular: phase A checked it
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int k = 0; k < 3; ++k) {
                L(3 * r + k) = S.m[r][k];
                L(9 + 3 * r + k) = Sinv.m[r][k];
                L(18 + 3 * r + k) = obs.m[r][k];
              }
            q = gauss_times(t);
            t_mid = t[1];
            dt01 = t[0] - t[1];
            dt21 = t[2] - t[1];
          }
        }
      }
    }
    if (__ballot(cand >= 0) == 0ull && __ballot(!exhausted) == 0ull) break;
    if (cand < 0) continue;

    // ---- one step of this lane's candidate -------------------------------------------
    Mat3 S, Sinv, obs;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S.m[r][k] = L(3 * r + k);
        Sinv.m[r][k] = L(9 + 3 * r + k);
        obs.m[r][k] = L(18 + 3 * r + k);
      }
    CorrStep step = kCorrContinue;
    bool finished = false;
    if (need_init) {
      // next accepted root in solver order (gauss.rs:1152-1163)
      bool accepted = false;
      while (slot < n_filt) {
        const double r2 = a.W.roots[(unsigned long long)slot * n + (unsigned long long)cand];
        double e_pre;
        if (accept_root(obs, S, Sinv, t_mid, q, a.P, r2, st.pos, st.vel, e_pre)) {
          accepted = true;
          L(27) = st.pos.m[0][1]; L(28) = st.pos.m[1][1]; L(29) = st.pos.m[2][1];
          L(30) = st.vel.x; L(31) = st.vel.y; L(32) = st.vel.z; L(33) = e_pre;
          st.epoch = 0.0; st.chi01 = 0.0; st.chi21 = 0.0; st.has01 = false; st.has21 = false; st.it = 0;
          break;
        }
        ++slot;
      }
      need_init = false;
      if (!accepted) finished = true;
      else if (fabs(dt01) <= kEps || fabs(dt21) <= kEps) step = kCorrNone;  // gauss.rs:1311-1313
    } else {
      step = correction_iteration(obs, S, Sinv, t_mid, dt01, dt21, a.P, st);
    }
    const unsigned long long ucand = (unsigned long long)cand;
    if (step == kCorrSome) {
      a.W.state[0 * n + ucand] = st.pos.m[0][1];
      a.W.state[1 * n + ucand] = st.pos.m[1][1];
      a.W.state[2 * n + ucand] = st.pos.m[2][1];
      a.W.state[3 * n + ucand] = st.vel.x;
      a.W.state[4 * n + ucand] = st.vel.y;
      a.W.state[5 * n + ucand] = st.vel.z;
      a.W.state[6 * n + ucand] = st.epoch;
      a.W.flags[ucand] = kFlagOk | kFlagCorrected;
      cand = -1;
    } else if (step == kCorrNone) {
      if (!have_prelim) {  // the first PrelimOrbit is the fallback (gauss.rs:1245-1246)
        have_prelim = true;
#pragma unroll
        for (unsigned k = 0; k < 7; ++k) a.W.state[(unsigned long long)k * n + ucand] = L(27 + k);
      }
      ++pushes;
      ++slot;
      if (pushes >= a.P.max_tested_solutions || slot >= n_filt) finished = true;
      else need_init = true;
    }
    if (finished) {
      a.W.flags[ucand] = have_prelim ? kFlagOk : (unsigned)kCauseNoRoots;
      cand = -1;
    }
  }
}

// ------------------------------------------------------------------------------
// K2c: Cartesian state -> elements -> equinoctial -> sum of squared normalised residuals
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

__global__ void __launch_bounds__(256)
k_orbit_rms(RmsArgs a) {
  const unsigned long long cid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= a.I.n_cand) return;
  const unsigned f0 = a.W.flags[cid];
  if ((f0 & kFlagEmpty) || !(f0 & kFlagOk)) return;
  const unsigned long long n = a.I.n_cand;
  const CandidateRef c = decode_candidate(a.I, cid);
  const Vec3 r2 = {a.W.state[0 * n + cid], a.W.state[1 * n + cid], a.W.state[2 * n + cid]};
  const Vec3 v2 = {a.W.state[3 * n + cid], a.W.state[4 * n + cid], a.W.state[5 * n + cid]};
  const double epoch = a.W.state[6 * n + cid];

  // build_result -> compute_orbit_from_state -> ccek1 (gauss.rs:906-923,1063-1076)
  const Elements el = ccek1(equ_to_ecl(r2), equ_to_ecl(v2), epoch);
  unsigned flags = (f0 & (kFlagOk | kFlagCorrected)) | (el.kind ? kFlagCometary : 0u);
#pragma unroll
  for (int k = 0; k < 6; ++k) a.W.el[(unsigned long long)k * n + cid] = el.e[k];

  Equinoctial eq;
  const int conv = to_equinoctial(el, eq);
  if (conv != 0) {  // `?` at trajectory.rs:493 aborts the trajectory: K3 replays that
    a.W.flags[cid] = flags | ((unsigned)conv << kFlagConvShift);
    return;
  }
  // rms_orbit_error (trajectory.rs:352-427): full left-fold sum; pruning is replayed in K3
  unsigned long long i_start, i_end;
  select_rms_interval(a.I.mjd + c.base, c.n_obs, c.idx[0], c.idx[2], a.extf, a.dtmax, i_start, i_end);
  a.W.n_win[cid] = (unsigned)(i_end - i_start + 1);
  double acc = 0.0;
  bool err = false;
  if (sqrt(eq.h * eq.h + eq.k * eq.k) >= 1.0) {
    err = true;  // check_elliptical_orbit (observation_ephemeris.rs:288-296)
  } else {
    const TwoBodyFrame F = twobody_frame(eq);
    for (unsigned long long i = i_start; i <= i_end; ++i) {
      const unsigned long long o = c.base + i;
      const Vec3 op = {a.obs_pos[3 * o], a.obs_pos[3 * o + 1], a.obs_pos[3 * o + 2]};
      double term;
      if (!ephemeris_error(eq, F, a.I.mjd[o], a.I.ra[o], a.I.dec[o], a.I.sra[o], a.I.sdec[o], op, term)) {
        err = true;
        break;
      }
      acc = acc + term;
    }
  }
  a.W.sum[cid] = acc;
  a.W.flags[cid] = flags | (err ? kFlagRmsErr : 0u);
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
    const unsigned f = a.W.flags[cid];
    attempts += 1;
    if (!(f & kFlagOk)) { last_error = (int)(f & 0xFu); continue; }  // prelim_orbit Err
    const unsigned conv = (f >> kFlagConvShift) & 0x7u;
    if (conv) { out.status = (int)conv; aborted = true; break; }      // to_equinoctial()? (quirk Q2)
    // rms_orbit_error with prune_if_rms_ge = Some(best): the partial sums are
    // non-decreasing, so "some partial sum >= cutoff" == "final sum >= cutoff".
    const double denom = 2.0 * (double)a.W.n_win[cid];
    const double cutoff = isfinite(best) ? best * best * denom : INFINITY;
    double rms;
    if (f & kFlagRmsErr) rms = best;
    else {
      const double s = a.W.sum[cid];
      rms = (s >= cutoff) ? best : sqrt(s / denom);
    }
    if (!isfinite(rms)) { last_error = 4; continue; }  // NonFiniteScore
    if (rms < best) { best = rms; best_c = (long long)c; }
  }
  out.attempts = attempts;
  if (!aborted) {
    if (best_c >= 0) {
      const unsigned long long cid = c0 + (unsigned long long)best_c;
      const unsigned f = a.W.flags[cid];
      out.status = 0;
      out.stage = (f & kFlagCorrected) ? 1 : 0;
      out.kind = (f & kFlagCometary) ? 1 : 0;
      out.best_triplet = (unsigned)best_c / a.n_real;
      out.best_realization = (unsigned)best_c % a.n_real;
      out.epoch = a.W.state[6 * a.n_cand + cid];
      for (int k = 0; k < 6; ++k) out.elem[k] = a.W.el[(unsigned long long)k * a.n_cand + cid];
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
hipError_t launch_gauss_roots(const RootsArgs& a, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.I.n_cand + 255) / 256);
  hipLaunchKernelGGL(k_gauss_roots, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gauss_correct(const CorrectArgs& a, unsigned n_cu, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  // persistent grid: two 256-lane workgroups per CU (LDS: 34 doubles per lane), never more
  // workgroups than there is work; every wave exits once the queue is drained
  const unsigned long long want = (a.I.n_cand + kCorrectBlock - 1) / kCorrectBlock;
  const unsigned grid = (unsigned)(want < 2ull * n_cu ? want : 2ull * n_cu);
  const size_t lds = (size_t)kCorrectLdsDoubles * kCorrectBlock * sizeof(double);
  hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gauss_correct, dim3(grid), dim3(kCorrectBlock), lds, s, a);
  return hipGetLastError();
}
hipError_t launch_orbit_rms(const RmsArgs& a, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.I.n_cand + 255) / 256);
  hipLaunchKernelGGL(k_orbit_rms, dim3(grid), dim3(256), 0, s, a);
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
