// diffcor.hip — K4 `k_diffcor`: differential orbit correction (Newton weighted least squares with
// two-body partials, chi-square outlier rejection, covariance rescaling); persistent lane groups, one
// trajectory per group at a time, trajectories handed out by a work queue.
// Replaces `differential_correction` (differential_orbit_correction/mod.rs:66-112) as driven by
// FitLSQ::fit_lsq (obs_dataset_api.rs:57-130); device code in device/diffcor.hpp.
#include <algorithm>

#include "kernels.h"
#include "device/diffcor.hpp"

namespace outfit {

#ifdef OUTFIT_DC_STATS  // diagnostic build (tools/perf_lsq.py with a stats library): (wave passes, active lanes) per phase
__device__ unsigned long long g_dc_stat[32];
#define DCSTAT(slot)                                                                \
  do {                                                                              \
    const unsigned long long m__ = __ballot(true);                                  \
    if (__lane_id() == (unsigned)__ffsll((long long)m__) - 1u) {                    \
      atomicAdd(&g_dc_stat[2 * (slot)], 1ull);                                      \
      atomicAdd(&g_dc_stat[2 * (slot) + 1], (unsigned long long)__popcll(m__));     \
    }                                                                               \
  } while (0)
#else
#define DCSTAT(slot) do { } while (0)
#endif

namespace {
constexpr double kF64Max = 1.7976931348623157e308;

// acc += v on a lane-private LDS slot as one DS_ADD_F64 (IEEE add, no returned value): the
// accumulators never occupy registers
__device__ __forceinline__ void lds_add(double* acc, double v) {
  (void)__hip_atomic_fetch_add(acc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// sum of an integer over the kG consecutive lanes of a group (all of them alive and converged here)
template <unsigned kG>
__device__ __forceinline__ unsigned group_sum(unsigned v) {
#pragma unroll
  for (unsigned m = 1; m < kG; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
}  // namespace

// Per-observation scratch (global, indexed by observation row o):
//   res[b][c][o]       : (residual_ra, residual_dec, chi) double-buffered: b = cur is written by the
//                        running iteration, b = cur^1 holds the last ACCEPTED iteration (ObsFitData)
//   sel[o]             : ObsSelection; written by the first outlier pass (all Active before it)
// The observation equations (12 partials per observation) are NOT kept: the outlier pass, which only
// follows an accepted iteration, recomputes them from that iteration's input elements — the same
// deterministic arithmetic, hence the same bits — instead of 12 scattered stores per observation
// in every Newton iteration (measured: -28 % kernel time).
// Per-trajectory scratch: cov[b][36][n_traj] = OrbitalUncertainty.covariance, double-buffered like `res`
// (b = cur: inverse of the running iteration, b = cur^1: last accepted iteration).
//
// kG lanes share one trajectory (kG = 1, 2, 4, 8): lane `sub` evaluates observations i = sub (mod kG), the
// group then adds its kG terms to the shared LDS accumulators one lane after the other, in observation
// order, so every accumulator still sums the observations left to right exactly as the reference's
// sequential loop does. Everything after the observation loop (inverse, bookkeeping) is computed redundantly
// by the kG lanes on identical inputs.
//
// Persistent groups + work queue: trajectories need 1..max_newton_iterations Newton iterations (bench
// workload: mean 2.5, mean of the per-wave maximum 4.0), so a wave bound to 64 / kG fixed trajectories idles
// 37 % of its lane-iterations. Here every group pulls its next trajectory from a global queue as soon as it
// has written the previous one, and one trip of the wave's loop is one single_iteration (or one outlier pass)
// of whatever trajectory each group holds — the reference's nested loops flattened into (outer, inner, phase)
// registers. Per-trajectory arithmetic and its order are unchanged; which group serves a trajectory is not
// observable in the results. Every wave leaves once the queue is drained and its groups are idle.
// One wave per SIMD: with the correctly-rounded functions inlined the kernel wants ~480 registers (256 VGPR +
// AGPRs). Measured at (64, 2), 111 spilled VGPRs with the three big helpers out of line: the same 1.99 ms per
// 100 k objects as one wave — the kernel is bound by fp64 issue, not by latency, so occupancy buys nothing.
template <unsigned kG, int kFn>  // kFn: kFnCr (correctly-rounded math mode) | kFnGlibc (glibc-faithful mode), diffcor.hpp
__global__ void __launch_bounds__(64, 1)
k_diffcor(DiffCorArgs A) {
  // normal matrix (36), right-hand side (6) and weighted square sum (1) accumulate in LDS, [entry][slot]
  constexpr unsigned kSlots = 64 / kG;
  __shared__ double sN[43][kSlots];
  const unsigned lane = threadIdx.x / kG;   // LDS slot of this lane's group
  const unsigned sub = threadIdx.x % kG;
  const DiffCorConfig& C = A.cfg;
  const unsigned long long RB = 3ull * A.n_obs;  // one residual buffer
  constexpr unsigned long long kNone = ~0ull;
  enum : int { kIterate = 0, kOutliers = 1 };

  // ---- the trajectory this group holds (identical in its kG lanes) --------------------------------------
  unsigned long long t = kNone, base = 0;
  unsigned n = 0;
  int status = 0, dc_error = 0, conv_error = 0, phase = kIterate;
  unsigned total_iters = 0, outer_started = 0, last_num_meas = 0;
  unsigned outer = 0, inner = 0, stagnation = 0, cur = 0;
  double last_rms = kF64Max, prev_rms = kF64Max;
  bool have_accepted = false, sel_written = false, exhausted = false;
  Equinoctial q{}, q_in{};  // running elements; input elements of the last accepted iteration

  // GaussResultRec the corrector starts from: Some(Ok(orbit)) of the caller's seeds, else the session's IOD
  auto seed_of = [&](unsigned long long tt) {
    GaussResultRec seed = A.iod[tt];
    if (A.initial) {
      const GaussResultRec given = A.initial[tt];
      if (given.status == 0) seed = given;  // anything else => run_iod_on_observations
    }
    return seed;
  };

  // ---- output of a finished trajectory, then the group is idle ------------------------------------------
  auto finish = [&]() {
    if (dc_error) status = 1;  // OUTFIT_LSQ_IOD_KEPT: Err(_) => Ok(initial_orbit)
    const double* ra_ = A.res + (unsigned long long)(cur ^ 1u) * RB;
    const bool ok = status == 0;
    unsigned n_rejected = 0;
    for (unsigned i = sub; i < n; i += kG) {
      const unsigned long long o = base + i;
      const unsigned char s = (ok && sel_written) ? A.sel[o] : (unsigned char)0;
      n_rejected += s != 0;
      if (A.obs_fit) {
        ObsFitRec f;
        const bool have = ok && have_accepted;
        f.residual_ra = have ? ra_[o] : 0.0;
        f.residual_dec = have ? ra_[A.n_obs + o] : 0.0;
        f.chi = have ? ra_[2 * A.n_obs + o] : 0.0;
        f.selection = s;
        f.pad = 0;
        A.obs_fit[o] = f;
      }
    }
    n_rejected = group_sum<kG>(n_rejected);
    if (sub == 0) {  // one lane of the group writes the trajectory's record
      LsqResultRec* out = A.out + t;
      out->iod = seed_of(t);
      out->status = status;
      out->dc_error = dc_error;
      out->conv_error = conv_error;
      out->total_newton_iterations = total_iters;
      out->outer_passes = outer_started;
      out->pad = 0;
      out->epoch = (status == 0 || status == 1) ? q.epoch : 0.0;
      out->elem[0] = ok ? q.a : 0.0; out->elem[1] = ok ? q.h : 0.0; out->elem[2] = ok ? q.k : 0.0;
      out->elem[3] = ok ? q.p : 0.0; out->elem[4] = ok ? q.q : 0.0; out->elem[5] = ok ? q.lambda : 0.0;
      out->normalised_rms = ok ? last_rms : 0.0;
      out->num_measurements = ok ? last_num_meas : 0u;
      // rescale_covariance (least_square.rs:341-362)
      double mu2 = 1.0;
      if (ok) {
        double mu = 1.0;
        if ((unsigned)C.n_free < last_num_meas) {
          const double factor = sqrt((double)last_num_meas / (double)(last_num_meas - (unsigned)C.n_free));
          mu = last_rms > 1.0 ? last_rms * factor : factor;
        }
        mu2 = mu * mu;
      }
#pragma unroll
      for (int i = 0; i < 36; ++i)
        out->covariance[i] = ok ? A.cov[((unsigned long long)(cur ^ 1u) * 36ull + i) * A.n_traj + t] * mu2 : 0.0;
      out->n_rejected = n_rejected;
    }
    t = kNone;
  };

  // start of one pass of the outer loop (diff_cor.rs:335-392): false when it has no Newton iteration to run
  auto begin_outer = [&]() {
    outer_started += 1;
    prev_rms = kF64Max;
    stagnation = 0;
    inner = 0;
    phase = kIterate;
    return C.max_newton_iterations > 0;
  };
  // after the inner loop: true when the outlier pass follows, false when the trajectory is finished
  auto inner_ended = [&](bool inner_converged) {
    if (!C.enable_outlier_rejection) return false;
    if (outer == 0 && last_rms < C.convergence_before_rejection_threshold) return false;
    return inner_converged;
  };

  for (;;) {
    // ---- idle groups pull their next trajectory -----------------------------------------------------------
    for (unsigned tries = 0; tries < 4 && t == kNone && !exhausted; ++tries) {
      unsigned long long g = 0;
      if (sub == 0) g = atomicAdd(A.queue, 1ull);
      g = __shfl(g, (int)(threadIdx.x - sub));
      if (g >= A.n_traj) { exhausted = true; break; }
      DCSTAT(0);  // refill
      t = g;
      base = A.traj_offset[t];
      n = (unsigned)(A.traj_offset[t + 1] - base);
      status = 0; dc_error = 0; conv_error = 0;
      total_iters = 0; outer_started = 0; last_num_meas = 0;
      last_rms = kF64Max;
      cur = 0; outer = 0;
      // ObsFitData::new (obs_fit_data.rs:103-116): zero residuals, all Active — kept implicit until the
      // first accepted iteration / first outlier pass writes the scratch
      have_accepted = false; sel_written = false;
      q = Equinoctial{}; q_in = Equinoctial{};
      const GaussResultRec seed = seed_of(t);
      if (seed.status != 0) {
        status = 2;  // OUTFIT_LSQ_IOD_ERROR
      } else {
        Elements el;
        el.kind = seed.kind;
        el.epoch = seed.epoch;
#pragma unroll
        for (int k = 0; k < 6; ++k) el.e[k] = seed.elem[k];
        conv_error = to_equinoctial<kFn>(el, q);
        if (conv_error) status = 3;  // OUTFIT_LSQ_CONVERSION_ERROR
      }
      if (status == 0) {
#pragma unroll
        for (int i = 0; i < 36; ++i) A.cov[(36ull + i) * A.n_traj + t] = 0.0;  // accepted half (cur ^ 1 = 1)
        if (!begin_outer()) finish();
      } else {
        finish();
      }
    }
    if (__ballot(t != kNone) == 0) break;  // queue drained and every group of the wave idle
    DCSTAT(1);  // trip, all lanes of the wave that are still in the kernel
    if (t == kNone) continue;
    DCSTAT(2);  // trip, lanes holding a trajectory

    bool finished = false;
    if (phase == kIterate) {
      DCSTAT(3);  // Newton iteration
      total_iters += 1;
      // ---- single_iteration (single_iteration.rs:158-328) ----------------------------------------------
      const PartialsFrame P = partials_frame<kFn>(q);
#pragma unroll
      for (int i = 0; i < 43; ++i) sN[i][lane] = 0.0;
      unsigned active = 0;
      double* rc_ = A.res + (unsigned long long)cur * RB;
      const double* ra_ = A.res + (unsigned long long)(cur ^ 1u) * RB;
#pragma unroll 1
      for (unsigned i0 = 0; i0 < n; i0 += kG) {
        const unsigned i = i0 + sub;
        const unsigned long long o = base + i;
        bool eq_active = false;
        ObsPartials ob;
        double rra = 0.0, rdec = 0.0, wa = 0.0, wd = 0.0;
        DCSTAT(4);  // observation pass
        if (i < n) {
          const unsigned char s = sel_written ? A.sel[o] : (unsigned char)0;
          if (s == 0) {
            DCSTAT(5);  // obs_and_partials
            const Vec3 op = {A.obs_pos[3 * o], A.obs_pos[3 * o + 1], A.obs_pos[3 * o + 2]};
            if (obs_and_partials<kFn>(q, P, A.mjd[o], op, ob)) {
              eq_active = true;
              const double sra = A.sra[o], sdec = A.sdec[o];
              rra = angular_diff(A.ra[o] - 0.0, ob.ra);
              rdec = (A.dec[o] - 0.0) - ob.dec;
              const double ta = rra / sra, td = rdec / sdec;
              const double chi = sqrt(ta * ta + td * td);
              wa = 1.0 / (sra * sra);
              wd = 1.0 / (sdec * sdec);
              active += 1;
              rc_[o] = rra;
              rc_[A.n_obs + o] = rdec;
              rc_[2 * A.n_obs + o] = chi;
            }
          }
          if (!eq_active) {  // fit_data.clone(): the accepted residuals carry over
            rc_[o] = have_accepted ? ra_[o] : 0.0;
            rc_[A.n_obs + o] = have_accepted ? ra_[A.n_obs + o] : 0.0;
            rc_[2 * A.n_obs + o] = have_accepted ? ra_[2 * A.n_obs + o] : 0.0;
          }
        }
        // ordered accumulation: observation i0 first, then i0 + 1, ... (one lane of the group at a time)
#pragma unroll 1
        for (unsigned turn = 0; turn < kG; ++turn) {
          if (turn == sub && eq_active) {
            constexpr double wc = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
              for (int k = 0; k < 6; ++k)
                lds_add(&sN[6 * j + k][lane], ob.ga[j] * wa * ob.ga[k] + ob.gd[j] * wd * ob.gd[k] +
                                                  wc * (ob.gd[j] * ob.ga[k] + ob.ga[j] * ob.gd[k]));
              lds_add(&sN[36 + j][lane], (ob.ga[j] * wa + ob.gd[j] * wc) * rra + (ob.ga[j] * wc + ob.gd[j] * wd) * rdec);
            }
            lds_add(&sN[42][lane], wa * rra * rra + wd * rdec * rdec + 2.0 * wc * rra * rdec);
          }
        }
      }
      active = group_sum<kG>(active);
      DCSTAT(6);  // solve
      const double wss = sN[42][lane];
      // ---- solve_weighted_least_squares (least_square.rs:211-303) ---------------------------------------
      const unsigned num_meas = active * 2;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (!C.free_el[j]) {
#pragma unroll
          for (int k = 0; k < 6; ++k) { sN[6 * j + k][lane] = 0.0; sN[6 * k + j][lane] = 0.0; }
          sN[6 * j + j][lane] = 1.0;
          sN[36 + j][lane] = 0.0;
        }
      }
      // the inverse goes column by column into the staging half of the covariance scratch
      double* cov_new = A.cov + (unsigned long long)cur * 36ull * A.n_traj + t;
      bool inv_ok = cholesky_inverse6_lds(&sN[0][lane], kSlots, cov_new, A.n_traj);
      if (!inv_ok) DCSTAT(7);  // QR fallback
      if (!inv_ok)  // every lane of the group works on its own scratch column
        inv_ok = qr_inverse6(A.qr_work + t * kG + sub, A.n_traj * kG, &sN[0][lane], kSlots, cov_new, A.n_traj);
      if (!inv_ok) {
        dc_error = 1;  // DifferentialCorrectionFailed
        finished = true;
      } else {
        double dx[6], cdx[6], rhs[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) rhs[i] = sN[36 + i][lane];
#pragma unroll
        for (int r = 0; r < 6; ++r) {  // covariance * rhs (gemv6 order) from the staged inverse
          double acc = cov_new[(unsigned long long)(6 * r) * A.n_traj] * rhs[0];
#pragma unroll
          for (int j = 1; j < 6; ++j) acc = cov_new[(unsigned long long)(6 * r + j) * A.n_traj] * rhs[j] + acc;
          dx[r] = acc;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (!C.free_el[j]) dx[j] = 0.0;
        const double new_rms = num_meas > 0 ? sqrt(wss / (double)num_meas) : 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {  // normal_matrix * dx
          double acc = sN[6 * r][lane] * dx[0];
#pragma unroll
          for (int j = 1; j < 6; ++j) acc = sN[6 * r + j][lane] * dx[j] + acc;
          cdx[r] = acc;
        }
        const double cnorm = sqrt(dot6(dx, cdx));
        Equinoctial corrected = q;
        if (C.free_el[0]) corrected.a = q.a + dx[0];
        if (C.free_el[1]) corrected.h = q.h + dx[1];
        if (C.free_el[2]) corrected.k = q.k + dx[2];
        if (C.free_el[3]) corrected.p = q.p + dx[3];
        if (C.free_el[4]) corrected.q = q.q + dx[4];
        if (C.free_el[5]) corrected.lambda = q.lambda + dx[5];
        // ---- run_differential_correction bookkeeping (diff_cor.rs:335-392) -----------------------------
        bool inner_over = false, inner_converged = false;
        if (is_bizarre(corrected, C)) {
          dc_error = 2;
          finished = true;
        } else if (prev_rms < kF64Max && new_rms / prev_rms >= C.rms_divergence_ratio) {
          dc_error = 3;
          finished = true;
        } else {
          const bool stagnated = prev_rms < kF64Max && new_rms / prev_rms >= C.rms_stagnation_ratio;
          if (stagnated) stagnation += 1; else stagnation = 0;
          if (stagnated && stagnation >= C.max_stagnation_iterations) {
            inner_over = true;  // leaves the inner loop without accepting this iteration
          } else {
            last_rms = new_rms;
            last_num_meas = num_meas;
            q_in = q;
            have_accepted = true;
            q = corrected;
            cur ^= 1u;  // the buffer just written becomes the accepted one
            prev_rms = new_rms;
            inner += 1;
            if (cnorm < C.convergence_threshold) { inner_converged = true; inner_over = true; }
            else if (inner >= C.max_newton_iterations) inner_over = true;
          }
        }
        if (inner_over) {
          if (inner_ended(inner_converged)) phase = kOutliers;
          else finished = true;
        }
      }
    } else {
      // ---- update_observation_selection (outlier_rejection.rs:132-235) ---------------------------------
      DCSTAT(8);  // outlier pass
      double cov[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) cov[i] = A.cov[((unsigned long long)(cur ^ 1u) * 36ull + i) * A.n_traj + t];
      const double* ra_ = A.res + (unsigned long long)(cur ^ 1u) * RB;
      const PartialsFrame Pin = partials_frame<kFn>(q_in);
      unsigned changes = 0;
#pragma unroll 1
      for (unsigned i = sub; i < n; i += kG) {
        const unsigned long long o = base + i;
        const unsigned char s = sel_written ? A.sel[o] : (unsigned char)0;
        if (!sel_written) A.sel[o] = 0;
        if (s == 2) continue;  // ForcedOut
        // last_equations[i] (diff_cor.rs:377): zero partials and unit weights unless the observation was
        // active and its propagation succeeded in the accepted iteration
        bool eq_active = false;
        ObsPartials ob;
        if (s == 0) {
          const Vec3 op = {A.obs_pos[3 * o], A.obs_pos[3 * o + 1], A.obs_pos[3 * o + 2]};
          eq_active = obs_and_partials<kFn>(q_in, Pin, A.mjd[o], op, ob);
        }
        double ga[6], gd[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          ga[j] = eq_active ? ob.ga[j] : 0.0;
          gd[j] = eq_active ? ob.gd[j] : 0.0;
        }
        const double sra = A.sra[o], sdec = A.sdec[o];
        const double wa = eq_active ? 1.0 / (sra * sra) : 1.0, wd = eq_active ? 1.0 / (sdec * sdec) : 1.0;
        const double var_ra = sra * sra, var_dec = sdec * sdec;
        const double cov_cross = -sra * sdec * 0.0 / (wa * wd);
        double gga[6], ggd[6];
        gemv6(cov, ga, gga);
        gemv6(cov, gd, ggd);
        const double paa = dot6(ga, gga), pdd = dot6(gd, ggd), pad = dot6(ga, ggd);
        const double p00 = var_ra - paa, p01 = cov_cross - pad, p10 = cov_cross - pad, p11 = var_dec - pdd;
        const double det = p00 * p11 - p01 * p01;
        const double scale = fmax(fabs(p00), fabs(p11));
        const double thr = kEps * scale * scale;
        if (fabs(det) < thr || scale == 0.0) continue;
        const double i00 = p11 / det, i01 = -p01 / det, i10 = -p10 / det, i11 = p00 / det;
        const double r0 = ra_[o], r1 = ra_[A.n_obs + o];
        const double v0 = i00 * r0 + i01 * r1, v1 = i10 * r0 + i11 * r1;
        const double chi2 = r0 * v0 + r1 * v1;
        if (s == 0 && chi2 > C.chi2_reject) { A.sel[o] = 1; changes += 1; }
        else if (s == 1 && chi2 <= C.chi2_recover) { A.sel[o] = 0; changes += 1; }
      }
      sel_written = true;
      changes = group_sum<kG>(changes);
      outer += 1;
      if (changes == 0 || outer > C.max_outlier_rejection_passes) finished = true;
      else if (!begin_outer()) finished = true;
    }
    if (finished) { DCSTAT(9); finish(); }
  }
}

hipError_t launch_diffcor(const DiffCorArgs& a, unsigned group, unsigned n_cu, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
  // persistent waves: one per SIMD (the kernel's register budget allows no more), never more than there is work
  const unsigned slots = 64 / group;
  const unsigned long long need = (a.n_traj + slots - 1) / slots;
  const unsigned grid = (unsigned)std::min<unsigned long long>(need, 4ull * (n_cu ? n_cu : 256u));
  hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  const bool glibc = a.math_mode == kMathGlibc;
  switch (group) {
    case 1: if (glibc) hipLaunchKernelGGL((k_diffcor<1, kFnGlibc>), dim3(grid), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((k_diffcor<1, kFnCr>), dim3(grid), dim3(64), 0, s, a); break;
    case 2: if (glibc) hipLaunchKernelGGL((k_diffcor<2, kFnGlibc>), dim3(grid), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((k_diffcor<2, kFnCr>), dim3(grid), dim3(64), 0, s, a); break;
    case 4: if (glibc) hipLaunchKernelGGL((k_diffcor<4, kFnGlibc>), dim3(grid), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((k_diffcor<4, kFnCr>), dim3(grid), dim3(64), 0, s, a); break;
    case 8: if (glibc) hipLaunchKernelGGL((k_diffcor<8, kFnGlibc>), dim3(grid), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((k_diffcor<8, kFnCr>), dim3(grid), dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
#ifdef OUTFIT_DC_STATS
  {
    unsigned long long k[32] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(k, HIP_SYMBOL(g_dc_stat), sizeof k);
    static const unsigned long long zero[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dc_stat), zero, sizeof zero);
    const char* names[10] = {"refill", "trip (lanes in the kernel)", "trip (lanes with a trajectory)", "Newton iteration",
                             "observation pass", "obs_and_partials", "solve", "QR fallback", "outlier pass", "finish"};
    for (int i = 0; i < 10; ++i)
      fprintf(stderr, "[dc stats] %-32s wave passes %10llu  lanes/pass %5.2f\n", names[i], k[2 * i],
              k[2 * i] ? (double)k[2 * i + 1] / k[2 * i] : 0.0);
  }
#endif
  return hipGetLastError();
}

}  // namespace outfit
