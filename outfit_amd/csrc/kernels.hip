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
#include <cstddef>
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include "kernels.h"
#include "device/ephem.hpp"
#include "device/earthrot.hpp"
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
  if (wave_form_owns(n_obs, a.P.max_keep, a.wave_min_obs)) return;  // k_triplet_select_wave's
  // heap slot s of lane lt: scratch[(s * n_lanes_padded) + lt]
  HeapView h{a.heap_w + lt, a.heap_id + lt, a.heap_stride};
  const unsigned n = select_triplets(a.mjd + base, n_obs, a.P, h);
  a.n_triplets[lt] = n;
  for (unsigned s = 0; s < n; ++s) a.triplets[lt * a.P.k_cap + s] = h.I(s);
}

// Same selection with the heap in LDS ([slot][lane]: conflict-free, ~100-cycle latency instead
// of a round trip to L2 for every sift step). Chosen by the host when k_cap * 12 B * 64 lanes
// fits and no reduced trajectory exceeds 1024 observations (10-bit indices).
__global__ void __launch_bounds__(64)
k_triplet_select_lds(TripletArgs a) {
  extern __shared__ double lds_heap[];  // [k_cap][64] weights, then [k_cap][64] 32-bit ids
  const unsigned long long lt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lt >= a.n_traj) return;
  const unsigned long long traj = a.traj0 + lt;
  const unsigned long long base = a.traj_offset[traj];
  const unsigned long long n_obs = a.traj_offset[traj + 1] - base;
  if (wave_form_owns(n_obs, a.P.max_keep, a.wave_min_obs)) return;  // k_triplet_select_wave's
  LdsHeapView h{lds_heap + threadIdx.x, reinterpret_cast<unsigned*>(lds_heap + (size_t)a.P.k_cap * 64) + threadIdx.x,
                64u};
  const unsigned n = select_triplets(a.mjd + base, n_obs, a.P, h);
  a.n_triplets[lt] = n;
  for (unsigned s = 0; s < n; ++s) a.triplets[lt * a.P.k_cap + s] = h.I(s);
}

// One wave per LONG trajectory (device/triplets.hpp, select_triplets_wave): 64 lanes score a row of the generator at
// once, only the survivors of the ballot go through the literal heap operations. LDS: reduced epochs [n], then the
// heap's weights [k_cap] and 64-bit ids [k_cap].
__global__ void __launch_bounds__(64)
k_triplet_select_wave(TripletArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_wave[];
  const unsigned long long traj = a.wave_list[blockIdx.x];
  const unsigned long long lt = traj - a.traj0;
  const unsigned long long base = a.traj_offset[traj];
  const unsigned long long n_obs = a.traj_offset[traj + 1] - base;
  const unsigned n = (unsigned)reduced_len(n_obs, a.P.max_keep);
  double* ep = lds_wave;
  WaveHeap h{reinterpret_cast<double2*>(lds_wave + ((n + 1u) & ~1u))};   // 16-byte aligned behind the epochs
  const unsigned len = select_triplets_wave(a.mjd + base, n_obs, a.P, ep, h, a.triplets + lt * a.P.k_cap);
  if (threadIdx.x == 0) a.n_triplets[lt] = len;
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
  if (live && (a.parts & kPrepassUnit)) {
    // Column of unit_matrix (gauss.rs:488-502) for the noise-free observation, with the correctly-rounded
    // functions: every noise-free candidate that uses this observation (~15 triplets each) reads the
    // three direction cosines instead of re-evaluating two double-double sincos
    double sr, cr, sd, cd;
    sincos_mode(a.math_mode, a.ra[o], sr, cr);
    sincos_mode(a.math_mode, a.dec[o], sd, cd);
    a.unit[o] = cr * cd;
    a.unit[a.n_obs + o] = sr * cd;
    a.unit[2 * a.n_obs + o] = sd;
  }
  if (!(a.parts & kPrepassObserver)) return;
  // cos(dec_obs) of ephemeris_error (observation_ephemeris.rs:412) depends on the observation only. It is an ISOLATED
  // cosine in the reference: the standalone libm entry point (glm::cos1 in the glibc-faithful mode; glibcmath.hpp)
  double cd = 0.0;
  if (live) {
    cd = a.math_mode == kMathGlibc ? glm::cos1(a.dec[o]) : cos(a.dec[o]);
    a.cos_dec[o] = cd;
    // the screen's record (a zero uncertainty gives Inf / NaN weights: a non-finite sum, i.e. the exact loop)
    double* rec = a.obs_rec + (size_t)kObsRecDoubles * o;
    pack_screen_obs(rec, a.mjd[o], a.ra[o], a.dec[o], cd / a.sra[o], 1.0 / a.sdec[o]);
    if (!a.has_table) { rec[8] = a.helio[3 * o]; rec[9] = a.helio[3 * o + 1]; rec[10] = a.helio[3 * o + 2]; }
  }
  if (!a.has_table) return;
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
  double* rec = a.obs_rec + (size_t)kObsRecDoubles * o;
  rec[8] = out.x; rec[9] = out.y; rec[10] = out.z;
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
    double zd[6];
    const double* z = zd;
    if (I.noise_device) {
      const unsigned long long key = I.traj_key ? I.traj_key[traj] : I.traj_key_base + traj;
      noise::normals6(I.noise_seed, key, c.t, c.r, zd);
    } else {
      z = I.noise_z + ((traj * I.k_cap + c.t) * (unsigned long long)(I.n_real - 1) + (c.r - 1)) * 6;
    }
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
// K2a': first accepted root and the cold-start values of its two universal-Kepler
// solves, one lane per candidate (dense: every lane runs the expensive correctly-rounded
// cold start at the same time instead of inside the persistent kernel's divergent trips).
// The first iteration of pos_and_vel_correction calls solve_kepuni_with_guess with
// psi_guess = None, i.e. run_newton(prelim_kepuni(params)) (newton_solver.rs:151-161);
// the value prelim_kepuni(params) is what is stored here.
// ------------------------------------------------------------------------------
__device__ __forceinline__ void append_slow(const CandidateScratch& W, unsigned long long cid) {
  const unsigned long long at = atomicAdd(W.slow_count, 1ull);
  W.slow_list[at] = cid;
}

struct CandidateGeometry {
  Mat3 S, Sinv, obs;
  double t[3];
  GaussGeometry q;
};
__device__ __forceinline__ CandidateGeometry load_geometry(const CandidateIndex& I, const CandidateScratch& W,
                                                           unsigned long long cid) {
  CandidateGeometry g;
  const CandidateRef c = decode_candidate(I, cid);
  gather_times_observer(I, c, g.t, g.obs);
  if (c.r == 0) {
    // noise-free realisation: the direction matrix is three columns of K1's per-observation table (48 MB for 2x10^6
    // observations: served by L2 / Infinity Cache) — it is not stored per candidate (72 B x 10^7 per step)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = c.base + c.idx[k];
      g.S.m[0][k] = I.unit[o];
      g.S.m[1][k] = I.unit[I.n_obs_total + o];
      g.S.m[2][k] = I.unit[2 * I.n_obs_total + o];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) g.S.m[r][k] = W.S[(unsigned long long)(3 * r + k) * I.n_cand + cid];
  }
  (void)try_inverse(g.S, g.Sinv);  // non-singular: K2a checked it
  g.q = gauss_times(g.t);
  return g;
}

// ------------------------------------------------------------------------------
// K2a: direction matrix, 8th-degree polynomial, Aberth roots, root filter
// ------------------------------------------------------------------------------
// S (direction matrix), epochs and observer matrix of a candidate as K2a takes them: the noise-free columns from K1's
// table, a noisy realisation's from its own angles
struct CandidateGeometryInputs {
  Mat3 S, obs;
  double t[3];
};
template <class Args>
__device__ __forceinline__ CandidateGeometryInputs gather_inputs(const Args& a, const CandidateRef& c) {
  CandidateGeometryInputs g;
  if (c.r == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = c.base + c.idx[k];
      g.S.m[0][k] = a.I.unit[o];
      g.S.m[1][k] = a.I.unit[a.I.n_obs_total + o];
      g.S.m[2][k] = a.I.unit[2 * a.I.n_obs_total + o];
    }
  } else {
    double ra[3], dec[3];
    gather_angles(a.I, c, ra, dec);
    g.S = unit_matrix(ra, dec, a.P.math_mode);
  }
  gather_times_observer(a.I, c, g.t, g.obs);
  return g;
}

#ifndef OUTFIT_ROOTS_WAVES
#define OUTFIT_ROOTS_WAVES 4
#endif
#ifndef OUTFIT_ROOTS_BLOCK
#define OUTFIT_ROOTS_BLOCK 256
#endif
__global__ void __launch_bounds__(OUTFIT_ROOTS_BLOCK, OUTFIT_ROOTS_WAVES)
k_gauss_roots(RootsArgs a) {
  const unsigned long long lid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lid >= a.cand_count) return;
  const unsigned long long cid = a.cand_begin + lid;
  const CandidateRef c = decode_candidate(a.I, cid);
  if (!c.live) { a.W.flags[cid] = kFlagEmpty; return; }
  const unsigned long long n = a.I.n_cand;
  K2ASTAT(0);

  double ra[3], dec[3], t[3];
  Mat3 obs, Sinv;
  Mat3 S;
  if (c.r == 0) {  // noise-free realisation: the columns were evaluated once per observation by K1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = c.base + c.idx[k];
      S.m[0][k] = a.I.unit[o];
      S.m[1][k] = a.I.unit[a.I.n_obs_total + o];
      S.m[2][k] = a.I.unit[2 * a.I.n_obs_total + o];
    }
  } else {
    gather_angles(a.I, c, ra, dec);
    S = unit_matrix(ra, dec, a.P.math_mode);
  }
  if (c.r != 0) {  // noisy realisation: its direction matrix costs six correctly-rounded sincos — kept for K2b
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) a.W.S[(unsigned long long)(3 * r + k) * n + cid] = S.m[r][k];
  }
  if (!try_inverse(S, Sinv)) { a.W.flags[cid] = (unsigned)kCauseSingular; return; }
  gather_times_observer(a.I, c, t, obs);
  const GaussGeometry q = gauss_times(t);
  double c6, c3, c0;
  coeff_eight_poly(obs, S, Sinv, q, c6, c3, c0);
  if (descartes_upper_bound(c0, c3, c6) == 0) { a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }

  Cx z[8];
  int ab = aberth_deg8<true>(c0, c3, c6, a.P.aberth_max_iter, a.P.aberth_eps, z);
  if (ab == 3) ab = aberth_deg8<false>(c0, c3, c6, a.P.aberth_max_iter, a.P.aberth_eps, z);
  if (ab == 2) {
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
  if (nf == 0) { a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }
  K2ASTAT(2);
  // ---- K2a' fused: first accepted root and the cold-start values of its two universal-Kepler solves. The first
  // iteration of pos_and_vel_correction calls solve_kepuni_with_guess with psi_guess = None, i.e.
  // run_newton(prelim_kepuni(params)) (newton_solver.rs:151-161); prelim_kepuni(params) is what is stored.
  // The geometry (S, S^-1, observer matrix, times) is gathered AGAIN here instead of being kept alive across the
  // Aberth sweeps: held in registers it cost 70 spilled VGPRs at four waves per SIMD — 336 B of scratch per lane, 6 GB
  // of HBM traffic per step (measured) — while the gathers hit L2. The filtered roots are read back from this lane's
  // own stores (dynamic indexing of the iterates would push them to scratch).
  const CandidateGeometry g = load_geometry(a.I, a.W, cid);
  Mat3 pos;
  Vec3 vel;
  double e_pre;
  VelCommon vc;
  unsigned slot = 0;
  bool accepted = false;
#pragma unroll 1
  for (; slot < nf; ++slot) {
    const double r2 = a.W.roots[(unsigned long long)slot * n + cid];
    if (accept_root(g.obs, g.S, g.Sinv, g.t[1], g.q, a.P, r2, pos, vel, e_pre, vc)) { accepted = true; break; }
  }
  if (!accepted) { a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }  // no solution at all
  const double dt01 = g.t[0] - g.t[1], dt21 = g.t[2] - g.t[1];
  const bool slow = !vc.ok || fabs(dt01) <= kEps || fabs(dt21) <= kEps;
  K2ASTAT(3);
  if (!slow) {
    double psi01, psi21;  // the two cold starts share everything but dt (kepler.hpp prelim_kepuni_pair)
    prelim_kepuni_pair(velocity_problem(vc, dt01, a.P.kepler_eps, a.P.math_mode), dt01, dt21, psi01, psi21);
    a.W.psi0[cid] = psi01;
    a.W.psi0[n + cid] = psi21;
  }
  a.W.flags[cid] = kFlagOk | (nf << kFlagRootsShift) | (slot << kFlagFirstShift) | (slow ? kFlagSlow : 0u);
  if (slow) append_slow(a.W, cid);
}

// K2a, root-per-lane form (small batches): eight consecutive lanes per candidate, one Aberth iterate each
// (gauss.hpp aberth_deg8_lanes). Everything else of k_gauss_roots is evaluated by the group's lanes redundantly (same
// inputs, same values), the root filter compacts through a ballot in solver order, lanes 0 and 1 of a group take one
// cold start each. Writes exactly what k_gauss_roots writes.
__global__ void __launch_bounds__(256)
k_gauss_roots_lanes(RootsArgs a) {
  const unsigned long long lid = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  if (lid >= a.cand_count) return;  // whole groups leave together
  const unsigned lane = __lane_id(), r = lane & 7u, base = lane & ~7u;
  const unsigned long long cid = a.cand_begin + lid;
  const CandidateRef c = decode_candidate(a.I, cid);
  if (!c.live) { if (r == 0) a.W.flags[cid] = kFlagEmpty; return; }
  const unsigned long long n = a.I.n_cand;
  const CandidateGeometryInputs in = gather_inputs(a, c);
  Mat3 S = in.S, Sinv;
  if (c.r != 0 && r == 0) {
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int k = 0; k < 3; ++k) a.W.S[(unsigned long long)(3 * rr + k) * n + cid] = S.m[rr][k];
  }
  if (!try_inverse(S, Sinv)) { if (r == 0) a.W.flags[cid] = (unsigned)kCauseSingular; return; }
  const GaussGeometry q = gauss_times(in.t);
  double c6, c3, c0;
  coeff_eight_poly(in.obs, S, Sinv, q, c6, c3, c0);
  if (descartes_upper_bound(c0, c3, c6) == 0) { if (r == 0) a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }
  Cx z;
  const int ab = aberth_deg8_lanes(c0, c3, c6, a.P.aberth_max_iter, a.P.aberth_eps, z);
  if (ab == 2) { if (r == 0) a.W.flags[cid] = (unsigned)kCauseRootFinding; return; }
  // visit_real_positive_roots + the r2 window (gauss.rs:964-986,1148): solver order = lane order within the group
  const bool pass = z.re > 0.0 && fabs(z.im) < a.P.root_imag_eps && z.re >= a.P.r2_min_au && z.re <= a.P.r2_max_au;
  const unsigned bits = (unsigned)(__ballot(pass) >> base) & 0xFFu;
  const unsigned nf = (unsigned)__popc(bits);
  if (pass) a.W.roots[(unsigned long long)__popc(bits & ((1u << r) - 1u)) * n + cid] = z.re;
  if (nf == 0) { if (r == 0) a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }
  // first accepted root, in solver order (the roots come from their lanes, not back from memory)
  Mat3 pos;
  Vec3 vel;
  double e_pre;
  VelCommon vc;
  unsigned slot = 0;
  bool accepted = false;
#pragma unroll 1
  for (unsigned k = 0; k < 8; ++k) {
    const double r2 = __shfl(z.re, (int)(base + k));
    if (!((bits >> k) & 1u)) continue;
    if (accept_root(in.obs, S, Sinv, in.t[1], q, a.P, r2, pos, vel, e_pre, vc)) { accepted = true; break; }
    ++slot;
  }
  if (!accepted) { if (r == 0) a.W.flags[cid] = (unsigned)kCauseNoRoots; return; }
  const double dt01 = in.t[0] - in.t[1], dt21 = in.t[2] - in.t[1];
  const bool slow = !vc.ok || fabs(dt01) <= kEps || fabs(dt21) <= kEps;
  if (!slow && r < 2u)
    a.W.psi0[(unsigned long long)r * n + cid] = prelim_kepuni(velocity_problem(vc, r ? dt21 : dt01, a.P.kepler_eps, a.P.math_mode));
  if (r == 0) {
    a.W.flags[cid] = kFlagOk | (nf << kFlagRootsShift) | (slot << kFlagFirstShift) | (slow ? kFlagSlow : 0u);
    if (slow) append_slow(a.W, cid);
  }
}

// ------------------------------------------------------------------------------
// K2b (fast): persistent lanes, regular candidates only. A lane owns one candidate at a
// time, restarts from the first accepted root (recomputed: cheaper than storing it) with the
// precomputed cold-start values, and runs warm iterations until the loop ends (Some). Any
// irregular event hands the candidate to the general kernel through the slow list.
// ------------------------------------------------------------------------------
#ifndef OUTFIT_K2B_THRESHOLD
#define OUTFIT_K2B_THRESHOLD 24  // measured (round 3, 100 000 objects): 8 -> 7.48 ms, 16 -> 6.64, 24 -> 6.50, 32 -> 6.57
#endif
constexpr unsigned kRefillThreshold = OUTFIT_K2B_THRESHOLD;  // idle lanes per wave before the queue is touched
#ifndef OUTFIT_PAIR_MAX_CANDIDATES
#define OUTFIT_PAIR_MAX_CANDIDATES (3u << 17)
#endif
// up to this many candidates in a chunk the candidate kernels run in their latency forms (lane pairs, root per lane,
// 16-lane re-evaluation; 0 = never). Measured (tools/perf_form_threshold.py, step time latency / throughput forms):
// 25 000 candidates 0.78 / 1.16 ms, 100 000: 1.05 / 1.45, 200 000: 1.35 / 1.65, 300 000: 1.71 / 1.80, 500 000: 2.15 / 2.19,
// 1 000 000: 3.47 / 2.79 — beyond ~400 000 the lanes are better spent on candidates than on parts of candidates
constexpr unsigned long long kPairMaxCandidates = OUTFIT_PAIR_MAX_CANDIDATES;
// OUTFIT_PAIR_MODE=0 / 1 forces the form whatever the batch size (testing aid: the two forms must give the same bytes)
static bool use_pair_form(unsigned long long n_cand) {
  if (const char* e = std::getenv("OUTFIT_PAIR_MODE")) {
    if (e[0] == '0') return false;
    if (e[0] == '1') return true;
  }
  return n_cand <= kPairMaxCandidates;
}

// The kernel's own argument block, re-read from the kernarg segment at the point of use. The persistent kernels hold
// ~35 pointers and sizes (CandidateIndex, CandidateScratch) that only their REFILL path needs; kept live across the
// iteration loop they overflow the scalar register file and come back through v_readlane in the hot loop (K2b: 97
// spilled SGPRs, 31 of the 424 instructions of its iteration core). The empty asm makes the pointer opaque at this
// point, so the loads below it cannot be hoisted out of the branch they are in: the refill path pays ~10 scalar loads
// (cached, once per ~7 trips) and the loop keeps only what it uses.
template <class T>
__device__ __forceinline__ T kernarg_field(unsigned offset) {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  T out;
  __builtin_memcpy(&out, p + offset, sizeof(T));
  return out;
}
#ifndef OUTFIT_KERNARG_RELOAD       // the refill path's pointers (CandidateIndex, CandidateScratch) re-read at the refill
#define OUTFIT_KERNARG_RELOAD 1
#endif
// Measured (round 4, 100 000 objects, three runs): spilled SGPRs 97 -> 48 with the refill reload, -> 26 with the lazy
// parameters as well; k_gauss_correct_fast 6.51-6.60 ms without either, 6.51 with the reload, 6.65-6.71 with both — a
// v_readlane costs no more than the s_load + wait that replaces it. The reload is kept (not slower, half the spills), the
// lazy parameters are off.
// Measured and dropped (round 4): the fast kernel's queue in triplet-rank-major order (a wave's lanes = the same rank of 64
// trajectories) — 6.46 -> 8.60 ms: round passes per trip 4.94 -> 6.05 at 34.6 instead of 47.8 lanes. How many Newton rounds
// and series terms a solve takes is a property of the TRAJECTORY (its orbit), not of the triplet's gaps: the
// trajectory-major order already is the one that puts like with like.
#ifndef OUTFIT_K2B_RANK_MAJOR
#define OUTFIT_K2B_RANK_MAJOR 0
#endif
#ifndef OUTFIT_KERNARG_LAZY_PARAMS  // GaussParams fields and the result pointers re-read where they are used
#define OUTFIT_KERNARG_LAZY_PARAMS 0
#endif
// GaussParams of a kernel whose argument struct holds it at byte `kOff`, every field read from the kernarg segment where
// it is used (an s_load on the scalar memory path, not a v_readlane on the VALU): same field names as GaussParams,
// so the device functions templated on the parameter type read `P.max_ecc` either way.
template <unsigned kOff>
struct KernargGaussParams {
  template <class T, unsigned kField>
  struct Lazy {
    __device__ __forceinline__ operator T() const { return kernarg_field<T>(kOff + kField); }
  };
#define OUTFIT_LAZY(type, name) Lazy<type, (unsigned)offsetof(GaussParams, name)> name
  OUTFIT_LAZY(double, max_ecc); OUTFIT_LAZY(double, max_perihelion_au); OUTFIT_LAZY(double, min_rho2_au);
  OUTFIT_LAZY(double, aberth_eps); OUTFIT_LAZY(double, kepler_eps); OUTFIT_LAZY(double, r2_min_au);
  OUTFIT_LAZY(double, r2_max_au); OUTFIT_LAZY(double, newton_eps); OUTFIT_LAZY(double, root_imag_eps);
  OUTFIT_LAZY(double, peri_lo); OUTFIT_LAZY(double, peri_hi); OUTFIT_LAZY(double, eps2_lo); OUTFIT_LAZY(double, eps2_hi);
  OUTFIT_LAZY(unsigned, aberth_max_iter); OUTFIT_LAZY(unsigned, max_tested_solutions);
  OUTFIT_LAZY(unsigned, newton_max_it); OUTFIT_LAZY(int, math_mode);
#undef OUTFIT_LAZY
};

// The fast kernel's parameters, split by where an iteration reads them (gauss.hpp tail_params): what the two Kepler
// solves need stays in scalar registers; what the tail needs (thresholds of position_and_epoch, eccentricity_control and
// the convergence test: 16 doubles with the derived bounds) is fetched from the kernarg segment behind the solves.
#ifndef OUTFIT_K2B_SPLIT_PARAMS
#define OUTFIT_K2B_SPLIT_PARAMS 1
#endif
template <unsigned kOff>
struct SplitParams {
  double kepler_eps;
  unsigned newton_max_it;
  int math_mode;
  __device__ __forceinline__ GaussParams tail() const { return kernarg_field<GaussParams>(kOff); }
};

// wave-wide grab of the next queue entries for the lanes in need_mask
__device__ __forceinline__ unsigned long long wave_grab(unsigned long long* queue, unsigned long long need_mask) {
  const unsigned lane = __lane_id();
  const unsigned leader = (unsigned)__ffsll((long long)need_mask) - 1u;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(queue, (unsigned long long)__popcll(need_mask));
  base = __shfl(base, (int)leader);
  return base + (unsigned long long)__popcll(need_mask & ((1ull << lane) - 1ull));
}

// lane-pair form (gauss.hpp, correction_iteration<.., kPair>): lanes 2k and 2k+1 take the SAME queue entry; needy lanes
// always come in pairs (both lanes of a pair finish together), so the pair's index is half the even lane's rank
__device__ __forceinline__ unsigned long long wave_grab_pairs(unsigned long long* queue, unsigned long long need_mask) {
  const unsigned lane = __lane_id();
  const unsigned leader = (unsigned)__ffsll((long long)need_mask) - 1u;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(queue, (unsigned long long)(__popcll(need_mask) / 2));
  base = __shfl(base, (int)leader);
  return base + (unsigned long long)(__popcll(need_mask & ((1ull << lane) - 1ull)) / 2);
}

// kPair: two lanes per candidate, one Lagrange correction each (small batches: the launcher chooses)
template <bool kPair>
__global__ void __launch_bounds__(kFastBlock, kFastWavesPerSimd)
k_gauss_correct_fast(CorrectArgs a) {
  extern __shared__ double lds[];  // [27][kFastBlock]: S, S^-1, observer matrix
  // the block size is the launcher's constant: with a runtime blockDim.x every L(k) paid a scalar offset (27 of
  // them, spilled to VGPR lanes) and a v_add_u32; as immediates they are the ds_read/ds_write offset field
  const unsigned tid = threadIdx.x;
  constexpr unsigned bs = kFastBlock;
  const unsigned long long n = a.I.n_cand;
  auto L = [&](unsigned k) -> double& { return lds[k * bs + tid]; };

  long long cand = -1;
  bool exhausted = false;
  double t_mid = 0.0, dt01 = 0.0, dt21 = 0.0;
  CorrectionState st{};
#ifdef OUTFIT_K2B_STATS  // diagnostic build only (tools/perf_k2b_stats.py): where a wave's time goes
  unsigned long long st_trips = 0, st_refills = 0, st_active = 0, st_needy = 0, st_cyc_refill = 0;
  const unsigned long long st_t0 = clock64();
#endif

  for (;;) {
    const bool needy = (cand < 0) && !exhausted;
    const unsigned long long need_mask = __ballot(needy);
    const unsigned long long active_mask = __ballot(cand >= 0);
    if (need_mask && ((unsigned)__popcll(need_mask) >= kRefillThreshold || active_mask == 0ull)) {
#ifdef OUTFIT_K2B_STATS
      const unsigned long long st_r0 = clock64();
      st_refills += 1; st_needy += (unsigned)__popcll(need_mask);
#endif
      unsigned long long my = kPair ? wave_grab_pairs(a.queue, need_mask) : wave_grab(a.queue, need_mask);
      if (needy) {
        if (my >= n) {
          exhausted = true;
        } else {
#if OUTFIT_K2B_RANK_MAJOR
          // queue order = triplet rank first, trajectory second: the lanes a wave refills together hold the SAME rank of
          // 64 different trajectories — triplets of similar weight, i.e. similar time gaps, hence similar numbers of
          // Newton rounds and series terms — instead of ~2/3 of one trajectory's ranks (gaps from 0.02 d to 20 d)
          {
            const unsigned long long n_t = n / a.I.cand_per_traj;
            my = (my % n_t) * a.I.cand_per_traj + my / n_t;
          }
#endif
#if OUTFIT_KERNARG_RELOAD
          const CandidateIndex I = kernarg_field<CandidateIndex>((unsigned)offsetof(CorrectArgs, I));
          const CandidateScratch W = kernarg_field<CandidateScratch>((unsigned)offsetof(CorrectArgs, W));
#else
          const CandidateIndex& I = a.I;
          const CandidateScratch& W = a.W;
#endif
          const unsigned f = W.flags[my];
          if ((f & kFlagOk) && !(f & (kFlagEmpty | kFlagSlow))) {
            const CandidateGeometry g = load_geometry(I, W, my);
            const unsigned slot = (f >> kFlagFirstShift) & 0xFu;
            double e_pre;
            // same inputs as in K2a' -> same (pos, vel, vc); accepted there, so accepted here
            (void)accept_root(g.obs, g.S, g.Sinv, g.t[1], g.q, a.P, W.roots[(unsigned long long)slot * n + my],
                              st.pos, st.vel, e_pre, st.vc);
            if constexpr (!kPair) prepare_guarded(st);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int k = 0; k < 3; ++k) {
                L(3 * r + k) = g.S.m[r][k];
                L(9 + 3 * r + k) = g.Sinv.m[r][k];
                L(18 + 3 * r + k) = g.obs.m[r][k];
              }
            t_mid = g.t[1];
            dt01 = g.t[0] - g.t[1];
            dt21 = g.t[2] - g.t[1];
            st.epoch = 0.0; st.it = 0;
            st.has01 = true; st.chi01 = W.psi0[my];
            st.has21 = true; st.chi21 = W.psi0[n + my];
            cand = (long long)my;
          }
        }
      }
#ifdef OUTFIT_K2B_STATS
      st_cyc_refill += clock64() - st_r0;
#endif
    }
    if (__ballot(cand >= 0) == 0ull && __ballot(!exhausted) == 0ull) break;
#ifdef OUTFIT_K2B_STATS
    st_trips += 1; st_active += (unsigned)__popcll(__ballot(cand >= 0));
#endif
    if (cand < 0) continue;

    Mat3 S, Sinv, obs;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S.m[r][k] = L(3 * r + k);
        Sinv.m[r][k] = L(9 + 3 * r + k);
        obs.m[r][k] = L(18 + 3 * r + k);
      }
#if OUTFIT_KERNARG_LAZY_PARAMS
    const KernargGaussParams<(unsigned)offsetof(CorrectArgs, P)> P;
#elif OUTFIT_K2B_SPLIT_PARAMS
    const SplitParams<(unsigned)offsetof(CorrectArgs, P)> P{a.P.kepler_eps, a.P.newton_max_it, a.P.math_mode};
#else
    const GaussParams& P = a.P;
#endif
    CorrStep step = correction_iteration<true, true, std::remove_cv_t<std::remove_reference_t<decltype(P)>>, kPair>(
        obs, S, Sinv, t_mid, dt01, dt21, P, st);
    const unsigned long long ucand = (unsigned long long)cand;
    if (a.test_redo_mod && (ucand + st.it) % a.test_redo_mod == 0) step = kCorrRedo;  // test aid, see CorrectArgs
    if (step == kCorrSome || step == kCorrSlow || step == kCorrRedo) {
#if OUTFIT_KERNARG_LAZY_PARAMS
      const CandidateScratch W = kernarg_field<CandidateScratch>((unsigned)offsetof(CorrectArgs, W));
#else
      const CandidateScratch& W = a.W;
#endif
      const bool writer = !kPair || (tid & 1u) == 0u;  // lane-pair form: both lanes hold the same result
      if (!writer) {
      } else if (step == kCorrSome) {
        W.state[0 * n + ucand] = st.pos.m[0][1];
        W.state[1 * n + ucand] = st.pos.m[1][1];
        W.state[2 * n + ucand] = st.pos.m[2][1];
        W.state[3 * n + ucand] = st.vel.x;
        W.state[4 * n + ucand] = st.vel.y;
        W.state[5 * n + ucand] = st.vel.z;
        W.state[6 * n + ucand] = st.epoch;
        W.flags[ucand] = kFlagOk | kFlagCorrected;
      } else if (step == kCorrSlow) {
        // None inside the loop (eccentricity_control rejected an iterate): the root walk continues in the general
        // kernel, which does not replay this root's loop; kFlagSlow also selects the second RMS pass
        W.flags[ucand] |= kFlagSlow | kFlagFirstNone;
        append_slow(W, ucand);
      } else {
        // kCorrRedo: a guarded division met an operand outside its window (gauss.hpp): the general kernel replays the
        // candidate from its first root with plain divisions
        W.flags[ucand] |= kFlagSlow;
        append_slow(W, ucand);
      }
      cand = -1;
    }
  }
#ifdef OUTFIT_K2B_STATS
  if (__lane_id() == 0) {  // a.queue[8..15]: waves, trips, refills, active lanes, needy lanes, cycles in refill, cycles
    atomicAdd(a.queue + 8, 1ull); atomicAdd(a.queue + 9, st_trips); atomicAdd(a.queue + 10, st_refills);
    atomicAdd(a.queue + 11, st_active); atomicAdd(a.queue + 12, st_needy); atomicAdd(a.queue + 13, st_cyc_refill);
    atomicAdd(a.queue + 14, (unsigned long long)(clock64() - st_t0));
  }
#endif
}

// ------------------------------------------------------------------------------
// K2b (general): the complete root walk of the closure at gauss.rs:1147-1197 for the
// candidates on the slow list: accept_root -> correction loop -> first CorrectedOrbit
// wins (prelim_orbit, gauss.rs:1238-1247), a root whose loop returns None contributes a
// PrelimOrbit, the walk stops after max_tested_solutions pushes. Persistent lanes as above.
// ------------------------------------------------------------------------------
// kWaves (waves per SIMD the kernel is compiled for): 2 = 256 VGPRs, the shortest chain — small batches, where this kernel
// is the floor of the run; 3 = 168 VGPRs (+120 spilled): beside K2c, whose waves hold 3 x 154 of a SIMD's 512 registers, a
// 256-register wave practically never finds room (measured: 3.4 ms beside a 3.0 ms K2c for 0.45 ms of work, i.e. K3 waited
// for it), a 168-register one does as soon as one K2c wave retires (0.5 ms, hidden).
template <int kWaves, bool kPair = false>  // kPair: the lane-pair form (gauss.hpp), small batches
__global__ void __launch_bounds__(kCorrectBlock, kWaves)
k_gauss_correct(CorrectArgs a) {
  extern __shared__ double lds[];  // [kCorrectLdsDoubles][kCorrectBlock]
  const unsigned tid = threadIdx.x;
  constexpr unsigned bs = kCorrectBlock;
  const unsigned long long n = a.I.n_cand;
  const unsigned long long n_items = *a.W.slow_count;
  auto L = [&](unsigned k) -> double& { return lds[k * bs + tid]; };

  // A short list is spread over the waves of the grid instead of filling the first ones: lanes of one wave walk
  // their roots in lockstep, so every cold start or long loop of a neighbour lengthens a candidate's own chain —
  // which is what a small batch waits for. `quota` lanes per wave take work (64 when the list is long).
  const unsigned long long n_waves = (unsigned long long)gridDim.x * (bs / 64u);
  const unsigned long long per_wave = (n_items + n_waves - 1) / n_waves;
  constexpr unsigned kLanesPer = kPair ? 2u : 1u;     // lanes per candidate
  constexpr unsigned kSlots = 64u / kLanesPer;        // candidates a wave can hold
  const unsigned quota_c = per_wave >= kSlots ? kSlots : (per_wave < 1ull ? 1u : (unsigned)per_wave);
  const unsigned quota = quota_c * kLanesPer;         // lanes of a wave that take work
  const unsigned refill_at = quota >= 4u * kRefillThreshold ? kRefillThreshold : (quota_c + 3u) / 4u * kLanesPer;

  long long cand = -1;
  bool exhausted = (tid % 64u) >= quota, need_init = false, have_prelim = false, resume = false;
  unsigned slot = 0, n_filt = 0, pushes = 0;
  double t_mid = 0.0, dt01 = 0.0, dt21 = 0.0;
  GaussGeometry q{};
  CorrectionState st{};

  for (;;) {
    const bool needy = (cand < 0) && !exhausted;
    const unsigned long long need_mask = __ballot(needy);
    const unsigned long long active_mask = __ballot(cand >= 0);
    if (need_mask && ((unsigned)__popcll(need_mask) >= refill_at || active_mask == 0ull)) {
      const unsigned long long my = kPair ? wave_grab_pairs(a.queue, need_mask) : wave_grab(a.queue, need_mask);
      if (needy) {
        if (my >= n_items) {
          exhausted = true;
        } else {
          const unsigned long long cid = a.W.slow_list[my];
          const unsigned f = a.W.flags[cid];
          const CandidateGeometry g = load_geometry(a.I, a.W, cid);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              L(3 * r + k) = g.S.m[r][k];
              L(9 + 3 * r + k) = g.Sinv.m[r][k];
              L(18 + 3 * r + k) = g.obs.m[r][k];
            }
          q = g.q;
          t_mid = g.t[1];
          dt01 = g.t[0] - g.t[1];
          dt21 = g.t[2] - g.t[1];
          cand = (long long)cid;
          n_filt = (f >> kFlagRootsShift) & 0xFu;
          // a candidate handed over by the fast kernel resumes at its first accepted root, whose loop is known to
          // have ended in None (the roots before it were not accepted: same inputs, same answer)
          resume = (f & kFlagFirstNone) != 0u;
          slot = resume ? ((f >> kFlagFirstShift) & 0xFu) : 0u;
          pushes = 0; have_prelim = false; need_init = true;
        }
      }
    }
    if (__ballot(cand >= 0) == 0ull && __ballot(!exhausted) == 0ull) break;
    if (cand < 0) continue;

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
        if (accept_root(obs, S, Sinv, t_mid, q, a.P, r2, st.pos, st.vel, e_pre, st.vc)) {
          accepted = true;
          L(27) = st.pos.m[0][1]; L(28) = st.pos.m[1][1]; L(29) = st.pos.m[2][1];
          L(30) = st.vel.x; L(31) = st.vel.y; L(32) = st.vel.z; L(33) = e_pre;
          st.epoch = 0.0; st.has01 = false; st.has21 = false; st.it = 0;
          // cold starts of the first iteration's two solves (newton_solver.rs:151-161), evaluated here so that the
          // loop body below is the warm iteration only
          st.chi01 = st.chi21 = 0.0;
          if (st.vc.ok && !resume) {
            if constexpr (kPair) {  // one cold start per lane of the pair, exchanged
              const bool odd = (tid & 1u) != 0u;
              const double mine = prelim_kepuni(velocity_problem(st.vc, odd ? dt21 : dt01, a.P.kepler_eps, a.P.math_mode));
              const double other = pair_swap(mine);
              st.chi01 = odd ? other : mine;
              st.chi21 = odd ? mine : other;
            } else {
              prelim_kepuni_pair(velocity_problem(st.vc, dt01, a.P.kepler_eps, a.P.math_mode), dt01, dt21, st.chi01, st.chi21);
            }
          }
          break;
        }
        ++slot;
      }
      need_init = false;
      if (!accepted) finished = true;
      else if (resume) step = kCorrNone;  // this root's loop ran in the fast kernel: PrelimOrbit, next root
      else if (fabs(dt01) <= kEps || fabs(dt21) <= kEps) step = kCorrNone;  // gauss.rs:1311-1313
      resume = false;
    } else {
      step = correction_iteration<false, true, GaussParams, kPair>(obs, S, Sinv, t_mid, dt01, dt21, a.P, st);
    }
    const unsigned long long ucand = (unsigned long long)cand;
    const bool writer = !kPair || (tid & 1u) == 0u;  // lane-pair form: both lanes hold the same result
    if (step == kCorrSome) {
      if (writer) {
      a.W.state[0 * n + ucand] = st.pos.m[0][1];
      a.W.state[1 * n + ucand] = st.pos.m[1][1];
      a.W.state[2 * n + ucand] = st.pos.m[2][1];
      a.W.state[3 * n + ucand] = st.vel.x;
      a.W.state[4 * n + ucand] = st.vel.y;
      a.W.state[5 * n + ucand] = st.vel.z;
      a.W.state[6 * n + ucand] = st.epoch;
      a.W.flags[ucand] = kFlagOk | kFlagCorrected | kFlagSlow;
      }
      cand = -1;
    } else if (step == kCorrNone) {
      if (!have_prelim) {  // the first PrelimOrbit is the fallback (gauss.rs:1245-1246)
        have_prelim = true;
        if (writer) {
#pragma unroll
          for (unsigned k = 0; k < 7; ++k) a.W.state[(unsigned long long)k * n + ucand] = L(27 + k);
        }
      }
      ++pushes;
      ++slot;
      if (pushes >= a.P.max_tested_solutions || slot >= n_filt) finished = true;
      else need_init = true;
    }
    if (finished) {
      if (writer) a.W.flags[ucand] = have_prelim ? (kFlagOk | kFlagSlow) : (unsigned)kCauseNoRoots;
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

// One candidate of K2c. kSlowList = false skips the candidates marked kFlagSlow (their state is being produced
// concurrently by the general correction kernel); true handles exactly those.
// kScreen: the residual loop in the screening arithmetic of device/screen.hpp (glibc-faithful mode: K3b re-evaluates
// the contenders exactly).
// kDirect (short trajectories, the launcher's choice): the elements come straight from the state (screen_prologue) and
//   the loop starts warm from the epoch; the < 0.1 % of candidates that route declines go to the redo list — the tail of
//   slow_list, evaluated by the slow-list pass (kRedo) — with bit 63 set: "the screening loop behind the reference's
//   route". In place they cost the main pass its registers (the reference's route inlined: 26 spilled VGPRs, 2.08 ->
//   2.35 ms; as a call: 52, 2.31 ms). On LONG trajectories the redo list backfires — a near-parabolic candidate runs the
//   reference's Kepler iteration at every one of its 100 - 300 observations, and a pass of a few hundred lanes lasts as
//   long as its longest chain (10 000 x 300 observations: slow-list pass 0.8 -> 1.5 ms) while the conversion is a
//   small part of such a candidate — so those batches run with kDirect = false: every candidate through the angles.
// A sum that is not finite (absurd inputs: overflow, 0 / 0, a zero uncertainty) takes the reference's expressions: in
// the slow-list pass in place, from the main pass through the redo list (bit 63 clear; a candidate is on one list at
// most), which keeps that second loop out of the main pass's register budget.
constexpr unsigned long long kRedoScreenBit = 1ull << 63;
template <bool kSlowList, int kFn, bool kScreen = false, bool kDirect = false, bool kRedo = false>
__device__ __forceinline__ void rms_candidate(const RmsArgs& a, unsigned long long cid) {
  const unsigned f0 = a.W.flags[cid];
  if ((f0 & kFlagEmpty) || !(f0 & kFlagOk)) return;
  if (!kSlowList && (f0 & kFlagSlow)) return;
  const unsigned long long n = a.I.n_cand;
  const CandidateRef c = decode_candidate(a.I, cid);
  const Vec3 r2 = {a.W.state[0 * n + cid], a.W.state[1 * n + cid], a.W.state[2 * n + cid]};
  const Vec3 v2 = {a.W.state[3 * n + cid], a.W.state[4 * n + cid], a.W.state[5 * n + cid]};
  const double epoch = a.W.state[6 * n + cid];
  constexpr bool kInPlace = kSlowList || kRedo;  // evaluated by the slow-list pass: no list to hand a candidate to
  if constexpr (kScreen && kDirect) {
    const ScreenConsts SCR = screen_consts();
    Equinoctial eq;
    TwoBodyFrame F;
    ScreenKepler S;
    const int cls = screen_prologue(equ_to_ecl(r2), equ_to_ecl(v2), epoch, eq, F, S, SCR);
    bool fallback = cls == kScreenDivert;
    if (!fallback) {
      unsigned flags = f0 & (kFlagOk | kFlagCorrected | kFlagSlow);
      unsigned long long i_start, i_end;
      select_rms_interval(a.I.mjd + c.base, c.n_obs, c.idx[0], c.idx[2], a.extf, a.dtmax, i_start, i_end);
      if (cls == kScreenHyperbolic) {  // Cometary, converted, e >= 1: check_elliptical_orbit's Err before any residual
        a.W.n_win[cid] = (unsigned)(i_end - i_start + 1);
        a.W.sum[cid] = 0.0;
        a.W.flags[cid] = flags | kFlagCometary | kFlagRmsErr;
        return;
      }
      const ScreenFrame G = screen_frame(eq, F);
      double acc = 0.0;
      bool err = false;
#ifdef OUTFIT_K2C_PROLOGUE_ONLY  // measurement aid: the loop sees one observation (what the conversion in front of it costs)
      i_end = i_start;
#endif
      for (unsigned long long i = i_start; i <= i_end; ++i) {
        // one record per observation (kernels.h kObsRecDoubles): 16-byte loads from one address
        const ScreenObs ob = load_screen_obs(a.obs_rec + (size_t)kObsRecDoubles * (c.base + i));
        if (!screen_kepler(eq, F, ob.mjd - eq.epoch, S, a.sc, SCR, true)) {
          err = true;
          break;
        }
        // (the generic residual stays inline behind its test: a second, generic loop for the rest of the window was
        // measured — 2.23 instead of 2.08 ms, the register allocator spills more with two loops)
        acc += screen_term<true>(eq, F, G, S.sx, S.cx, ob, SCR, S.inv_rho_max);
      }
      if (err || fabs(acc) < 1e300) {
        a.W.n_win[cid] = (unsigned)(i_end - i_start + 1);
        a.W.sum[cid] = acc;
        a.W.flags[cid] = flags | (screen_level(eq.a, S.inv_rho_max, S.dmin) << kFlagLevelShift) | (err ? kFlagRmsErr : 0u);
        return;
      }
    }
    if (!kInPlace) {
      const unsigned long long at = atomicAdd(a.W.slow_count + 1, 1ull);
      a.W.slow_list[n - 1 - at] = cid | (fallback ? kRedoScreenBit : 0ull);
      return;
    }
    // slow-list pass: the reference's route below, in place (its screened value is what the redo list would give)
  }

  // build_result -> compute_orbit_from_state -> ccek1 (gauss.rs:906-923,1063-1076)
  // kFn = kFnPlatform (correctly-rounded mode): these elements only feed the candidate's RMS, the selected candidate's
  // are recomputed correctly rounded in K3 from the same state; kFn = kFnGlibc: glibc's bits all the way (orbit.hpp)
  // (the conversion in front of a screening loop takes the screen's own atan2 / sincos: orbit.hpp kFnScreen)
  constexpr int kConv = kScreen ? kFnScreen : kFn;
  const Elements el = ccek1<kConv>(equ_to_ecl(r2), equ_to_ecl(v2), epoch);
  unsigned flags = (f0 & (kFlagOk | kFlagCorrected | kFlagSlow)) | (el.kind ? kFlagCometary : 0u);

  Equinoctial eq;
  const int conv = to_equinoctial<kConv>(el, eq);
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
    const TwoBodyFrame F = twobody_frame<kConv>(eq);
    bool exact = !kScreen;
    if (kScreen) {
      const ScreenFrame G = screen_frame(eq, F);
      const ScreenConsts SCR = screen_consts();  // compile-time literals (screen.hpp: why not kernel arguments)
      ScreenKepler S;
      S.warm = false;
      S.dmin = 1.0;
      S.inv_rho_max = 0.0;
#ifdef OUTFIT_K2C_PROLOGUE_ONLY
      i_end = i_start;
#endif
      for (unsigned long long i = i_start; i <= i_end; ++i) {
        const ScreenObs ob = load_screen_obs(a.obs_rec + (size_t)kObsRecDoubles * (c.base + i));
        if (!screen_kepler(eq, F, ob.mjd - eq.epoch, S, a.sc, SCR)) {
          err = true;
          break;
        }
        acc += screen_term(eq, F, G, S.sx, S.cx, ob, SCR, S.inv_rho_max);
      }
      flags |= screen_level(eq.a, S.inv_rho_max, S.dmin) << kFlagLevelShift;
      if (!err && !(fabs(acc) < 1e300)) {
        if (!kInPlace) {
          const unsigned long long at = atomicAdd(a.W.slow_count + 1, 1ull);
          a.W.slow_list[n - 1 - at] = cid;
          return;
        }
        exact = true;
        acc = 0.0;
      }
    }
    if (exact) {
      for (unsigned long long i = i_start; i <= i_end; ++i) {
        const unsigned long long o = c.base + i;
        const Vec3 op = {a.obs_pos[3 * o], a.obs_pos[3 * o + 1], a.obs_pos[3 * o + 2]};
        double term;
        if (!ephemeris_error<kFn>(eq, F, a.I.mjd[o], a.I.ra[o], a.I.dec[o], a.cos_dec[o], a.I.sra[o], a.I.sdec[o], op,
                                  term, kFn == kFnPlatform ? &a.sc : nullptr)) {
          err = true;
          break;
        }
        acc = acc + term;
      }
    }
  }
  a.W.sum[cid] = acc;
  a.W.flags[cid] = flags | (err ? kFlagRmsErr : 0u);
}

// kSlowList = false: every candidate of the chunk, one lane each. true: exactly the slow list — its length lives on the
// device, so the grid covers every candidate and the lanes beyond the list return at once (0.16 ms; a modest grid
// striding over the list was measured in round 3: the same 0.16 ms — the floor is the 20-observation chain of a handful
// of waves — and the loop cost 200 spilled registers).
// kMode: kRmsExact — platform functions, every rounding of the reference's expressions (correctly-rounded mode);
//        kRmsGlibc — the glibc-faithful functions (orbit.hpp kFnGlibc; OUTFIT_RMS_EXACT_ALL=1);
//        kRmsScreen — the screening arithmetic (glibc-faithful mode, followed by K3b).
constexpr int kRmsExact = 0, kRmsGlibc = 1, kRmsScreen = 2;
#ifndef OUTFIT_RMS_WAVES
#define OUTFIT_RMS_WAVES 3
#endif
// (the slow-list pass — few lanes, three routes in one kernel, a chain of up to 30 k instructions — is compiled for two
// waves per SIMD: 256 registers instead of 168 + spills)
// kDirect (kRmsScreen only): the screen's direct route from the state to the elements (rms_candidate; RmsArgs::screen_direct)
template <bool kSlowList, int kMode = kRmsExact, bool kDirect = false>
__global__ void __launch_bounds__(256, kSlowList ? 2 : OUTFIT_RMS_WAVES)
k_orbit_rms(RmsArgs a) {
  constexpr int kFn = kMode == kRmsGlibc ? kFnGlibc : kFnPlatform;
  constexpr bool kScr = kMode == kRmsScreen;
  const unsigned long long lid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (kSlowList) {
    const unsigned long long n_slow = a.W.slow_count[0];
    if (lid < n_slow) {
      rms_candidate<true, kFn, kScr, kDirect>(a, a.W.slow_list[lid]);
    } else if (kScr && lid - n_slow < a.W.slow_count[1]) {  // the main pass's redo list
      const unsigned long long e = a.W.slow_list[a.I.n_cand - 1 - (lid - n_slow)];
      if (kDirect && (e & kRedoScreenBit)) rms_candidate<false, kFn, true, false, true>(a, e & ~kRedoScreenBit);
      else rms_candidate<false, kFn, false>(a, e & ~kRedoScreenBit);
    }
  } else if (lid < a.I.n_cand) {
    rms_candidate<false, kFn, kScr, kDirect>(a, lid);
  }
}

// The mapping BASELINE.json's north_star names — ONE WAVEFRONT PER CANDIDATE — for the RMS stage, as a measurement
// aid (OUTFIT_RMS_WAVE=1; never the default): lane i evaluates observation i of the window, the terms are then folded
// left to right exactly as the reference's try_fold does (wave-uniform loop over the lanes, first Err ends it), so the
// bytes are those of k_orbit_rms<false>. The per-candidate prologue (ccek1, to_equinoctial, frame) is evaluated
// redundantly by the lanes. A candidate has 20 observations here: 20 of 64 lanes work. Measured: DESIGN.md §3.
__global__ void __launch_bounds__(256, 3)
k_orbit_rms_wave(RmsArgs a) {
  const unsigned long long cid = (unsigned long long)blockIdx.x * 4 + threadIdx.x / 64;
  const unsigned lane = threadIdx.x % 64;
  if (cid >= a.I.n_cand) return;
  const unsigned f0 = a.W.flags[cid];
  if ((f0 & kFlagEmpty) || !(f0 & kFlagOk) || (f0 & kFlagSlow)) return;
  const unsigned long long n = a.I.n_cand;
  const CandidateRef c = decode_candidate(a.I, cid);
  const Vec3 r2 = {a.W.state[0 * n + cid], a.W.state[1 * n + cid], a.W.state[2 * n + cid]};
  const Vec3 v2 = {a.W.state[3 * n + cid], a.W.state[4 * n + cid], a.W.state[5 * n + cid]};
  const double epoch = a.W.state[6 * n + cid];
  const Elements el = ccek1<kFnPlatform>(equ_to_ecl(r2), equ_to_ecl(v2), epoch);
  const unsigned flags = (f0 & (kFlagOk | kFlagCorrected | kFlagSlow)) | (el.kind ? kFlagCometary : 0u);
  Equinoctial eq;
  const int conv = to_equinoctial(el, eq);
  if (conv != 0) {
    if (lane == 0) a.W.flags[cid] = flags | ((unsigned)conv << kFlagConvShift);
    return;
  }
  unsigned long long i_start, i_end;
  select_rms_interval(a.I.mjd + c.base, c.n_obs, c.idx[0], c.idx[2], a.extf, a.dtmax, i_start, i_end);
  double acc = 0.0;
  bool err = false;
  if (sqrt(eq.h * eq.h + eq.k * eq.k) >= 1.0) {
    err = true;
  } else {
    const TwoBodyFrame F = twobody_frame(eq);
    for (unsigned long long b0 = i_start; b0 <= i_end && !err; b0 += 64) {
      const unsigned long long i = b0 + lane;
      double term = 0.0;
      int bad = 0;
      if (i <= i_end) {
        const unsigned long long o = c.base + i;
        const Vec3 op = {a.obs_pos[3 * o], a.obs_pos[3 * o + 1], a.obs_pos[3 * o + 2]};
        bad = !ephemeris_error(eq, F, a.I.mjd[o], a.I.ra[o], a.I.dec[o], a.cos_dec[o], a.I.sra[o], a.I.sdec[o], op, term,
                               &a.sc);
      }
      const unsigned cnt = (unsigned)((i_end - b0 + 1) < 64 ? (i_end - b0 + 1) : 64);
      for (unsigned k = 0; k < cnt; ++k) {  // the reference's left fold, in observation order
        if (__shfl(bad, (int)k)) { err = true; break; }
        acc = acc + __shfl(term, (int)k);
      }
    }
  }
  if (lane == 0) {
    a.W.n_win[cid] = (unsigned)(i_end - i_start + 1);
    a.W.sum[cid] = acc;
    a.W.flags[cid] = flags | (err ? kFlagRmsErr : 0u);
  }
}

// ------------------------------------------------------------------------------
// K3: replay of the sequential loop of estimate_best_orbit over the candidates of
// one trajectory, in (triplet, realisation) order.
// ------------------------------------------------------------------------------
// One wave = 64 trajectories; their candidates are contiguous in the scratch arrays, so the wave stages
// them through LDS in tiles of kSelTile candidates per trajectory (two rows of 32 contiguous words per load
// instruction) instead of 64 lanes each walking its own row (one cache line per lane and load).
#ifndef OUTFIT_CONTENDER_WINDOW
#define OUTFIT_CONTENDER_WINDOW 1.0e-4
#endif
constexpr double kContenderWindow = OUTFIT_CONTENDER_WINDOW;  // relative part of K3b's contender rule (see k_refine_best)
constexpr unsigned kSelTile = 16;
// Launched with 256 threads per 64 trajectories for throughput, with 1024 for small batches: the staging phase of a tile
// — loads, one division and one square root per candidate — runs on 4 / 16 waves instead of one (a lone wave issues an
// instruction every ~5 cycles: with 16 workgroups on the whole device K3 was 0.11 ms of a 1.05 ms step at 1 000
// trajectories, now 0.05; 0.19 -> 0.16 ms at 100 000); the fold itself stays with the first wave, one lane per trajectory.
__global__ void __launch_bounds__(1024)
k_select_best(SelectArgs a) {
  __shared__ unsigned s_list[64][kContenderListCap];  // K3b's contenders, collected against the running bound
  __shared__ double s_sum[64][kSelTile + 1];
  __shared__ double s_den[64][kSelTile + 1];   // 2 n_win
  __shared__ double s_rms[64][kSelTile + 1];   // sqrt(sum / (2 n_win)): evaluated by the staging lanes, see below
  __shared__ unsigned s_flags[64][kSelTile + 1];
  const unsigned tl = threadIdx.x;
  const unsigned long long blk0 = (unsigned long long)blockIdx.x * 64;
  const unsigned long long lt = blk0 + (tl & 63u);
  const bool live = tl < 64u && lt < a.n_traj;   // threads beyond the first wave only stage
  GaussResultRec out;
  out.status = 0; out.cause = 0; out.stage = 0; out.kind = 0;
  out.attempts = 0; out.best_triplet = 0; out.best_realization = 0;
  out.epoch = 0.0; out.rms = 0.0;
  for (int k = 0; k < 6; ++k) out.elem[k] = 0.0;
  const unsigned nt = live ? a.n_triplets[lt] : 0u;
  out.n_triplets = nt;
  double best = INFINITY;
  long long best_c = -1;
  int last_error = 0;
  unsigned attempts = 0;
  bool aborted = false;
  const unsigned long long c0 = lt * a.cand_per_traj;
  const unsigned n_c = nt * a.n_real;
  // for K3b: the upper bound of the exact minimum, min_c (screened_c + slack_c), slack_c = screen_slack_rad / sigma_min
  double inv_sigma = 0.0, bound = INFINITY;
  unsigned n_list = 0;  // kContenderOverflow once more than kContenderListCap candidates have passed
  if (a.bound && live) {
    double sig = INFINITY;
    // four observations per pass (indices clamped to the last one: a repeated element does not change a minimum), so
    // that the eight loads of a pass are in flight together instead of one dependent load per observation
    const unsigned long long o0 = a.traj_offset[a.traj0 + lt], o1 = a.traj_offset[a.traj0 + lt + 1];
    for (unsigned long long o = o0; o < o1; o += 4) {
      double m[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long ok = o + k < o1 ? o + k : o1 - 1;
        m[k] = fmin(fabs(a.sra[ok]), fabs(a.sdec[ok]));
      }
      sig = fmin(sig, fmin(fmin(m[0], m[1]), fmin(m[2], m[3])));
    }
    inv_sigma = 1.0 / sig;
  }
  for (unsigned cb = 0; cb < a.cand_per_traj; cb += kSelTile) {
    // does any trajectory of this wave still need candidates from this tile on?
    const bool want = live && !aborted && cb < n_c;
    if (!__syncthreads_or(want)) break;   // (also the barrier between the previous tile's fold and this tile's staging)
    for (unsigned idx = tl; idx < 64u * kSelTile; idx += blockDim.x) {
      const unsigned row = idx / kSelTile, cc = idx % kSelTile;
      const unsigned long long g = (blk0 + row) * a.cand_per_traj + cb + cc;
      const bool in = (blk0 + row) < a.n_traj && cb + cc < a.cand_per_traj;
      s_flags[row][cc] = in ? a.W.flags[g] : kFlagEmpty;
      // the candidate's RMS — the division and the square root of the fold below — is evaluated HERE, by the 64 lanes
      // on independent candidates, instead of inside the fold's dependent chain (one lane, ~100 candidates in a row,
      // each waiting for the previous `best`): the operations and operands are the fold's, so are the bits; a value the
      // fold does not use (candidate without a sum, pruned, trajectory aborted) is simply not read
      const double den = 2.0 * (double)(in ? a.W.n_win[g] : 0u);
      const double sm = in ? a.W.sum[g] : 0.0;
      s_sum[row][cc] = sm;
      s_den[row][cc] = den;
      s_rms[row][cc] = sqrt(sm / den);
    }
    __syncthreads();
    if (want) {
      const unsigned hi = n_c - cb < kSelTile ? n_c - cb : kSelTile;
      for (unsigned cc = 0; cc < hi; ++cc) {
        const unsigned c = cb + cc;
        const unsigned f = s_flags[tl][cc];
        attempts += 1;
        if (!(f & kFlagOk)) { last_error = (int)(f & 0xFu); continue; }  // prelim_orbit Err
        const unsigned conv = (f >> kFlagConvShift) & 0x7u;
        if (conv) { out.status = (int)conv; aborted = true; break; }      // to_equinoctial()? (quirk Q2)
        // rms_orbit_error with prune_if_rms_ge = Some(best): the partial sums are
        // non-decreasing, so "some partial sum >= cutoff" == "final sum >= cutoff".
        const double denom = s_den[tl][cc];
        const double cutoff = isfinite(best) ? best * best * denom : INFINITY;
        double rms;
        if (f & kFlagRmsErr) rms = best;
        else rms = (s_sum[tl][cc] >= cutoff) ? best : s_rms[tl][cc];  // == sqrt(sum / denom)
        if (!isfinite(rms)) { last_error = 4; continue; }  // NonFiniteScore
        if (a.bound && !(f & kFlagRmsErr)) {
          // K3b's contender rule (see k_refine_best), applied with the bound as it stands: the final bound is no larger,
          // so whatever passes the final test passes this one — the list is a superset of the contenders
          const double screened = s_rms[tl][cc];
          const double slack = screen_slack_rad((f & kFlagLevelMask) >> kFlagLevelShift) * inv_sigma;
          bound = fmin(bound, screened + slack);
          if (screened <= bound * (1.0 + kContenderWindow) + slack) {
            if (n_list < kContenderListCap) s_list[tl][n_list++] = c;
            else n_list = kContenderOverflow;
          }
        }
        if (rms < best) { best = rms; best_c = (long long)c; }
      }
    }
  }
  if (!live) return;
  if (a.bound) {
    a.bound[2 * lt] = bound;
    a.bound[2 * lt + 1] = inv_sigma;
    unsigned* row = a.list + lt * (kContenderListCap + 1u);
    row[0] = n_list;
    for (unsigned k = 0; k < kContenderListCap; ++k) row[1 + k] = k < n_list ? s_list[tl][k] : 0u;
  }
  if (nt == 0) {
    out.status = 1;  // NoFeasibleTriplets (trajectory.rs:439-451)
    a.out[a.traj0 + lt] = out;
    return;
  }
  out.attempts = attempts;
  if (!aborted) {
    if (best_c >= 0) {
      const unsigned long long cid = c0 + (unsigned long long)best_c;
      const unsigned f = a.W.flags[cid];
      out.status = 0;
      out.stage = (f & kFlagCorrected) ? 1 : 0;
      out.best_triplet = (unsigned)best_c / a.n_real;
      out.best_realization = (unsigned)best_c % a.n_real;
      // build_result -> compute_orbit_from_state -> ccek1 (gauss.rs:906-923,1063-1076) for the winner, with the
      // correctly-rounded functions: these are the elements the caller gets
      out.rms = best;
      if (!a.bound) {  // (with K3b behind it the winner, its epoch, rms and elements are K3b's to write: no gather here)
        const unsigned long long nc = a.n_cand;
        const Vec3 r2 = {a.W.state[0 * nc + cid], a.W.state[1 * nc + cid], a.W.state[2 * nc + cid]};
        const Vec3 v2 = {a.W.state[3 * nc + cid], a.W.state[4 * nc + cid], a.W.state[5 * nc + cid]};
        out.epoch = a.W.state[6 * nc + cid];
        // correctly rounded, or glibc's bits (atan2 included) in the glibc-faithful mode
        const Elements el = a.math_mode == kMathGlibc ? ccek1<kFnGlibc>(equ_to_ecl(r2), equ_to_ecl(v2), out.epoch)
                                                      : ccek1<kFnCr>(equ_to_ecl(r2), equ_to_ecl(v2), out.epoch, kMathCr);
        out.kind = el.kind;
        for (int k = 0; k < 6; ++k) out.elem[k] = el.e[k];
      }
    } else {
      out.status = 2;  // NoViableOrbit{cause, attempts}
      out.cause = last_error;
    }
  }
  a.out[a.traj0 + lt] = out;
}

// ------------------------------------------------------------------------------
// K3b (glibc-faithful mode): exact re-evaluation of the contenders.
//
// In OUTFIT_MATH_GLIBC the selection key — the RMS of estimate_best_orbit's fold (trajectory.rs:455-544) — must be the
// bits a glibc run of the reference computes, which takes glibc's atan2 / hypot / tan / sincos (glibcmath*.hpp): ~2.2x
// the instructions of the platform functions in the residual loop (measured: K2c 5.7 -> 12.5 ms per 10^7 candidates).
// Two stages instead: K2c screens every candidate (device/screen.hpp: its own cheap arithmetic, with a per-candidate error
// level), K3 folds those values and the bound m = min_c (screened_c + slack_c), and this kernel re-evaluates EXACTLY — with
// the glibc-faithful functions — every candidate that can still be the exact minimum (screened_c <= m (1 + 1e-4) +
// slack_c: 1.005 candidates per trajectory on the bench workload) and keeps the smallest (rms, index), i.e. the first
// minimum of the reference's strict-`<` fold. The winner, its `rms` and its elements are then those of the exhaustive
// exact evaluation as long as every candidate's screening error stays inside its bound (tests/test_gpu_glibc_mode.py:
// all of 1.5 x 10^7 candidates within a quarter of it); OUTFIT_RMS_EXACT_ALL=1 runs the exhaustive evaluation instead, and
// the same test holds the two against each other byte for byte on 135 000 trajectories.
// Four lanes per evaluation (a lane per trajectory was a 45 k-instruction chain on 1.5 waves per SIMD: 1.6 ms;
// four lanes 0.6 ms), and the contenders of a wave's sixteen trajectories form one work list, so a trajectory with many
// of them is spread over the wave.
// ------------------------------------------------------------------------------
// lanes per trajectory (kRefineGroup, a template parameter of the two functions below): 4 for throughput — a lane per
// trajectory was a 45 k-instruction chain on 1.5 waves per SIMD: 1.6 ms; four lanes 0.6 ms —, 16 for small batches, where
// the chain of the few waves there are is what the run waits for (1 000 trajectories: 0.22 -> see DESIGN.md §3).
// The terms are added in observation order whatever the group size: same bytes.

// rms_orbit_error of one candidate with the glibc-faithful functions, evaluated by the kRefineGroup lanes of a group:
// lane `sub` takes the observations i_start + sub, + kRefineGroup, ...; the terms are then added one after the other in
// observation order by every lane (the reference's left fold, bit for bit). NaN = the candidate cannot be selected
// (conversion error, e >= 1, an Err of ephemeris_error). The prologue is evaluated redundantly by the lanes.
template <int kFn, unsigned kRefineGroup>
__device__ __forceinline__ double exact_candidate_rms(const RmsArgs& a, unsigned long long cid, unsigned sub,
                                                      unsigned group_base) {
  const unsigned long long n = a.I.n_cand;
  const CandidateRef c = decode_candidate(a.I, cid);
  const Vec3 r2 = {a.W.state[0 * n + cid], a.W.state[1 * n + cid], a.W.state[2 * n + cid]};
  const Vec3 v2 = {a.W.state[3 * n + cid], a.W.state[4 * n + cid], a.W.state[5 * n + cid]};
  Equinoctial eq;
  {
    const Elements el = ccek1<kFn>(equ_to_ecl(r2), equ_to_ecl(v2), a.W.state[6 * n + cid]);
    if (to_equinoctial<kFn>(el, eq) != 0) return nan("");
  }
  unsigned long long i_start, i_end;
  select_rms_interval(a.I.mjd + c.base, c.n_obs, c.idx[0], c.idx[2], a.extf, a.dtmax, i_start, i_end);
  if (sqrt(eq.h * eq.h + eq.k * eq.k) >= 1.0) return nan("");
  const TwoBodyFrame F = twobody_frame<kFn>(eq);
  double acc = 0.0;
  bool err = false;
  for (unsigned long long b0 = i_start; b0 <= i_end && !err; b0 += kRefineGroup) {
    const unsigned long long i = b0 + sub;
    double term = 0.0;
    int bad = 0;
    if (i <= i_end) {
      const unsigned long long o = c.base + i;
      const Vec3 op = {a.obs_pos[3 * o], a.obs_pos[3 * o + 1], a.obs_pos[3 * o + 2]};
      bad = !ephemeris_error<kFn>(eq, F, a.I.mjd[o], a.I.ra[o], a.I.dec[o], a.cos_dec[o], a.I.sra[o], a.I.sdec[o], op, term,
                                  kFn == kFnPlatform ? &a.sc : nullptr);
    }
    const unsigned cnt = (unsigned)((i_end - b0 + 1) < kRefineGroup ? (i_end - b0 + 1) : kRefineGroup);
    for (unsigned k = 0; k < cnt; ++k) {
      if (__shfl(bad, (int)(group_base + k))) { err = true; break; }
      acc = acc + __shfl(term, (int)(group_base + k));
    }
  }
  return err ? nan("") : sqrt(acc / (2.0 * (double)(i_end - i_start + 1)));
}

#ifdef OUTFIT_REFINE_STATS  // diagnostic build: trajectories, contenders, list flushes, largest list of one wave
__device__ unsigned long long g_refine_stats[4];
#endif
constexpr unsigned kRefineList = 256;  // contenders a wave holds between two evaluation passes
// Compiled for 3 waves per SIMD (168 VGPRs + 220 spilled): 0.60 ms against 0.65 ms at 2 waves without spills — the kernel is
// latency bound (dependent double-double chains, table look-ups), a third wave hides more than the spills cost.
// Measured and dropped (round 3): the fold and this kernel run slice by slice on the side stream under the next slice's
// K2c — same step time (18.05 ms): each of its long-lived waves displaces a K2c wave (registers), K2c 3.05 -> 3.65 ms.
// inject_fail: test aid (OUTFIT_TEST_EXACT_RMS_FAILS=1, read by the launcher): every exact evaluation reports an Err,
// so that the path "the screen accepted candidates the exact evaluation rejects" can be exercised
// (tests/test_gpu_glibc_mode.py); always 0 otherwise.
// (the small-batch form, 16 lanes per trajectory, is compiled for 2 waves per SIMD: there are few waves and their chain is
// what counts, so no spills)
template <int kFn, unsigned kRefineGroup>  // kFnGlibc: the glibc-faithful mode; kFnPlatform: the correctly-rounded mode's RMS (platform functions)
#ifndef OUTFIT_REFINE_WAVES
#define OUTFIT_REFINE_WAVES 2
#endif
__global__ void __launch_bounds__(64, kRefineGroup == 16u ? 2 : OUTFIT_REFINE_WAVES)
k_refine_best(RmsArgs r, GaussResultRec* out_all, const double* bound, const unsigned* list, unsigned long long n_traj,
              int inject_fail) {
  constexpr unsigned kRefineTraj = 64 / kRefineGroup;       // trajectories per wave
  __shared__ unsigned s_item_c[kRefineList];      // candidate index within its trajectory
  __shared__ unsigned char s_item_t[kRefineList]; // trajectory of the wave (0..kRefineTraj-1)
  __shared__ double s_item_rms[kRefineList];      // its exact RMS (NaN: cannot be selected)
  __shared__ double s_window[kRefineTraj], s_inv_sigma[kRefineTraj];
  __shared__ unsigned s_nc[kRefineTraj];
  const unsigned tl = threadIdx.x, g = tl / kRefineGroup, sub = tl % kRefineGroup, group_base = g * kRefineGroup;
  const unsigned long long blk0 = (unsigned long long)blockIdx.x * kRefineTraj;
  const unsigned long long lt = blk0 + g;
  const bool live = lt < n_traj;
  const bool ok = live && out_all[r.I.traj0 + lt].status == 0;
  if (sub == 0) s_nc[g] = ok ? r.I.n_triplets[lt] * r.I.n_real : 0u;
  __syncthreads();
  // The contender rule. The screen's error is an error in the computed ANGLES (its own bound per candidate, device/screen.hpp:
  // screen_slack_rad of the level it stored), so in the normalised RMS it has an absolute part — that angle over the smallest
  // uncertainty of the trajectory — beside the relative one (roundings: kContenderWindow). With slack_c = screen_slack_rad(level_c) /
  // sigma_min the exact RMS of candidate c lies in screened_c (1 -+ w) -+ slack_c, so the exact minimum is at most
  //   m = min_c (screened_c + slack_c)                          (folded by K3: bound[])
  // and c can be that minimum only if   screened_c <= m (1 + kContenderWindow) + slack_c:   the contenders.
  if (sub == 0) {
    s_window[g] = ok ? bound[2 * lt] * (1.0 + kContenderWindow) : -1.0;
    s_inv_sigma[g] = ok ? bound[2 * lt + 1] : 0.0;
  }
  __syncthreads();
  // The wave alternates two phases until its kRefineTraj trajectories are exhausted:
  //  1. scan: the 64 lanes sweep one trajectory's candidates at a time (coalesced) and append the contenders —
  //     screened RMS within the window of the screened minimum — to ONE list of the wave (ballot compaction, the
  //     count is wave-uniform), until the list could overflow on the next sweep;
  //  2. evaluate: the groups take the list's entries sixteen at a time, whichever trajectory they belong to (one call
  //     site: the function is ~12 k instructions per lane), then each trajectory's group folds the exact values of its
  //     entries: the smallest (rms, index) wins, i.e. the first minimum of the reference's strict-`<` fold.
  // A trajectory with many contenders (seen: 77 of 210 noisy candidates within 1e-4) is thus shared by the whole wave
  // instead of serialising on its own four lanes (that one trajectory used to cost 3.7 ms of a 4.3 ms launch).
  // A second round of the two phases, with an unbounded window, serves the trajectories none of whose contenders has
  // a finite exact RMS (the exact evaluation fails where the screen did not): all their candidates are then evaluated.
  double best = INFINITY;
  long long best_c = -1;
  for (int round = 0; round < 2; ++round) {
  unsigned count = 0, t = 0, sweep = 0;  // wave-uniform
  for (;;) {
    while (t < kRefineTraj) {
      const unsigned n_c = s_nc[t];
      if (sweep >= n_c) { ++t; sweep = 0; continue; }
      if (count + 64u > kRefineList) break;
      // Round 0 takes the trajectory's candidates from the list K3 collected against its running bound (a superset of
      // the contenders, ~1 - 3 entries) instead of sweeping all of them; a trajectory whose list overflowed, and the
      // exhaustive second round, sweep as before. Wave-uniform: t, round and the row's count are the same for all lanes.
      unsigned c = sweep + tl;
      bool from_list = false, mine = c < n_c;
      if (round == 0 && list != nullptr) {
        const unsigned* row = list + (blk0 + t) * (unsigned long long)(kContenderListCap + 1u);
        const unsigned n_l = row[0];
        if (n_l != kContenderOverflow) {
          from_list = true;
          mine = tl < n_l;
          c = mine ? row[1u + tl] : 0u;
        }
      }
      bool is = false;
      if (mine) {
        const unsigned long long cid = (blk0 + t) * r.I.cand_per_traj + c;
        const unsigned f = r.W.flags[cid];
        if ((f & kFlagOk) && !((f >> kFlagConvShift) & 0x7u) && !(f & kFlagRmsErr))
          is = sqrt(r.W.sum[cid] / (2.0 * (double)r.W.n_win[cid])) <=
               s_window[t] + screen_slack_rad((f & kFlagLevelMask) >> kFlagLevelShift) * s_inv_sigma[t];
      }
      const unsigned long long mask = __ballot(is);
      if (is) {
        const unsigned slot = count + (unsigned)__popcll(mask & ((1ull << tl) - 1ull));
        s_item_c[slot] = c;
        s_item_t[slot] = (unsigned char)t;
      }
      count += (unsigned)__popcll(mask);
      if (from_list) { ++t; sweep = 0; }
      else sweep += 64u;
    }
    if (count == 0u) break;
    __syncthreads();
#ifdef OUTFIT_REFINE_STATS
    if (tl == 0) { atomicAdd(&g_refine_stats[1], (unsigned long long)count); atomicAdd(&g_refine_stats[2], 1ull);
                   atomicMax(&g_refine_stats[3], (unsigned long long)count); }
#endif
#ifndef OUTFIT_REFINE_SCAN_ONLY  // (measurement aid)
    for (unsigned k0 = 0; k0 < count; k0 += kRefineTraj) {
      const unsigned k = k0 + g;
      if (k < count) {
        const unsigned long long cid = (blk0 + s_item_t[k]) * r.I.cand_per_traj + s_item_c[k];
        const double e = exact_candidate_rms<kFn, kRefineGroup>(r, cid, sub, group_base);
        if (sub == 0) s_item_rms[k] = inject_fail ? nan("") : e;
      }
    }
    __syncthreads();
    if (sub == 0)
      for (unsigned k = 0; k < count; ++k) {
        if (s_item_t[k] != g) continue;
        const double e = s_item_rms[k];
        const long long c = (long long)s_item_c[k];
        if (isfinite(e) && (e < best || (e == best && c < best_c))) { best = e; best_c = c; }
      }
    __syncthreads();
#endif
    count = 0u;
    if (t >= kRefineTraj) break;
  }
  const bool again = ok && best_c < 0 && sub == 0;
  if (round == 1 || __ballot(again) == 0ull) break;
  __syncthreads();
  if (sub == 0) {
    s_window[g] = again ? INFINITY : -1.0;
    if (!again) s_nc[g] = 0u;
  }
  __syncthreads();
  }
#ifdef OUTFIT_REFINE_STATS
  if (ok && sub == 0) atomicAdd(&g_refine_stats[0], 1ull);
#endif
  const unsigned long long c0 = lt * r.I.cand_per_traj;
  if (!ok || sub != 0) return;
  GaussResultRec* out = out_all + r.I.traj0 + lt;
  if (best_c < 0) {
    // No candidate has a finite exact RMS, not even in the exhaustive second round: the exact evaluation rejects
    // what the screen accepted. The reference skips every Err / non-finite score and ends in
    // NoViableOrbit{cause: last error, attempts} (trajectory.rs:499-541) — so does this record: K3 counted the
    // attempts (no `?` abort, or status would not be 0); the cause is the LAST candidate's error in fold order: a
    // prelim_orbit Err keeps its code, a candidate that reached the RMS now scores non-finite (NonFiniteScore).
    int last_error = 0;
    const unsigned n_c = r.I.n_triplets[lt] * r.I.n_real;
    for (unsigned c = 0; c < n_c; ++c) {
      const unsigned f = r.W.flags[c0 + c];
      last_error = (f & kFlagOk) ? 4 : (int)(f & 0xFu);
    }
    out->status = 2;
    out->cause = last_error;
    out->stage = 0; out->kind = 0;
    out->best_triplet = 0; out->best_realization = 0;
    out->epoch = 0.0; out->rms = 0.0;
    for (int k = 0; k < 6; ++k) out->elem[k] = 0.0;
    return;
  }
  // the winner's record: build_result -> compute_orbit_from_state -> ccek1 with glibc's bits (K3 filled the counters)
  const unsigned long long cid = c0 + (unsigned long long)best_c, nc = r.I.n_cand;
  const unsigned f = r.W.flags[cid];
  const Vec3 r2 = {r.W.state[0 * nc + cid], r.W.state[1 * nc + cid], r.W.state[2 * nc + cid]};
  const Vec3 v2 = {r.W.state[3 * nc + cid], r.W.state[4 * nc + cid], r.W.state[5 * nc + cid]};
  const Elements el = kFn == kFnGlibc ? ccek1<kFnGlibc>(equ_to_ecl(r2), equ_to_ecl(v2), r.W.state[6 * nc + cid])
                                      : ccek1<kFnCr>(equ_to_ecl(r2), equ_to_ecl(v2), r.W.state[6 * nc + cid], kMathCr);
  out->stage = (f & kFlagCorrected) ? 1 : 0;
  out->best_triplet = (unsigned)best_c / r.I.n_real;
  out->best_realization = (unsigned)best_c % r.I.n_real;
  out->epoch = el.epoch;
  out->kind = el.kind;
  for (int k = 0; k < 6; ++k) out->elem[k] = el.e[k];
  out->rms = best;
}
hipError_t launch_refine_best(const SelectArgs& a, const RmsArgs& r, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
#ifdef OUTFIT_REFINE_STATS
  { static const unsigned long long z[4] = {}; (void)hipStreamSynchronize(s); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_refine_stats), z, sizeof z); }
#endif
  static const int inject_fail = std::getenv("OUTFIT_TEST_EXACT_RMS_FAILS") != nullptr;  // test aid, see the kernel
  // small batch (the chunk's candidates, as for the correction kernels' forms): 16 lanes per trajectory
  static const char* forced_group = std::getenv("OUTFIT_REFINE_GROUP");  // measurement aid: 4 or 16 lanes per evaluation
  // 16 lanes up to 16 384 trajectories (measured, ms at 4 / 16 lanes: 10 000 trajectories 0.23 / 0.16, 20 000 0.24 / 0.22,
  // 40 000 0.32 / 0.41, 100 000 0.64 / 0.93)
  const unsigned group = forced_group ? (std::atoi(forced_group) == 16 ? 16u : 4u)
                                      : (use_pair_form(a.n_cand) || a.n_traj <= 16384) ? 16u : 4u;
  const dim3 grid((unsigned)((a.n_traj + 64 / group - 1) / (64 / group)));
  const bool glibc = a.math_mode == kMathGlibc;
  static const bool use_list = std::getenv("OUTFIT_REFINE_SCAN_ALL") == nullptr;  // measurement / test aid: sweep every candidate
  if (group == 16u) {
    if (glibc) hipLaunchKernelGGL((k_refine_best<kFnGlibc, 16>), grid, dim3(64), 0, s, r, a.out, a.bound, use_list ? a.list : nullptr, a.n_traj, inject_fail);
    else hipLaunchKernelGGL((k_refine_best<kFnPlatform, 16>), grid, dim3(64), 0, s, r, a.out, a.bound, use_list ? a.list : nullptr, a.n_traj, inject_fail);
  } else {
    if (glibc) hipLaunchKernelGGL((k_refine_best<kFnGlibc, 4>), grid, dim3(64), 0, s, r, a.out, a.bound, use_list ? a.list : nullptr, a.n_traj, inject_fail);
    else hipLaunchKernelGGL((k_refine_best<kFnPlatform, 4>), grid, dim3(64), 0, s, r, a.out, a.bound, use_list ? a.list : nullptr, a.n_traj, inject_fail);
  }
#ifdef OUTFIT_REFINE_STATS
  { unsigned long long h[4] = {}; (void)hipStreamSynchronize(s); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_refine_stats), sizeof h);
    fprintf(stderr, "[refine stats] trajectories %llu contenders %llu (%.4f each) list passes %llu largest list of a wave %llu\n", h[0], h[1], h[0] ? (double)h[1] / h[0] : 0.0, h[2], h[3]); }
#endif
  return hipGetLastError();
}

// ------------------------------------------------------------------------------
// kepler::propagate_universal, batched
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_propagate_universal(unsigned long long n, const double* rv, const double* t0, const double* t1,
                      SolverSpec sv, int math_mode, double* out, int* status) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double o[11];
  for (int k = 0; k < 11; ++k) o[k] = nan("");
  const Vec3 r = {rv[6 * i], rv[6 * i + 1], rv[6 * i + 2]};
  const Vec3 v = {rv[6 * i + 3], rv[6 * i + 4], rv[6 * i + 5]};
  status[i] = propagate_universal(r, v, t0[i], t1[i], sv, o, math_mode);
  for (int k = 0; k < 11; ++k) out[11 * i + k] = o[k];
}

// ------------------------------------------------------------------------------
// self-test: the fast divisions of fastdiv.hpp against the compiler's `/`, bit for bit
// ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// random double with a uniformly random mantissa and sign, exponent in [-e_range, e_range]
__device__ __forceinline__ double random_double(unsigned long long& s, int e_range) {
  const unsigned long long bits = splitmix64(s);
  const int e = (int)(splitmix64(s) % (unsigned long long)(2 * e_range + 1)) - e_range;
  const unsigned long long mant = bits & 0x000FFFFFFFFFFFFFull;
  const unsigned long long sign = bits & 0x8000000000000000ull;
  return __longlong_as_double((long long)(sign | ((unsigned long long)(1023 + e) << 52) | mant));
}

__global__ void __launch_bounds__(256)
k_selftest_division(unsigned long long n_per_thread, unsigned long long seed, unsigned long long* mismatches) {
  unsigned long long s = seed + 0x632BE59BD9B4E019ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long bad = 0;
  for (unsigned long long it = 0; it < n_per_thread; ++it) {
    // wide exponents every 8th round so that the out-of-window fallback is exercised too
    const int er = (it & 7) == 7 ? 600 : 40;
    const double a1 = random_double(s, er), a2 = (it & 15) == 3 ? 0.0 : random_double(s, er);
    const double b = random_double(s, er);
    double q1, q2;
    fdiv::div2_shared(a1, a2, b, q1, q2);
    const double r1 = a1 / b, r2 = a2 / b;
    bad += (__double_as_longlong(q1) != __double_as_longlong(r1));
    bad += (__double_as_longlong(q2) != __double_as_longlong(r2));
    const int j = 1 + (int)(it % 70);
    const double beta = random_double(s, (it & 7) == 7 ? 400 : 8);
    const double d2 = fdiv::kSeries.d2[j], d3 = fdiv::kSeries.d3[j];
    const double c2 = fdiv::div_const(beta, d2, fdiv::kSeries.r2[j]);
    const double c3 = fdiv::div_const(beta, d3, fdiv::kSeries.r3[j]);
    bad += (__double_as_longlong(c2) != __double_as_longlong(beta / d2));
    bad += (__double_as_longlong(c3) != __double_as_longlong(beta / d3));
    // the guarded forms of the correction loop (kepler.hpp div_by_*, s_funct's / 3, velocity_finish's three quotients
    // over one g, eccentricity_control's pair): inside the guard's window they are the bits of `/`, and the guard
    // reports every operand outside it (zero, tiny, huge, Inf, NaN)
    {
      fdiv::Guard g;
      g.reset();
      const double sqrt_mu = sqrt(kMu);
      const double v1 = div_by_sqrt_mu(beta, &g), v2 = div_by_mu(beta, &g), v3 = div_by_vlight(beta, &g);
      const double v4 = fdiv::div_const(beta, 3.0, 1.0 / 3.0, &g);
      double num[3], q[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) num[k] = random_double(s, er);
      if ((it & 63) == 9) num[2] = INFINITY;
      if ((it & 63) == 10) num[1] = nan("");
      if ((it & 63) == 11) num[0] = 0.0;
      if ((it & 63) == 12) num[0] = -0.0;
      fdiv::divn_shared<3>(num, b, q, &g);
      const double one_mu[2] = {1.0, kMu};
      double over[2];
      fdiv::divn_shared<2>(one_mu, b, over, &g);
      bool in = fdiv::in_window(beta) && fdiv::in_window(b);
#pragma unroll
      for (int k = 0; k < 3; ++k) in = in && fdiv::in_window(num[k]);
      if (g.ok()) {
        bad += !in;  // the guard may only pass when every operand is inside the window
        bad += (__double_as_longlong(v1) != __double_as_longlong(beta / sqrt_mu));
        bad += (__double_as_longlong(v2) != __double_as_longlong(beta / kMu));
        bad += (__double_as_longlong(v3) != __double_as_longlong(beta / kVlightAU));
        bad += (__double_as_longlong(v4) != __double_as_longlong(beta / 3.0));
#pragma unroll
        for (int k = 0; k < 3; ++k) bad += (__double_as_longlong(q[k]) != __double_as_longlong(num[k] / b));
        bad += (__double_as_longlong(over[0]) != __double_as_longlong(1.0 / b));
        bad += (__double_as_longlong(over[1]) != __double_as_longlong(kMu / b));
      }
      // (the converse is not required: the guard tests the SUM of the magnitudes against the upper limit, which is
      //  conservative; with operands of ordinary size it passes — the bench run hands no candidate over for it)
      // without a guard: the plain divisions
      bad += (__double_as_longlong(div_by_sqrt_mu(beta, nullptr)) != __double_as_longlong(beta / sqrt_mu));
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

// Row n2: observer geocentric (ecliptic J2000) and heliocentric (equatorial J2000) positions from the
// Earth-fixed site vector and the two time scales, one lane per observation (pvobs + helio_position).
__global__ void __launch_bounds__(256)
k_observer_geometry(GeometryArgs a) {
  const unsigned long long o = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_obs) return;
  const double tt = a.mjd_tt[o];
  const Vec3 geo = erot::pvobs(tt, a.mjd_ut1[o], {a.body_fixed[3 * o], a.body_fixed[3 * o + 1], a.body_fixed[3 * o + 2]});
  a.geo_ecl[3 * o] = geo.x; a.geo_ecl[3 * o + 1] = geo.y; a.geo_ecl[3 * o + 2] = geo.z;
  unsigned long long nr;
  double tau;
  Vec3 h = {nan(""), nan(""), nan("")};
  if (record_index(a.T, tt, nr, tau)) h = erot::helio_position(earth_from_block(a.T, a.T.coef + nr * a.T.per_block, tau), geo);
  else atomicExch(a.error_flag, 1);
  a.helio_equ[3 * o] = h.x; a.helio_equ[3 * o + 1] = h.y; a.helio_equ[3 * o + 2] = h.z;
}

// outfit_hip_device_normals: the generator's output in the noise_z layout
__global__ void __launch_bounds__(256)
k_device_normals(unsigned long long seed, const unsigned long long* traj_key, unsigned long long key_base,
                 unsigned long long n_traj, unsigned k_cap, unsigned n_noise, double* out) {
  const unsigned long long id = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long per_traj = (unsigned long long)k_cap * n_noise;
  if (id >= n_traj * per_traj) return;
  const unsigned long long traj = id / per_traj;
  const unsigned within = (unsigned)(id % per_traj);
  const unsigned t = within / n_noise, r = within % n_noise + 1;
  double z[6];
  noise::normals6(seed, traj_key ? traj_key[traj] : key_base + traj, t, r, z);
#pragma unroll
  for (int k = 0; k < 6; ++k) out[id * 6 + k] = z[k];
}

// Self-test of the shared-reciprocal Aberth sweep against the plain divisions: same return code and the same
// bits in all 8 iterates, on random sparse octics x^8 + c6 x^6 + c3 x^3 + c0 (the sign pattern and size of the
// Gauss polynomial, gauss.rs:585-614, plus arbitrary signs). out[0] = polynomials that differ, out[1] = guard trips.
__global__ void __launch_bounds__(256)
k_selftest_aberth(unsigned long long n_per_thread, unsigned long long seed, unsigned long long* out) {
  unsigned long long s = seed + 0xA24BAED4963EE407ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long bad = 0, trips = 0;
  for (unsigned long long it = 0; it < n_per_thread; ++it) {
    const int er = (it & 3) == 3 ? 12 : 3;
    double c6 = random_double(s, er), c3 = random_double(s, er), c0 = random_double(s, er);
    if ((it & 1) == 0) { c6 = -fabs(c6); c0 = -fabs(c0); }  // physical case: c6 < 0, c0 < 0
    if ((it & 31) == 5) c3 = 0.0;                            // even polynomial: roots in +- and conjugate quadruples
    if ((it & 31) == 9) c6 = 0.0;
    if ((it & 63) == 17) { c3 = 0.0; c6 = 0.0; }             // z^8 + c0: eight roots on a circle
    Cx za[8], zb[8];
    const int ra = aberth_deg8<true>(c0, c3, c6, 50u, 1e-6, za);
    const int rb = aberth_deg8<false>(c0, c3, c6, 50u, 1e-6, zb);
    if (ra == 3) { trips += 1; continue; }
    bool same = ra == rb;
    if (ra != 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        same = same && __double_as_longlong(za[k].re) == __double_as_longlong(zb[k].re) &&
               __double_as_longlong(za[k].im) == __double_as_longlong(zb[k].im);
    }
    bad += !same;
  }
  if (bad) atomicAdd(out, bad);
  if (trips) atomicAdd(out + 1, trips);
}

// Self-test of the correctly-rounded functions: the fast (table + rounding test) path against the full
// double-double evaluation, bit for bit. out[0] = arguments whose sin or cos differ, out[1] = arguments whose
// acos differs, out[2] = sin/cos arguments that needed the full evaluation, out[3] = same for acos.
__global__ void __launch_bounds__(256)
k_selftest_crmath(unsigned long long n_per_thread, unsigned long long seed, unsigned long long* out) {
  unsigned long long s = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long bad_sc = 0, bad_ac = 0, slow_sc = 0, bad_hy = 0;
  for (unsigned long long it = 0; it < n_per_thread; ++it) {
    double x;
    const unsigned mode = (unsigned)(it & 7);
    const double u = (double)(splitmix64(s) >> 11) * 0x1p-53;  // [0, 1)
    if (mode < 3) x = (u - 0.5) * 64.0;                                         // the angles the path sees
    else if (mode == 3) x = random_double(s, 19);                               // all magnitudes below 2^20
    else if (mode == 4) x = random_double(s, 30) * 0x1p-25;                     // small arguments
    else if (mode == 5) x = (double)((long long)(u * 4096.0) - 2048) * 1.5707963267948966 + random_double(s, 20) * 0x1p-40;
    else if (mode == 6) x = (double)((long long)(u * 256.0) - 128) * 0.0078125 + random_double(s, 20) * 0x1p-45;
    else x = (u - 0.5) * 2.0e5;
    double s1, c1, s2, c2;
    crm::sincos(x, s1, c1);
    crm::sincos_slow(x, s2, c2);
    bad_sc += (__double_as_longlong(s1) != __double_as_longlong(s2)) || (__double_as_longlong(c1) != __double_as_longlong(c2));
    if (fabs(x) < 1048576.0 && x != 0.0) {
      int q;
      const crm::dd r = crm::reduce_pio2(x, q);
      crm::dd sr, cr;
      crm::sincos_fast_dd(r, sr, cr);
      slow_sc += !(crm::rounds_safely(sr, crm::kFastRelErr * fabs(sr.hi)) &&
                   crm::rounds_safely(cr, crm::kFastRelErr * fabs(cr.hi)));
    }
    double y;
    if (mode < 4) y = 2.0 * u - 1.0;
    else if (mode == 4) y = copysign(1.0 - random_double(s, 10) * random_double(s, 10) * 0x1p-40, u - 0.5);  // near +-1
    else if (mode == 5) y = random_double(s, 25) * 0x1p-30;                                                   // near 0
    else y = cos((u - 0.5) * 6.0);
    if (fabs(y) <= 1.0) {
      const double a1 = crm::acos(y), a2 = crm::acos_slow(y);
      bad_ac += __double_as_longlong(a1) != __double_as_longlong(a2);
    }
    // sinh / cosh / log (hyperbolic cold start): |f| < 15 as the call site guarantees, plus tiny and large arguments
    double f;
    if (mode < 3) f = (u - 0.5) * 30.0;
    else if (mode == 3) f = (u - 0.5) * 0.5;                                     // around the 2^-3 switch
    else if (mode == 4) f = random_double(s, 40) * 0x1p-20;
    else if (mode == 5) f = (double)((long long)(u * 128.0) - 64) * 0.6931471805599453 + random_double(s, 12) * 0x1p-30;
    else if (mode == 6) f = (double)((long long)(u * 4096.0) - 2048) * 0.0078125 + random_double(s, 12) * 0x1p-40;
    else f = (u - 0.5) * 1400.0;
    double h1, g1, h2, g2;
    crm::sinhcosh(f, h1, g1);
    crm::sinhcosh_slow(f, h2, g2);
    bad_hy += (__double_as_longlong(h1) != __double_as_longlong(h2)) || (__double_as_longlong(g1) != __double_as_longlong(g2));
    double yl;
    if (mode < 3) yl = 1.0 + u * 50.0;
    else if (mode == 3) yl = 1.0 + fabs(random_double(s, 30)) * 0x1p-10;        // just above 1 (c -> 1+)
    else if (mode == 4) yl = exp(fabs(f));
    else if (mode == 5) yl = 1.0 + fabs(random_double(s, 8));
    else yl = fabs(random_double(s, 200)) + 1.0;
    const double l1 = crm::log(yl), l2 = crm::log_slow(yl);
    bad_hy += __double_as_longlong(l1) != __double_as_longlong(l2);
    // atan2: all quadrants, near the axes (tiny ratios), unit-vector components
    double ay, ax;
    if (mode < 3) { ::sincos((u - 0.5) * 6.283185307179586, &ay, &ax); ax *= 1.0 + fabs(f); ay *= 1.0 + fabs(f); }
    else if (mode == 3) { ay = random_double(s, 60) * 0x1p-30; ax = u - 0.5; }
    else if (mode == 4) { ax = random_double(s, 60) * 0x1p-30; ay = u - 0.5; }
    else { ay = random_double(s, 100); ax = random_double(s, 100); }
    const double t1 = crm::atan2(ay, ax), t2 = crm::atan2_slow(ay, ax);
    bad_hy += __double_as_longlong(t1) != __double_as_longlong(t2);
  }
  if (bad_sc) atomicAdd(out, bad_sc);
  if (bad_ac) atomicAdd(out + 1, bad_ac);
  if (slow_sc) atomicAdd(out + 2, slow_sc);
  if (bad_hy) atomicAdd(out + 3, bad_hy);
}

// outfit_hip_crmath_eval: the device's correctly-rounded functions on given arguments (checked against
// high-precision known answers by the tests). fn & 15: 0 sin, 1 cos, 2 acos, 3 sinh, 4 cosh, 5 log, 6 atan2 (x holds n pairs y, x), 7 / 8 sin / cos of K2c's sincos_k; fn & 16: the
// full double-double evaluation instead of Ziv's two steps.
__global__ void __launch_bounds__(256)
k_crmath_eval(int fn, unsigned long long n, const double* x, double* out, SinCosConsts sc) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool slow = (fn & 16) != 0;
  if ((fn & 15) == 6) {  // atan2 on pairs (y, x)
    out[i] = slow ? crm::atan2_slow(x[2 * i], x[2 * i + 1]) : crm::atan2(x[2 * i], x[2 * i + 1]);
    return;
  }
  const double v = x[i];
  double a = 0.0, b = 0.0, r;
  if ((fn & 15) >= 9) {  // the glibc-faithful functions of OUTFIT_MATH_GLIBC (glibcmath.hpp)
    const int g = fn & 15;
    out[i] = g == 9 ? glm::sin(v) : g == 10 ? glm::cos(v) : g == 11 ? glm::sinh(v) : glm::cosh(v);
    return;
  }
  if ((fn & 15) == 7 || (fn & 15) == 8) {  // K2c's kernel-argument-constant sincos (not correctly rounded: ~1 ulp)
    sincos_k(v, sc, a, b);
    out[i] = (fn & 15) == 7 ? a : b;
    return;
  }
  switch (fn & 15) {
    case 0: case 1:
      if (slow) crm::sincos_slow(v, a, b); else crm::sincos(v, a, b);
      r = (fn & 15) == 0 ? a : b;
      break;
    case 2: r = slow ? crm::acos_slow(v) : crm::acos(v); break;
    case 3: case 4:
      if (slow) crm::sinhcosh_slow(v, a, b); else crm::sinhcosh(v, a, b);
      r = (fn & 15) == 3 ? a : b;
      break;
    default: r = slow ? crm::log_slow(v) : crm::log(v); break;
  }
  out[i] = r;
}

// outfit_hip_glibc_eval: the glibc-faithful functions of round 3 on given arguments (checked against the oracle's
// restatement and, on glibc hosts, against the platform libm by the tests). fn: 0 sin1, 1 cos1 (the standalone
// entry points), 2 tan, 3 atan2(x[i], y[i]), 4 hypot(x[i], y[i]), 5 acos, 6 log.
__global__ void __launch_bounds__(256)
k_glibc_eval(int fn, unsigned long long n, const double* x, const double* y, double* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i], b = (fn == 3 || fn == 4) ? y[i] : 0.0;
  out[i] = fn == 0 ? glm::sin1(a) : fn == 1 ? glm::cos1(a) : fn == 2 ? glm::tan(a) : fn == 3 ? glm::atan2(a, b) : fn == 4 ? glm::hypot(a, b) : fn == 5 ? glm::acos(a) : glm::log(a);
}
hipError_t launch_glibc_eval(int fn, unsigned long long n, const double* x, const double* y, double* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_glibc_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, n, x, y, out);
  return hipGetLastError();
}

// Calibration of the HBM byte counters for THIS library's access width: every kernel here reads and writes 8 bytes
// per lane (SoA doubles), a width the rocprofv3 FETCH_SIZE / WRITE_SIZE derivations are not calibrated for on gfx950
// (MI355X_MICROARCH.md, HBM section). This kernel streams n doubles in and n doubles out, 8 B per lane, coalesced;
// with n * 8 B far beyond the 256 MiB Infinity Cache the true traffic is n * 8 B each way.
__global__ void __launch_bounds__(256)
k_calibration_stream(unsigned long long n, const double* in, double* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 2.0;
}
hipError_t launch_calibration_stream(unsigned long long n, const double* in, double* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_calibration_stream, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, in, out);
  return hipGetLastError();
}

// outfit_hip_prelim_kepuni: the cold-start value of the universal Kepler solve (kepler/params.rs:185-191) for
// explicit problems (dt, r0, sig0, alpha, e0, convergency), in the given math mode
__global__ void __launch_bounds__(256)
k_prelim_kepuni(unsigned long long n, const double* prob, int math_mode, double* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  KeplerProblem p;
  p.dt = prob[6 * i]; p.r0 = prob[6 * i + 1]; p.sig0 = prob[6 * i + 2]; p.alpha = prob[6 * i + 3];
  p.e0 = prob[6 * i + 4]; p.convergency = prob[6 * i + 5]; p.math_mode = math_mode;
  out[i] = prelim_kepuni(p);
}
hipError_t launch_prelim_kepuni(unsigned long long n, const double* prob, int math_mode, double* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_prelim_kepuni, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, prob, math_mode, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------
hipError_t launch_observer_geometry(const GeometryArgs& a, hipStream_t s) {
  if (a.n_obs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_observer_geometry, dim3((unsigned)((a.n_obs + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_device_normals(unsigned long long seed, const unsigned long long* traj_key,
                                 unsigned long long key_base, unsigned long long n_traj, unsigned k_cap,
                                 unsigned n_noise, double* out, hipStream_t s) {
  const unsigned long long n = n_traj * k_cap * n_noise;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_device_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, traj_key, key_base,
                     n_traj, k_cap, n_noise, out);
  return hipGetLastError();
}
hipError_t launch_selftest_aberth(unsigned long long n_threads, unsigned long long n_per_thread,
                                  unsigned long long seed, unsigned long long* d_out, hipStream_t s) {
  const unsigned grid = (unsigned)((n_threads + 255) / 256);
  hipLaunchKernelGGL(k_selftest_aberth, dim3(grid), dim3(256), 0, s, n_per_thread, seed, d_out);
  return hipGetLastError();
}
hipError_t launch_selftest_crmath(unsigned long long n_threads, unsigned long long n_per_thread,
                                  unsigned long long seed, unsigned long long* d_out, hipStream_t s) {
  const unsigned grid = (unsigned)((n_threads + 255) / 256);
  hipLaunchKernelGGL(k_selftest_crmath, dim3(grid), dim3(256), 0, s, n_per_thread, seed, d_out);
  return hipGetLastError();
}
hipError_t launch_crmath_eval(int fn, unsigned long long n, const double* x, double* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_crmath_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, n, x, out, sincos_consts());
  return hipGetLastError();
}
hipError_t launch_selftest_division(unsigned long long n_threads, unsigned long long n_per_thread,
                                    unsigned long long seed, unsigned long long* d_mismatches, hipStream_t s) {
  const unsigned grid = (unsigned)((n_threads + 255) / 256);
  hipLaunchKernelGGL(k_selftest_division, dim3(grid), dim3(256), 0, s, n_per_thread, seed, d_mismatches);
  return hipGetLastError();
}
hipError_t launch_triplet_select(const TripletArgs& a, bool lds_heap, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_traj + 63) / 64);
  if (lds_heap) {
    const size_t lds = (size_t)a.P.k_cap * 64 * (sizeof(double) + sizeof(unsigned));
    hipLaunchKernelGGL(k_triplet_select_lds, dim3(grid), dim3(64), lds, s, a);
  } else {
    hipLaunchKernelGGL(k_triplet_select, dim3(grid), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}
hipError_t launch_triplet_select_wave(const TripletArgs& a, size_t lds_bytes, hipStream_t s) {
  if (a.n_wave_list == 0) return hipSuccess;
#ifdef OUTFIT_K0_STATS
  {
    static const unsigned long long zero[16] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k0_stat), zero, sizeof zero);
  }
#endif
  hipLaunchKernelGGL(k_triplet_select_wave, dim3(a.n_wave_list), dim3(64), lds_bytes, s, a);
#ifdef OUTFIT_K0_STATS
  {
    unsigned long long k[16] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(k, HIP_SYMBOL(g_k0_stat), sizeof k);
    const double t = k[0] ? (double)k[0] : 1.0;
    fprintf(stderr, "[k0 stats] trajectories %llu | per trajectory: middle groups %.1f  rows enumerated %.1f  last-chunks scored %.1f  "
                    "chunks with a survivor %.1f  survivors %.1f (fill pushes %.1f, replacements %.1f, dropped %.1f)  "
                    "sift-down levels %.1f  sift-up levels %.1f  cycles %.3e\n",
            k[0], k[1] / t, k[2] / t, k[3] / t, k[4] / t, k[5] / t, k[6] / t, k[7] / t, k[8] / t, k[9] / t, k[10] / t, k[11] / t);
  }
#endif
  return hipGetLastError();
}
hipError_t launch_observer_prepass(const PrepassArgs& a, hipStream_t s) {
  if (a.n_obs == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_obs + 255) / 256);
  hipLaunchKernelGGL(k_observer_prepass, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gauss_roots(const RootsArgs& a, hipStream_t s) {
  if (a.cand_count == 0) return hipSuccess;
  if (use_pair_form(a.I.n_cand)) {  // small batch: one Aberth iterate per lane, eight lanes per candidate
    const unsigned long long lanes = a.cand_count * 8ull;
    hipLaunchKernelGGL(k_gauss_roots_lanes, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((a.cand_count + OUTFIT_ROOTS_BLOCK - 1) / OUTFIT_ROOTS_BLOCK);
  hipLaunchKernelGGL(k_gauss_roots, dim3(grid), dim3(OUTFIT_ROOTS_BLOCK), 0, s, a);
  return hipGetLastError();
}
// the slow list is filled by K2a (fused cold start): reset it before the first K2a slice of a chunk
hipError_t reset_slow_list(const CandidateScratch& w, hipStream_t s) {
  return hipMemsetAsync(w.slow_count, 0, 2 * sizeof(unsigned long long), s);  // slow list, redo list (K2c screen)
}
hipError_t launch_gauss_correct_fast(const CorrectArgs& a, unsigned n_cu, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  // persistent grid sized by residency (LDS: 27 doubles per lane; kFastWavesPerSimd waves per
  // SIMD), never more workgroups than there is work; every wave exits once the queue is drained
  // small batch: two lanes per candidate (lane-pair form) — twice the lanes, each candidate's chain ~ a quarter shorter
  const bool pair = use_pair_form(a.I.n_cand);
  const unsigned long long per_cu = (4ull * kFastWavesPerSimd * 64ull) / kFastBlock;
  const unsigned long long lanes = pair ? 2ull * a.I.n_cand : a.I.n_cand;
  const unsigned long long want = (lanes + kFastBlock - 1) / kFastBlock;
  const unsigned grid = (unsigned)(want < per_cu * n_cu ? want : per_cu * n_cu);
  const size_t lds = (size_t)27 * kFastBlock * sizeof(double);
  hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
#ifdef OUTFIT_K2A_STATS  // the K2a slices of this chunk have been enqueued before this launch: read and reset their counters
  {
    unsigned long long k[16] = {};
    static const unsigned long long zero[16] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(k, HIP_SYMBOL(g_k2a_stat), sizeof k);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2a_stat), zero, sizeof zero);
    const char* names[5] = {"kernel body (live candidates)", "Aberth sweep", "after the root filter", "cold-start section", "elliptic cold start: Newton iteration"};
    for (int i = 0; i < 5; ++i)
      fprintf(stderr, "[k2a stats]   %-40s wave passes %12llu (%.3f per wave)   lanes/pass %5.2f\n", names[i], k[2 * i],
              k[0] ? (double)k[2 * i] / (double)k[0] : 0.0, k[2 * i] ? (double)k[2 * i + 1] / (double)k[2 * i] : 0.0);
  }
#endif
#ifdef OUTFIT_K2B_STATS
  e = hipMemsetAsync(a.queue + 8, 0, 8 * sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  {
    static const unsigned long long zero[32] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kstat), zero, sizeof zero);
  }
#endif
  const char* redo = std::getenv("OUTFIT_TEST_GUARD_REDO");  // test aid (CorrectArgs::test_redo_mod), read per launch
  CorrectArgs b = a;
  b.test_redo_mod = redo ? (unsigned)std::strtoul(redo, nullptr, 10) : 0u;
  if (pair) hipLaunchKernelGGL(k_gauss_correct_fast<true>, dim3(grid), dim3(kFastBlock), lds, s, b);
  else hipLaunchKernelGGL(k_gauss_correct_fast<false>, dim3(grid), dim3(kFastBlock), lds, s, b);
#ifdef OUTFIT_K2B_STATS
  {
    unsigned long long h[8] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, a.queue + 8, sizeof h, hipMemcpyDeviceToHost);
    if (h[0]) fprintf(stderr, "[k2b stats] waves %llu  trips/wave %.1f  refills/wave %.1f  active lanes/trip %.2f  needy lanes/refill %.2f  "
                      "refill cycle share %.3f  cycles/wave %.3e\n", h[0], (double)h[1] / h[0], (double)h[2] / h[0],
                      (double)h[3] / h[1], (double)h[4] / (h[2] ? h[2] : 1), (double)h[5] / h[6], (double)h[6] / h[0]);
    unsigned long long k[32] = {};
    (void)hipMemcpyFromSymbol(k, HIP_SYMBOL(g_kstat), sizeof k);
    const char* names[7] = {"newton round", "series term", "re-evaluation", "iteration entry", "after both solves", "before position_and_epoch", "tail (ecc control ..)"};
    for (int i = 0; i < 7; ++i)
      fprintf(stderr, "[k2b stats]   %-28s wave passes/trip %7.3f   lanes/pass %5.2f\n", names[i], (double)k[2 * i] / h[1],
              k[2 * i] ? (double)k[2 * i + 1] / k[2 * i] : 0.0);
    for (int c = 0; c < 2; ++c) {
      unsigned long long tot = 0;
      for (int r = 0; r < 8; ++r) tot += k[16 + 8 * c + r];
      fprintf(stderr, "[k2b stats]   Newton rounds per solve, %s:", c ? "first iteration (cold start)" : "later iterations (warm start)");
      for (int r = 0; r < 8; ++r) fprintf(stderr, " %d:%.4f", r, tot ? (double)k[16 + 8 * c + r] / tot : 0.0);
      fprintf(stderr, "  (n = %llu)\n", tot);
    }
  }
#endif
  return hipGetLastError();
}
hipError_t launch_gauss_correct(const CorrectArgs& a, unsigned n_cu, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  // general kernel over the slow list (its length is only known on the device): a modest
  // persistent grid, two 256-lane workgroups per CU at most
  const unsigned long long want = (a.I.n_cand + kCorrectBlock - 1) / kCorrectBlock;
  const unsigned long long cap = 2ull * n_cu * (256u / kCorrectBlock);  // as many waves as two 256-lane workgroups per CU
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  const size_t lds = (size_t)kCorrectLdsDoubles * kCorrectBlock * sizeof(double);
  hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  if (a.I.n_cand >= (1ull << 20)) hipLaunchKernelGGL(k_gauss_correct<3>, dim3(grid), dim3(kCorrectBlock), lds, s, a);
  else if (use_pair_form(a.I.n_cand)) hipLaunchKernelGGL((k_gauss_correct<2, true>), dim3(grid), dim3(kCorrectBlock), lds, s, a);
  else hipLaunchKernelGGL(k_gauss_correct<2>, dim3(grid), dim3(kCorrectBlock), lds, s, a);
  return hipGetLastError();
}
hipError_t launch_orbit_rms(const RmsArgs& a, bool slow_list, hipStream_t s) {
  if (a.I.n_cand == 0) return hipSuccess;
  // the direct route shortens the main pass and lengthens the slow-list pass (the redo list's chains): a small batch —
  // the chunks that run the latency forms of the correction kernels — is the sum of its chains (1 000 objects: 0.95 ->
  // 1.00 ms per step with it), so it keeps the one route
  // (the size alone decides, not OUTFIT_PAIR_MODE: the test aid that forces a form must not change the screen with it)
  const bool direct = a.screen_direct != 0 && a.I.n_cand > kPairMaxCandidates;
  if (slow_list) {
    const unsigned grid = (unsigned)((a.I.n_cand + 255) / 256);  // the list length lives on the device
    if (a.rms_mode == kRmsGlibc) hipLaunchKernelGGL((k_orbit_rms<true, kRmsGlibc>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rms_mode == kRmsScreen && direct) hipLaunchKernelGGL((k_orbit_rms<true, kRmsScreen, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rms_mode == kRmsScreen) hipLaunchKernelGGL((k_orbit_rms<true, kRmsScreen>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_orbit_rms<true, kRmsExact>), dim3(grid), dim3(256), 0, s, a);
  } else {
    static const bool wave_per_candidate = std::getenv("OUTFIT_RMS_WAVE") != nullptr;  // measurement aid, see k_orbit_rms_wave
    if (wave_per_candidate) {
      hipLaunchKernelGGL(k_orbit_rms_wave, dim3((unsigned)((a.I.n_cand + 3) / 4)), dim3(256), 0, s, a);
      return hipGetLastError();
    }
    const unsigned grid = (unsigned)((a.I.n_cand + 255) / 256);
    if (a.rms_mode == kRmsGlibc) hipLaunchKernelGGL((k_orbit_rms<false, kRmsGlibc>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rms_mode == kRmsScreen && direct) hipLaunchKernelGGL((k_orbit_rms<false, kRmsScreen, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rms_mode == kRmsScreen) hipLaunchKernelGGL((k_orbit_rms<false, kRmsScreen>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_orbit_rms<false, kRmsExact>), dim3(grid), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}
hipError_t launch_select_best(const SelectArgs& a, hipStream_t s) {
  if (a.n_traj == 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.n_traj + 63) / 64);
  // small batches: 16 waves stage each tile (see the kernel); OUTFIT_K3_THREADS = 64 .. 1024 forces a shape (measurement aid)
  static const char* forced_once = std::getenv("OUTFIT_K3_THREADS");
  const char* forced = forced_once ? forced_once : std::getenv("OUTFIT_K3_THREADS_TEST");  // (_TEST: re-read per launch)
  // (100 000 trajectories: 64 threads 0.31 ms, 128 0.22, 256 0.16, 1024 0.36; 10 000: 0.11 / 0.06 at 64 / 256)
  unsigned threads = a.n_traj <= 4096 ? 1024u : 256u;
  if (forced) {
    const long v = std::strtol(forced, nullptr, 10);
    if (v >= 64 && v <= 1024 && v % 64 == 0) threads = (unsigned)v;
  }
  hipLaunchKernelGGL(k_select_best, dim3(grid), dim3(threads), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_propagate_universal(unsigned long long n, const double* rv, const double* t0,
                                      const double* t1, const SolverSpec& sv, int math_mode, double* out, int* status,
                                      hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_propagate_universal, dim3(grid), dim3(256), 0, s, n, rv, t0, t1, sv, math_mode,
                     out, status);
  return hipGetLastError();
}

}  // namespace outfit
