// gauss.hpp — device-side Gauss preliminary orbit, split in the three phases the
// kernels run (kernels.hip):
//
//   phase A  gauss_geometry + aberth_deg8 + root filter          (one lane / candidate)
//   phase B  accept_root + pos_and_vel_correction, one ITERATION (persistent lanes)
//   phase C  state -> elements -> RMS                            (orbit.hpp)
//
// Replaces GaussObs::prelim_orbit (reference
// src/initial_orbit_determination/gauss.rs:1119-1247) and what it calls:
// gauss_prelim :532, coeff_eight_poly :585, descartes_upper_bound_deg8_sparse
// :214, aberth::aberth (crate aberth 0.4.1, call site :971), accept_root :816,
// position_vector_and_reference_epoch :702, gibbs_correction :754,
// pos_and_vel_correction :1284.
#pragma once
#include "kepler.hpp"

namespace outfit {

struct Mat3 {
  // m[row][col]; for the observer/position matrices the columns are the three epochs
  double m[3][3];
};

// nalgebra gemv: ((A[:,0]*x0) + A[:,1]*x1) + A[:,2]*x2
__device__ __forceinline__ Vec3 gemv(const Mat3& A, const Vec3& x) {
  return {(A.m[0][0] * x.x + A.m[0][1] * x.y) + A.m[0][2] * x.z,
          (A.m[1][0] * x.x + A.m[1][1] * x.y) + A.m[1][2] * x.z,
          (A.m[2][0] * x.x + A.m[2][1] * x.y) + A.m[2][2] * x.z};
}
__device__ __forceinline__ Vec3 column(const Mat3& A, int c) { return {A.m[0][c], A.m[1][c], A.m[2][c]}; }

// The scalar tunables of IODParams the candidate kernels need (mod.rs:225-266).
struct GaussParams {
  double max_ecc, max_perihelion_au, min_rho2_au;
  double aberth_eps, kepler_eps, r2_min_au, r2_max_au, newton_eps, root_imag_eps;
  unsigned aberth_max_iter;
  unsigned max_tested_solutions;  // saturated to 2^32-1 on the host
  unsigned newton_max_it;         // saturated likewise
  int math_mode;                  // kMathCr / kMathGlibc (kepler.hpp)
  // Derived by derive() on the host: the outer bounds of the two-sided pre-tests (kepler.hpp `eccentricity_control`,
  // `correction_iteration` below). NaN switches a pre-test off — every compare is then false and the literal
  // quotient decides — which is what derive() stores when a parameter is outside the range the error analysis covers.
  // (NaN until derive() has run: a GaussParams filled in by hand is slower, never wrong)
  double peri_lo = __builtin_nan(""), peri_hi = __builtin_nan("");   // max_perihelion_au (1 -/+ 2^-49)
  double eps2_lo = __builtin_nan(""), eps2_hi = __builtin_nan("");   // newton_eps^2 (1 -/+ 2^-45)
  __host__ __device__ void derive() {
    const double nan = __builtin_nan("");
    const bool peri_sane = max_perihelion_au >= 1e-100 && max_perihelion_au <= 1e100 && max_ecc <= 1e100;
    peri_lo = peri_sane ? max_perihelion_au * (1.0 - 0x1p-49) : nan;
    peri_hi = peri_sane ? max_perihelion_au * (1.0 + 0x1p-49) : nan;
    const bool eps_sane = newton_eps >= 1e-30 && newton_eps <= 1.0;
    eps2_lo = eps_sane ? (newton_eps * newton_eps) * (1.0 - 0x1p-45) : nan;
    eps2_hi = eps_sane ? (newton_eps * newton_eps) * (1.0 + 0x1p-45) : nan;
  }
};

// error codes = OUTFIT_CAUSE_* of include/outfit_hip.h
constexpr int kCauseNoRoots = 1, kCauseSingular = 2, kCauseRootFinding = 3;

// nalgebra 0.34 Matrix3::try_inverse (adjugate / determinant), gauss.rs:544-546
__device__ __forceinline__ bool try_inverse(const Mat3& A, Mat3& o) {
  const double m11 = A.m[0][0], m12 = A.m[0][1], m13 = A.m[0][2];
  const double m21 = A.m[1][0], m22 = A.m[1][1], m23 = A.m[1][2];
  const double m31 = A.m[2][0], m32 = A.m[2][1], m33 = A.m[2][2];
  const double minor_m12_m23 = m22 * m33 - m32 * m23;
  const double minor_m11_m23 = m21 * m33 - m31 * m23;
  const double minor_m11_m22 = m21 * m32 - m31 * m22;
  const double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
  if (det == 0.0) return false;
  o.m[0][0] = minor_m12_m23 / det;
  o.m[0][1] = (m13 * m32 - m33 * m12) / det;
  o.m[0][2] = (m12 * m23 - m22 * m13) / det;
  o.m[1][0] = -minor_m11_m23 / det;
  o.m[1][1] = (m11 * m33 - m31 * m13) / det;
  o.m[1][2] = (m13 * m21 - m23 * m11) / det;
  o.m[2][0] = minor_m11_m22 / det;
  o.m[2][1] = (m12 * m31 - m32 * m11) / det;
  o.m[2][2] = (m11 * m22 - m21 * m12) / det;
  return true;
}

// ---- complex helpers with num-complex 0.4 operation order ---------------------
struct Cx {
  double re, im;
};
__device__ __forceinline__ Cx cadd(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cx csub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cx cmul(Cx a, Cx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ Cx cdiv(Cx a, Cx b) {
  const double n = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n};
}
// The same quotient with ONE refined reciprocal of |b|^2 shared by the two components — bit-identical to
// cdiv as long as |b|^2 (and with it both numerators, bounded by |a| |b|) stays far from the exponent limits,
// where the compiler's division would rescale (fastdiv.hpp; device self-test on 2^27 operand pairs).
// `lo` / `hi` collect the range of |b|^2 seen so that the caller can verify that condition afterwards.
__device__ __forceinline__ Cx cdiv_shared(Cx a, Cx b, double& lo, double& hi) {
  const double n = b.re * b.re + b.im * b.im;
  lo = fmin(lo, n);
  hi = fmax(hi, n);
  const double y = fdiv::refined_rcp(n);
  return {fdiv::div_with_rcp(a.re * b.re + a.im * b.im, n, y), fdiv::div_with_rcp(a.im * b.re - a.re * b.im, n, y)};
}
// (1 + 0i) / b as num-complex evaluates it: numerators 1*b.re + 0*b.im and 0*b.re - 1*b.im. A product by 0.0 is exact
// (+-0, or NaN for a non-finite factor), so fusing it into the addition changes nothing: fma(0.0, b.im, b.re) IS
// (0.0 * b.im) + b.re, fma(0.0, b.re, -b.im) IS (0.0 * b.re) - b.im, for every input incl. zeros, infinities and NaN —
// one instruction per numerator instead of two (the compiler may not do this itself: contraction is off).
__device__ __forceinline__ Cx crecip(Cx b) {
  const double n = b.re * b.re + b.im * b.im;
  return {fma(0.0, b.im, b.re) / n, fma(0.0, b.re, -b.im) / n};
}
__device__ __forceinline__ Cx crecip_shared(Cx b, double& lo, double& hi) {
  const double n = b.re * b.re + b.im * b.im;
  lo = fmin(lo, n);
  hi = fmax(hi, n);
  const double y = fdiv::refined_rcp(n);
  return {fdiv::div_with_rcp(fma(0.0, b.im, b.re), n, y), fdiv::div_with_rcp(fma(0.0, b.re, -b.im), n, y)};
}
// Complex::mul_add with a real addend: (a*b) + c, fused where num-complex fuses
__device__ __forceinline__ Cx cmul_add_real(Cx a, Cx b, double c) {
  return {fma(a.re, b.re, c) - (a.im * b.im), fma(a.re, b.im, fma(a.im, b.re, 0.0))};
}

// The first Horner step of p and of p': (1 + 0i) z + 0 and (8 + 0i) z + 0 through Complex::mul_add, for a FINITE z
// (every iterate is: a non-finite one ends the iteration in the sweep that produced it). Written out,
//   re = fma(1, z.re, 0.0) - 0.0 * z.im,   im = fma(1, z.im, fma(0.0, z.re, 0.0)):
// 0.0 * z.im is a zero, and A - (+-0) = A for every A that fma(1, z.re, 0.0) = z.re + 0.0 can be (a non-zero number, or
// +0: never -0); fma(0.0, z.re, 0.0) = (+-0) + (+0) = +0. Hence exactly {z.re + 0.0, z.im + 0.0} — the additions stay:
// they turn a -0 component into +0 as the crate's arithmetic does (a third of the polynomials end with an exactly zero
// imaginary part) — and likewise {fma(8, z.re, 0.0), fma(8, z.im, 0.0)}. Two instructions each instead of five.
__device__ __forceinline__ Cx horner_first_step_unit(Cx z) { return {z.re + 0.0, z.im + 0.0}; }
__device__ __forceinline__ Cx horner_first_step_8(Cx z) { return {fma(8.0, z.re, 0.0), fma(8.0, z.im, 0.0)}; }

// aberth::aberth::<9, f64>(&[c0,0,0,c3,0,0,c6,0,1], max_iter, eps): Aberth–Ehrlich
// sweeps from Aberth's initial circle, every root updated from the previous sweep's
// iterates (Jacobi). All 8 complex iterates and the 8 running sums stay in VGPRs
// (the loops are fully unrolled, indices are static).
//
// Exact work saving: the crate evaluates 1/(z_i - z_k) for all 56 ordered pairs; since
// 1/(z_k - z_i) == -(1/(z_i - z_k)) bit for bit (negation is exact in every operation
// of num-complex's division), 28 reciprocals suffice. The sums are accumulated in the
// crate's order (k ascending) so every partial sum is bit-identical.
// Returns 0 converged / 1 max-iteration / 2 failed (NaN or Inf iterate).
//
// kSharedRcp: the 36 complex divisions of a sweep use cdiv_shared (11 instead of 22 division
// instructions each). Its precondition is checked per sweep on the collected range of |denominator|^2;
// if it ever fails (two iterates closer than 1e-100 or farther than 1e100 apart: not seen on any input)
// the function returns 3 and the caller restarts with kSharedRcp = false, the plain divisions.
template <bool kSharedRcp>
__device__ __forceinline__ int aberth_deg8(double c0, double c3, double c6, unsigned max_iter, double eps,
                                           Cx (&z)[8]) {
  // initial guesses: the polynomial is monic with c7 = 0, so the centre is -0.0 and the
  // Pascal shift is the identity; r0 = first positive integer with
  // S(r0) = r0^8 - |c6| r0^6 - |c3| r0^3 - |c0| > 0 (dense Horner with fused steps).
  const double a6 = -fabs(c6), a3 = -fabs(c3), a0 = -fabs(c0);
  double r0 = 1.0;
  for (int guard = 0; guard < 4096; ++guard) {
    double re = 1.0;
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a6);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a3);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a0);
    if (re > 0.0) break;
    r0 += 1.0;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double s, c;  // cos/sin of theta_k = fma(2pi/8, k, (pi/2)/8), correctly rounded
    crm::aberth_dir(k, c, s);
    z[k] = {fma(r0, c, -0.0), r0 * s};
  }
  const double c6x6 = 6.0 * c6, c3x3 = 3.0 * c3;
  for (unsigned it = 1; it <= max_iter; ++it) {
    K2ASTAT(1);
    double n_lo = 1.0, n_hi = 1.0;
    Cx sum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum[i] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int k = i + 1; k < 8; ++k) {
        const Cx w = kSharedRcp ? crecip_shared(csub(z[i], z[k]), n_lo, n_hi)
                                : crecip(csub(z[i], z[k]));  // 1/(z_i - z_k)
        sum[i] = cadd(sum[i], w);
        sum[k] = cadd(sum[k], {-w.re, -w.im});             // 1/(z_k - z_i), exactly
      }
    }
    bool converged = true, failed = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const Cx zi = z[i];
      // p(z): dense Horner over the 9 coefficients (zeros included, as the crate does)
      Cx p = horner_first_step_unit(zi);  // == cmul_add_real({1.0, 0.0}, zi, 0.0)
      p = cmul_add_real(p, zi, c6);
      p = cmul_add_real(p, zi, 0.0);
      p = cmul_add_real(p, zi, 0.0);
      p = cmul_add_real(p, zi, c3);
      p = cmul_add_real(p, zi, 0.0);
      p = cmul_add_real(p, zi, 0.0);
      p = cmul_add_real(p, zi, c0);
      // p'(z): coefficients [0,0,3c3,0,0,6c6,0,8]
      Cx dp = horner_first_step_8(zi);   // == cmul_add_real({8.0, 0.0}, zi, 0.0)
      dp = cmul_add_real(dp, zi, c6x6);
      dp = cmul_add_real(dp, zi, 0.0);
      dp = cmul_add_real(dp, zi, 0.0);
      dp = cmul_add_real(dp, zi, c3x3);
      dp = cmul_add_real(dp, zi, 0.0);
      dp = cmul_add_real(dp, zi, 0.0);
      const Cx den = csub(cmul(p, sum[i]), dp);
      const Cx zn = cadd(zi, kSharedRcp ? cdiv_shared(p, den, n_lo, n_hi) : cdiv(p, den));
      z[i] = zn;  // in place: the old z_i is no longer needed by any other root
      if (!failed) {
        // the crate returns Failed at the FIRST non-finite iterate; nothing after it matters
        if (isnan(zn.re) || isnan(zn.im) || isinf(zn.re) || isinf(zn.im)) failed = true;
        else if (!(fabs(zn.re - zi.re) < eps && fabs(zn.im - zi.im) < eps)) converged = false;
      }
    }
    // outside the window nothing of this sweep is trusted (coincident iterates, |d|^2 = 0, land here too and
    // are then reported as failed by the plain divisions)
    if (kSharedRcp && !(n_lo > 1e-200 && n_hi < 1e200)) return 3;
    if (failed) return 2;
    if (converged) return 0;
  }
  return 1;
}

// The same iteration with ONE ROOT PER LANE (north_star's "8 lanes <-> 8 roots"): eight consecutive lanes hold the eight
// iterates of one polynomial. A sweep is Jacobi (every root from the previous sweep's iterates, see above), so the
// eight updates are independent: each lane gathers the group's iterates, forms ITS sum of 1/(z_i - z_k) over k
// ascending (the crate's literal form: plain divisions, no symmetry trick), its Horner pair and its update; the
// group then agrees on failed / converged through a ballot. A candidate's chain is one root's work per sweep instead
// of eight: the form for small batches, where the chain is what the run waits for (the lane-per-candidate form above
// issues half the divisions and is the one for throughput). Same operations on the same operands: same bits
// (tests/test_gpu_small_batch.py holds the two forms against each other). `z` = this lane's root (index lane & 7).
// Returns 0 converged / 1 max-iteration / 2 failed, uniform over the group.
__device__ __forceinline__ int aberth_deg8_lanes(double c0, double c3, double c6, unsigned max_iter, double eps, Cx& z) {
  const unsigned lane = __lane_id(), r = lane & 7u, base = lane & ~7u;
  const double a6 = -fabs(c6), a3 = -fabs(c3), a0 = -fabs(c0);
  double r0 = 1.0;
  for (int guard = 0; guard < 4096; ++guard) {
    double re = 1.0;
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a6);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a3);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a0);
    if (re > 0.0) break;
    r0 += 1.0;
  }
  {
    double s, c;
    crm::aberth_dir((int)r, c, s);
    z = {fma(r0, c, -0.0), r0 * s};
  }
  const double c6x6 = 6.0 * c6, c3x3 = 3.0 * c3;
  for (unsigned it = 1; it <= max_iter; ++it) {
    Cx sum = {0.0, 0.0};
#pragma unroll
    for (unsigned k = 0; k < 8; ++k) {
      const Cx zk = {__shfl(z.re, (int)(base + k)), __shfl(z.im, (int)(base + k))};
      const Cx w = crecip(csub(z, zk));  // 1/(z_i - z_k); for k == i a division by zero, discarded
      const Cx added = cadd(sum, w);
      if (k != r) sum = added;
    }
    const Cx zi = z;
    Cx p = horner_first_step_unit(zi);
    p = cmul_add_real(p, zi, c6);
    p = cmul_add_real(p, zi, 0.0);
    p = cmul_add_real(p, zi, 0.0);
    p = cmul_add_real(p, zi, c3);
    p = cmul_add_real(p, zi, 0.0);
    p = cmul_add_real(p, zi, 0.0);
    p = cmul_add_real(p, zi, c0);
    Cx dp = horner_first_step_8(zi);
    dp = cmul_add_real(dp, zi, c6x6);
    dp = cmul_add_real(dp, zi, 0.0);
    dp = cmul_add_real(dp, zi, 0.0);
    dp = cmul_add_real(dp, zi, c3x3);
    dp = cmul_add_real(dp, zi, 0.0);
    dp = cmul_add_real(dp, zi, 0.0);
    const Cx den = csub(cmul(p, sum), dp);
    const Cx zn = cadd(zi, cdiv(p, den));
    z = zn;
    const bool bad = isnan(zn.re) || isnan(zn.im) || isinf(zn.re) || isinf(zn.im);
    const bool moved = !(fabs(zn.re - zi.re) < eps && fabs(zn.im - zi.im) < eps);
    const unsigned any_bad = (unsigned)(__ballot(bad) >> base) & 0xFFu;
    const unsigned any_moved = (unsigned)(__ballot(moved) >> base) & 0xFFu;
    if (any_bad) return 2;
    if (!any_moved) return 0;
  }
  return 1;
}

struct GaussTimes {
  double t[3];
};

struct GaussGeometry {
  double tau1, tau3;
  double a0, a2, b0, b2;  // vector_a = (a0,-1,a2), vector_b = (b0,0,b2)
};

// gauss.rs:532-542 (time part of gauss_prelim)
__device__ __forceinline__ GaussGeometry gauss_times(const double (&t)[3]) {
  GaussGeometry q;
  const double tau1 = kGaussGrav * (t[0] - t[1]);
  const double tau3 = kGaussGrav * (t[2] - t[1]);
  const double tau13 = tau3 - tau1;
  q.tau1 = tau1; q.tau3 = tau3;
  q.a0 = tau3 / tau13;
  q.a2 = -(tau1 / tau13);
  q.b0 = q.a0 * (tau13 * tau13 - tau3 * tau3) / 6.0;
  q.b2 = q.a2 * (tau13 * tau13 - tau1 * tau1) / 6.0;
  return q;
}

// unit_matrix (gauss.rs:488-502) with correctly-rounded sin/cos (crmath.hpp)
__device__ __forceinline__ Mat3 unit_matrix(const double (&ra)[3], const double (&dec)[3], int math_mode) {
  Mat3 S;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double sr, cr, sd, cd;
    sincos_mode(math_mode, ra[c], sr, cr);
    sincos_mode(math_mode, dec[c], sd, cd);
    S.m[0][c] = cr * cd;
    S.m[1][c] = sr * cd;
    S.m[2][c] = sd;
  }
  return S;
}

// coeff_eight_poly (gauss.rs:585-614)
__device__ __forceinline__ void coeff_eight_poly(const Mat3& obs, const Mat3& S, const Mat3& Sinv,
                                                 const GaussGeometry& q, double& c6, double& c3, double& c0) {
  const Vec3 ra = gemv(obs, {q.a0, -1.0, q.a2});
  const Vec3 rb = gemv(obs, {q.b0, 0.0, q.b2});
  const Vec3 row1 = {Sinv.m[1][0], Sinv.m[1][1], Sinv.m[1][2]};
  const double a2s = dot(row1, ra), b2s = dot(row1, rb);
  const Vec3 R2 = column(obs, 1);
  const double r22 = ((0.0 + R2.x * R2.x) + R2.y * R2.y) + R2.z * R2.z;
  const double s2r2 = ((0.0 + S.m[0][1] * R2.x) + S.m[1][1] * R2.y) + S.m[2][1] * R2.z;
  c6 = -(a2s * a2s) - r22 - (2.0 * a2s * s2r2);
  c3 = -(2.0 * b2s * (a2s + s2r2));
  c0 = -(b2s * b2s);
}

// gauss.rs:214-240 with zero_eps = 0.0 (is_sign_positive tests the sign bit: quirk Q11)
__device__ __forceinline__ unsigned descartes_upper_bound(double c0, double c3, double c6) {
  int last = 1;
  unsigned count = 0;
  const double v[3] = {c6, c3, c0};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (fabs(v[i]) <= 0.0) continue;
    // a NaN coefficient: the reference reads its sign bit (is_sign_positive), which depends on where the NaN came from
    // and on the platform (x86 generates negative, this GPU positive NaNs) — taken as negative here and in the oracle
    const int s = (signbit(v[i]) || v[i] != v[i]) ? -1 : 1;
    if (s != last) ++count;
    last = s;
  }
  return count;
}

// gauss.rs:702-724 ; false = SpuriousRootDetected
__device__ __forceinline__ bool position_and_epoch(const Mat3& obs, const Mat3& S, const Mat3& Sinv,
                                                   double t_mid, const Vec3& c, double min_rho2, Mat3& pos,
                                                   double& epoch, fdiv::Guard* guard = nullptr) {
  const Vec3 gcap = gemv(obs, c);
  const Vec3 crhom = gemv(Sinv, gcap);
  const double rho0 = -(crhom.x / c.x), rho1 = -(crhom.y / c.y), rho2 = -(crhom.z / c.z);
  if (rho1 < min_rho2) return false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pos.m[r][0] = obs.m[r][0] + rho0 * S.m[r][0];
    pos.m[r][1] = obs.m[r][1] + rho1 * S.m[r][1];
    pos.m[r][2] = obs.m[r][2] + rho2 * S.m[r][2];
  }
  epoch = t_mid - div_by_vlight(rho1, guard);
  return true;
}

// gauss.rs:754-781
__device__ __forceinline__ Vec3 gibbs_correction(const Mat3& pos, double tau1, double tau3) {
  const double tau13 = tau3 - tau1;
  const double n1 = norm(column(pos, 0)), n2 = norm(column(pos, 1)), n3 = norm(column(pos, 2));
  const double r1m3 = 1.0 / (n1 * n1 * n1), r2m3 = 1.0 / (n2 * n2 * n2), r3m3 = 1.0 / (n3 * n3 * n3);
  const double d1 = tau3 * (r1m3 / 12.0 - 1.0 / (tau1 * tau13));
  const double d2 = (tau1 + tau3) * (r2m3 / 12.0 - 1.0 / (tau1 * tau3));
  const double d3 = -tau1 * (r3m3 / 12.0 + 1.0 / (tau3 * tau13));
  const double e0 = -d1, e1 = d2, e2 = d3;
  return {kGaussGrav * (((0.0 + pos.m[0][0] * e0) + pos.m[0][1] * e1) + pos.m[0][2] * e2),
          kGaussGrav * (((0.0 + pos.m[1][0] * e0) + pos.m[1][1] * e1) + pos.m[1][2] * e2),
          kGaussGrav * (((0.0 + pos.m[2][0] * e0) + pos.m[2][1] * e1) + pos.m[2][2] * e2)};
}

// Matrix3::norm(): sum over columns of (x^2+y^2)+z^2 accumulated from 0, then sqrt
__device__ __forceinline__ double frob_sq(const Mat3& A) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) s += (A.m[0][k] * A.m[0][k] + A.m[1][k] * A.m[1][k]) + A.m[2][k] * A.m[2][k];
  return s;
}
__device__ __forceinline__ double frob(const Mat3& A) { return sqrt(frob_sq(A)); }

// accept_root (gauss.rs:816-870): root -> positions at the three epochs, Gibbs velocity
__device__ __forceinline__ bool accept_root(const Mat3& obs, const Mat3& S, const Mat3& Sinv, double t_mid,
                                            const GaussGeometry& q, const GaussParams& P, double r2,
                                            Mat3& pos, Vec3& vel, double& epoch, VelCommon& vc) {
  const double r2m3 = 1.0 / (r2 * r2 * r2);
  const Vec3 c = {q.a0 + q.b0 * r2m3, -1.0, q.a2 + q.b2 * r2m3};
  if (!position_and_epoch(obs, S, Sinv, t_mid, c, P.min_rho2_au, pos, epoch)) return false;
  vel = gibbs_correction(pos, q.tau1, q.tau3);
  const Vec3 rm = column(pos, 1);
  const EccControl ec = eccentricity_control(rm, vel, P.max_perihelion_au, P.max_ecc, nullptr, P.peri_lo, P.peri_hi);
  vc = velocity_common_from(ec, rm, vel);
  return ec.some && ec.accepted;
}

// State of one pos_and_vel_correction loop (gauss.rs:1284-1418) between iterations.
struct CorrectionState {
  Mat3 pos;
  Vec3 vel;
  double epoch;          // 0.0 until an iteration commits (quirk Q4)
  double chi01, chi21;   // warm starts of the two universal-Kepler solves
  bool has01, has21;
  unsigned it;           // iterations consumed so far
  VelCommon vc;          // velocity.rs:105-129 evaluated on (pos[:,1], vel): shared by both sides
};

// |P_new - P|_F / |P_new|_F <= newton_eps (gauss.rs:1393-1411) from the two SQUARED norms; lo / hi = newton_eps^2
// (1 -/+ 2^-45) from GaussParams::derive(). The literal value is RN(RN(sqrt d) / RN(sqrt n)) = sqrt(d / n)(1 + t) with
// |t| < 2^-51; d <= RN(n lo) gives sqrt(d / n) <= newton_eps (1 - 2^-47), so the literal compare is true, and
// d >= RN(n hi) makes it false. In between (or with NaN bounds / operands) the literal expression decides. The caller
// has already excluded n <= kFrobSqMax and a non-finite n.
__device__ __forceinline__ bool rel_converged(double diff_sq, double denom_sq, double newton_eps, double lo, double hi) {
  if (diff_sq <= denom_sq * lo) return true;
  if (diff_sq >= denom_sq * hi) return false;
  return sqrt(diff_sq) / sqrt(denom_sq) <= newton_eps;
}

// Before the FIRST guarded iteration of a root (correction_iteration<kFast = true>, throughput form): accept_root
// evaluates eccentricity_control without a guard, so the carried reciprocal of r2 is not there yet.
__device__ __forceinline__ void prepare_guarded(CorrectionState& st) { st.vc.inv_r2 = fdiv::refined_rcp(st.vc.r2); }

// kCorrRedo (fast kernel only): an operand of one of the iteration's guarded divisions (fastdiv.hpp Guard) was zero,
// outside the window or not finite — nothing of this iteration is used, the general kernel (plain divisions) replays the
// candidate from its first root
enum CorrStep { kCorrContinue = 0, kCorrSome = 1, kCorrNone = 2, kCorrSlow = 3, kCorrRedo = 4 };

// ONE iteration of the loop at gauss.rs:1316-1415. `continue` in the reference = return
// kCorrContinue here (the iteration is consumed, the state is kept).
//
// kFast = true is the path of the persistent fast kernel, which walks ONE root: every outcome that ends or
// continues this root's loop is handled in place (the `continue`s of the reference, Some(state) at a fixed point
// or at the last iteration); only the rejection by eccentricity_control inside the loop — None: a PrelimOrbit is
// pushed and the NEXT root is tried — returns kCorrSlow, and the caller hands the candidate to the general
// (kFast = false) kernel, which replays it from its first root. Results are identical by construction.
//
// The eccentricity_control at the end of an iteration is evaluated on exactly the (r2, v2)
// the next iteration starts from (gauss.rs:1385 vs velocity.rs:112), so its outcome is kept in
// st.vc instead of being recomputed.
//
// kWarmOnly: both solves start from st.chi01 / st.chi21 whatever the has-flags say — the caller has put the
// cold-start value prelim_kepuni(params) there before the first iteration (solve_kepuni_with_guess(None) IS
// run_newton(prelim_kepuni(params)), and a first iteration whose solves fail ends the loop: the cold start is
// never needed again), so the correctly-rounded cold-start code stays out of the loop body.
// Params: GaussParams, or a type with the same field names whose fields convert to their values where they are read
// (kernels.hip KernargGaussParams: parameters re-read from the kernarg segment instead of being held in SGPRs).
//
// kPair: LANE-PAIR form for small batches, where a candidate's chain — not the device's throughput — is what the run
// waits for. Lanes 2k and 2k+1 hold the SAME candidate and the same state; each runs ONE of the two independent
// Lagrange corrections (even lane: t1 <- t2, odd lane: t3 <- t2), the two exchange their (v, f, g, chi) through the
// quad permute, and both evaluate the identical tail. Every operation is the one the single-lane form performs, on the
// same operands: same bits. The iteration's chain loses one of its two universal-Kepler solves (~ a quarter of it).
__device__ __forceinline__ double pair_swap(double x) {  // the partner lane's x (lanes 2k <-> 2k+1)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// The parameters the TAIL of an iteration reads (everything behind the two Kepler solves). For GaussParams that is the
// object itself; a kernel may pass a type whose tail() fetches them at that point instead (kernels.hip SplitParams: one
// block load from the kernarg segment per iteration, so that ~20 scalar registers are not held — or spilled into
// VGPR lanes and read back with v_readlane — across the solves' loops).
__device__ __forceinline__ const GaussParams& tail_params(const GaussParams& P) { return P; }
template <class Params>
__device__ __forceinline__ auto tail_params(const Params& P) -> decltype(P.tail()) { return P.tail(); }

#ifndef OUTFIT_K2B_UNROLL_SIDES   // measurement switch: two copies of the Kepler solver instead of a rolled loop over the sides
#define OUTFIT_K2B_UNROLL_SIDES 0
#endif
template <bool kFast, bool kWarmOnly = kFast, class Params = GaussParams, bool kPair = false>
__device__ __forceinline__ CorrStep correction_iteration_impl(const Mat3& obs, const Mat3& S, const Mat3& Sinv,
                                                              double t_mid, double dt01, double dt21,
                                                              const Params& P, CorrectionState& st, fdiv::Guard* G) {
  KSTAT(3);
  st.it += 1;
  const bool last = st.it >= P.newton_max_it;
  // An iteration is a pure function of (pos, vel, chi01, chi21, has01, has21). If it ends in a
  // `continue` and leaves that tuple bit for bit unchanged, every remaining iteration repeats it
  // and the loop ends with Some(current state) (gauss.rs:1417): return that now. This is the
  // common fate of spurious roots, which get stuck on rho2 < min_rho2 until itmax.
  // A `continue` before the chi update changes nothing at all, hence `rare` = Some directly.
  const CorrStep rare = kCorrSome;
  const Vec3 r1 = column(st.pos, 0), r2 = column(st.pos, 1), r3 = column(st.pos, 2);
  if (!st.vc.ok) return rare;
  if (G) G->see(st.vc.r2);  // divided by through its carried reciprocal (velocity_finish)
  const bool had_both = st.has01 && st.has21;
  const long long chi01_old = __double_as_longlong(st.chi01), chi21_old = __double_as_longlong(st.chi21);
  // the two Lagrange corrections (t1<-t2 and t3<-t2) are independent; a rolled loop keeps ONE
  // copy of the universal-Kepler solver in the instruction stream
  VelCorr a, b;
  if constexpr (kPair) {
    const bool odd = (__lane_id() & 1u) != 0u;
    const double dt = odd ? dt21 : dt01;
    const double chi = odd ? st.chi21 : st.chi01;
    const KeplerProblem p = velocity_problem(st.vc, dt, P.kepler_eps, P.math_mode);
    const KeplerSolution sol = kWarmOnly ? run_newton(p, chi) : solve_kepuni(p, odd ? st.has21 : st.has01, chi);
    const Vec3 x1 = {odd ? r3.x : r1.x, odd ? r3.y : r1.y, odd ? r3.z : r1.z};
    VelCorr mine = velocity_finish(sol, x1, r2, st.vc.r2, dt);
    if (!mine.ok) { mine.v = {0.0, 0.0, 0.0}; mine.f = mine.g = mine.chi = 0.0; }  // defined values to exchange
    VelCorr other;
    other.ok = __builtin_amdgcn_mov_dpp((int)mine.ok, 0xB1, 0xF, 0xF, false) != 0;
    other.v = {pair_swap(mine.v.x), pair_swap(mine.v.y), pair_swap(mine.v.z)};
    other.f = pair_swap(mine.f); other.g = pair_swap(mine.g); other.chi = pair_swap(mine.chi);
    a = odd ? other : mine;
    b = odd ? mine : other;
  } else {
#if OUTFIT_K2B_UNROLL_SIDES
#pragma unroll
#else
#pragma unroll 1
#endif
  for (int side = 0; side < 2; ++side) {
    const double dt = side ? dt21 : dt01;
    const double chi = side ? st.chi21 : st.chi01;
    const KeplerProblem p = velocity_problem(st.vc, dt, P.kepler_eps, P.math_mode);
    const KeplerSolution sol = kWarmOnly ? run_newton(p, chi, G) : solve_kepuni(p, side ? st.has21 : st.has01, chi, G);
#ifdef OUTFIT_K2B_STATS  // histogram of Newton rounds per solve: first iteration of a candidate (cold start) / later ones
    atomicAdd(&g_kstat[(st.it == 1 ? 24 : 16) + (sol.rounds < 7 ? sol.rounds : 7)], 1ull);
#endif
    const VelCorr vc = velocity_finish(sol, side ? r3 : r1, r2, st.vc.r2, dt, G, st.vc.inv_r2);
    if (side) b = vc; else a = vc;
  }
  }
  KSTAT(4);
  if (!(a.ok && b.ok)) return rare;
  const auto& T = tail_params(P);
  st.has01 = true; st.chi01 = a.chi;
  st.has21 = true; st.chi21 = b.chi;
  const bool fixed_point = had_both && __double_as_longlong(a.chi) == chi01_old &&
                           __double_as_longlong(b.chi) == chi21_old;
  const CorrStep skip = (fixed_point || last) ? kCorrSome : kCorrContinue;
  if (!isfinite(a.g) || !isfinite(b.g)) return skip;
  const Vec3 nv = {(a.v.x + b.v.x) * 0.5, (a.v.y + b.v.y) * 0.5, (a.v.z + b.v.z) * 0.5};
  const double fl = a.f * b.g - b.f * a.g;
  if (!isfinite(fl) || fabs(fl) < kEps) return skip;
  const double inv_f = 1.0 / fl;
  const Vec3 c = {b.g * inv_f, -1.0, -a.g * inv_f};
  Mat3 np;
  double ne;
  KSTAT(5);
  if (!position_and_epoch(obs, S, Sinv, t_mid, c, T.min_rho2_au, np, ne, G)) return skip;  // common
  KSTAT(6);
  const Vec3 nr2 = column(np, 1);
  const EccControl ec = eccentricity_control(nr2, nv, T.max_perihelion_au, T.max_ecc, G, T.peri_lo, T.peri_hi);
  if (!ec.some || !ec.accepted) return kFast ? kCorrSlow : kCorrNone;
  // denom = |P_new|_F: `!isfinite(denom) || denom <= eps` decided on its square (kepler.hpp kFrobSqMax)
  const double denom_sq = frob_sq(np);
  if (!isfinite(denom_sq) || denom_sq <= kFrobSqMax) return skip;
  Mat3 d;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) d.m[r][k] = np.m[r][k] - st.pos.m[r][k];
  // rel = |P_new - P|_F / |P_new|_F is only compared with newton_eps
  const bool converged = rel_converged(frob_sq(d), denom_sq, T.newton_eps, T.eps2_lo, T.eps2_hi);
  st.pos = np;
  st.vel = nv;
  st.epoch = ne;
  st.vc = velocity_common_from(ec, nr2, nv, G);
  if (converged) return kCorrSome;
  return last ? kCorrSome : kCorrContinue;
}
// The throughput form of the fast kernel runs the iteration's divisions by constants and by shared denominators in their
// short forms under a guard (fastdiv.hpp) and asks for a replay with plain divisions if the guard saw an operand outside
// the window in which the short forms are the bits of `/`; every other caller gets the plain divisions.
#ifndef OUTFIT_GUARDED_DIVISIONS
#define OUTFIT_GUARDED_DIVISIONS 1
#endif
template <bool kFast, bool kWarmOnly = kFast, class Params = GaussParams, bool kPair = false>
__device__ __forceinline__ CorrStep correction_iteration(const Mat3& obs, const Mat3& S, const Mat3& Sinv,
                                                         double t_mid, double dt01, double dt21,
                                                         const Params& P, CorrectionState& st) {
  if constexpr (kFast && !kPair && OUTFIT_GUARDED_DIVISIONS) {
    fdiv::Guard guard;
    guard.reset();
    const CorrStep step = correction_iteration_impl<kFast, kWarmOnly, Params, kPair>(obs, S, Sinv, t_mid, dt01, dt21, P, st, &guard);
    return guard.ok() ? step : kCorrRedo;
  } else {
    return correction_iteration_impl<kFast, kWarmOnly, Params, kPair>(obs, S, Sinv, t_mid, dt01, dt21, P, st, nullptr);
  }
}

}  // namespace outfit
