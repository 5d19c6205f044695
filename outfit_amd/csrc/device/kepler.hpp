// kepler.hpp — device-side universal-variable two-body routines (gfx950, fp64).
//
// Replaces, on the GPU, the reference's src/kepler/ as used by the Gauss
// correction loop (gauss.rs:1325-1344 -> velocity.rs:94-211 -> params.rs:130 ->
// newton_solver.rs:151-352 -> stumpff.rs:78-297) and by
// kepler::propagate_universal (propagation.rs:114-207).
//
// Everything lives in registers: one lane owns one Kepler problem. The
// arithmetic keeps the reference's operation order and the translation unit is
// compiled with -ffp-contract=off, so apart from libm-vs-ocml transcendentals
// the results follow the Rust code to the last bit.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "crmath.hpp"
#include "glibcmath.hpp"
#include "glibcmath2.hpp"
#include "glibcmath3.hpp"
#include "fastdiv.hpp"

namespace outfit {

#ifdef OUTFIT_K2B_STATS  // diagnostic build: (wave passes, active lanes) at a few program points of the correction loop
__device__ unsigned long long g_kstat[32];
#define KSTAT(slot)                                                                 \
  do {                                                                              \
    const unsigned long long m__ = __ballot(true);                                  \
    if (__lane_id() == (unsigned)__ffsll((long long)m__) - 1u) {                    \
      atomicAdd(&g_kstat[2 * (slot)], 1ull);                                        \
      atomicAdd(&g_kstat[2 * (slot) + 1], (unsigned long long)__popcll(m__));       \
    }                                                                               \
  } while (0)
#else
#define KSTAT(slot) do { } while (0)
#endif

#ifdef OUTFIT_K2A_STATS  // diagnostic build: (wave passes, active lanes) at a few program points of k_gauss_roots
__device__ unsigned long long g_k2a_stat[16];
#define K2ASTAT(slot)                                                               \
  do {                                                                              \
    const unsigned long long m__ = __ballot(true);                                  \
    if (__lane_id() == (unsigned)__ffsll((long long)m__) - 1u) {                    \
      atomicAdd(&g_k2a_stat[2 * (slot)], 1ull);                                     \
      atomicAdd(&g_k2a_stat[2 * (slot) + 1], (unsigned long long)__popcll(m__));    \
    }                                                                               \
  } while (0)
#else
#define K2ASTAT(slot) do { } while (0)
#endif

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 6.283185307179586476925286766559;  // constants.rs:31
constexpr double kEps = 2.220446049250313e-16;                // f64::EPSILON
constexpr double kGaussGrav = 0.01720209895;                  // constants.rs:70
constexpr double kMu = kGaussGrav * kGaussGrav;               // constants.rs:73
constexpr double kAU = 149597870.7;                           // constants.rs:37
constexpr double kVlightAU = 2.99792458e5 / kAU * 86400.0;    // constants.rs:79

// a / c for the reference's physical constants through a guard (fastdiv.hpp: 3 operations + 2 of bookkeeping instead of
// the 11-instruction, 58-cycle division sequence, profiles/round04_issue_cost_gfx950.json); plain `/` without one.
// sqrt(kMu) is folded by the compiler (correctly rounded: the value the device's own square root returns).
__device__ __forceinline__ double div_by_sqrt_mu(double a, fdiv::Guard* g) {
  const double c = sqrt(kMu);
  return fdiv::div_const(a, c, 1.0 / c, g);
}
__device__ __forceinline__ double div_by_mu(double a, fdiv::Guard* g) { return fdiv::div_const(a, kMu, 1.0 / kMu, g); }
__device__ __forceinline__ double div_by_vlight(double a, fdiv::Guard* g) {
  return fdiv::div_const(a, kVlightAU, 1.0 / kVlightAU, g);
}

struct Vec3 {
  double x, y, z;
};

__device__ __forceinline__ double dot(const Vec3& a, const Vec3& b) {
  // nalgebra 3-vector dot: (a0*b0 + a1*b1) + a2*b2
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
__device__ __forceinline__ double norm(const Vec3& a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ Vec3 cross(const Vec3& a, const Vec3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);  // f64::clamp (NaN passes through)
}

// f64::clamp(x, -m, m) for m > 0: one compare on the magnitude and a sign transfer instead of two compares and two
// selects per word — x < -m gives -m, x > m gives m, anything else (NaN included: the compare is false) passes through
__device__ __forceinline__ double clamp_symmetric(double x, double m) {
  return fabs(x) > m ? copysign(m, x) : x;
}

// f64::rem_euclid(2*pi)   (kepler/angles.rs:10-12)
__device__ __forceinline__ double principal_angle(double a) {
  double r = fmod(a, kTwoPi);
  return r < 0.0 ? r + kTwoPi : r;
}

struct Stumpff {
  double s0, s1, s2, s3;
};

// kepler/stumpff.rs:78-297. Power series while |alpha psi^2| < 100, otherwise
// halve psi, expand s0/s1 and apply the duplication formulas.
__device__ __forceinline__ Stumpff s_funct(double psi, double alpha, fdiv::Guard* guard = nullptr) {
  constexpr double tol = 100.0 * kEps, ovf = 1.0 / kEps;
  if (psi == 0.0) return {1.0, 0.0, 0.0, 0.0};
  const double psi2 = psi * psi;
  const double beta = alpha * psi2;
  if (fabs(beta) < 100.0) {
    double s2 = 0.5 * psi2, t2 = s2;
    double s3 = fdiv::div_const(s2 * psi, 3.0, 1.0 / 3.0, guard), t3 = s3;  // (s2 psi) / 3
    // term j divides by the integer constants (2j+1)(2j+2) and (2j+2)(2j+3): exact quotient
    // from the tabulated correctly-rounded reciprocal (fastdiv.hpp), same bits as `/`
    const bool fast = fdiv::in_window(beta);
    auto term = [&](int j, auto fast_tag) -> bool {  // adds term j; true = the series has converged (or overflowed)
      KSTAT(1);
      const double d2 = fdiv::kSeries.d2[j], d3 = fdiv::kSeries.d3[j];
      double x2, x3;
      if (decltype(fast_tag)::value) {
        // (every constant loaded into its scalar pair where it is used, see fdiv::scalar_const: held across the loop the 32
        //  words of the eight unrolled terms' reciprocals overflow the scalar file and come back through v_readlane)
        const double rc2 = fdiv::scalar_const(fdiv::kSeries.r2[j]), rc3 = fdiv::scalar_const(fdiv::kSeries.r3[j]);
        const double q2 = beta * rc2, q3 = beta * rc3;
        x2 = fma(fma(-q2, fdiv::scalar_const(d2), beta), rc2, q2);
        x3 = fma(fma(-q3, fdiv::scalar_const(d3), beta), rc3, q3);
      } else {
        x2 = beta / d2;
        x3 = beta / d3;
      }
      t2 *= x2;
      s2 += t2;
      t3 *= x3;
      s3 += t3;
      const double a2 = fabs(t2), a3 = fabs(t3);
      if (decltype(fast_tag)::value) {
        // Inside the window neither term can be NaN: beta is finite, non-zero and normal, so psi is finite and non-zero,
        // t2 starts finite, t3 starts finite or +-Inf, and every factor x is finite and non-zero (Inf * x stays Inf,
        // 0 * x stays 0). Without NaNs (a2 < tol && a3 < tol) is max(a2, a3) < tol and (a2 > ovf || a3 > ovf) is
        // max(a2, a3) > ovf: one v_max_f64 and two compares instead of four compares, the same decisions.
        const double m = fmax(a2, a3);
        return m < tol || m > ovf;
      }
      return (a2 < tol && a3 < tol) || a2 > ovf || a3 > ovf;
    };
    using yes = std::true_type;
    using no = std::false_type;
#ifndef OUTFIT_SFUNCT_PER_TERM_WINDOW
    // The window test is hoisted out of the terms (one branch per evaluation instead of one per term: the compiler
    // does not unswitch it, and each per-term branch cost ~8 scalar instructions of exec bookkeeping). Outside the
    // window (beta == 0 for a parabolic alpha, or denormal) the plain divisions run, rolled: the same operations.
    if (fast) {
      // The first terms are unrolled so that their four constants are literals: inside a rolled loop each
      // term paid two scalar table loads (address arithmetic + memory latency) for six arithmetic
      // instructions. The orbit-sized arguments of this path converge within 5-6 terms.
      constexpr int kUnrolled = 8;
      bool done = false;
#pragma unroll
      for (int j = 1; j <= kUnrolled; ++j) {
        if (!done) done = term(j, yes{});
      }
      if (!done) {
#pragma unroll 1
        for (int j = kUnrolled + 1; j <= 70; ++j) {
          if (term(j, yes{})) break;
        }
      }
    } else {
#pragma unroll 1
      for (int j = 1; j <= 70; ++j) {
        if (term(j, no{})) break;
      }
    }
#else
    constexpr int kUnrolled = 8;
    bool done = false;
#pragma unroll
    for (int j = 1; j <= kUnrolled; ++j) {
      if (!done) done = fast ? term(j, yes{}) : term(j, no{});
    }
    if (!done) {
#pragma unroll 1
      for (int j = kUnrolled + 1; j <= 70; ++j) {
        if (fast ? term(j, yes{}) : term(j, no{})) break;
      }
    }
#endif
    return {1.0 + alpha * s2, psi + alpha * s3, s2, s3};
  }
  double rp = psi, rb = beta;
  int nh = 0;
  while (fabs(rb) >= 100.0 && nh < 30) { rp *= 0.5; rb *= 0.25; ++nh; }
  double s0 = 1.0, s1 = rp, t0 = 1.0, t1 = rp;
  for (int j = 1; j <= 70; ++j) {
    t0 *= rb / ((double)(2 * j - 1) * (double)(2 * j));
    s0 += t0;
    if (fabs(t0) < tol || fabs(t0) > ovf) break;
  }
  for (int j = 1; j <= 70; ++j) {
    t1 *= rb / ((double)(2 * j) * (double)(2 * j + 1));
    s1 += t1;
    if (fabs(t1) < tol || fabs(t1) > ovf) break;
  }
  for (int i = 0; i < nh; ++i) {
    const double c = s0, s = s1;
    s0 = 2.0 * c * c - 1.0;
    s1 = 2.0 * c * s;
  }
  return {s0, s1, (s0 - 1.0) / alpha, (s1 - psi) / alpha};
}

// Math mode of the elementary functions that feed the correction loop and the returned elements
// (include/outfit_hip.h OUTFIT_MATH_*): 0 = correctly rounded (crmath.hpp), 1 = sin/cos/sinh/cosh with
// glibc 2.35's bits (glibcmath.hpp), the rest correctly rounded. Always wave-uniform (a kernel argument).
constexpr int kMathCr = 0, kMathGlibc = 1;
__device__ __forceinline__ void sincos_mode(int mode, double x, double& s, double& c) {
  if (mode == kMathGlibc) glm::sincos(x, s, c); else crm::sincos(x, s, c);
}
__device__ __forceinline__ void sinhcosh_mode(int mode, double x, double& s, double& c) {
  if (mode == kMathGlibc) glm::sinhcosh(x, s, c); else crm::sinhcosh(x, s, c);
}

struct KeplerProblem {
  double dt, r0, sig0, alpha, e0, convergency;
  int math_mode;
  // SolverParams (kepler/params.rs:24-44); the IOD path always uses the defaults below (compile-time constants there)
  int max_iter_prelim = 20;   // max_iter_prelim_kepuni
  int parabolic_newton = 0;   // ParabolicPrelimMethod::Newton instead of Cardano
};

// kepler/prelim_kepler/prelim_elliptic.rs:9-134
__device__ __forceinline__ double prelim_elliptic(const KeplerProblem& p) {
  const double contr = p.convergency;
  const double a0 = -1.0 / p.alpha;
  const double n = sqrt(kMu) * sqrt(-(p.alpha * p.alpha * p.alpha));
  if (p.e0 < contr) return n * p.dt / sqrt(-p.alpha);
  const double c = (1.0 - p.r0 / a0) / p.e0;
  double u0;
  if (fabs(c) <= 1.0) u0 = p.math_mode == kMathGlibc ? glm::acos(c) : crm::acos(c);  // glibc's bits, or correctly rounded (crmath.hpp)
  else if (c >= 1.0) u0 = 0.0;
  else u0 = kPi;
  if (p.sig0 < 0.0) u0 = -u0;
  u0 = principal_angle(u0);
  // `u0.sin()` is an ISOLATED sine in the reference (prelim_elliptic.rs:97): the standalone libm entry point, whose
  // bits differ from sincos's in glibc (glibcmath.hpp sin1); correctly rounded otherwise
  double su0, cu0;
  if (p.math_mode == kMathGlibc) su0 = glm::sin1(u0); else crm::sincos(u0, su0, cu0);
  const double m0 = principal_angle(u0 - p.e0 * su0);
  const double m = m0 + n * p.dt;
  double u = m;
  // not unrolled: the body inlines the correctly-rounded functions with their second steps (~1-2 k instructions);
  // 20 copies of it made this the largest code of the library for no measurable gain
#pragma unroll 1
  for (int i = 0; i < p.max_iter_prelim; ++i) {
    K2ASTAT(4);
    double su, cu;
    sincos_mode(p.math_mode, u, su, cu);
    const double res = u - p.e0 * su - m;
    const double der = 1.0 - p.e0 * cu;
    const double step = -res / der;
    u += step;
    if (fabs(step) < contr * 1e3) break;
  }
  return (u - u0) / sqrt(-p.alpha);
}

// kepler/prelim_kepler/prelim_hyperbolic.rs:45-141 (loop exit tests |F|, quirk Q5)
__device__ __forceinline__ double prelim_hyperbolic(const KeplerProblem& p) {
  const double a0 = -1.0 / p.alpha;
  const double n = sqrt(kMu) * sqrt(p.alpha * p.alpha * p.alpha);
  const double c = (1.0 - p.r0 / a0) / p.e0;
  // log / sinh / cosh correctly rounded (crmath.hpp) or glibc's bits (glibcmath*.hpp): this value seeds the chaotically sensitive loop
  double f0 = (c > 1.0) ? (p.math_mode == kMathGlibc ? glm::log(c + sqrt(c * c - 1.0)) : crm::log(c + sqrt(c * c - 1.0))) : 0.0;
  if (p.sig0 < 0.0) f0 = -f0;
  double sh, ch;
  sinhcosh_mode(p.math_mode, f0, sh, ch);
  const double m = (p.e0 * sh - f0) + n * p.dt;
  double f = 0.0;
  // not unrolled: the body inlines the correctly-rounded functions with their second steps (~1-2 k instructions);
  // 20 copies of it made this the largest code of the library for no measurable gain
#pragma unroll 1
  for (int i = 0; i < p.max_iter_prelim; ++i) {
    if (fabs(f) < 15.0) {
      sinhcosh_mode(p.math_mode, f, sh, ch);
      const double res = p.e0 * sh - f - m;
      const double der = p.e0 * ch - 1.0;
      const double cand = f + (-res / der);
      f = (f * cand < 0.0) ? f / 2.0 : cand;
    } else {
      f /= 2.0;
    }
    if (fabs(f) < p.convergency * 1e3) break;
  }
  return (f - f0) / sqrt(p.alpha);
}

__device__ __forceinline__ void cubic_res_der(double psi, double r0, double sig0, double smdt,
                                              double& res, double& der) {
  res = (psi * psi * psi) / 6.0 + sig0 / 2.0 * (psi * psi) + r0 * psi - smdt;
  der = (psi * psi) / 2.0 + sig0 * psi + r0;
}

// kepler/prelim_kepler/prelim_parabolic.rs:264-477 (Cardano; reached only for alpha == 0)
__device__ __forceinline__ double prelim_parabolic_cardano(const KeplerProblem& p) {
  const double r0 = p.r0, sig0 = p.sig0;
  const double smdt = sqrt(kMu) * p.dt;
  if (p.dt == 0.0) return 0.0;
  const double lead = 1.0 / 6.0;
  const double qc = (sig0 / 2.0) / lead, lc = r0 / lead, cc = -smdt / lead;
  const double shift = qc / 3.0;
  const double dl = lc - qc * shift;
  const double dc = 2.0 * (shift * shift * shift) - lc * shift + cc;
  const double half = dc / 2.0;
  const double third = dl / 3.0;
  const double disc = half * half + third * third * third;
  double c0, c1 = 0.0, c2 = 0.0;
  int nc = 1;
  if (disc > 0.0) {
    const double sq = sqrt(disc);
    c0 = (cbrt(-half + sq) + cbrt(-half - sq)) - shift;
  } else {
    const double arg = clampd((3.0 * dc) / (2.0 * dl) * sqrt(-3.0 / dl), -1.0, 1.0);
    const double base = acos(arg) / 3.0;
    const double amp = 2.0 * sqrt(-dl / 3.0);
    c0 = amp * cos(base) - shift;
    c1 = amp * cos(base - 2.0 * kPi / 3.0) - shift;
    c2 = amp * cos(base - 4.0 * kPi / 3.0) - shift;
    nc = 3;
  }
  // select_physical_root: derivative >= 0 first, then closest to smdt/r0 (first minimum wins)
  const double lin = smdt / r0;
  double res, d0, d1 = -1.0, d2 = -1.0;
  cubic_res_der(c0, r0, sig0, smdt, res, d0);
  if (nc == 3) {
    cubic_res_der(c1, r0, sig0, smdt, res, d1);
    cubic_res_der(c2, r0, sig0, smdt, res, d2);
  }
  const bool any_mono = (d0 >= 0.0) || (nc == 3 && (d1 >= 0.0 || d2 >= 0.0));
  double sel = 0.0;
  bool have = false;
  auto consider = [&](double cand, double der) {
    if (any_mono && !(der >= 0.0)) return;
    if (!have || fabs(cand - lin) < fabs(sel - lin)) { sel = cand; have = true; }
  };
  consider(c0, d0);
  if (nc == 3) { consider(c1, d1); consider(c2, d2); }
  double psi = sel;
  for (int i = 0; i < 2; ++i) {
    double der;
    cubic_res_der(psi, r0, sig0, smdt, res, der);
    if (der == 0.0 || !isfinite(der)) break;
    psi -= res / der;
  }
  return psi;
}

// kepler/prelim_kepler/prelim_parabolic.rs:198-246 (ParabolicPrelimMethod::Newton; Cardano when the cubic is not monotonic)
__device__ __forceinline__ double prelim_parabolic_newton(const KeplerProblem& p) {
  const double smdt = sqrt(kMu) * p.dt;
  if (p.dt == 0.0) return 0.0;
  if (p.sig0 * p.sig0 > 2.0 * p.r0) return prelim_parabolic_cardano(p);
  double psi = smdt / p.r0;
  for (int i = 0; i < p.max_iter_prelim; ++i) {
    double res, der;
    cubic_res_der(psi, p.r0, p.sig0, smdt, res, der);
    if (!isfinite(der) || fabs(der) < 10.0 * kEps) { psi *= 0.5; continue; }
    const double mx = 2.0 * (1.0 + fabs(psi));
    const double step = clampd(-res / der, -mx, mx);
    psi += step;
    if (fabs(step) < p.convergency) break;
  }
  return psi;
}
__device__ __forceinline__ double prelim_parabolic(const KeplerProblem& p) {
  return p.parabolic_newton ? prelim_parabolic_newton(p) : prelim_parabolic_cardano(p);
}

// kepler/params.rs:185-191
__device__ __forceinline__ double prelim_kepuni(const KeplerProblem& p) {
  if (p.alpha < 0.0) return prelim_elliptic(p);
  if (p.alpha > 0.0) return prelim_hyperbolic(p);
  return prelim_parabolic(p);
}

// prelim_kepuni for the TWO solves of one correction iteration's cold start (t1 <- t2 and t3 <- t2): the problems differ
// in dt only, and everything of the elliptic / hyperbolic cold start that does not depend on dt — acos(c) or ln(c + ..),
// the sine / hyperbolic pair at the start value, the reference mean anomaly — is the same value in both calls of the
// reference (prelim_elliptic.rs:72-134, prelim_hyperbolic.rs:45-141); it is evaluated once here. Same bits as two calls
// of prelim_kepuni.
__device__ __forceinline__ void prelim_kepuni_pair(const KeplerProblem& p, double dt_a, double dt_b, double& out_a,
                                                   double& out_b) {
  if (p.alpha < 0.0) {
    const double contr = p.convergency;
    const double a0 = -1.0 / p.alpha;
    const double n = sqrt(kMu) * sqrt(-(p.alpha * p.alpha * p.alpha));
    if (p.e0 < contr) {
      out_a = n * dt_a / sqrt(-p.alpha);
      out_b = n * dt_b / sqrt(-p.alpha);
      return;
    }
    const double c = (1.0 - p.r0 / a0) / p.e0;
    double u0;
    if (fabs(c) <= 1.0) u0 = p.math_mode == kMathGlibc ? glm::acos(c) : crm::acos(c);
    else if (c >= 1.0) u0 = 0.0;
    else u0 = kPi;
    if (p.sig0 < 0.0) u0 = -u0;
    u0 = principal_angle(u0);
    double su0, cu0;
    if (p.math_mode == kMathGlibc) su0 = glm::sin1(u0); else crm::sincos(u0, su0, cu0);
    const double m0 = principal_angle(u0 - p.e0 * su0);
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
      const double m = m0 + n * (side ? dt_b : dt_a);
      double u = m;
#pragma unroll 1
      for (int i = 0; i < p.max_iter_prelim; ++i) {
        K2ASTAT(4);
        double su, cu;
        sincos_mode(p.math_mode, u, su, cu);
        const double res = u - p.e0 * su - m;
        const double der = 1.0 - p.e0 * cu;
        const double step = -res / der;
        u += step;
        if (fabs(step) < contr * 1e3) break;
      }
      const double r = (u - u0) / sqrt(-p.alpha);
      if (side) out_b = r; else out_a = r;
    }
    return;
  }
  if (p.alpha > 0.0) {
    const double a0 = -1.0 / p.alpha;
    const double n = sqrt(kMu) * sqrt(p.alpha * p.alpha * p.alpha);
    const double c = (1.0 - p.r0 / a0) / p.e0;
    double f0 = (c > 1.0) ? (p.math_mode == kMathGlibc ? glm::log(c + sqrt(c * c - 1.0)) : crm::log(c + sqrt(c * c - 1.0))) : 0.0;
    if (p.sig0 < 0.0) f0 = -f0;
    double sh0, ch0;
    sinhcosh_mode(p.math_mode, f0, sh0, ch0);
    const double m_base = p.e0 * sh0 - f0;
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
      const double m = m_base + n * (side ? dt_b : dt_a);
      double f = 0.0, sh, ch;
#pragma unroll 1
      for (int i = 0; i < p.max_iter_prelim; ++i) {
        if (fabs(f) < 15.0) {
          sinhcosh_mode(p.math_mode, f, sh, ch);
          const double res = p.e0 * sh - f - m;
          const double der = p.e0 * ch - 1.0;
          const double cand = f + (-res / der);
          f = (f * cand < 0.0) ? f / 2.0 : cand;
        } else {
          f /= 2.0;
        }
        if (fabs(f) < p.convergency * 1e3) break;
      }
      const double r = (f - f0) / sqrt(p.alpha);
      if (side) out_b = r; else out_a = r;
    }
    return;
  }
  KeplerProblem q = p;
  q.dt = dt_a;
  out_a = prelim_parabolic(q);
  q.dt = dt_b;
  out_b = prelim_parabolic(q);
}

struct KeplerSolution {
  double psi;
  Stumpff s;
  bool ok;
#ifdef OUTFIT_K2B_STATS
  int rounds = 0;
#endif
};

// kepler/newton_solver.rs:151-161,240-352 (SolverKind::NewtonRaphson)
// run_newton (newton_solver.rs:240-352) from a given start value
__device__ __forceinline__ KeplerSolution run_newton(const KeplerProblem& p, double psi, fdiv::Guard* guard = nullptr) {
  const double sqrt_mu = sqrt(kMu);
  const double res_tol = 10.0 * kEps * (1.0 + fabs(sqrt_mu * p.dt));
  // Every exit of the reference's loop returns the CURRENT psi with the Stumpff values computed LAST (the residual exit:
  // this round's; the absolute-step exit, quirk Q3: the new psi with this round's values; the relative-step exit: the
  // new psi and values recomputed at it), so the loop's own variables are the result — no copies into a second set of
  // registers at every exit (round 5: 14 v_mov_b64 per round in the compiled loop).
  KeplerSolution out;
  out.ok = false;
  out.s = {1.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < 50; ++it) {
    KSTAT(0);
#ifdef OUTFIT_K2B_STATS
    out.rounds = it + 1;
#endif
    if (!isfinite(psi)) { psi = 0.5; continue; }
    out.s = s_funct(psi, p.alpha, guard);
    const double residual = p.r0 * out.s.s1 + p.sig0 * out.s.s2 + out.s.s3 - sqrt_mu * p.dt;
    const double derivative = p.r0 * out.s.s0 + p.sig0 * out.s.s1 + out.s.s2;
    if (fabs(residual) <= res_tol) { out.ok = true; break; }
    if (!isfinite(derivative) || fabs(derivative) < 10.0 * kEps) { psi *= 0.5; continue; }
    const double mx = 2.0 * (1.0 + fabs(psi));
    const double step = clamp_symmetric(-residual / derivative, mx);
    const double cand = psi + step;
    psi = (cand * psi < 0.0) ? 0.5 * psi : cand;
    const double sa = fabs(step);
    if (sa <= p.convergency) {  // quirk Q3: new psi with the previous Stumpff values
      out.ok = true; break;
    }
    if (sa <= p.convergency * (1.0 + fabs(psi))) {
      KSTAT(2);
      out.s = s_funct(psi, p.alpha, guard); out.ok = true; break;
    }
  }
  out.psi = psi;
  return out;
}

// solve_kepuni_with_guess (newton_solver.rs:151-161): psi_guess or the cold-start value
__device__ __forceinline__ KeplerSolution solve_kepuni(const KeplerProblem& p, bool has_guess, double guess,
                                                       fdiv::Guard* guard = nullptr) {
  return run_newton(p, has_guess ? guess : prelim_kepuni(p), guard);
}

// kepler/brent_dekker_solver.rs:59-62
__device__ __forceinline__ double kepler_residual(double psi, const KeplerProblem& p) {
  const Stumpff s = s_funct(psi, p.alpha);
  return p.r0 * s.s1 + p.sig0 * s.s2 + s.s3 - sqrt(kMu) * p.dt;
}

// solve_kepuni_brent_dekker (kepler/brent_dekker_solver.rs:150-177,214-233,469-526): bracket around the start
// value (guess or prelim_kepuni), golden-ratio expansion of the end with the smaller |f| (<= 60 times), then
// Brent's iteration (inverse quadratic interpolation / secant, accepted only strictly between (3a+b)/4 and b and
// when it at least halves the reference step; else bisection), <= 100 iterations.
__device__ __forceinline__ KeplerSolution solve_kepuni_brent_dekker(const KeplerProblem& p, bool has_guess, double guess) {
  constexpr double kPhi = 1.618033988749895;
  KeplerSolution out;
  out.ok = false; out.psi = 0.0; out.s = {1.0, 0.0, 0.0, 0.0};
  const double psi0 = has_guess ? guess : prelim_kepuni(p);
  const double hw = fabs(psi0) > 1.0 ? fabs(psi0) : 1.0;
  double lo = psi0 - hw, hi = psi0 + hw;
  double flo = kepler_residual(lo, p), fhi = kepler_residual(hi, p);
  bool ok = false;
#pragma unroll 1
  for (int i = 0; i < 60; ++i) {
    if (flo * fhi <= 0.0) { ok = true; break; }
    const double w = hi - lo;
    if (fabs(flo) < fabs(fhi)) { lo = lo - kPhi * w; flo = kepler_residual(lo, p); }
    else { hi = hi + kPhi * w; fhi = kepler_residual(hi, p); }
  }
  if (!ok) return out;
  double a = lo, fa = kepler_residual(lo, p), b = hi, fb = kepler_residual(hi, p);
  if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  double c = a, fc = fa, prev_len = fabs(hi - lo);
  bool prev_bis = true;
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
    if (fabs(fb) <= p.convergency || 0.5 * fabs(b - a) <= p.convergency) {
      out.psi = b; out.s = s_funct(b, p.alpha); out.ok = true;
      return out;
    }
    double interp;
    if (fabs(fa - fc) > kEps && fabs(fb - fc) > kEps) {
      const double ta = a * fb * fc / ((fa - fb) * (fa - fc));
      const double tb = b * fa * fc / ((fb - fa) * (fb - fc));
      const double tc = c * fa * fb / ((fc - fa) * (fc - fb));
      interp = ta + tb + tc;
    } else {
      interp = b - fb * (b - a) / (fb - fa);
    }
    const double ref_len = prev_bis ? fabs(b - c) : prev_len;
    const double q3 = (3.0 * a + b) / 4.0;
    const bool inside = (q3 < b) ? (interp > q3 && interp < b) : (interp > b && interp < q3);
    const bool progress = fabs(interp - b) < 0.5 * ref_len;
    const bool take = inside && progress;
    const double next = take ? interp : 0.5 * (a + b);
    const double fnext = kepler_residual(next, p);
    prev_len = fabs(b - c);
    prev_bis = !take;
    c = b; fc = fb;
    if (fa * fnext < 0.0) { b = next; fb = fnext; }
    else { a = next; fa = fnext; }
    if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  }
  return out;
}

// UniversalKeplerParams::solve (kepler/params.rs:130-142); kind = OUTFIT_SOLVER_*
__device__ __forceinline__ KeplerSolution solve_kepuni_kind(const KeplerProblem& p, int kind, bool has_guess, double guess) {
  if (kind == 1) return solve_kepuni_brent_dekker(p, has_guess, guess);
  const KeplerSolution n = solve_kepuni(p, has_guess, guess);
  if (kind == 0 || n.ok) return n;
  return solve_kepuni_brent_dekker(p, has_guess, guess);
}

struct EccControl {
  bool some;      // false = None (zero angular momentum)
  bool accepted;
  double ecc, energy;
  double dist, h2;  // |r| and |r x v|^2 as computed on the way (reused by velocity_common_from)
  double inv_dist;  // with a guard: the refined reciprocal of |r| (fastdiv.hpp), reused for s2 / r2 in the next iteration
};

// Decisions that the reference takes on a SQUARE ROOT or on a QUOTIENT, taken here on the operands — same outcome for
// every input, without the 20-instruction sqrt or the division (round 5):
//  * sqrt is correctly rounded and monotone, so RN(sqrt(x)) <= c  <=>  x <= T(c), T(c) the largest double whose rounded
//    root is <= c (no ties: the midpoint above c squared is not a double). tools/sqrt_thresholds.py derives the two
//    constants with exact rational arithmetic and tests/test_device_math_host.py checks them against sqrt around T.
//      |r x v| <= 1e6 eps   <=>  |r x v|^2 <= kHnormSqMax      (velocity.rs:167-178, the angular-momentum guard)
//      |P|_F <= eps         <=>  |P|_F^2 <= kFrobSqMax         (gauss.rs:1392, the denominator guard)
//    and RN(sqrt(x)) == 0 <=> x == 0, isfinite(RN(sqrt(x))) <=> isfinite(x) for x >= 0 or NaN.
//  * a quotient against a threshold, RN(a / b) < t: decided by two products when a lies clearly on one side of b t
//    (margins of 2^-49, far above the rounding errors of the products and of the quotient), by the division itself in
//    the sliver between — `two_sided` below.
constexpr double kHnormSqMax = 0x1.d1a94a2p-65;          // T(1e6 * 2^-52) = (1e6 * 2^-52)^2 exactly
constexpr double kFrobSqMax = 0x1.0000000000001p-104;    // T(2^-52)

// RN(h2 / den) < peri_max (orb_elem.rs:296-299: the perihelion distance is only ever compared), for den = mu (1 + e) of
// a finite e. lo / hi = peri_max (1 -/+ 2^-49): with q = RN(h2 / den) = (h2 / den)(1 + t), |t| <= 2^-53, and the
// products below carrying 2^-53 each, h2 <= RN(den lo) puts h2 / den below peri_max (1 - 2^-50), hence q < peri_max by
// monotonicity of the rounding; h2 >= RN(den hi) likewise puts q above. NaN bounds (pre-test off) or a NaN operand
// fail both compares and the quotient decides.
__device__ __forceinline__ bool peri_below(double h2, double den, double peri_max, double lo, double hi) {
  if (h2 <= den * lo) return true;
  if (h2 >= den * hi) return false;
  return (h2 / den) < peri_max;
}

// orb_elem.rs:257-301. peri_lo / peri_hi: peri_max (1 -/+ 2^-49) from GaussParams::derive(), or NaN (the default: every
// compare with them is false and the division decides).
__device__ __forceinline__ EccControl eccentricity_control(const Vec3& r, const Vec3& v, double peri_max,
                                                  double ecc_max, fdiv::Guard* guard = nullptr,
                                                  double peri_lo = __builtin_nan(""), double peri_hi = __builtin_nan("")) {
  EccControl o;
  const double v2 = dot(v, v);
  const double dist = norm(r);
  const Vec3 h = cross(r, v);
  const double h2 = dot(h, h);
  o.dist = dist;
  o.h2 = h2;
  o.some = !(h2 == 0.0);   // h_norm == 0.0 with h_norm = sqrt(h2)
  const Vec3 vxh = cross(v, h);
  // 1 / |r| and mu / |r| (the energy, below) share their denominator
  const double inv_mu = 1.0 / kMu;
  const double one_mu[2] = {1.0, kMu};
  double over_d[2];
  o.inv_dist = fdiv::divn_shared<2>(one_mu, dist, over_d, guard);
  const double inv_d = over_d[0];
  const Vec3 lenz = {vxh.x * inv_mu - r.x * inv_d, vxh.y * inv_mu - r.y * inv_d,
                     vxh.z * inv_mu - r.z * inv_d};
  o.ecc = norm(lenz);
  // peri = h2 / (mu (1 + e)) is only ever compared with peri_max
  const bool peri_ok = peri_below(h2, kMu * (1.0 + o.ecc), peri_max, peri_lo, peri_hi);
  o.energy = v2 / 2.0 - over_d[1];
  o.accepted = (o.ecc < ecc_max) && peri_ok;
  return o;
}

struct VelCorr {
  bool ok;
  Vec3 v;
  double f, g, chi;
};

// kepler/velocity.rs:94-211, split in two. The reference calls velocity_correction_with_guess
// twice per iteration with the SAME (x2, v2) (gauss.rs:1325-1344): everything up to the
// Kepler problem's (r0, sig0, alpha, e0) is identical for both calls, so it is computed once.
// Quirk Q7: only the None of eccentricity_control fails.
struct VelCommon {
  bool ok;
  double r2, sig0, alpha, e0;
  double inv_r2;  // guarded path only: refined reciprocal of r2 (fastdiv.hpp refined_rcp), else unused
};
__device__ __forceinline__ VelCommon velocity_common(const Vec3& x2, const Vec3& v2, double peri_max,
                                                     double ecc_max) {
  VelCommon c;
  c.ok = false;
  c.r2 = norm(x2);
  c.inv_r2 = 0.0;
  c.sig0 = dot(x2, v2) / sqrt(kMu);
  c.alpha = 0.0; c.e0 = 0.0;
  const double hn = norm(cross(x2, v2));
  if (!isfinite(hn) || hn <= 1e6 * kEps) return c;
  const EccControl ec = eccentricity_control(x2, v2, peri_max, ecc_max);
  if (!ec.some) return c;
  c.alpha = 2.0 * ec.energy / kMu;
  c.e0 = ec.ecc;
  c.ok = true;
  return c;
}
// the same, from an eccentricity_control already evaluated on (x2, v2): norm(x2) and
// norm(x2 x v2) are the very values eccentricity_control computed (same operation order)
__device__ __forceinline__ VelCommon velocity_common_from(const EccControl& ec, const Vec3& x2, const Vec3& v2,
                                                          fdiv::Guard* guard = nullptr) {
  VelCommon c;
  c.r2 = ec.dist;
  c.inv_r2 = ec.inv_dist;
  c.sig0 = div_by_sqrt_mu(dot(x2, v2), guard);
  c.alpha = div_by_mu(2.0 * ec.energy, guard);
  c.e0 = ec.ecc;
  // isfinite(h_norm) && !(h_norm <= 1e6 eps) with h_norm = sqrt(h2): decided on h2 (see kHnormSqMax)
  c.ok = isfinite(ec.h2) && !(ec.h2 <= kHnormSqMax) && ec.some;
  return c;
}
__device__ __forceinline__ KeplerProblem velocity_problem(const VelCommon& c, double dt, double eps, int math_mode) {
  KeplerProblem p;
  p.dt = dt; p.r0 = c.r2; p.sig0 = c.sig0; p.alpha = c.alpha; p.e0 = c.e0; p.convergency = eps;
  p.math_mode = math_mode;
  return p;
}
// Lagrange f, g and the corrected velocity from a Kepler solution (velocity.rs:144-161)
// With a guard, `inv_r2` is the refined reciprocal of r2 (r2 itself recorded by the caller): the quotient's last three
// operations only, the same bits as `/` inside the guard's window.
__device__ __forceinline__ VelCorr velocity_finish(const KeplerSolution& sol, const Vec3& x1, const Vec3& x2,
                                                   double r2, double dt, fdiv::Guard* guard = nullptr, double inv_r2 = 0.0) {
  VelCorr o;
  o.ok = false;
  if (!sol.ok) return o;
  double s2_over_r2;
  if (guard) { guard->see(sol.s.s2); s2_over_r2 = fdiv::div_with_rcp(sol.s.s2, r2, inv_r2); }
  else s2_over_r2 = sol.s.s2 / r2;
  const double f = 1.0 - s2_over_r2;
  const double g = dt - div_by_sqrt_mu(sol.s.s3, guard);
  const double ga = fabs(g);
  if (!isfinite(ga) || ga < 100.0 * kEps * (1.0 + fabs(dt))) return o;
  const double num[3] = {(-f) * x2.x + x1.x, (-f) * x2.y + x1.y, (-f) * x2.z + x1.z};
  double q[3];
  fdiv::divn_shared<3>(num, g, q, guard);  // three components over the same g
  o.v = {q[0], q[1], q[2]};
  o.f = f; o.g = g; o.chi = sol.psi;
  o.ok = true;
  return o;
}
__device__ __forceinline__ VelCorr velocity_correction(const Vec3& x1, const Vec3& x2, const Vec3& v2, double dt,
                                              double peri_max, double ecc_max, bool has_guess,
                                              double chi_guess, double eps, int math_mode = kMathCr) {
  const VelCommon c = velocity_common(x2, v2, peri_max, ecc_max);
  if (!c.ok) { VelCorr o; o.ok = false; return o; }
  return velocity_finish(solve_kepuni(velocity_problem(c, dt, eps, math_mode), has_guess, chi_guess), x1, x2, c.r2, dt);
}

// kepler/propagation.rs:114-207 ; out = r1[3] v1[3] f g fdot gdot psi ; returns the status code
// documented at outfit_hip_propagate_universal (include/outfit_hip.h).
struct SolverSpec {  // SolverType (kepler/params.rs:55-73), flattened
  int kind = 0, parabolic_newton = 0, max_iter_prelim = 20;
  bool has_guess = false;
  double convergency = 100.0 * 2.220446049250313e-16, guess = 0.0;
};
__device__ __forceinline__ int propagate_universal(const Vec3& r, const Vec3& v, double t0, double t1,
                                          const SolverSpec& sv, double* out, int math_mode = kMathCr) {
  const double r0 = norm(r);
  if (r0 < kEps) return 1;
  const double v2 = dot(v, v);
  const double sqrt_mu = sqrt(kMu);
  KeplerProblem p;
  p.sig0 = dot(r, v) / sqrt_mu;
  p.alpha = (v2 - 2.0 * kMu / r0) / kMu;
  const Vec3 h = cross(r, v);
  double e = sqrt(1.0 + p.alpha * dot(h, h) / kMu);
  if (!(e >= 0.0)) e = 0.0;  // f64::max(NaN, 0.0) == 0.0
  p.e0 = e; p.r0 = r0; p.dt = t1 - t0; p.convergency = sv.convergency; p.math_mode = math_mode;
  p.max_iter_prelim = sv.max_iter_prelim; p.parabolic_newton = sv.parabolic_newton;
  const KeplerSolution s = solve_kepuni_kind(p, sv.kind, sv.has_guess, sv.guess);
  if (!s.ok) return sv.kind == 0 ? 2 : 4;
  const double r1 = r0 * s.s.s0 + p.sig0 * s.s.s1 + s.s.s2;
  if (r1 < kEps) return 3;
  const double f = 1.0 - s.s.s2 / r0;
  const double g = (r0 * s.s.s1 + p.sig0 * s.s.s2) / sqrt_mu;
  const double fd = -(sqrt_mu / (r0 * r1)) * s.s.s1;
  const double gd = 1.0 - s.s.s2 / r1;
  out[0] = f * r.x + g * v.x; out[1] = f * r.y + g * v.y; out[2] = f * r.z + g * v.z;
  out[3] = fd * r.x + gd * v.x; out[4] = fd * r.y + gd * v.y; out[5] = fd * r.z + gd * v.z;
  out[6] = f; out[7] = g; out[8] = fd; out[9] = gd; out[10] = s.psi;
  return 0;
}

}  // namespace outfit
