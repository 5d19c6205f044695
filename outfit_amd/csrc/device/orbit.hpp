// orbit.hpp — device-side state -> elements conversion and the per-observation
// astrometric residual, one lane per candidate.
//
// Replaces: compute_orbit_from_state (gauss.rs:906-923) -> ccek1
// (orb_elem.rs:58-226); OrbitalElements::to_equinoctial (orbit_type/mod.rs:399-443,
// equinoctial_element.rs:285-313, cometary_element.rs:224-289);
// EquinoctialElements::propagate_twobody (equinoctial_element.rs:809-867) with the
// `roots` 0.0.8 Newton–Raphson (equinoctial_element.rs:326-348);
// Observation::ephemeris_error (observation_ephemeris.rs:388-416) with
// topocentric_radec_and_partials :222-275 and correct_aberration_first_order
// (aberration.rs:139-145).
#pragma once
#include "kepler.hpp"
#include "screen_math.hpp"

namespace outfit {

constexpr double kCosEps = 9.174820620691818e-1;  // constants.rs:93-127
constexpr double kSinEps = 3.977771559319137e-1;

// ROT_EQUMJ2000_TO_ECLMJ2000 * v with nalgebra's gemv order (zero terms kept)
__device__ __forceinline__ Vec3 equ_to_ecl(const Vec3& v) {
  return {(1.0 * v.x + 0.0 * v.y) + 0.0 * v.z,
          (0.0 * v.x + kCosEps * v.y) + kSinEps * v.z,
          (0.0 * v.x + (-kSinEps) * v.y) + kCosEps * v.z};
}
__device__ __forceinline__ Vec3 ecl_to_equ(const Vec3& v) {
  return {(1.0 * v.x + 0.0 * v.y) + 0.0 * v.z,
          (0.0 * v.x + kCosEps * v.y) + (-kSinEps) * v.z,
          (0.0 * v.x + kSinEps * v.y) + kCosEps * v.z};
}
// The same product for FINITE vectors without the exact-zero terms (x + 0*y + 0*z == x,
// 0*x + c*y == c*y whenever x, y, z are finite): used in the residual loop, where a non-finite
// position ends in a non-finite score either way.
__device__ __forceinline__ Vec3 ecl_to_equ_finite(const Vec3& v) {
  return {v.x, kCosEps * v.y + (-kSinEps) * v.z, kSinEps * v.y + kCosEps * v.z};
}
// fmod(a, 2*pi), exactly, without the generic remainder loop when |a| < 4*pi: for
// 2*pi <= |a| < 4*pi the subtraction is exact (Sterbenz) and equals the IEEE remainder.
__device__ __forceinline__ double fmod_2pi(double a) {
  const double aa = fabs(a);
  if (aa < kTwoPi) return a;
  if (aa < 2.0 * kTwoPi) return a < 0.0 ? a + kTwoPi : a - kTwoPi;
  return fmod(a, kTwoPi);
}
__device__ __forceinline__ double principal_angle_fast(double a) {
  const double r = fmod_2pi(a);
  return r < 0.0 ? r + kTwoPi : r;
}

struct Elements {
  int kind;      // 0 Keplerian (a,e,i,Om,om,M) ; 1 Cometary (q,e,i,Om,om,nu)
  double epoch;
  double e[6];
};

// ---- sin and cos for the residual loop of K2c -------------------------------------------------------------
// The platform sincos costs ~100 VALU instructions per call here, ~40 of them v_mov: its polynomial constants are
// 64-bit literals, which this ISA can only feed to a VALU instruction from a register, and the compiler
// materialises them in VGPRs inside the loop. Constants that arrive as KERNEL ARGUMENTS sit in SGPRs and are read
// as operands directly (one scalar operand per instruction is all a Horner step needs): same ~1 ulp accuracy in ~40
// instructions. Cody-Waite by pi/2 in three parts (k * P1 and k * P2 exact for |k| < 2^20), fdlibm's minimax
// polynomials (k_sin.c / k_cos.c, degree 13 / 14 on [-pi/4, pi/4]). Only where a last place does not reach the
// returned elements: the RMS of a candidate (it moves by <= 1e-9 relative against the platform functions).
struct SinCosConsts {
  double two_over_pi, p1, p2, p3;
  double s1, s2, s3, s4, s5, s6;
  double c1, c2, c3, c4, c5, c6;
};
__host__ __device__ inline SinCosConsts sincos_consts() {
  return {0.6366197723675814, 1.5707963267341256, 6.077100506303966e-11, 2.0222662487959506e-21,
          -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
          2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
          4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
          -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
}
__device__ __forceinline__ void sincos_k(double x, const SinCosConsts& K, double& s, double& c) {
  if (!(fabs(x) < 1048576.0)) { ::sincos(x, &s, &c); return; }
  const double k = rint(x * K.two_over_pi);
  double r = fma(-k, K.p1, x);
  r = fma(-k, K.p2, r);
  r = fma(-k, K.p3, r);
  const int q = (int)k;
  const double z = r * r;
  const double ps = K.s2 + z * (K.s3 + z * (K.s4 + z * (K.s5 + z * K.s6)));
  const double sr = r + (z * r) * (K.s1 + z * ps);
  const double pc = z * (K.c1 + z * (K.c2 + z * (K.c3 + z * (K.c4 + z * (K.c5 + z * K.c6)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * pc);
  const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}


// Which elementary functions a caller of the conversions below gets (template argument kFn):
//   kFnPlatform  ROCm's — the candidate RMS in the correctly-rounded math mode (K2c, 2x10^8 observation evaluations
//                per step): they move the RMS by <= 1e-6 relative, never the returned elements;
//   kFnCr        correctly rounded (sin/cos by `math_mode`) — the returned elements (K3) in that mode, and the
//                differential corrector in both modes (ill-conditioned normal equations amplify a last place up to
//                1e13 times; bit-identical to the oracle);
//   kFnScreen    the screening arithmetic (screen_math.hpp: own atan2, sincos_k with literal constants; tan and hypot the
//                platform's) — the candidate -> equinoctial conversion in front of K2c's screening loop: ~1e-16 from the
//                platform functions, no decision of the conversion depends on the angles it changes;
//   kFnGlibc     glibc 2.35's bits (glibcmath.hpp, glibcmath2.hpp) — the whole candidate -> RMS chain and the returned
//                elements in OUTFIT_MATH_GLIBC: `rms`, the selection and the elements then follow a glibc run of the
//                reference bit for bit. Every sin/cos pair here shares its operand, i.e. is ONE sincos call in the
//                reference (glm::sincos); the one isolated cosine, cos(dec_obs), is K1's (glm::cos1).
constexpr int kFnPlatform = 0, kFnCr = 1, kFnGlibc = 2, kFnScreen = 3;
template <int kFn> __device__ __forceinline__ void sincos_sel(double x, double& s, double& c, int math_mode = kMathCr) {
  if constexpr (kFn == kFnCr) sincos_mode(math_mode, x, s, c);
  else if constexpr (kFn == kFnGlibc) glm::sincos(x, s, c);
  else if constexpr (kFn == kFnScreen) sincos_k(x, sincos_consts(), s, c);
  else ::sincos(x, &s, &c);
}
template <int kFn> __device__ __forceinline__ double atan2_sel(double y, double x) {
  if constexpr (kFn == kFnCr) return crm::atan2(y, x);
  else if constexpr (kFn == kFnGlibc) return glm::atan2(y, x);
  else if constexpr (kFn == kFnScreen) return atan2_screen(y, x, screen_consts());
  else return ::atan2(y, x);
}
template <int kFn> __device__ __forceinline__ double tan_sel(double x) {
  if constexpr (kFn == kFnCr) return crm::tan(x);
  else if constexpr (kFn == kFnGlibc) return glm::tan(x);
  else return ::tan(x);
}
template <int kFn> __device__ __forceinline__ double hypot_sel(double x, double y) {
  if constexpr (kFn == kFnCr) return crm::hypot(x, y);
  else if constexpr (kFn == kFnGlibc) return glm::hypot(x, y);
  else return ::hypot(x, y);
}

// orb_elem.rs:58-226 (input already rotated to the ecliptic frame). The angles and the rotation's sines and
// cosines are correctly rounded (crmath.hpp): a last-place difference of i or Omega is amplified ~a/r in the
// near-parabolic 1/a and ~1/e in omega and M of a near-circular orbit, and these are the returned elements.
// kCr = false (platform functions, ~900 instructions cheaper) serves the RMS evaluation of every candidate in K2c;
// the elements that are RETURNED are recomputed with kCr = true for the selected candidate only (K3).
template <int kCr = kFnCr>
__device__ __forceinline__ Elements ccek1(const Vec3& pos, const Vec3& vel, double epoch, int math_mode = kMathCr) {
  constexpr double EPS_EQ = 1e-15, EPS_PARAB = 1e-12, EPS_E = 5e-15;
  Elements out;
  out.epoch = epoch;
  const Vec3 h = cross(pos, vel);
  const double h2 = dot(h, h);
  const double hn = sqrt(h2);
  const Vec3 hu = {h.x / hn, h.y / hn, h.z / hn};
  const double sin_i = sqrt(hu.x * hu.x + hu.y * hu.y);
  double inc = principal_angle(atan2_sel<kCr>(sin_i, hu.z));
  double node;
  if (sin_i <= EPS_EQ) { inc = 0.0; node = 0.0; }
  else node = principal_angle(atan2_sel<kCr>(hu.x, -hu.y));
  double si, ci, sn, cn;
  sincos_sel<kCr>(inc, si, ci, math_mode);
  sincos_sel<kCr>(node, sn, cn, math_mode);
  // rotmt(i,0)^T * rotmt(Om,2)^T: every entry is a single product (ref_system.rs:453-462)
  const double m10 = ci * (-sn), m11 = ci * cn, m20 = (-si) * (-sn), m21 = (-si) * cn;
  const Vec3 po = {(cn * pos.x + sn * pos.y) + 0.0 * pos.z, (m10 * pos.x + m11 * pos.y) + si * pos.z,
                   (m20 * pos.x + m21 * pos.y) + ci * pos.z};
  const Vec3 vo = {(cn * vel.x + sn * vel.y) + 0.0 * vel.z, (m10 * vel.x + m11 * vel.y) + si * vel.z,
                   (m20 * vel.x + m21 * vel.y) + ci * vel.z};
  const double rv = po.x * vo.x + po.y * vo.y;
  const double rad = sqrt(po.x * po.x + po.y * po.y);
  const double vsq = vo.x * vo.x + vo.y * vo.y;
  const double inv_a = 2.0 / rad - vsq / kMu;
  bool parabolic = false;
  if (inv_a > EPS_PARAB) {
    const double a = 1.0 / inv_a;
    const double n = sqrt(kMu / (a * a * a));
    const double esin = rv / (n * a * a);
    const double ecos = vsq * rad / kMu - 1.0;
    const double e = sqrt(esin * esin + ecos * ecos);
    if (fabs(e - 1.0) < EPS_E) {
      parabolic = true;
    } else {
      const double ea = atan2_sel<kCr>(esin, ecos);
      double sea, cea;
      sincos_sel<kCr>(ea, sea, cea, math_mode);
      const double ma = principal_angle(ea - e * sea);
      const double x1 = cea - e;
      const double x2 = sqrt(1.0 - e * e) * sea;
      const double nm = sqrt(x1 * x1 + x2 * x2);
      const double x1n = x1 / nm, x2n = x2 / nm;
      const double sp = x1n * po.y - x2n * po.x;
      const double cp = x1n * po.x + x2n * po.y;
      out.kind = 0;
      out.e[0] = a; out.e[1] = e; out.e[2] = inc; out.e[3] = node;
      out.e[4] = principal_angle(atan2_sel<kCr>(sp, cp));
      out.e[5] = ma;
      return out;
    }
  } else if (fabs(inv_a) <= EPS_PARAB) {
    parabolic = true;
  } else {
    const double p = h2 / kMu;
    const double ecos = p / rad - 1.0;
    const double esin = rv * p / (hn * rad);
    const double nu = atan2_sel<kCr>(esin, ecos);
    const double e = sqrt(ecos * ecos + esin * esin);
    if (fabs(e - 1.0) < EPS_E) {
      parabolic = true;
    } else {
      out.kind = 1;
      out.e[0] = p / (1.0 + e); out.e[1] = e; out.e[2] = inc; out.e[3] = node;
      out.e[4] = principal_angle(atan2_sel<kCr>(po.y, po.x) - nu);
      out.e[5] = nu;
      return out;
    }
  }
  // parabolic_solution (orb_elem.rs:122-145)
  (void)parabolic;
  const double p = h2 / kMu;
  const double cnu = p / rad - 1.0;
  const double snu = rv * p / (rad * hn);
  const double nu = atan2_sel<kCr>(snu, cnu);
  out.kind = 1;
  out.e[0] = p / 2.0; out.e[1] = 1.0; out.e[2] = inc; out.e[3] = node;
  out.e[4] = principal_angle(atan2_sel<kCr>(po.y, po.x) - nu);
  out.e[5] = nu;
  return out;
}

struct Equinoctial {
  double epoch, a, h, k, p, q, lambda;
};

// equinoctial_element.rs:285-313
template <int kCr = kFnPlatform>
__device__ __forceinline__ Equinoctial from_kepler(double epoch, double a, double e, double inc, double node,
                                          double argp, double ma) {
  Equinoctial q;
  const double dig = node + argp;
  double sd, cd, sn, cn;
  sincos_sel<kCr>(dig, sd, cd);
  sincos_sel<kCr>(node, sn, cn);
  q.h = e * sd;
  q.k = e * cd;
  const double t = tan_sel<kCr>(inc / 2.0);
  q.p = t * sn;
  q.q = t * cn;
  q.lambda = principal_angle(dig + ma);
  q.epoch = epoch;
  q.a = a;
  return q;
}

// orbit_type/mod.rs:399-443 ; returns 0, 3 (InvalidConversion) or 4 (InvalidOrbit) =
// OUTFIT_TRAJ_* of include/outfit_hip.h
template <int kCr = kFnPlatform>
__device__ __forceinline__ int to_equinoctial(const Elements& el, Equinoctial& eq) {
  if (el.kind == 0) {
    eq = from_kepler<kCr>(el.epoch, el.e[0], el.e[1], el.e[2], el.e[3], el.e[4], el.e[5]);
    return 0;
  }
  const double e = el.e[1];
  if (fabs(e - 1.0) < 1e-12) return 3;
  const double p = el.e[0] * (1.0 + e);
  const double a = -p / (e * e - 1.0);
  if (e <= 1.0) return 4;
  const double s = sqrt((e - 1.0) / (e + 1.0));
  const double x = clampd(s * tan_sel<kCr>(0.5 * el.e[5]), -1.0 + 1e-15, 1.0 - 1e-15);
  const double hh = 2.0 * atanh(x);
  eq = from_kepler<kCr>(el.epoch, a, e, el.e[2], el.e[3], el.e[4], e * sinh(hh) - hh);
  return 0;
}

// Candidate-constant part of propagate_twobody, hoisted out of the observation loop.
struct TwoBodyFrame {
  double n, e2, varpi, beta, bet, one_bh2, one_bk2;
  double x0, sx0, cx0;  // Newton start pi + varpi (equinoctial_element.rs:337) and its sine / cosine:
                        // the same for every observation of the candidate, evaluated once
  Vec3 f, g;
};

template <int kCr = kFnPlatform>
__device__ __forceinline__ TwoBodyFrame twobody_frame(const Equinoctial& q) {
  TwoBodyFrame F;
  F.n = sqrt(kMu / (q.a * q.a * q.a));
  F.e2 = q.h * q.h + q.k * q.k;
  F.varpi = 0.0;
  if (F.e2 > kEps * 1e2) F.varpi = principal_angle(atan2_sel<kCr>(q.h, q.k));
  F.beta = 1.0 / (1.0 + sqrt(1.0 - F.e2));
  F.bet = F.beta * q.h * q.k;
  F.one_bh2 = 1.0 - F.beta * (q.h * q.h);
  F.one_bk2 = 1.0 - F.beta * (q.k * q.k);
  const double u = 1.0 + q.p * q.p + q.q * q.q;
  const double inv_u = 1.0 / u;
  const double cc = 2.0 * q.p * q.q * inv_u;
  F.f = {(1.0 - q.p * q.p + q.q * q.q) * inv_u, cc, -2.0 * q.p * inv_u};
  F.g = {cc, (1.0 + q.p * q.p - q.q * q.q) * inv_u, 2.0 * q.q * inv_u};
  F.x0 = kPi + F.varpi;
  sincos_sel<kCr>(F.x0, F.sx0, F.cx0);
  return F;
}

// equinoctial_element.rs:326-348: roots::find_root_newton_raphson on F - k sin F + h cos F = lambda1 from
// F0 = pi + varpi, SimpleConvergency{eps: 100*EPSILON, max_iter: 25}; also returns sin F, cos F.
// false = Err(RootFindingError)
template <int kCr = kFnPlatform>
__device__ __forceinline__ bool solve_kepler_equation(const Equinoctial& q, const TwoBodyFrame& F, double lam,
                                                      double& x, double& sx, double& cx,
                                                      const SinCosConsts* K = nullptr) {
  constexpr double eps = kEps * 1e2;
  x = F.x0; sx = F.sx0; cx = F.cx0;
  int iter = 0;
  for (;;) {
    const double f = x - q.k * sx + q.h * cx - lam;
    const double d = 1.0 - q.k * cx - q.h * sx;
    if (fabs(f) < eps) return true;
    if (fabs(d) < eps) {
      if (iter != 0) return false;
      x = x + 1.0;
      iter = 1;
    } else {
      const double x1 = x - f / d;
      const bool conv = fabs(x - x1) < eps;
      x = x1;
      if (conv) { if (K) sincos_k(x, *K, sx, cx); else sincos_sel<kCr>(x, sx, cx); return true; }
      iter = iter + 1;
      if (iter >= 25) return false;
    }
    if (K) sincos_k(x, *K, sx, cx); else sincos_sel<kCr>(x, sx, cx);
  }
}

// equinoctial_element.rs:809-867 with t0 = 0, t1 = dt ; false = Err(RootFindingError)
template <int kCr = kFnPlatform>
__device__ __forceinline__ bool propagate_twobody(const Equinoctial& q, const TwoBodyFrame& F, double dt,
                                         Vec3& pos, Vec3& vel, const SinCosConsts* K = nullptr) {
  double lam = principal_angle_fast(q.lambda + F.n * (dt - 0.0));
  if (lam < F.varpi) lam += kTwoPi;
  // roots::find_root_newton_raphson, SimpleConvergency{eps: 100*EPSILON, max_iter: 25}
  double x, sx, cx;
  if (!solve_kepler_equation<kCr>(q, F, lam, x, sx, cx, K)) return false;
  const double xe = q.a * (F.one_bh2 * cx + F.bet * sx - q.k);
  const double ye = q.a * (F.one_bk2 * sx + F.bet * cx - q.h);
  pos = {xe * F.f.x + ye * F.g.x, xe * F.f.y + ye * F.g.y, xe * F.f.z + ye * F.g.z};
  const double vc = F.n * (q.a * q.a) / sqrt(xe * xe + ye * ye);
  const double vxe = vc * (F.bet * cx - F.one_bh2 * sx);
  const double vye = vc * (F.one_bk2 * cx - F.bet * sx);
  vel = {vxe * F.f.x + vye * F.g.x, vxe * F.f.y + vye * F.g.y, vxe * F.f.z + vye * F.g.z};
  return true;
}

// observation_ephemeris.rs:369-416 for one observation; false = Err(_)
template <int kCr = kFnPlatform>
__device__ __forceinline__ bool ephemeris_error(const Equinoctial& q, const TwoBodyFrame& F, double mjd, double ra,
                                       double dec, double cos_dec, double sra, double sdec,
                                       const Vec3& obs_pos_equ, double& term, const SinCosConsts* K = nullptr) {
  Vec3 pe, ve;
  if (!propagate_twobody<kCr>(q, F, mjd - q.epoch, pe, ve, K)) return false;
  const Vec3 pq = ecl_to_equ_finite(pe), vq = ecl_to_equ_finite(ve);
  const Vec3 rel = {pq.x - obs_pos_equ.x, pq.y - obs_pos_equ.y, pq.z - obs_pos_equ.z};
  const double lt = norm(rel) / kVlightAU;
  const Vec3 cor = {rel.x - lt * vq.x, rel.y - lt * vq.y, rel.z - lt * vq.z};
  const double delta = atan2_sel<kCr>(cor.z, hypot_sel<kCr>(cor.x, cor.y));
  double alpha = atan2_sel<kCr>(cor.y, cor.x);  // |atan2| <= pi < 2*pi: rem_euclid(TAU) only adds 2*pi when negative
  if (alpha < 0.0) alpha += kTwoPi;
  double da = fmod_2pi(ra - alpha);  // Rust `%`: sign of the dividend is kept (quirk Q6)
  if (da > kPi) da -= kTwoPi;
  const double t1 = cos_dec * (da / sra);  // cos(dec_obs) is per observation: precomputed by K1
  const double t2 = (dec - delta) / sdec;
  term = t1 * t1 + t2 * t2;
  return true;
}

}  // namespace outfit
