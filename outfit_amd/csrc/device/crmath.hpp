// crmath.hpp — correctly-rounded sin / cos / acos on the device (double-double).
//
// Why: the reference's Gauss correction loop (gauss.rs:1284-1418) is chaotically
// sensitive to a 1-ulp change of the direction cosines (gauss.rs:488-502) or of
// the Kepler start value (prelim_elliptic.rs:9-134): the loop stops on a
// 1e-10 relative threshold that sits on the noise floor of the f/g series, so a
// last-bit difference can change the iteration count and move the returned
// elements by ~1e-10. ROCm's ocml and the host libm differ in the last bit for
// ~0.1 % of arguments, so the three functions that feed that loop are evaluated
// here to the correctly rounded result — what an ideal libm returns on any
// platform. Cost: ~300 flops per call, a few dozen calls per candidate.
//
// Method: Cody–Waite reduction by pi/2 carried in double-double (pi/2 split in
// 3 x 33-bit parts + remainder, exact products for |x| < 2^20), Taylor series on
// [-pi/4, pi/4] with the head in double-double and the sub-ulp tail in double;
// acos is one double-double Newton step on cos(y) = x from the ocml seed.
// Constants: tools/gen_crmath_constants.py (mpmath).
#pragma once
#include <hip/hip_runtime.h>
#include "crmath_table.hpp"

namespace outfit {
namespace crm {

struct dd {
  double hi, lo;
};

__device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd fast_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
__device__ __forceinline__ dd add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return fast_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd add_d(dd a, double b) {
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return fast_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd neg(dd a) { return {-a.hi, -a.lo}; }
__device__ __forceinline__ dd div(dd a, dd b) {
  const double q1 = a.hi / b.hi;
  dd r = add(a, neg(mul(b, {q1, 0.0})));
  const double q2 = r.hi / b.hi;
  r = add(r, neg(mul(b, {q2, 0.0})));
  const double q3 = r.hi / b.hi;
  return add_d(fast_two_sum(q1, q2), q3);
}

constexpr double kP1 = 1.5707963267341256;      // 0x1.921fb54400000p+0
constexpr double kP2 = 6.077100506303966e-11;   // 0x1.0b4611a600000p-34
constexpr double kP3 = 2.0222662487111665e-21;  // 0x1.3198a2e000000p-69
constexpr double kP4 = 8.4784276603689e-32;     // 0x1.b839a252049c1p-104
constexpr double kTwoOverPi = 0.6366197723675814;
constexpr double kPiHi = 3.141592653589793;

// (-1)^j / (2j+1)!  (sin) and (-1)^j / (2j)!  (cos), double-double heads, double tails
__device__ __forceinline__ dd sin_coef(int j) {
  constexpr double c[8][2] = {
      {1.0, 0.0},
      {-0.16666666666666666, -9.25185853854297e-18},
      {0.008333333333333333, 1.1564823173178714e-19},
      {-0.0001984126984126984, -1.7209558293420705e-22},
      {2.7557319223985893e-06, -1.858393274046472e-22},
      {-2.505210838544172e-08, 1.448814070935912e-24},
      {1.6059043836821613e-10, 1.2585294588752098e-26},
      {-7.647163731819816e-13, -7.03872877733453e-30}};
  return {c[j][0], c[j][1]};
}
__device__ __forceinline__ dd cos_coef(int j) {
  constexpr double c[9][2] = {
      {1.0, 0.0},
      {-0.5, 0.0},
      {0.041666666666666664, 2.3129646346357427e-18},
      {-0.001388888888888889, 5.300543954373577e-20},
      {2.48015873015873e-05, 2.1511947866775882e-23},
      {-2.755731922398589e-07, -2.3767714622250297e-23},
      {2.08767569878681e-09, -1.20734505911326e-25},
      {-1.1470745597729725e-11, -2.0655512752830745e-28},
      {4.779477332387385e-14, 4.399205485834081e-31}};
  return {c[j][0], c[j][1]};
}

// r = x - k*pi/2 (|r| <= pi/4), valid for |x| < 2^20; q = k mod 4
__device__ __forceinline__ dd reduce_pio2(double x, int& q) {
  const double k = rint(x * kTwoOverPi);
  q = (int)((long long)k & 3);
  if (k == 0.0) return {x, 0.0};
  const double t = x - k * kP1;  // exact
  const dd a = two_sum(t, -(k * kP2));
  const dd b = two_sum(a.hi, -(k * kP3));
  const double lo = (a.lo + b.lo) - k * kP4;
  return two_sum(b.hi, lo);
}

__device__ __forceinline__ void sincos_kernel(dd r, dd& s, dd& c) {
  const dd z = mul(r, r);
  const double zh = z.hi;
  // tails: sin j = 14..8, cos j = 15..9 (below 2^-53 of the sums)
  double ts = 1.1309962886447716e-31;                 // +1/29!
  ts = ts * zh + -9.183689863795546e-29;              // -1/27!
  ts = ts * zh + 6.446950284384474e-26;               // +1/25!
  ts = ts * zh + -3.868170170630684e-23;              // -1/23!
  ts = ts * zh + 1.9572941063391263e-20;              // +1/21!
  ts = ts * zh + -8.22063524662433e-18;               // -1/19!
  ts = ts * zh + 2.8114572543455206e-15;              // +1/17!
  dd p = {ts, 0.0};
#pragma unroll
  for (int j = 7; j >= 0; --j) p = add(mul(p, z), sin_coef(j));
  s = mul(r, p);
  double tc = -3.7699876288159054e-33;                // -1/30!
  tc = tc * zh + 3.279889237069838e-30;               // +1/28!
  tc = tc * zh + -2.4795962632247976e-27;             // -1/26!
  tc = tc * zh + 1.6117375710961184e-24;              // +1/24!
  tc = tc * zh + -8.896791392450574e-22;              // -1/22!
  tc = tc * zh + 4.110317623312165e-19;               // +1/20!
  tc = tc * zh + -1.5619206968586225e-16;             // -1/18!
  dd qd = {tc, 0.0};
#pragma unroll
  for (int j = 8; j >= 0; --j) qd = add(mul(qd, z), cos_coef(j));
  c = qd;
}

__device__ __forceinline__ void sincos_dd(double x, dd& s, dd& c) {
  int q;
  const dd r = reduce_pio2(x, q);
  dd sr, cr;
  sincos_kernel(r, sr, cr);
  if (q == 0) { s = sr; c = cr; }
  else if (q == 1) { s = cr; c = neg(sr); }
  else if (q == 2) { s = neg(sr); c = neg(cr); }
  else { s = neg(cr); c = sr; }
}

// ---- fast path (Ziv's strategy) ---------------------------------------------------------------
// sin/cos of the reduced argument from a 102-entry double-double table at j/128 and short polynomials
// in the remainder |d| <= 2^-8:
//   sin(a + d) = S + [S (cos d - 1) + C sin d],   cos(a + d) = C + [C (cos d - 1) - S sin d]
// with S, C double-double, the O(d) product exact (two_prod) and everything of relative size
// <= 2^-17 in plain double. Error of the double-double result: < 2^-67 relative (largest contribution:
// the rounding of S (cos d - 1), 2^-53 * 2^-17.5). The result is accepted only when adding that bound to
// the low word cannot change the rounded high word; otherwise (probability ~2^-12 per call) the full
// double-double Taylor evaluation below decides. Either way the returned double is the correctly
// rounded value, so the two paths — and the oracle's (oracle/crmath.c) — agree bit for bit;
// outfit_hip_selftest_crmath checks this on the device for ~10^9 arguments.
constexpr double kFastRelErr = 0x1p-66;  // 2x the derived bound

// double-double sin and cos of r = rh + rl, |r| <= pi/4 + 2^-9; relative error < 2^-67
__device__ __forceinline__ void sincos_fast_dd(dd r, dd& s, dd& c) {
  const double kf = rint(r.hi * 128.0);
  const int k = (int)kf;
  const int j = k < 0 ? -k : k;
  const double sgn = k < 0 ? -1.0 : 1.0;
  const double sh = sgn * kSinCosTab[j][0], sl = sgn * kSinCosTab[j][1];
  const double ch = kSinCosTab[j][2], cl = kSinCosTab[j][3];
  const dd d = fast_two_sum(r.hi - kf * 0.0078125, r.lo);  // r.hi - k/128 is exact
  const double d2 = d.hi * d.hi;
  // sin d = d.hi + d.lo + ps ; cos d - 1 = pc
  const double ps = d.hi * d2 * (-0.16666666666666666 + d2 * (0.008333333333333333 + d2 * -0.0001984126984126984));
  const double pc = d2 * (-0.5 + d2 * (0.041666666666666664 + d2 * -0.001388888888888889)) - d.hi * d.lo;
  {
    const dd p = two_prod(ch, d.hi);
    const double rest = ((sl + p.lo) + (ch * d.lo + cl * d.hi)) + (sh * pc + ch * ps);
    dd t = two_sum(sh, p.hi);
    t.lo += rest;
    s = fast_two_sum(t.hi, t.lo);
  }
  {
    const dd p = two_prod(sh, d.hi);
    const double rest = ((cl - p.lo) - (sh * d.lo + sl * d.hi)) + (ch * pc - sh * ps);
    dd t = two_sum(ch, -p.hi);
    t.lo += rest;
    c = fast_two_sum(t.hi, t.lo);
  }
}

// true when the high word of v is the correctly rounded value given an absolute error bound e on v
__device__ __forceinline__ bool rounds_safely(dd v, double e) { return v.hi + (v.lo + copysign(e, v.lo)) == v.hi; }

__device__ __forceinline__ void apply_quadrant(int q, dd sr, dd cr, dd& s, dd& c) {
  if (q == 0) { s = sr; c = cr; }
  else if (q == 1) { s = cr; c = neg(sr); }
  else if (q == 2) { s = neg(sr); c = neg(cr); }
  else { s = neg(cr); c = sr; }
}

// correctly-rounded sin and cos
__device__ __forceinline__ void sincos(double x, double& s, double& c) {
  if (!(fabs(x) < 1048576.0)) { ::sincos(x, &s, &c); return; }
  if (x == 0.0) { s = x; c = 1.0; return; }
  int q;
  const dd r = reduce_pio2(x, q);
  dd sr, cr, sd, cd;
  sincos_fast_dd(r, sr, cr);
  if (!(rounds_safely(sr, kFastRelErr * fabs(sr.hi)) && rounds_safely(cr, kFastRelErr * fabs(cr.hi))))
    sincos_kernel(r, sr, cr);
  apply_quadrant(q, sr, cr, sd, cd);
  s = sd.hi;
  c = cd.hi;
}
// the full-precision evaluation alone (reference for the self-test)
__device__ __forceinline__ void sincos_slow(double x, double& s, double& c) {
  if (!(fabs(x) < 1048576.0)) { ::sincos(x, &s, &c); return; }
  if (x == 0.0) { s = x; c = 1.0; return; }
  dd sd, cd;
  sincos_dd(x, sd, cd);
  s = sd.hi;
  c = cd.hi;
}

// correctly-rounded acos for |x| <= 1 (callers guard the domain): one Newton step on cos(y) = x
__device__ __forceinline__ double acos_slow(double x) {
  if (!(fabs(x) <= 1.0)) return ::acos(x);
  if (x == 1.0) return 0.0;
  if (x == -1.0) return kPiHi;
  const double y0 = ::acos(x);
  dd sd, cd;
  sincos_dd(y0, sd, cd);
  const dd q = div(add_d(cd, -x), sd);
  return add_d(q, y0).hi;
}
__device__ __forceinline__ double acos(double x) {
  if (!(fabs(x) <= 1.0)) return ::acos(x);
  if (x == 1.0) return 0.0;
  if (x == -1.0) return kPiHi;
  const double y0 = ::acos(x);  // within a few ulp; y0 in (0, pi)
  int q;
  const dd r = reduce_pio2(y0, q);
  dd sr, cr, sd, cd;
  sincos_fast_dd(r, sr, cr);
  apply_quadrant(q, sr, cr, sd, cd);
  // y = y0 + (cos y0 - x) / sin y0 ; cos y0 - x is a few ulp of x: the high parts cancel exactly
  const double num = (cd.hi - x) + cd.lo;
  const double corr = num / sd.hi;
  const dd y = fast_two_sum(y0, corr);
  // error of the correction: the 2^-66 absolute error of cos y0 (|cos| <= 1) divided by sin y0, plus the
  // relative error of the quotient and the neglected second-order Newton term (corr^2 cot(y0) / 2)
  const double e = (0x1p-66 + corr * corr) / fabs(sd.hi) + 0x1p-50 * fabs(corr);
  if (rounds_safely(y, e)) return y.hi;
  return acos_slow(x);
}

// ---- exp in double-double, and from it correctly-rounded sinh / cosh / log -------------------------
// Why: the hyperbolic cold start of the universal Kepler solve (prelim_hyperbolic.rs:45-141) evaluates
// log, sinh and cosh, and its value enters the same chaotically sensitive correction loop as the elliptic
// one (measured: glibc's 1-ulp sinh/cosh moved a = 14 AU by 1.6e-9 on 1 object in 30 000). Same remedy as
// for sin/cos: the correctly rounded value, here and in the oracle (oracle/crmath.c, via binary128).
//
// x = k ln2 + j/128 + d, |d| <= 2^-8 (+ rounding slack): exp(x) = 2^k * T_j * exp(d), T_j a double-double table.
struct ExpArg {
  dd d;
  int k, j;
};
__device__ __forceinline__ ExpArg exp_reduce(double x) {  // |x| < 2^10
  const double kf = rint(x * kInvLn2);
  dd r = {x, 0.0};
  if (kf != 0.0) {
    const double t = x - kf * kLn2_1;  // exact
    const dd a = two_sum(t, -(kf * kLn2_2));
    const dd b = two_sum(a.hi, -(kf * kLn2_3));
    r = two_sum(b.hi, (a.lo + b.lo) - kf * kLn2_4);
  }
  const double jf = rint(r.hi * 128.0);
  return {fast_two_sum(r.hi - jf * 0.0078125, r.lo), (int)kf, (int)jf};
}
// exp(+-(j/128 + d)) as double-double, unscaled (the caller applies 2^(+-k)); `err` = absolute error bound.
// Fast: exp(d) = 1 + d.hi + qs, qs = d.lo (1 + d.hi) + d^2 (A(d^2) + d B(d^2)) in plain double (|qs| <= 2^-17):
// the O(d) product with the table entry is exact (two_prod), the rest has relative size <= 2^-17, so the
// result carries an absolute error < 2^-50 |qs| T (Horner roundings of qs, its product with T, one more
// addition) + the table's own 2^-106 T; the Taylor remainder d^8/8! is below that for every d.
__device__ __forceinline__ void exp_pair_fast(const ExpArg& a, dd& ep, dd& em, double& err_p, double& err_m) {
  const double dh = a.d.hi, dl = a.d.lo, z = dh * dh;
  const double A = 0.5 + z * (0.041666666666666664 + z * 0.001388888888888889);
  const double B = 0.16666666666666666 + z * (0.008333333333333333 + z * 0.0001984126984126984);
#pragma unroll
  for (int sgn = 0; sgn < 2; ++sgn) {
    const double h = sgn ? -dh : dh, l = sgn ? -dl : dl;
    const double qs = (l + l * h) + z * (A + h * B);
    const int idx = (sgn ? -a.j : a.j) + 45;
    const double th = kExpTab[idx][0], tl = kExpTab[idx][1];
    const dd pr = two_prod(th, h);
    const double rest = (tl + pr.lo) + (th * qs + tl * (h + qs));
    dd t = two_sum(th, pr.hi);
    t.lo += rest;
    const dd e = fast_two_sum(t.hi, t.lo);
    const double bound = (0x1p-50 * fabs(qs) + 0x1p-103) * th;
    if (sgn) { em = e; err_m = bound; } else { ep = e; err_p = bound; }
  }
}
// Full double-double: exp(d) by Horner over 1/n!, n <= 13 (|d|^14/14! < 2^-148); relative error < 2^-100
__device__ __forceinline__ dd exp_slow_dd(const ExpArg& a, bool negate) {
  const dd d = negate ? neg(a.d) : a.d;
  dd p = {kInvFact[13][0], kInvFact[13][1]};
#pragma unroll
  for (int n = 12; n >= 0; --n) p = add(mul(p, d), {kInvFact[n][0], kInvFact[n][1]});
  const int idx = (negate ? -a.j : a.j) + 45;
  return mul({kExpTab[idx][0], kExpTab[idx][1]}, p);
}
__device__ __forceinline__ dd scale2(dd v, int k) { return {ldexp(v.hi, k), ldexp(v.lo, k)}; }

// |f| < 2^-3: sinh f = f + f z (1/6 + z (1/120 + ...)), cosh f = 1 + z (1/2 + z (1/24 + ...)), z = f^2
__device__ __forceinline__ void sinhcosh_tiny(double f, dd& s, dd& c, double& err_s, double& err_c) {
  const dd z = two_prod(f, f);
  const double zh = z.hi;
  const double rs = zh * (0.008333333333333333 + zh * (0.0001984126984126984 + zh * (2.7557319223985893e-06 + zh * (2.505210838544172e-08 + zh * (1.6059043836821613e-10 + zh * 7.647163731819816e-13)))));
  const dd ps = fast_two_sum(kInvFact[3][0], kInvFact[3][1] + rs);
  const dd v = mul(mul(z, ps), {f, 0.0});
  dd t = two_sum(f, v.hi);
  t.lo += v.lo;
  s = fast_two_sum(t.hi, t.lo);
  err_s = 0x1p-59 * fabs(v.hi);
  const double rc = zh * (0.041666666666666664 + zh * (0.001388888888888889 + zh * (2.48015873015873e-05 + zh * (2.755731922398589e-07 + zh * (2.08767569878681e-09 + zh * 1.1470745597729725e-11)))));
  const dd pc = fast_two_sum(0.5, rc);
  const dd w = mul(z, pc);
  t = two_sum(1.0, w.hi);
  t.lo += w.lo;
  c = fast_two_sum(t.hi, t.lo);
  err_c = 0x1p-59 * fabs(w.hi);
}
__device__ __forceinline__ void sinhcosh_tiny_slow(double f, dd& s, dd& c) {
  const dd z = two_prod(f, f);
  dd ps = {kInvFact[19][0], kInvFact[19][1]}, pc = {kInvFact[20][0], kInvFact[20][1]};
#pragma unroll
  for (int n = 8; n >= 0; --n) ps = add(mul(ps, z), {kInvFact[2 * n + 1][0], kInvFact[2 * n + 1][1]});
#pragma unroll
  for (int n = 9; n >= 0; --n) pc = add(mul(pc, z), {kInvFact[2 * n][0], kInvFact[2 * n][1]});
  s = mul(ps, {f, 0.0});
  c = pc;
}

// correctly-rounded sinh and cosh (both at once: they share exp(+-|f|))
__device__ __forceinline__ void sinhcosh_slow(double f, double& s, double& c) {
  const double a = fabs(f);
  if (!(a <= 700.0)) { s = ::sinh(f); c = ::cosh(f); return; }
  if (f == 0.0) { s = f; c = 1.0; return; }  // keeps the sign of -0
  dd sd, cd;
  if (a < 0.125) {
    sinhcosh_tiny_slow(f, sd, cd);
    s = sd.hi; c = cd.hi;
    return;
  }
  const ExpArg r = exp_reduce(a);
  const dd ep = scale2(exp_slow_dd(r, false), r.k - 1), em = scale2(exp_slow_dd(r, true), -r.k - 1);
  sd = add(ep, neg(em));
  cd = add(ep, em);
  s = copysign(sd.hi, f);
  c = cd.hi;
}
__device__ __forceinline__ void sinhcosh(double f, double& s, double& c) {
  const double a = fabs(f);
  if (!(a <= 700.0)) { s = ::sinh(f); c = ::cosh(f); return; }
  if (f == 0.0) { s = f; c = 1.0; return; }  // keeps the sign of -0
  dd sd, cd;
  double es, ec;
  if (a < 0.125) {
    sinhcosh_tiny(f, sd, cd, es, ec);
    if (rounds_safely(sd, es) && rounds_safely(cd, ec)) { s = sd.hi; c = cd.hi; return; }
    sinhcosh_slow(f, s, c);
    return;
  }
  const ExpArg r = exp_reduce(a);
  dd ep, em;
  double bp, bm;
  exp_pair_fast(r, ep, em, bp, bm);
  ep = scale2(ep, r.k - 1);   // e^a / 2
  em = scale2(em, -r.k - 1);  // e^-a / 2
  const double e = ldexp(bp, r.k - 1) + ldexp(bm, -r.k - 1) + 0x1p-103 * ep.hi;
  sd = add(ep, neg(em));
  cd = add(ep, em);
  if (rounds_safely(sd, e) && rounds_safely(cd, e)) { s = copysign(sd.hi, f); c = cd.hi; return; }
  sinhcosh_slow(f, s, c);
}

// correctly-rounded log for finite y > 1 (the call site's domain; anything else goes to the platform log):
// one Newton step on exp(l) = y from the platform's value l0, log y = l0 + log1p(y e^-l0 - 1)
__device__ __forceinline__ double log_slow(double y) {
  if (!(y > 1.0 && y < 1.0e300)) return ::log(y);
  const double l0 = ::log(y);
  const ExpArg r = exp_reduce(-l0);
  const dd e = scale2(exp_slow_dd(r, false), r.k);
  const dd t = add_d(mul(e, {y, 0.0}), -1.0);             // y e^-l0 - 1, a few ulp of 1
  const dd corr = add(t, mul(mul(t, t), {-0.5, 0.0}));   // log1p(t) = t - t^2/2 + O(t^3), |t|^3 < 2^-150
  return add_d(corr, l0).hi;
}
__device__ __forceinline__ double log(double y) {
  if (!(y > 1.0 && y < 1.0e300)) return ::log(y);
  const double l0 = ::log(y);
  const ExpArg r = exp_reduce(-l0);
  dd e, unused;
  double be, bu;
  exp_pair_fast(r, e, unused, be, bu);
  e = scale2(e, r.k);
  be = ldexp(be, r.k);
  const dd p = two_prod(y, e.hi);
  const double t = (p.hi - 1.0) + (p.lo + y * e.lo);  // p.hi - 1 is exact (p.hi within 2^-40 of 1)
  const double corr = t - 0.5 * t * t;
  const dd l = fast_two_sum(l0, corr);
  const double err = y * be + 0x1p-51 * fabs(t) + 0x1p-105;  // + rounding of t and corr, log1p's t^3/3 (|t| < 2^-40)
  if (fabs(t) < 0x1p-40 && rounds_safely(l, err)) return l.hi;
  return log_slow(y);
}

// correctly-rounded atan2 for finite, non-zero y and x of ordinary magnitude (everything else — zeros,
// infinities, NaN, |x| or |y| outside [2^-400, 2^400] — goes to the platform function, whose results there
// are exact constants or irrelevant): one Newton step from the platform's value t0,
//   atan2(y, x) = t0 + atan((y cos t0 - x sin t0) / (x cos t0 + y sin t0)),
// the numerator from double-double sin/cos with exact leading products (they cancel to a few ulp).
// Used where a last-place difference of an angle is amplified before it is returned: the rotations of the
// state -> elements conversion (orb_elem.rs:58-226; near-parabolic a and near-circular omega, M).
__device__ __forceinline__ bool atan2_ordinary(double y, double x) {
  const double ay = fabs(y), ax = fabs(x);
  return ay > 0x1p-400 && ay < 0x1p400 && ax > 0x1p-400 && ax < 0x1p400;
}
__device__ __forceinline__ double atan2_slow(double y, double x) {
  if (!atan2_ordinary(y, x)) return ::atan2(y, x);
  const double t0 = ::atan2(y, x);
  dd s, c;
  sincos_dd(t0, s, c);
  const dd num = add(mul(c, {y, 0.0}), neg(mul(s, {x, 0.0})));
  const dd den = add(mul(c, {x, 0.0}), mul(s, {y, 0.0}));
  const dd q = div(num, den);  // tan(delta); delta - tan(delta) ~ delta^3 / 3 < 2^-150
  return add_d(q, t0).hi;
}
__device__ __forceinline__ double atan2(double y, double x) {
  if (!atan2_ordinary(y, x)) return ::atan2(y, x);
  const double t0 = ::atan2(y, x);
  int q;
  const dd r = reduce_pio2(t0, q);
  dd sr, cr, s, c;
  sincos_fast_dd(r, sr, cr);
  apply_quadrant(q, sr, cr, s, c);
  const dd p1 = two_prod(y, c.hi), p2 = two_prod(x, s.hi);
  const double num = (p1.hi - p2.hi) + ((p1.lo - p2.lo) + (y * c.lo - x * s.lo));
  const double den = x * c.hi + y * s.hi;  // = hypot(x, y) up to rounding, > 0
  const double delta = num / den;
  const dd t = fast_two_sum(t0, delta);
  // errors of sin/cos (2^-67 relative each) through the numerator, the quotient's and the sum's roundings
  const double err = 0x1p-66 * (fabs(p1.hi) + fabs(p2.hi)) / den + 0x1p-51 * fabs(delta);
  if (fabs(delta) < 0x1p-40 * fabs(t0) && rounds_safely(t, err)) return t.hi;
  return atan2_slow(y, x);
}

// correctly-rounded tan for |x| < 2^20: double-double sin / cos and one double-double quotient
__device__ __forceinline__ double tan_slow(double x) {
  if (!(fabs(x) < 1048576.0)) return ::tan(x);
  if (x == 0.0) return x;
  dd s, c;
  sincos_dd(x, s, c);
  return div(s, c).hi;
}
__device__ __forceinline__ double tan(double x) {
  if (!(fabs(x) < 1048576.0)) return ::tan(x);
  if (x == 0.0) return x;
  int q;
  const dd r = reduce_pio2(x, q);
  dd sr, cr;
  sincos_fast_dd(r, sr, cr);
  const dd num = (q & 1) ? neg(cr) : sr, den = (q & 1) ? sr : cr;  // tan(r + pi/2) = -cos r / sin r
  const double t0 = num.hi / den.hi;
  const dd p = two_prod(t0, den.hi);
  const double rem = ((num.hi - p.hi) - p.lo) + (num.lo - t0 * den.lo);
  const dd t = fast_two_sum(t0, rem / den.hi);
  // 2^-67 relative from each of sin and cos, plus the roundings of the remainder and its quotient
  if (rounds_safely(t, 0x1p-65 * fabs(t0))) return t.hi;
  return tan_slow(x);
}

// correctly-rounded hypot for finite arguments of ordinary magnitude (else the platform function): x^2 + y^2
// exactly as a double-double, its square root by one Newton correction of the IEEE sqrt of the high word.
// (An exact tie — x^2 + y^2 the square of a 54-bit number — cannot be decided by an error bound; the nearest
// candidate is returned.)
__device__ __forceinline__ double hypot(double x, double y) {
  const double ax = fabs(x), ay = fabs(y);
  if (!(ax < 0x1p400 && ay < 0x1p400 && (ax > 0x1p-400 || ay > 0x1p-400))) return ::hypot(x, y);
  const dd s = add(two_prod(x, x), two_prod(y, y));
  const double r0 = sqrt(s.hi);
  const dd p = two_prod(r0, r0);
  const double e = ((s.hi - p.hi) - p.lo) + s.lo;
  const dd r = fast_two_sum(r0, e / (2.0 * r0));
  return r.hi;
}

// Aberth's initial-guess directions: theta_k = fma(2pi/8, k, (pi/2)/8), correctly rounded
__device__ __forceinline__ void aberth_dir(int k, double& c, double& s) {
  constexpr double t[8][2] = {
      {0.9807852804032304, 0.19509032201612825},   {0.5555702330196023, 0.8314696123025452},
      {-0.1950903220161282, 0.9807852804032304},   {-0.8314696123025453, 0.5555702330196022},
      {-0.9807852804032304, -0.19509032201612836}, {-0.5555702330196022, -0.8314696123025452},
      {0.1950903220161283, -0.9807852804032304},   {0.8314696123025452, -0.5555702330196022}};
  c = t[k][0];
  s = t[k][1];
}

}  // namespace crm
}  // namespace outfit
