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
