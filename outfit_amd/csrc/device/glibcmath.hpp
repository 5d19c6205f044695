// SPDX-License-Identifier: LGPL-2.1-or-later
// This file restates algorithms and/or carries constant tables of the GNU C Library (glibc) 2.35 math library:
//   Copyright (C) 1993-2022 Free Software Foundation, Inc.
//   sin/cos/tan/atan2/acos: IBM Accurate Mathematical Library, written by International Business Machines Corp.,
//     Copyright (C) 2001-2022 Free Software Foundation, Inc.
//   exp/log: contributed to glibc by Arm Ltd. (Szabolcs Nagy), Copyright (C) 2018-2022 Free Software Foundation, Inc.
//   sinh/cosh/expm1: derived from FreeBSD/Sun fdlibm — Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//     Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.
// glibc is free software; you can redistribute it and/or modify it under the terms of the GNU Lesser General Public
// License as published by the Free Software Foundation; either version 2.1 of the License, or (at your option) any later
// version. It is distributed WITHOUT ANY WARRANTY. The license text is in LICENSES/LGPL-2.1.txt; what that means for
// liboutfit_hip.so, which compiles this file in, is in NOTICE.
//
// glibcmath.hpp — sin, cos, sinh, cosh with the bits glibc 2.35 (x86-64, FMA-capable CPU) returns.
//
// Why: the reference calls f64::sin/cos/sinh/cosh = the platform libm, glibc on the Linux machines it runs on.
// glibc is not correctly rounded for 0.15 % (sin, cos) and 13-20 % (sinh, cosh) of arguments, and the Gauss
// correction loop amplifies a last-place difference: against a glibc run of the same path the
// correctly-rounded default (crmath.hpp) leaves ~0.8 % of the orbits beyond 1e-12 rad; with these four functions
// (math mode OUTFIT_MATH_GLIBC) none of 9 944 + 3 984 measured orbits is (tests/test_gpu_glibc_mode.py).
//
// What is restated is the published method of glibc's dbl-64 routines — IBM Accurate Mathematical Library
// sin/cos (table at 1/128 steps, three-piece pi/2 reduction), Sun fdlibm expm1/sinh/cosh, ARM optimized-routines
// exp (N = 128, the fused shape of glibc's FMA/AVX2 variant) — operation for operation as in oracle/glibcmath.c,
// which is pinned bit for bit against the installed libm (1.5e9 / 1e8 arguments). This translation unit is
// compiled with -ffp-contract=off: only the explicit fma() calls below are fused.
// Outside the restated ranges (|x| >= 105414350; |x| >= 22 or tiny for sinh/cosh) the platform function is
// used; the path's call sites (angles, |F| < 15) never get there.
#pragma once
#include <hip/hip_runtime.h>
#include "glibc_tables.hpp"

namespace outfit {
namespace glm {

__device__ __forceinline__ double taylor_sin(double xx, double x, double dx) {
  const double p = (((-0x1.ADDFFC2FCDF59p-26 * xx + 0x1.71DE27B9A7ED9p-19) * xx + -0x1.A01A019DB08B8p-13) * xx +
                    0x1.1111111110ECEp-7) * xx + -0x1.5555555555555p-3;
  const double t = (p * x - 0.5 * dx) * xx + dx;
  return x + t;
}
constexpr double kSn3 = -1.66666666666664880952546298448555E-01, kSn5 = 8.33333214285722277379541354343671E-03,
                 kCs2 = 4.99999999999999999999950396842453E-01, kCs4 = -4.16666666666664434524222570944589E-02,
                 kCs6 = 1.38888874007937613028114285595617E-03;
constexpr double kBig = 0x1.8p45, kHp0 = 0x1.921FB54442D18p0, kHp1 = 0x1.1A62633145C07p-54;

__device__ __forceinline__ double do_cos(double x, double dx) {
  if (x < 0) dx = -dx;
  const double u = kBig + fabs(x);
  x = fabs(x) - (u - kBig) + dx;
  const double* T = kGlibcSinCosTab + __double2loint(u) * 4;
  const double xx = x * x;
  const double s = x + x * xx * (kSn3 + xx * kSn5);
  const double c = xx * (kCs2 + xx * (kCs4 + xx * kCs6));
  const double cor = (T[3] - s * T[1] - T[2] * c) - T[0] * s;
  return T[2] + cor;
}
__device__ __forceinline__ double do_sin(double x, double dx) {
  const double xold = x;
  if (fabs(x) < 0.126) return taylor_sin(x * x, x, dx);
  if (x <= 0) dx = -dx;
  const double u = kBig + fabs(x);
  x = fabs(x) - (u - kBig);
  const double* T = kGlibcSinCosTab + __double2loint(u) * 4;
  const double xx = x * x;
  const double s = x + (dx + x * xx * (kSn3 + xx * kSn5));
  const double c = x * dx + xx * (kCs2 + xx * (kCs4 + xx * kCs6));
  const double cor = (T[1] + s * T[3] - T[0] * c) + T[2] * s;
  return copysign(T[0] + cor, xold);
}
__device__ __forceinline__ int reduce_sincos(double x, double& a, double& da) {
  const double t = x * 0x1.45F306DC9C883p-1 + 0x1.8p52;
  const double xn = t - 0x1.8p52;
  const double y = (x - xn * 0x1.921FB58000000p0) - xn * -0x1.DDE973C000000p-27;
  const int n = __double2loint(t) & 3;
  double t1 = xn * -0x1.CB3B398000000p-55;
  const double t2 = y - t1;
  double db = (y - t2) - t1;
  t1 = xn * -0x1.d747f23e32ed7p-83;
  const double b = t2 - t1;
  db += (t2 - b) - t1;
  a = b;
  da = db;
  return n;
}
__device__ __forceinline__ double do_sincos(double a, double da, int n) {
  const double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
  return (n & 2) ? -r : r;
}
__device__ __forceinline__ double sin(double x) {
  const unsigned k = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (k < 0x3e500000u) return x;
  if (k < 0x3feb6000u) return do_sin(x, 0.0);
  if (k < 0x400368fdu) return copysign(do_cos(kHp0 - fabs(x), kHp1), x);
  if (k < 0x419921FBu) { double a, da; const int n = reduce_sincos(x, a, da); return do_sincos(a, da, n); }
  return ::sin(x);
}
__device__ __forceinline__ double cos(double x) {
  const unsigned k = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (k < 0x3e400000u) return 1.0;
  if (k < 0x3feb6000u) return do_cos(x, 0.0);
  if (k < 0x400368fdu) { const double y = kHp0 - fabs(x), a = y + kHp1, da = (y - a) + kHp1; return do_sin(a, da); }
  if (k < 0x419921FBu) { double a, da; const int n = reduce_sincos(x, a, da); return do_sincos(a, da, n + 1); }
  return ::cos(x);
}
// x.sin() and x.cos() of one operand = ONE sincos() call of the platform libm (see sin1 / cos1 below): these two are
// its restatement, bit for bit
// sincos(x) = (sin(x), cos(x)) of the two functions above — written WITHOUT their range branches (round 5). In each
// range a call evaluates do_sin once and do_cos once, on range-specific arguments:
//   |x| < 0.855..   sin = do_sin(x, 0)                          cos = do_cos(x, 0)
//   |x| < 2.426..   sin = copysign(do_cos(hp0 - |x|, hp1), x)   cos = do_sin(a', da'), a' = (hp0 - |x|) + hp1, da' = ((hp0 - |x|) - a') + hp1
//   |x| < 1.05e8    (a, da, n) = reduce_sincos(x):  sin = do_sincos(a, da, n), cos = do_sincos(a, da, n + 1)
// so the arguments are SELECTED per lane and the two kernels evaluated once for the whole wave. Every lane performs
// exactly the operations of its own range on the same operands — the same bits as sin(x), cos(x) above — but a wave
// whose lanes fall into all three ranges (any wave of angles spread over [0, 2 pi): 14 / 25 / 61 %) issues ~120
// instructions instead of the ~260 of the branches executed one after the other. Huge arguments keep their branch.
__device__ __forceinline__ void sincos(double x, double& s, double& c) {
  const unsigned k = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (k >= 0x419921FBu) { s = ::sin(x); c = ::cos(x); return; }
  const double ax = fabs(x);
  double a, da;
  const int n = reduce_sincos(x, a, da);
  const double y = kHp0 - ax, ac = y + kHp1, dac = (y - ac) + kHp1;
  const bool in_b = k < 0x3feb6000u, in_c = !in_b && k < 0x400368fdu;
  const double ts = in_b ? x : (in_c ? ac : a), dts = in_b ? 0.0 : (in_c ? dac : da);
  const double tc = in_b ? x : (in_c ? y : a), dtc = in_b ? 0.0 : (in_c ? kHp1 : da);
  const double vs = do_sin(ts, dts), vc = do_cos(tc, dtc);
  // the reduced ranges: n for the sine, n + 1 for the cosine (do_sincos)
  const double rs = (n & 1) ? vc : vs, rc = ((n + 1) & 1) ? vc : vs;
  const double ds = (n & 2) ? -rs : rs, dc = ((n + 1) & 2) ? -rc : rc;
  s = in_b ? vs : (in_c ? copysign(vc, x) : ds);
  c = in_b ? vc : (in_c ? vs : dc);
  if (k < 0x3e500000u) s = x;
  if (k < 0x3e400000u) c = 1.0;
}

// ---- sin / cos as the STANDALONE libm entry points return them (round 3) -------------------------------------
// glibc 2.35 x86-64: sincos() (one SSE2 variant, nothing fused: the functions above) and sin() / cos() (IFUNCs whose
// FMA variants are the same source compiled with contraction) differ in the last bit for 0.07 % of arguments. The
// reference calls sincos wherever x.sin() and x.cos() of one operand share a basic block (LLVM's FSINCOS combine on
// *-linux-gnu) and the standalone function at an isolated call: on the path `u0.sin()` in prelim_elliptic.rs:97 and
// `dec.cos()` in observation_ephemeris.rs:412. Operation for operation as orc_glibc_sin1 / cos1 (oracle/glibcmath.c),
// which are pinned against the installed libm's sin / cos.
__device__ __forceinline__ double taylor_sin_fma(double xx, double x, double dx) {
  const double p = fma(fma(fma(fma(-0x1.ADDFFC2FCDF59p-26, xx, 0x1.71DE27B9A7ED9p-19), xx, -0x1.A01A019DB08B8p-13), xx,
                           0x1.1111111110ECEp-7), xx, -0x1.5555555555555p-3);
  const double t = fma(fma(p, x, -(0.5 * dx)), xx, dx);
  return x + t;
}
__device__ __forceinline__ double do_cos_fma(double x, double dx) {
  if (x < 0) dx = -dx;
  const double u = kBig + fabs(x);
  x = fabs(x) - (u - kBig) + dx;
  const double* T = kGlibcSinCosTab + __double2loint(u) * 4;
  const double xx = x * x;
  const double s = fma(x * xx, fma(xx, kSn5, kSn3), x);
  const double c = xx * fma(xx, fma(xx, kCs6, kCs4), kCs2);
  const double cor = fma(-s, T[0], fma(-c, T[2], fma(-s, T[1], T[3])));
  return T[2] + cor;
}
__device__ __forceinline__ double do_sin_fma(double x, double dx) {
  const double xold = x;
  if (fabs(x) < 0.126) return taylor_sin_fma(x * x, x, dx);
  if (x <= 0) dx = -dx;
  const double u = kBig + fabs(x);
  x = fabs(x) - (u - kBig);
  const double* T = kGlibcSinCosTab + __double2loint(u) * 4;
  const double xx = x * x;
  const double s = x + fma(x * xx, fma(xx, kSn5, kSn3), dx);
  const double c = fma(x, dx, xx * fma(xx, fma(xx, kCs6, kCs4), kCs2));
  const double cor = fma(s, T[2], fma(-c, T[0], fma(s, T[3], T[1])));
  return copysign(T[0] + cor, xold);
}
__device__ __forceinline__ int reduce_sincos_fma(double x, double& a, double& da) {
  const double t = fma(x, 0x1.45F306DC9C883p-1, 0x1.8p52);
  const double xn = t - 0x1.8p52;
  const double y = fma(-xn, -0x1.DDE973C000000p-27, fma(-xn, 0x1.921FB58000000p0, x));
  const int n = __double2loint(t) & 3;
  const double t2 = fma(-xn, -0x1.CB3B398000000p-55, y);
  double db = fma(-xn, -0x1.CB3B398000000p-55, y - t2);
  const double b = fma(-xn, -0x1.d747f23e32ed7p-83, t2);
  db += fma(-xn, -0x1.d747f23e32ed7p-83, t2 - b);
  a = b;
  da = db;
  return n;
}
__device__ __forceinline__ double do_sincos_fma(double a, double da, int n) {
  const double r = (n & 1) ? do_cos_fma(a, da) : do_sin_fma(a, da);
  return (n & 2) ? -r : r;
}
__device__ __forceinline__ double sin1(double x) {
  const unsigned k = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (k < 0x3e500000u) return x;
  if (k < 0x3feb6000u) return do_sin_fma(x, 0.0);
  if (k < 0x400368fdu) return copysign(do_cos_fma(kHp0 - fabs(x), kHp1), x);
  if (k < 0x419921FBu) { double a, da; const int n = reduce_sincos_fma(x, a, da); return do_sincos_fma(a, da, n); }
  return ::sin(x);
}
__device__ __forceinline__ double cos1(double x) {
  const unsigned k = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (k < 0x3e400000u) return 1.0;
  if (k < 0x3feb6000u) return do_cos_fma(x, 0.0);
  if (k < 0x400368fdu) { const double y = kHp0 - fabs(x), a = y + kHp1, da = (y - a) + kHp1; return do_sin_fma(a, da); }
  if (k < 0x419921FBu) { double a, da; const int n = reduce_sincos_fma(x, a, da); return do_sincos_fma(a, da, n + 1); }
  return ::cos(x);
}

__device__ __forceinline__ double add_exponent(double y, int k) {
  return __hiloint2double(__double2hiint(y) + (k << 20), __double2loint(y));
}
// 2^-54 <= |x| < 56 ln 2 (callers guard the range)
__device__ __forceinline__ double expm1_core(double x) {
  constexpr double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10,
                   kInvLn2 = 1.44269504088896338700e+00, Q1 = -3.33333333333331316428e-02,
                   Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                   Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
  double hi, lo, c = 0.0, t, e, y;
  int k;
  unsigned hx = (unsigned)__double2hiint(x);
  const bool neg = (hx & 0x80000000u) != 0u;
  hx &= 0x7fffffffu;
  if (hx > 0x3fd62e42u) {
    if (hx < 0x3FF0A2B2u) {
      if (!neg) { hi = x - kLn2Hi; lo = kLn2Lo; k = 1; } else { hi = x + kLn2Hi; lo = -kLn2Lo; k = -1; }
    } else {
      k = (int)(kInvLn2 * x + (neg ? -0.5 : 0.5));
      t = (double)k;
      hi = x - t * kLn2Hi;
      lo = t * kLn2Lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else {
    k = 0;
  }
  const double hfx = 0.5 * x, hxs = x * hfx;
  const double R1 = 1.0 + hxs * Q1, h2 = hxs * hxs, R2 = Q2 + hxs * Q3, h4 = h2 * h2, R3 = Q4 + hxs * Q5;
  const double r1 = (R1 + R2 * h2) + R3 * h4;
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) return (x < -0.25) ? -2.0 * (e - (x + 0.5)) : 1.0 + 2.0 * (x - e);
  if (k <= -2 || k > 56) { y = 1.0 - (e - x); return add_exponent(y, k) - 1.0; }
  if (k < 20) { t = __hiloint2double(0x3ff00000 - (0x200000 >> k), 0); return add_exponent(t - (e - x), k); }
  t = __hiloint2double((0x3ff - k) << 20, 0);
  y = x - (e + t);
  y += 1.0;
  return add_exponent(y, k);
}
// 2^-54 <= |x| < 512 (callers guard the range); the FMA variant's fused shape
__device__ __forceinline__ double exp_core(double x) {
  constexpr double kInvLn2N = 0x1.71547652b82fep0 * 128, kNegLn2HiN = -0x1.62e42fefa0000p-8,
                   kNegLn2LoN = -0x1.cf79abc9e3b3ap-47, kShift = 0x1.8p52, C2 = 0x1.ffffffffffdbdp-2,
                   C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  double kd = fma(x, kInvLn2N, kShift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double r = fma(kd, kNegLn2LoN, fma(kd, kNegLn2HiN, x));
  const unsigned idx = 2u * (unsigned)(ki & 127ull);
  const double tail = __longlong_as_double((long long)kGlibcExpTab[idx]);
  const double scale = __longlong_as_double((long long)(kGlibcExpTab[idx + 1] + (ki << 45)));
  const double r2 = r * r;
  const double tmp = fma(r2 * r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r));
  return fma(scale, tmp, scale);
}
__device__ __forceinline__ double sinh(double x) {
  const unsigned jx = (unsigned)__double2hiint(x), ix = jx & 0x7fffffffu;
  if (ix >= 0x40360000u || ix < 0x3e300000u) return ::sinh(x);
  const double h = (jx & 0x80000000u) ? -0.5 : 0.5;
  const double t = expm1_core(fabs(x));
  if (ix < 0x3ff00000u) return h * (2.0 * t - t * t / (t + 1.0));
  return h * (t + t / (t + 1.0));
}
__device__ __forceinline__ double cosh(double x) {
  const unsigned ix = (unsigned)__double2hiint(x) & 0x7fffffffu;
  if (ix >= 0x40360000u) return ::cosh(x);
  if (ix < 0x3fd62e43u) {
    if (ix < 0x3c800000u) return 1.0;
    const double t = expm1_core(fabs(x)), w = 1.0 + t;
    return 1.0 + (t * t) / (w + w);
  }
  const double t = exp_core(fabs(x));
  return 0.5 * t + 0.5 / t;
}
__device__ __forceinline__ void sinhcosh(double x, double& s, double& c) {
  s = sinh(x);
  c = cosh(x);
}

}  // namespace glm
}  // namespace outfit
