/*
 * SPDX-License-Identifier: LGPL-2.1-or-later
 * This file restates algorithms and/or carries constant tables of the GNU C Library (glibc) 2.35 math library:
 *   Copyright (C) 1993-2022 Free Software Foundation, Inc.
 *   sin/cos/tan/atan2/acos: IBM Accurate Mathematical Library, written by International Business Machines Corp.,
 *     Copyright (C) 2001-2022 Free Software Foundation, Inc.
 *   exp/log: contributed to glibc by Arm Ltd. (Szabolcs Nagy), Copyright (C) 2018-2022 Free Software Foundation, Inc.
 *   sinh/cosh/expm1: derived from FreeBSD/Sun fdlibm — Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *     Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.
 * glibc is free software; you can redistribute it and/or modify it under the terms of the GNU Lesser General Public
 * License as published by the Free Software Foundation; either version 2.1 of the License, or (at your option) any later
 * version. It is distributed WITHOUT ANY WARRANTY. The license text is in LICENSES/LGPL-2.1.txt; what that means for
 * liboutfit_hip.so, which compiles this file in, is in NOTICE.
 */
/*
 * glibcmath.c — sin, cos, sinh, cosh as glibc 2.35 (x86-64, FMA-capable CPU) evaluates them, restated.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Why: the reference calls f64::sin / cos / sinh / cosh, i.e. the platform libm — glibc on the Linux machines
 * its KATs were recorded on. glibc's results are NOT correctly rounded for 0.15 % (sin, cos) and 13-20 %
 * (sinh, cosh) of arguments, and the Gauss correction loop amplifies a last-place difference:
 * measured on BASELINE configs[1], the correctly-rounded evaluation (crmath.c) leaves 14 of 994 orbits not
 * bit-identical to a glibc run, 7 of them beyond 1e-12 rad — all of it through sin/cos. This file restates
 * the four functions so that the oracle (orc_math_mode = 2) and the device (OUTFIT_MATH_GLIBC,
 * outfit_amd/csrc/device/glibcmath.hpp, same operation order) return glibc's bits without calling glibc.
 *
 * The algorithms live in a third-party dependency of the reference's toolchain (glibc 2.35, not under
 * /root/reference); what is restated is the published method, operation order read off the behaviour and the
 * machine code of the installed libm.so.6:
 *   sin, cos    sysdeps/ieee754/dbl-64/s_sin.c  (IBM Accurate Mathematical Library): Taylor below 0.126,
 *               else table at 1/128 steps + degree-5/6 polynomials; pi/2 reduction in three pieces for
 *               2.43 <= |x| < 105414350. No fused operations (the file has no explicit fma and glibc builds
 *               it with -ffp-contract=off), so every x86-64 variant returns these bits.
 *   expm1       sysdeps/ieee754/dbl-64/s_expm1.c (Sun fdlibm, with glibc's three-way split of the polynomial).
 *   exp         sysdeps/ieee754/dbl-64/e_exp.c   (ARM optimized routines, S. Nagy): N = 128 table, degree-5
 *               polynomial, in the shape glibc's FMA/AVX2 ifunc variant runs (seven fused operations).
 *   sinh, cosh  sysdeps/ieee754/dbl-64/e_sinh.c / e_cosh.c (fdlibm) on top of the two.
 * Pin: tests/test_glibcmath.py compares them with the host's libm bit for bit wherever that is glibc (zero
 * differences on 1.5e9 sin/cos and 1e8 exp/expm1/sinh/cosh arguments when this was written; tools/glibc_pin.c);
 * the hyperbolic cold-start KAT of the reference (kepler/params.rs:689-725, assert_eq!) is bit-exact in this mode.
 * Outside the restated ranges (|x| >= 105414350 for sin/cos, |x| >= 22 for sinh/cosh, tiny arguments) the
 * platform function is called: unreachable from the path's call sites (angles and |F| < 15).
 * Tables: tools/gen_glibc_tables.py (mpmath).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"
#include "glibc_tables.h"

static inline uint64_t asuint64(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double asdouble(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline uint32_t hiword(double x) { return (uint32_t)(asuint64(x) >> 32); }
static inline int32_t loword(double x) { return (int32_t)(uint32_t)asuint64(x); }

/* ---- sin / cos ------------------------------------------------------------------------------------------ */
static const double S1 = -0x1.5555555555555p-3, S2 = 0x1.1111111110ECEp-7, S3 = -0x1.A01A019DB08B8p-13,
                    S4 = 0x1.71DE27B9A7ED9p-19, S5 = -0x1.ADDFFC2FCDF59p-26;
static const double SN3 = -1.66666666666664880952546298448555E-01, SN5 = 8.33333214285722277379541354343671E-03,
                    CS2 = 4.99999999999999999999950396842453E-01, CS4 = -4.16666666666664434524222570944589E-02,
                    CS6 = 1.38888874007937613028114285595617E-03;
static const double BIG = 0x1.8p45, HP0 = 0x1.921FB54442D18p0, HP1 = 0x1.1A62633145C07p-54,
                    MP1 = 0x1.921FB58000000p0, MP2 = -0x1.DDE973C000000p-27, PP3 = -0x1.CB3B398000000p-55,
                    PP4 = -0x1.d747f23e32ed7p-83, HPINV = 0x1.45F306DC9C883p-1, TOINT = 0x1.8p52;

static inline double taylor_sin(double xx, double x, double dx) {
  double p = (((S5 * xx + S4) * xx + S3) * xx + S2) * xx + S1;
  double t = (p * x - 0.5 * dx) * xx + dx;
  return x + t;
}
static inline double do_cos(double x, double dx) {
  if (x < 0) dx = -dx;
  double u = BIG + fabs(x);
  x = fabs(x) - (u - BIG) + dx;
  const double *T = kGlibcSinCosTab + loword(u) * 4;
  double xx = x * x;
  double s = x + x * xx * (SN3 + xx * SN5);
  double c = xx * (CS2 + xx * (CS4 + xx * CS6));
  double cor = (T[3] - s * T[1] - T[2] * c) - T[0] * s;
  return T[2] + cor;
}
static inline double do_sin(double x, double dx) {
  double xold = x;
  if (fabs(x) < 0.126) return taylor_sin(x * x, x, dx);
  if (x <= 0) dx = -dx;
  double u = BIG + fabs(x);
  x = fabs(x) - (u - BIG);
  const double *T = kGlibcSinCosTab + loword(u) * 4;
  double xx = x * x;
  double s = x + (dx + x * xx * (SN3 + xx * SN5));
  double c = x * dx + xx * (CS2 + xx * (CS4 + xx * CS6));
  double cor = (T[1] + s * T[3] - T[0] * c) + T[2] * s;
  return copysign(T[0] + cor, xold);
}
static inline int reduce_sincos(double x, double *a, double *da) {
  double t = x * HPINV + TOINT;
  double xn = t - TOINT;
  double y = (x - xn * MP1) - xn * MP2;
  int n = loword(t) & 3;
  double t1 = xn * PP3, t2 = y - t1, db = (y - t2) - t1;
  t1 = xn * PP4;
  double b = t2 - t1;
  db += (t2 - b) - t1;
  *a = b;
  *da = db;
  return n;
}
static inline double do_sincos(double a, double da, int n) {
  double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
  return (n & 2) ? -r : r;
}
double orc_glibc_sin(double x) {
  uint32_t k = hiword(x) & 0x7fffffff;
  double a, da;
  if (k < 0x3e500000) return x;
  if (k < 0x3feb6000) return do_sin(x, 0);
  if (k < 0x400368fd) return copysign(do_cos(HP0 - fabs(x), HP1), x);
  if (k < 0x419921FB) { int n = reduce_sincos(x, &a, &da); return do_sincos(a, da, n); }
  return sin(x);
}
double orc_glibc_cos(double x) {
  uint32_t k = hiword(x) & 0x7fffffff;
  double a, da;
  if (k < 0x3e400000) return 1.0;
  if (k < 0x3feb6000) return do_cos(x, 0);
  if (k < 0x400368fd) { double y = HP0 - fabs(x); a = y + HP1; da = (y - a) + HP1; return do_sin(a, da); }
  if (k < 0x419921FB) { int n = reduce_sincos(x, &a, &da); return do_sincos(a, da, n + 1); }
  return cos(x);
}

/* ---- sin / cos as the STANDALONE entry points return them --------------------------------------------------
 * Found in round 3: glibc 2.35 x86-64 does NOT return the same bits from sin(x) / cos(x) and from sincos(x).
 * `sincos` has one variant (SSE2, no fused operation) — what the functions above restate; `sin` and `cos` are
 * IFUNCs whose FMA/AVX2 variants (__sin_fma, __cos_fma) are the same s_sin.c compiled with contraction: every
 * a*b+c of do_sin / do_cos / TAYLOR_SIN / reduce_sincos that GCC could fuse IS fused. The two differ in the last
 * bit for 0.07 % of arguments (tools/glibc_pin2.c). Which one the reference calls is decided by LLVM per call
 * site: x.sin() and x.cos() of the SAME operand in one basic block become ONE call of sincos on *-linux-gnu
 * (SelectionDAG's FSINCOS combine; Rust's f64::sin / cos are the llvm.sin / llvm.cos intrinsics), an isolated
 * x.sin() or x.cos() calls the standalone function. On the path the isolated ones are `u0.sin()` of the elliptic cold
 * start (prelim_elliptic.rs:97) and `dec.cos()` of the residual (observation_ephemeris.rs:412); every other site
 * pairs (unit_vector gauss.rs:464-469 after inlining, the Newton loops prelim_elliptic.rs:125-126 and
 * equinoctial_element.rs:332-337, :659-660, :295-300, orb_elem.rs:166-171, nalgebra's sin_cos in the rotations).
 * Operation order below: read off the machine code of __sin_fma / __cos_fma (Ubuntu GLIBC 2.35-0ubuntu3.12). */
static inline double taylor_sin_fma(double xx, double x, double dx) {
  double p = fma(fma(fma(fma(S5, xx, S4), xx, S3), xx, S2), xx, S1);
  double t = fma(fma(p, x, -(0.5 * dx)), xx, dx);
  return x + t;
}
static inline double do_cos_fma(double x, double dx) {
  if (x < 0) dx = -dx;
  double u = BIG + fabs(x);
  x = fabs(x) - (u - BIG) + dx;
  const double *T = kGlibcSinCosTab + loword(u) * 4;
  double xx = x * x;
  double s = fma(x * xx, fma(xx, SN5, SN3), x);
  double c = xx * fma(xx, fma(xx, CS6, CS4), CS2);
  double cor = fma(-s, T[0], fma(-c, T[2], fma(-s, T[1], T[3])));
  return T[2] + cor;
}
static inline double do_sin_fma(double x, double dx) {
  double xold = x;
  if (fabs(x) < 0.126) return taylor_sin_fma(x * x, x, dx);
  if (x <= 0) dx = -dx;
  double u = BIG + fabs(x);
  x = fabs(x) - (u - BIG);
  const double *T = kGlibcSinCosTab + loword(u) * 4;
  double xx = x * x;
  double s = x + fma(x * xx, fma(xx, SN5, SN3), dx);
  double c = fma(x, dx, xx * fma(xx, fma(xx, CS6, CS4), CS2));
  double cor = fma(s, T[2], fma(-c, T[0], fma(s, T[3], T[1])));
  return copysign(T[0] + cor, xold);
}
static inline int reduce_sincos_fma(double x, double *a, double *da) {
  double t = fma(x, HPINV, TOINT);
  double xn = t - TOINT;
  double y = fma(-xn, MP2, fma(-xn, MP1, x));
  int n = loword(t) & 3;
  double t2 = fma(-xn, PP3, y), db = fma(-xn, PP3, y - t2);
  double b = fma(-xn, PP4, t2);
  db += fma(-xn, PP4, t2 - b);
  *a = b;
  *da = db;
  return n;
}
static inline double do_sincos_fma(double a, double da, int n) {
  double r = (n & 1) ? do_cos_fma(a, da) : do_sin_fma(a, da);
  return (n & 2) ? -r : r;
}
double orc_glibc_sin1(double x) {
  uint32_t k = hiword(x) & 0x7fffffff;
  double a, da;
  if (k < 0x3e500000) return x;
  if (k < 0x3feb6000) return do_sin_fma(x, 0);
  if (k < 0x400368fd) return copysign(do_cos_fma(HP0 - fabs(x), HP1), x);
  if (k < 0x419921FB) { int n = reduce_sincos_fma(x, &a, &da); return do_sincos_fma(a, da, n); }
  return sin(x);
}
double orc_glibc_cos1(double x) {
  uint32_t k = hiword(x) & 0x7fffffff;
  double a, da;
  if (k < 0x3e400000) return 1.0;
  if (k < 0x3feb6000) return do_cos_fma(x, 0);
  if (k < 0x400368fd) { double y = HP0 - fabs(x); a = y + HP1; da = (y - a) + HP1; return do_sin_fma(a, da); }
  if (k < 0x419921FB) { int n = reduce_sincos_fma(x, &a, &da); return do_sincos_fma(a, da, n + 1); }
  return cos(x);
}

/* ---- expm1 (0x1p-54 <= |x| < 56 ln 2) --------------------------------------------------------------------- */
static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                    INVLN2 = 1.44269504088896338700e+00, Q1 = -3.33333333333331316428e-02,
                    Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                    Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
static inline double add_exponent(double y, int k) { return asdouble(asuint64(y) + ((uint64_t)(uint32_t)(k << 20) << 32)); }
double orc_glibc_expm1(double x) {
  double y, hi, lo, c = 0.0, t, e;
  int k;
  uint32_t hx = hiword(x);
  const int neg = (hx & 0x80000000u) != 0;
  hx &= 0x7fffffff;
  if (hx >= 0x4043687A || hx < 0x3c900000) return expm1(x);
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      if (!neg) { hi = x - LN2_HI; lo = LN2_LO; k = 1; } else { hi = x + LN2_HI; lo = -LN2_LO; k = -1; }
    } else {
      k = (int)(INVLN2 * x + (neg ? -0.5 : 0.5));
      t = k;
      hi = x - t * LN2_HI;
      lo = t * LN2_LO;
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
  if (k < 20) { t = asdouble((uint64_t)(uint32_t)(0x3ff00000 - (0x200000 >> k)) << 32); return add_exponent(t - (e - x), k); }
  t = asdouble((uint64_t)(uint32_t)((0x3ff - k) << 20) << 32);
  y = x - (e + t);
  y += 1.0;
  return add_exponent(y, k);
}

/* ---- exp (2^-54 <= |x| < 512), the FMA variant -------------------------------------------------------------- */
static const double INVLN2N = 0x1.71547652b82fep0 * 128, NEGLN2HIN = -0x1.62e42fefa0000p-8,
                    NEGLN2LON = -0x1.cf79abc9e3b3ap-47, SHIFT = 0x1.8p52, C2 = 0x1.ffffffffffdbdp-2,
                    C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
double orc_glibc_exp(double x) {
  const uint32_t abstop = (uint32_t)(asuint64(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9 >= 0x408 - 0x3c9) return exp(x);
  double kd = fma(x, INVLN2N, SHIFT);
  const uint64_t ki = asuint64(kd);
  kd -= SHIFT;
  const double r = fma(kd, NEGLN2LON, fma(kd, NEGLN2HIN, x));
  const uint64_t idx = 2 * (ki % 128);
  const double tail = asdouble(kGlibcExpTab[idx]);
  const double scale = asdouble(kGlibcExpTab[idx + 1] + (ki << 45));
  const double r2 = r * r;
  const double tmp = fma(r2 * r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r));
  return fma(scale, tmp, scale);
}

/* ---- sinh / cosh (|x| < 22) -------------------------------------------------------------------------------- */
double orc_glibc_sinh(double x) {
  const uint32_t jx = hiword(x), ix = jx & 0x7fffffff;
  if (ix >= 0x40360000 || ix < 0x3e300000) return sinh(x);
  const double h = (jx & 0x80000000u) ? -0.5 : 0.5;
  const double t = orc_glibc_expm1(fabs(x));
  if (ix < 0x3ff00000) return h * (2.0 * t - t * t / (t + 1.0));
  return h * (t + t / (t + 1.0));
}
double orc_glibc_cosh(double x) {
  const uint32_t ix = hiword(x) & 0x7fffffff;
  if (ix >= 0x40360000) return cosh(x);
  if (ix < 0x3fd62e43) {
    if (ix < 0x3c800000) return 1.0;
    const double t = orc_glibc_expm1(fabs(x)), w = 1.0 + t;
    return 1.0 + (t * t) / (w + w);
  }
  const double t = orc_glibc_exp(fabs(x));
  return 0.5 * t + 0.5 / t;
}

/* Pin: compares the restatement with the HOST libm on n pseudo-random arguments per function over the ranges the
 * path uses (and beyond); counts[k] = differing results for k = sin, cos, expm1, exp, sinh, cosh. All zero on a
 * glibc 2.35 x86-64 host with FMA; meaningless on another libm (the caller checks which one it is). */
void orc_glibc_compare_with_host(uint64_t n, uint64_t seed, uint64_t *counts) {
  uint64_t s = seed ? seed : 88172645463325252ull;
  for (int k = 0; k < 6; ++k) counts[k] = 0;
  for (uint64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) * 0x1p-53;
    double x, h;
    switch (i & 7) {
      case 0: x = (u - 0.5) * 1.72; h = (u - 0.5) * 0.8; break;
      case 1: x = (u - 0.5) * 4.86; h = (u - 0.5) * 2.2; break;
      case 2: x = (u - 0.5) * 64.0; h = (u - 0.5) * 44.0; break;
      case 3: x = (u - 0.5) * 2e5; h = (u - 0.5) * 30.0; break;
      case 4: x = (u - 0.5) * 2e8; h = (u - 0.5) * 80.0; break;
      case 5: x = ldexp(u - 0.5, -(int)(s % 40)); h = ldexp(u - 0.5, -(int)(s % 50)); break;
      case 6: x = (u - 0.5) * 14.0; h = (u - 0.5) * 6.0; break;
      default: x = (double)((long)(u * 4096) - 2048) * 1.5707963267948966 + (u - 0.5) * 1e-6;
               h = (double)((long)(u * 64) - 32) * 0.6931471805599453 + (u - 0.5) * 1e-3;
    }
    counts[0] += asuint64(sin(x)) != asuint64(orc_glibc_sin(x));
    counts[1] += asuint64(cos(x)) != asuint64(orc_glibc_cos(x));
    counts[2] += asuint64(expm1(h)) != asuint64(orc_glibc_expm1(h));
    counts[3] += asuint64(exp(h)) != asuint64(orc_glibc_exp(h));
    counts[4] += asuint64(sinh(h)) != asuint64(orc_glibc_sinh(h));
    counts[5] += asuint64(cosh(h)) != asuint64(orc_glibc_cosh(h));
  }
}
