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
 * glibcmath3.c — acos as glibc 2.35 evaluates it on x86-64 with FMA (`__ieee754_acos_fma`; IBM Accurate Mathematical
 * Library, sysdeps/ieee754/dbl-64/e_asin.c after the removal of its multi-precision slow paths), restated for the
 * oracle's glibc-faithful mode. TEST INFRASTRUCTURE ONLY (see oracle.h); the device twin is
 * outfit_amd/csrc/device/glibcmath3.hpp.
 *
 * Why: the reference calls f64::acos once per elliptic cold start (prelim_elliptic.rs:80-96), i.e. the platform libm.
 * glibc's acos is not correctly rounded; where it differs from the correctly-rounded value the cold start moves by a
 * last place, and in 1 of 200 000 objects (profiles/round03e) that reached a returned element (1 ulp). The algorithm
 * follows the published source; WHICH a*b+c are fused is read off the machine code of the installed build (the
 * function is compiled with contraction), as for atan2 / tan in glibcmath2.c; the tables (`asncs`: per node the Taylor
 * coefficients of asin and asin(node) in two parts; `inroot`: 1/sqrt seeds) are read out of the same binary by
 * tools/extract_glibc_tables.py. Pin: tools/glibc_pin3.c, zero differences against the platform acos.
 *
 * Intervals by the high word k = |x| bits >> 32 (e_asin.c `__ieee754_acos`):
 *   |x| < 2^-55            pi/2
 *   |x| < 0.125            pi/2 - x - x^3 P(x^2)
 *   0.125 <= |x| < 0.96875 Taylor expansion of asin around the table node below |x| (five layouts of entries),
 *                          acos = pi/2 -+ asin
 *   0.96875 <= |x| < 1     2 asin(sqrt((1 -+ x)/2)) with a division-free square root refined to double-double
 *   |x| = 1                0 or pi;   else NaN.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"
#include "glibc_tables3.h"

#define HP0 0x1.921fb54442d18p+0   /* pi/2, high */
#define HP1 0x1.1a62633145c07p-54  /* pi/2, low */
static const double F1 = 0.1666666666666641, F2 = 0.07500000000261227, F3 = 0.044642856142105974,
                    F4 = 0.03038212685821193, F5 = 0.022355121102652562, F6 = 0.018138290340456505;
static const double RT0 = 0.9999999998599908, RT1 = 0.4999999994959554, RT2 = 0.3750175008673452, RT3 = 0.31252362655451865;
#define T27 134217728.0

/* one table interval: entry at n holds node, c1, c2 .. c_top, the x^2-step coefficient, asin(node) high */
static inline double acos_table(double x, int32_t m, int n, int top) {
  const double *T = orc_glibc_asncs;
  const double xx = (m > 0 ? x : -x) - T[n];
  double t = T[n + top];
  for (int j = top - 1; j >= 2; --j) t = fma(t, xx, T[n + j]);
  const double x2 = xx * xx;
  t = fma(x2, t, T[n + top + 1]);
  t = fma(xx, T[n + 1], t);
  const double y = T[n + top + 2];
  return m > 0 ? (HP1 - t) + (HP0 - y) : (t + HP1) + (y + HP0);
}

double orc_glibc_acos(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  const int32_t m = (int32_t)(u >> 32), k = m & 0x7fffffff;
  if (k < 0x3c880000) return HP0;
  if (k < 0x3fc00000) {
    const double x2 = x * x;
    const double t = fma(fma(fma(fma(fma(F6, x2, F5), x2, F4), x2, F3), x2, F2), x2, F1);
    const double y = HP0 - x;
    const double eps = ((HP0 - y) - x) + HP1;
    return y + fma(-t, x * x2, eps);
  }
  if (k < 0x3fd00000) return acos_table(x, m, 11 * ((k >> 15) & 0x1f), 6);
  if (k < 0x3fe00000) return acos_table(x, m, 11 * ((k >> 14) & 0x3f) + 352, 6);
  if (k < 0x3fe80000) return acos_table(x, m, 12 * ((k >> 13) & 0x7f) + 1056, 7);
  if (k < 0x3fed8000) return acos_table(x, m, 13 * ((k >> 13) & 0x7f) + 992, 8);
  if (k < 0x3fee8000) return acos_table(x, m, 14 * ((k >> 13) & 0x7f) + 884, 9);
  if (k < 0x3fef0000) return acos_table(x, m, 15 * ((k >> 13) & 0x7f) + 768, 10);
  if (k < 0x3ff00000) {
    const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
    uint64_t zu;
    memcpy(&zu, &z, 8);
    double t = orc_glibc_inroot[(zu >> 46) & 0x7f] * orc_glibc_powtwo[511 - (int)(zu >> 53)];
    const double r = fma(-(t * t), z, 1.0);
    t = fma(fma(fma(RT3, r, RT2), r, RT1), r, RT0) * t;
    const double c = z * t;
    const double h = fma(-c, t * 0.5, 1.5);        /* 1.5 - 0.5 t c */
    const double y = fma(-T27, c, fma(c, T27, c));  /* (t27 c + c) - t27 c: the head of c */
    const double cc = fma(-y, y, z) / fma(h, c, y);
    const double p = fma(fma(fma(fma(fma(F6, z, F5), z, F4), z, F3), z, F2), z, F1) * z;
    const double q = p * (y + cc);
    if (m >= 0) { const double s = (cc + q) + y; return s + s; }
    const double s = ((HP1 - cc) - q) + (HP0 - y);
    return s + s;
  }
  if (k == 0x3ff00000 && (uint32_t)u == 0) return m > 0 ? 0.0 : 0x1.921fb54442d18p+1;
  if (k > 0x7ff00000 || (k == 0x7ff00000 && (uint32_t)u != 0)) return x + x;
  return (x - x) / (x - x);
}

/* ---- log: glibc 2.35's `__log_fma` (sysdeps/ieee754/dbl-64/e_log.c, from ARM's optimized routines; the build with a fast
 * fma computes r = fma(z, invc, -1) and never touches tab2). The reference calls f64::ln in the hyperbolic cold start
 * (prelim_hyperbolic.rs:45-141). Zero, negative, subnormal and non-finite arguments go to the platform function (the path
 * never produces them: the argument is c + sqrt(c^2 - 1) with c > 1). */
double orc_glibc_log(double x) {
  const double *D = orc_glibc_logdata, *A = D + 2, *B = D + 7, *T = D + 18;
  uint64_t ix;
  memcpy(&ix, &x, 8);
  const uint32_t top = (uint32_t)(ix >> 48);
  if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {   /* 1 - 0x1p-4 <= x < 1 + 0x1.09p-4 */
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0, r2 = r * r, r3 = r * r2;
    double y = fma(r2, B[9], fma(r, B[8], B[7]));
    y = fma(r3, B[10], y);
    y = fma(y, r3, fma(r2, B[6], fma(r, B[5], B[4])));
    y = fma(y, r3, fma(r2, B[3], fma(r, B[2], B[1])));
    const double rhi = fma(-0x1p27, r, fma(r, 0x1p27, r));
    const double rlo = r - rhi;
    const double sq = rhi * rhi;
    const double hi = fma(sq, B[0], r);
    const double lo = fma(B[0] * rlo, r + rhi, fma(sq, B[0], r - hi));
    return fma(y, r3, lo) + hi;
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) return log(x);
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int64_t k = (int64_t)tmp >> 52;
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  double z;
  memcpy(&z, &iz, 8);
  const double invc = T[2 * i], logc = T[2 * i + 1], kd = (double)k;
  const double r = fma(z, invc, -1.0);
  const double w = fma(kd, D[0], logc);
  const double hi = w + r;
  const double lo = fma(kd, D[1], (w - hi) + r);
  const double r2 = r * r;
  const double p = fma(fma(r, A[4], A[3]), r2, fma(r, A[2], A[1]));
  return fma(r * r2, p, fma(r2, A[0], lo)) + hi;
}
