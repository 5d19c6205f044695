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
// glibcmath3.hpp — acos and log with the bits glibc 2.35 (x86-64, FMA-capable CPU) returns. Companion of glibcmath.hpp (sin, cos,
// sinh, cosh) and glibcmath2.hpp (atan2, tan, hypot).
//
// Why: the elliptic cold start of the universal Kepler solver (prelim_elliptic.rs:9-31) calls f64::acos, i.e. the platform
// libm. glibc's acos is not correctly rounded, and where it differs from the correctly-rounded value used until round 3 the
// start value moves by a last place: on 200 000 objects one returned orbit differed from a glibc run by 1 ulp for that
// reason alone. Operation for operation as oracle/glibcmath3.c (pinned bit for bit against the installed libm on 2e9
// arguments; it documents the algorithm, where the fused operations were read and where the tables come from). This
// translation unit is compiled with -ffp-contract=off: only the explicit fma() calls are fused.
#pragma once
#include <hip/hip_runtime.h>
#include "glibc_tables3.hpp"

namespace outfit {
namespace glm {

constexpr double kAcHp0 = 0x1.921fb54442d18p+0, kAcHp1 = 0x1.1a62633145c07p-54;
constexpr double kAcF1 = 0.1666666666666641, kAcF2 = 0.07500000000261227, kAcF3 = 0.044642856142105974,
                 kAcF4 = 0.03038212685821193, kAcF5 = 0.022355121102652562, kAcF6 = 0.018138290340456505;

// one table interval of e_asin.c: the entry at n holds node, c1, c2 .. c_top, the x^2-step coefficient, asin(node) high
__device__ __forceinline__ double acos_table(double x, int m, int n, int top) {
  const double* T = kAsncs + n;
  const double xx = (m > 0 ? x : -x) - T[0];
  double t = T[top];
  for (int j = top - 1; j >= 2; --j) t = fma(t, xx, T[j]);
  const double x2 = xx * xx;
  t = fma(x2, t, T[top + 1]);
  t = fma(xx, T[1], t);
  const double y = T[top + 2];
  return m > 0 ? (kAcHp1 - t) + (kAcHp0 - y) : (t + kAcHp1) + (y + kAcHp0);
}

__device__ __forceinline__ double acos(double x) {
  const int m = __double2hiint(x), k = m & 0x7fffffff;
  if (k < 0x3c880000) return kAcHp0;
  if (k < 0x3fc00000) {
    const double x2 = x * x;
    const double t = fma(fma(fma(fma(fma(kAcF6, x2, kAcF5), x2, kAcF4), x2, kAcF3), x2, kAcF2), x2, kAcF1);
    const double y = kAcHp0 - x;
    const double eps = ((kAcHp0 - y) - x) + kAcHp1;
    return y + fma(-t, x * x2, eps);
  }
  if (k < 0x3fef0000) {
    int n, top;
    if (k < 0x3fd00000) { n = 11 * ((k >> 15) & 0x1f); top = 6; }
    else if (k < 0x3fe00000) { n = 11 * ((k >> 14) & 0x3f) + 352; top = 6; }
    else if (k < 0x3fe80000) { n = 12 * ((k >> 13) & 0x7f) + 1056; top = 7; }
    else if (k < 0x3fed8000) { n = 13 * ((k >> 13) & 0x7f) + 992; top = 8; }
    else if (k < 0x3fee8000) { n = 14 * ((k >> 13) & 0x7f) + 884; top = 9; }
    else { n = 15 * ((k >> 13) & 0x7f) + 768; top = 10; }
    return acos_table(x, m, n, top);
  }
  if (k < 0x3ff00000) {
    const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
    const unsigned long long zu = (unsigned long long)__double_as_longlong(z);
    // inroot[(bits >> 46) & 127] * 2^(511 - (bits >> 53))  (powtwo.tbl holds the powers of two)
    double t = kInroot[(zu >> 46) & 0x7f] * __longlong_as_double((long long)((1023ull + 511ull - (zu >> 53)) << 52));
    const double r = fma(-(t * t), z, 1.0);
    t = fma(fma(fma(0.31252362655451865, r, 0.3750175008673452), r, 0.4999999994959554), r, 0.9999999998599908) * t;
    const double c = z * t;
    const double h = fma(-c, t * 0.5, 1.5);
    const double y = fma(-134217728.0, c, fma(c, 134217728.0, c));
    const double cc = fma(-y, y, z) / fma(h, c, y);
    const double p = fma(fma(fma(fma(fma(kAcF6, z, kAcF5), z, kAcF4), z, kAcF3), z, kAcF2), z, kAcF1) * z;
    const double q = p * (y + cc);
    if (m >= 0) { const double s = (cc + q) + y; return s + s; }
    const double s = ((kAcHp1 - cc) - q) + (kAcHp0 - y);
    return s + s;
  }
  if (k == 0x3ff00000 && __double2loint(x) == 0) return m > 0 ? 0.0 : 0x1.921fb54442d18p+1;
  if (k > 0x7ff00000 || (k == 0x7ff00000 && __double2loint(x) != 0)) return x + x;
  return (x - x) / (x - x);
}

// ---- log: glibc 2.35's `__log_fma` (e_log.c; oracle/glibcmath3.c `orc_glibc_log`). Zero, negative, subnormal and non-finite
// arguments go to the platform function; the path's call site (hyperbolic cold start, prelim_hyperbolic.rs) passes > 1.
__device__ __forceinline__ double log(double x) {
  const double *D = kLogData, *A = D + 2, *B = D + 7, *T = D + 18;
  const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
  const unsigned top = (unsigned)(ix >> 48);
  if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {
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
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) return ::log(x);
  const unsigned long long tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const long long k = (long long)tmp >> 52;
  const double z = __longlong_as_double((long long)(ix - (tmp & 0xfff0000000000000ull)));
  const double invc = T[2 * i], logc = T[2 * i + 1], kd = (double)k;
  const double r = fma(z, invc, -1.0);
  const double w = fma(kd, D[0], logc);
  const double hi = w + r;
  const double lo = fma(kd, D[1], (w - hi) + r);
  const double r2 = r * r;
  const double p = fma(fma(r, A[4], A[3]), r2, fma(r, A[2], A[1]));
  return fma(r * r2, p, fma(r2, A[0], lo)) + hi;
}

}  // namespace glm
}  // namespace outfit
