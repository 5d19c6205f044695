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
// glibcmath2.hpp — atan2, tan, hypot with the bits glibc 2.35 (x86-64, FMA-capable CPU) returns. Companion of
// glibcmath.hpp (sin, cos, sinh, cosh).
//
// Why: the RMS stage of the path (ccek1 -> to_equinoctial -> propagate_twobody -> ephemeris_error; reference
// orb_elem.rs:58-226, orbit_type/mod.rs:399-443, equinoctial_element.rs:809-867, observation_ephemeris.rs:222-275,388-416)
// calls atan2 / hypot / tan / sin / cos of the platform libm per candidate and per observation; its sum is the `rms` field
// and the key of the selection. With these three functions next to sin/cos the whole chain runs on glibc's bits in
// OUTFIT_MATH_GLIBC: `rms` and the selection are then bit-identical to a glibc run, not merely within 1e-6.
//
// Operation for operation as in oracle/glibcmath2.c (which is pinned bit for bit against the installed libm on 2e9
// arguments per function and documents where the algorithm, the fused operations and the tables come from). This
// translation unit is compiled with -ffp-contract=off: only the explicit fma() calls below are fused — exactly the
// operations glibc's FMA variants fuse. Outside the restated range (|x| > 25 for tan) the platform function is used;
// the path's call site (tan of half an inclination) never gets there.
#pragma once
#include <hip/hip_runtime.h>
#include "glibc_tables2.hpp"

namespace outfit {
namespace glm {

// ---- atan2 (sysdeps/ieee754/dbl-64/e_atan2.c, the FMA variant) ------------------------------------------------
constexpr double kHpi = 0x1.921fb54442d18p+0, kOpi = 0x1.921fb54442d18p+1, kQpi = 0x1.921fb54442d18p-1,
                 kTqpi = 0x1.2d97c7f3321d2p+1, kHpi1 = 0x1.1a62633145c07p-54, kOpi1 = 0x1.1a62633145c07p-53;

__device__ __forceinline__ double atan_poly(double v) {
  return fma(fma(fma(fma(fma(0x1.375f08b31cbcep-4, v, -0x1.7458022b13c25p-4), v, 0x1.c71c6e5129a3bp-4), v,
                     -0x1.24924923f7603p-3), v, 0x1.99999999997fdp-3), v, -0x1.5555555555555p-2);
}
__device__ __forceinline__ double cij_poly(const double* c, double v) {
  return fma(fma(fma(fma(c[6], v, c[5]), v, c[4]), v, c[3]), v, c[2]);
}
__device__ __forceinline__ int cij_index(double u) { return (int)(fma(u, 256.0, 0x1p52) - 0x1p52) - 16; }

__device__ __forceinline__ double atan2(double y, double x) {
  const unsigned ux = (unsigned)__double2hiint(x), dx = (unsigned)__double2loint(x);
  const unsigned uy = (unsigned)__double2hiint(y), dy = (unsigned)__double2loint(y);
  if ((ux & 0x7ff00000u) == 0x7ff00000u && (((ux & 0x000fffffu) | dx) != 0)) return x + y;
  if ((uy & 0x7ff00000u) == 0x7ff00000u && (((uy & 0x000fffffu) | dy) != 0)) return y + y;
  if (uy == 0x00000000u) { if (dy == 0) return (ux & 0x80000000u) == 0 ? 0.0 : kOpi; }
  else if (uy == 0x80000000u) { if (dy == 0) return (ux & 0x80000000u) == 0 ? -0.0 : -kOpi; }
  if (x == 0.0) return (uy & 0x80000000u) == 0 ? kHpi : -kHpi;
  if (ux == 0x7ff00000u) {
    if (dx == 0) {
      if (uy == 0x7ff00000u) { if (dy == 0) return kQpi; }
      else if (uy == 0xfff00000u) { if (dy == 0) return -kQpi; }
      else return (uy & 0x80000000u) == 0 ? 0.0 : -0.0;
    }
  } else if (ux == 0xfff00000u) {
    if (dx == 0) {
      if (uy == 0x7ff00000u) { if (dy == 0) return kTqpi; }
      else if (uy == 0xfff00000u) { if (dy == 0) return -kTqpi; }
      else return (uy & 0x80000000u) == 0 ? kOpi : -kOpi;
    }
  }
  if (uy == 0x7ff00000u) { if (dy == 0) return kHpi; }
  else if (uy == 0xfff00000u) { if (dy == 0) return -kHpi; }

  double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
  const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
  if (de >= 59768832) return y > 0.0 ? kHpi : -kHpi;
  if (de <= -59768832) {
    if (x > 0.0) return copysign(ay / ax, y);
    return y > 0.0 ? kOpi : -kOpi;
  }
  if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
  if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }

  // Round 5: the quotient and the four quadrant cases WITHOUT their branches. glibc's code is eight straight-line
  // tails — (x > 0 | x <= 0) x (|y| < |x| | not) x (u < 1/16: polynomial | table) — and a wave of directions spread over
  // the circle used to run all of them one after the other. Operands and signs are selected per lane instead: the
  // quotient is lo / hi for lo, hi = the smaller and the larger magnitude (the two branches were the same
  // operations with the roles swapped); the three cases other than (x > 0, |y| < |x|) differ only in the constant
  // (pi/2 or pi) and in signs, and A - B is A + (-B) exactly; the table tails share their argument v and the inner
  // polynomial. What remains are two tails (polynomial | table) with a select on the case inside. Every lane performs
  // the operations of its own case on the same operands: the same bits
  // (tests/test_device_math_host.py, tests/test_glibcmath2.py).
  const bool y_lt = ay < ax;
  const double lo = y_lt ? ay : ax, hi = y_lt ? ax : ay;
  const double u = lo / hi;
  const double q0 = hi * u, q1 = fma(hi, u, -q0);
  const double du = ((lo - q0) - q1) / hi;
  const bool x_pos = x > 0.0;
  const bool case1 = x_pos && y_lt;                // z = atan(u)
  const bool case3 = !x_pos && ax < ay;            // z = pi/2 + atan(u); cases 2 and 4: pi/2 - atan(u), pi - atan(u)
  const bool case4 = !x_pos && !case3;
  const double K = case4 ? kOpi : kHpi, K1 = case4 ? kOpi1 : kHpi1;
  double z;
  if (u < 0.0625) {
    const double v = u * u, p = atan_poly(v), uv = u * v;
    const double z1 = u + fma(uv, p, du);
    const double zz = uv * p;
    const double su = case3 ? u : -u, sdu = case3 ? du : -du, szz = case3 ? zz : -zz;
    const double t2 = K + su;                      // K > |u| here (u < 1/16): always the first arm of glibc's `cor`
    const double cor = (K - t2) + su;
    const double zo = (((cor + K1) + sdu) + szz) + t2;
    z = case1 ? z1 : zo;
  } else {
    const double* c = kCij[cij_index(u)];
    const double t3 = u - c[0], v = t3 + du;
    const double c1 = c[1], c2 = c[2];
    const double p = fma(fma(fma(c[6], v, c[5]), v, c[4]), v, c[3]);
    const double dv = fabs(t3) > fabs(du) ? (t3 - v) + du : (du - v) + t3;
    const double z1 = fma(v, c2, fma(dv, c2, (v * v) * p)) + c1;
    const double P = fma(p, v, c2);                // cij_poly(c, v)
    const double sc1 = case3 ? c1 : -c1, sP = case3 ? P : -P;
    const double zo = (K + sc1) + fma(sP, v, K1);
    z = case1 ? z1 : zo;
  }
  return copysign(z, y);
}

// ---- tan (sysdeps/ieee754/dbl-64/s_tan.c, the FMA variant), |x| <= 25 -----------------------------------------
__device__ __forceinline__ double tan_poly(double x2) {
  return fma(fma(fma(fma(0x1.2385a3cf2e4eap-7, x2, 0x1.664ed49cfc666p-6), x2, 0x1.ba1ba1cdb8745p-5), x2,
                 0x1.11111111107c6p-3), x2, 0x1.5555555555555p-2);
}
__device__ __forceinline__ double tan(double x) {
  constexpr double g1 = 0x1.b096cp-27, g2 = 0x1.f212dp-5, g3 = 0x1.92f1ap-1, g4 = 25.0, mfftnhf = -15.5,
                   e0 = 0x1.5555555554dbdp-2, e1 = 0x1.11112e0a6b45fp-3;
  if (((unsigned)__double2hiint(x) & 0x7ff00000u) == 0x7ff00000u) return x - x;
  const double w = x < 0.0 ? -x : x;
  if (w <= g1) return x;
  if (w <= g2) {
    const double x2 = x * x;
    return fma(x * x2, tan_poly(x2), x);
  }
  if (w <= g3) {
    const int i = (int)fma(w, 256.0, mfftnhf);
    const double s = x < 0.0 ? -1.0 : 1.0;
    const double z = w - kXfg[i][0], z2 = z * z;
    const double pz = fma(z * z2, fma(z2, e1, e0), z);
    const double fi = kXfg[i][1], gi = kXfg[i][2];
    return (((fi + gi) * pz) / (gi - pz) + fi) * s;
  }
  if (!(w <= g4)) return ::tan(x);  // not restated: unreachable from the path
  const double t = fma(x, 0x1.45f306dc9c883p-1, 0x1.8p52), xn = t - 0x1.8p52;
  const int n = __double2loint(t) & 1;
  const double t1 = fma(-xn, -0x1.dde973cp-27, fma(-xn, 0x1.921fb58p+0, x));
  const double a = fma(-xn, -0x1.cb3b399d747f2p-55, t1);
  const double da = fma(-xn, -0x1.cb3b399d747f2p-55, t1 - a);
  double ya, yya, sy;
  if (a < 0.0) { ya = -a; yya = -da; sy = -1.0; } else { ya = a; yya = da; sy = 1.0; }
  if (ya <= g2) {
    const double a2 = a * a;
    const double t2 = fma(a * a2, tan_poly(a2), da);
    const double b = a + t2;
    if (!n) return b;
    const double db = fabs(a) > fabs(t2) ? (a - b) + t2 : (t2 - b) + a;
    const double c = 1.0 / b;
    const double uu0 = c * b, uu1 = fma(c, b, -uu0);
    const double cc = fma(-db, c, ((1.0 - uu0) - uu1) + 0.0) / b;
    const double zc = c + cc, zzc = (c - zc) + cc;
    return -(zzc + zc);
  }
  const int i = (int)fma(ya, 256.0, mfftnhf);
  const double z = (ya - kXfg[i][0]) + yya, z2 = z * z;
  const double pz = fma(z * z2, fma(z2, e1, e0), z);
  const double fi = kXfg[i][1], gi = kXfg[i][2];
  const double num = (fi + gi) * pz;
  if (n) return (gi - num / (pz + fi)) * -sy;
  return (num / (gi - pz) + fi) * sy;
}

// ---- hypot (sysdeps/ieee754/dbl-64/e_hypot.c of glibc 2.35; no fused operations in the x86-64 build) -----------
__device__ __forceinline__ double hypot_kernel(double ax, double ay) {
  double h = sqrt(ax * ax + ay * ay), t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t2 + t1) / (2.0 * h);
  return h;
}
__device__ __forceinline__ double hypot(double x, double y) {
  if (!isfinite(x) || !isfinite(y)) {
    if (isinf(x) || isinf(y)) return INFINITY;
    return x + y;
  }
  x = fabs(x);
  y = fabs(y);
  const double ax = x < y ? y : x, ay = x < y ? x : y;
  if (ax > 0x1p+511) {
    if (ay <= ax * 0x1p-54) return ax + ay;
    return hypot_kernel(ax * 0x1p-600, ay * 0x1p-600) * 0x1p+600;
  }
  if (ay < 0x1p-459) {
    if (ax >= ay * 0x1p+54) return ax + ay;
    return hypot_kernel(ax * 0x1p+600, ay * 0x1p+600) * 0x1p-600;
  }
  if (ay <= ax * 0x1p-54) return ax + ay;
  return hypot_kernel(ax, ay);
}

}  // namespace glm
}  // namespace outfit
