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
 * glibcmath2.c — atan2, tan, hypot as glibc 2.35 (x86-64, FMA-capable CPU) evaluates them, restated.
 * TEST INFRASTRUCTURE ONLY (see oracle.h). Companion of glibcmath.c (sin, cos, sinh, cosh).
 *
 * Why: the RMS stage of estimate_best_orbit (trajectory.rs:352-427 -> observation_ephemeris.rs:222-275,388-416,
 * equinoctial_element.rs:809-867, orb_elem.rs:58-226, orbit_type/mod.rs:399-443) calls f64::atan2 / hypot / tan / sin /
 * cos per observation, i.e. the platform libm. Its result is the `rms` field AND the argmin key of the selection. With
 * these three functions restated next to sin/cos, the device's OUTFIT_MATH_GLIBC mode evaluates the whole
 * candidate -> RMS chain with glibc's bits: the `rms` field and the selection become bit-identical to a glibc run of the
 * reference, not merely close.
 *
 * What is restated is the published algorithm of a third-party dependency of the reference's toolchain (glibc 2.35,
 * IBM Accurate Mathematical Library: sysdeps/ieee754/dbl-64/e_atan2.c, s_tan.c after the removal of their
 * multi-precision slow paths in 2.34/2.35; sysdeps/ieee754/dbl-64/e_hypot.c, the 2.35 rewrite after C. Borges), with the
 * operation order — in particular which a*b+c are fused — read off the machine code of the variant the dynamic linker
 * selects on FMA/AVX2 hardware (__ieee754_atan2_fma, __tan_fma; hypot has a single SSE2 variant without fused
 * operations) in Ubuntu GLIBC 2.35-0ubuntu3.12. The two lookup tables (`cij` 241 x 7, `xfg` 186 x 4) are accurate-table
 * constants that cannot be regenerated; tools/extract_glibc_tables.py reads them out of that libm.so.6.
 * Pin: tests/test_glibcmath2.py compares the three functions with the host's libm bit for bit wherever that is glibc on
 * x86-64 with FMA (zero differences on 4e9 atan2, 2e9 tan and 2e9 hypot arguments when this was written; tools/glibc_pin2.c).
 * Outside the restated range (|x| > 25 for tan) the platform function is called: unreachable from the path's call site
 * (tan of half an inclination).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"
#include "glibc_tables2.h"

static inline uint64_t asuint64(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline uint32_t hiword(double x) { return (uint32_t)(asuint64(x) >> 32); }
static inline uint32_t loword(double x) { return (uint32_t)asuint64(x); }

/* ---- atan2 (e_atan2.c, __ieee754_atan2_fma) ------------------------------------------------------------------ */
static const double HPI = 0x1.921fb54442d18p+0, OPI = 0x1.921fb54442d18p+1, QPI = 0x1.921fb54442d18p-1,
                    TQPI = 0x1.2d97c7f3321d2p+1, HPI1 = 0x1.1a62633145c07p-54, OPI1 = 0x1.1a62633145c07p-53,
                    TWOM500 = 0x1p-500, TWO500 = 0x1p500, INV16 = 0.0625, TWO52 = 0x1p52, TWO8 = 256.0;
static const double AD3 = -0x1.5555555555555p-2, AD5 = 0x1.99999999997fdp-3, AD7 = -0x1.24924923f7603p-3,
                    AD9 = 0x1.c71c6e5129a3bp-4, AD11 = -0x1.7458022b13c25p-4, AD13 = 0x1.375f08b31cbcep-4;

static inline double atan_poly(double v) { /* d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13)))), every step fused */
  return fma(fma(fma(fma(fma(AD13, v, AD11), v, AD9), v, AD7), v, AD5), v, AD3);
}
static inline double cij_poly(const double *c, double v) { /* c2 + v (c3 + v (c4 + v (c5 + v c6))), fused */
  return fma(fma(fma(fma(c[6], v, c[5]), v, c[4]), v, c[3]), v, c[2]);
}
static inline int cij_index(double u) { return (int)(fma(u, TWO8, TWO52) - TWO52) - 16; }

double orc_glibc_atan2(double y, double x) {
  const uint32_t ux = hiword(x), dx = loword(x), uy = hiword(y), dy = loword(y);
  /* x = NaN or y = NaN */
  if ((ux & 0x7ff00000u) == 0x7ff00000u && (((ux & 0x000fffffu) | dx) != 0)) return x + y;
  if ((uy & 0x7ff00000u) == 0x7ff00000u && (((uy & 0x000fffffu) | dy) != 0)) return y + y;
  /* y = +-0 */
  if (uy == 0x00000000u) { if (dy == 0) return (ux & 0x80000000u) == 0 ? 0.0 : OPI; }
  else if (uy == 0x80000000u) { if (dy == 0) return (ux & 0x80000000u) == 0 ? -0.0 : -OPI; }
  /* x = +-0 */
  if (x == 0.0) return (uy & 0x80000000u) == 0 ? HPI : -HPI;
  /* x = +-inf */
  if (ux == 0x7ff00000u) {
    if (dx == 0) {
      if (uy == 0x7ff00000u) { if (dy == 0) return QPI; }
      else if (uy == 0xfff00000u) { if (dy == 0) return -QPI; }
      else return (uy & 0x80000000u) == 0 ? 0.0 : -0.0;
    }
  } else if (ux == 0xfff00000u) {
    if (dx == 0) {
      if (uy == 0x7ff00000u) { if (dy == 0) return TQPI; }
      else if (uy == 0xfff00000u) { if (dy == 0) return -TQPI; }
      else return (uy & 0x80000000u) == 0 ? OPI : -OPI;
    }
  }
  /* y = +-inf */
  if (uy == 0x7ff00000u) { if (dy == 0) return HPI; }
  else if (uy == 0xfff00000u) { if (dy == 0) return -HPI; }

  double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
  const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
  if (de >= 59768832) return y > 0.0 ? HPI : -HPI;           /* |y/x| > 2^57 */
  if (de <= -59768832) {                                      /* |y/x| < 2^-57 */
    if (x > 0.0) return copysign(ay / ax, y);
    return y > 0.0 ? OPI : -OPI;
  }
  if (ax < TWOM500 || ay < TWOM500) { ax *= TWO500; ay *= TWO500; }
  if (ax > TWO500 || ay > TWO500) { ax *= TWOM500; ay *= TWOM500; }

  double u, du;
  if (ay < ax) {
    u = ay / ax;
    const double v = ax * u, vv = fma(ax, u, -v);             /* EMULV */
    du = ((ay - v) - vv) / ax;
  } else {
    u = ax / ay;
    const double v = ay * u, vv = fma(ay, u, -v);
    du = ((ax - v) - vv) / ay;
  }
  double z;
  if (x > 0.0) {
    if (ay < ax) {                                            /* (i) atan(ay/ax) */
      if (u < INV16) {
        const double v = u * u;
        z = u + fma(u * v, atan_poly(v), du);
      } else {
        const double *c = orc_glibc_cij[cij_index(u)];
        const double t3 = u - c[0], v = t3 + du;
        const double dv = fabs(t3) > fabs(du) ? (t3 - v) + du : (du - v) + t3;   /* EADD */
        const double p = fma(fma(fma(c[6], v, c[5]), v, c[4]), v, c[3]);
        z = fma(v, c[2], fma(dv, c[2], (v * v) * p)) + c[1];
      }
    } else {                                                  /* (ii) pi/2 - atan(ax/ay) */
      if (u < INV16) {
        const double v = u * u, zz = (u * v) * atan_poly(v);
        const double t2 = HPI - u;
        const double cor = HPI > fabs(u) ? (HPI - t2) - u : HPI - (u + t2);      /* ESUB */
        z = (((cor + HPI1) - du) - zz) + t2;
      } else {
        const double *c = orc_glibc_cij[cij_index(u)];
        const double v = (u - c[0]) + du;
        z = (HPI - c[1]) + fma(-cij_poly(c, v), v, HPI1);
      }
    }
  } else if (ax < ay) {                                       /* (iii) pi/2 + atan(ax/ay) */
    if (u < INV16) {
      const double v = u * u, zz = (v * u) * atan_poly(v);
      const double t2 = u + HPI;
      const double cor = HPI > fabs(u) ? (HPI - t2) + u : (u - t2) + HPI;        /* EADD */
      z = (((cor + HPI1) + du) + zz) + t2;
    } else {
      const double *c = orc_glibc_cij[cij_index(u)];
      const double v = (u - c[0]) + du;
      z = (HPI + c[1]) + fma(cij_poly(c, v), v, HPI1);
    }
  } else {                                                    /* (iv) pi - atan(ay/ax) */
    if (u < INV16) {
      const double v = u * u, zz = (v * u) * atan_poly(v);
      const double t2 = OPI - u;
      const double cor = OPI > fabs(u) ? (OPI - t2) - u : OPI - (t2 + u);        /* ESUB */
      z = (((cor + OPI1) - du) - zz) + t2;
    } else {
      const double *c = orc_glibc_cij[cij_index(u)];
      const double v = (u - c[0]) + du;
      z = (OPI - c[1]) + fma(-cij_poly(c, v), v, OPI1);
    }
  }
  return copysign(z, y);
}

/* ---- tan (s_tan.c, __tan_fma), |x| <= 25 ----------------------------------------------------------------------- */
static const double TG1 = 0x1.b096cp-27, TG2 = 0x1.f212dp-5, TG3 = 0x1.92f1ap-1, TG4 = 25.0, MFFTNHF = -15.5,
                    TD3 = 0x1.5555555555555p-2, TD5 = 0x1.11111111107c6p-3, TD7 = 0x1.ba1ba1cdb8745p-5,
                    TD9 = 0x1.664ed49cfc666p-6, TD11 = 0x1.2385a3cf2e4eap-7, TE0 = 0x1.5555555554dbdp-2,
                    TE1 = 0x1.11112e0a6b45fp-3, THPINV = 0x1.45f306dc9c883p-1, TTOINT = 0x1.8p52,
                    TMP1 = 0x1.921fb58p+0, TMP2 = -0x1.dde973cp-27, TMP3 = -0x1.cb3b399d747f2p-55;

static inline double tan_poly(double x2) { /* d3 + x2 (d5 + x2 (d7 + x2 (d9 + x2 d11))), fused */
  return fma(fma(fma(fma(TD11, x2, TD9), x2, TD7), x2, TD5), x2, TD3);
}

double orc_glibc_tan(double x) {
  if ((hiword(x) & 0x7ff00000u) == 0x7ff00000u) return x - x;
  const double w = x < 0.0 ? -x : x;
  if (w <= TG1) return x;                                                       /* (I) */
  if (w <= TG2) {                                                               /* (II) */
    const double x2 = x * x;
    return fma(x * x2, tan_poly(x2), x);
  }
  if (w <= TG3) {                                                               /* (III) */
    const int i = (int)fma(w, TWO8, MFFTNHF);
    const double s = x < 0.0 ? -1.0 : 1.0;
    const double z = w - orc_glibc_xfg[i][0], z2 = z * z;
    const double pz = fma(z * z2, fma(z2, TE1, TE0), z);
    const double fi = orc_glibc_xfg[i][1], gi = orc_glibc_xfg[i][2];
    return (((fi + gi) * pz) / (gi - pz) + fi) * s;
  }
  if (!(w <= TG4)) return tan(x);                                               /* not restated: unreachable here */
  /* range reduction by algorithm i */
  const double t = fma(x, THPINV, TTOINT), xn = t - TTOINT;
  const int n = (int)(loword(t) & 1u);
  const double t1 = fma(-xn, TMP2, fma(-xn, TMP1, x));
  const double a = fma(-xn, TMP3, t1);
  const double da = fma(-xn, TMP3, t1 - a);
  double ya, yya, sy;
  if (a < 0.0) { ya = -a; yya = -da; sy = -1.0; } else { ya = a; yya = da; sy = 1.0; }
  if (ya <= TG2) {                                                              /* |y| <= 0.0608 */
    const double a2 = a * a;
    const double t2 = fma(a * a2, tan_poly(a2), da);
    const double b = a + t2;
    if (!n) return b;                                                           /* tan */
    const double db = fabs(a) > fabs(t2) ? (a - b) + t2 : (t2 - b) + a;         /* EADD */
    /* -cot: DIV2 (1.0, 0.0, b, db) */
    const double c = 1.0 / b;
    const double uu0 = c * b, uu1 = fma(c, b, -uu0);
    const double cc = fma(-db, c, ((1.0 - uu0) - uu1) + 0.0) / b;
    const double zc = c + cc, zzc = (c - zc) + cc;
    return -(zzc + zc);
  }
  /* 0.0608 < |y| <= 0.787 */
  const int i = (int)fma(ya, TWO8, MFFTNHF);
  const double z = (ya - orc_glibc_xfg[i][0]) + yya, z2 = z * z;
  const double pz = fma(z * z2, fma(z2, TE1, TE0), z);
  const double fi = orc_glibc_xfg[i][1], gi = orc_glibc_xfg[i][2];
  const double num = (fi + gi) * pz;
  if (n) return (gi - num / (pz + fi)) * -sy;                                   /* -cot */
  return (num / (gi - pz) + fi) * sy;                                           /* tan */
}

/* ---- hypot (e_hypot.c of glibc 2.35; no fused operations in the x86-64 build) ------------------------------------ */
static inline double hypot_kernel(double ax, double ay) {
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
double orc_glibc_hypot(double x, double y) {
  if (!isfinite(x) || !isfinite(y)) {
    if (isinf(x) || isinf(y)) return INFINITY;   /* (signalling NaNs aside) */
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
