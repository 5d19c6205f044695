#!/usr/bin/env python3
"""Generates the constants of the correctly-rounded sin/cos/acos used by BOTH the oracle
(oracle/crmath.c) and the device code (outfit_amd/csrc/device/crmath.hpp):
  * pi/2 in four parts (3 x 33 significant bits + remainder) for exact Cody-Waite reduction,
  * 1/n! as double-double pairs for the Taylor kernels,
  * sin/cos of Aberth's 8 initial-guess angles theta_k = fma(2pi/8, k, (pi/2)/8), correctly rounded.
Needs mpmath (high precision); output is pasted into the two source files.
"""
import math
import struct

import mpmath as mp

mp.mp.prec = 400


def to_double(x):
    return float(mp.mpf(x))  # mpmath rounds to nearest


def hexd(d):
    return d.hex()


def split33(x):
    """leading 33 significant bits of x (truncated), as a double"""
    m, e = mp.frexp(x)           # x = m * 2^e, 0.5 <= m < 1
    t = mp.floor(m * mp.mpf(2) ** 33) / mp.mpf(2) ** 33
    return float(mp.ldexp(t, e))


pio2 = mp.pi / 2
p1 = split33(pio2)
p2 = split33(pio2 - mp.mpf(p1))
p3 = split33(pio2 - mp.mpf(p1) - mp.mpf(p2))
p4 = to_double(pio2 - mp.mpf(p1) - mp.mpf(p2) - mp.mpf(p3))
print("// pi/2 = P1 + P2 + P3 + P4 (P1..P3 carry 33 significant bits each)")
for n, v in (("P1", p1), ("P2", p2), ("P3", p3), ("P4", p4)):
    print(f"#define CRM_PIO2_{n} {v!r} /* {hexd(v)} */")
tw = to_double(2 / mp.pi)
print(f"#define CRM_TWO_OVER_PI {tw!r}")
pi_hi = to_double(mp.pi)
pi_lo = to_double(mp.pi - mp.mpf(pi_hi))
print(f"#define CRM_PI_HI {pi_hi!r}\n#define CRM_PI_LO {pi_lo!r}")

print("// 1/n! as double-double (hi, lo), n = 0..31")
rows = []
for n in range(32):
    v = 1 / mp.factorial(n)
    hi = to_double(v)
    lo = to_double(v - mp.mpf(hi))
    rows.append(f"  {{{hi!r}, {lo!r}}},")
print("static const double CRM_INV_FACT[32][2] = {")
print("\n".join(rows))
print("};")

print("// Aberth initial-guess angles: theta_k = fma(2pi/8, k, (pi/2)/8); (cos, sin) correctly rounded")
two_pi = 2.0 * math.pi
frac_2pi_n = two_pi / 8.0
frac_pi_2n = (math.pi / 2.0) / 8.0
print("static const double CRM_ABERTH_COS_SIN[8][2] = {")
for k in range(8):
    # fma(frac_2pi_n, k, frac_pi_2n) evaluated exactly then rounded once
    theta = to_double(mp.mpf(frac_2pi_n) * k + mp.mpf(frac_pi_2n))
    c = to_double(mp.cos(mp.mpf(theta)))
    s = to_double(mp.sin(mp.mpf(theta)))
    print(f"  {{{c!r}, {s!r}}},  /* theta = {theta!r} */")
print("};")
