#!/usr/bin/env python3
"""Generates tests/golden/crmath_hyp_vectors.json: correctly-rounded sinh / cosh / log reference values (mpmath,
400 bits, rounded to nearest) for the arguments the oracle's and the device's versions are tested on — the range of
the hyperbolic cold start (|f| < 15, y = c + sqrt(c^2 - 1) > 1), tiny and large arguments, the 2^-3 switch of the
device code, multiples of ln 2 and of 1/128 (table knots), and HARD cases: arguments, found by search, whose exact
value lies within 2^-12 ulp of a rounding boundary (they force the second step of Ziv's strategy on the device)."""
import json
import math
import os
import random

import mpmath as mp

mp.mp.prec = 400
random.seed(20261004)


def cr(v):
    return float(v)  # mpmath -> float rounds to nearest


def hardness(v):
    """distance of v from the nearest rounding boundary (midpoint of two doubles), in ulps"""
    d = cr(v)
    m, e = math.frexp(d)
    ulp = mp.ldexp(1, e - 53)
    frac = (v - mp.mpf(d)) / ulp  # in [-0.5, 0.5]
    return float(0.5 - abs(frac))


def f_args(n):
    out = []
    for i in range(n):
        k = i % 8
        if k < 3:
            f = random.uniform(-15, 15)
        elif k == 3:
            f = random.uniform(-0.25, 0.25)
        elif k == 4:
            f = random.uniform(-1, 1) * 2.0 ** random.uniform(-40, -3)
        elif k == 5:
            f = random.randint(-20, 20) * math.log(2) + random.uniform(-1e-9, 1e-9)
        elif k == 6:
            f = random.randint(-1900, 1900) / 128 + random.uniform(-1e-12, 1e-12)
        else:
            f = random.uniform(-700, 700)
        out.append(f)
    return out + [0.0, -0.0, 0.125, -0.125, math.nextafter(0.125, 0), 15.0, 1e-300, 700.0, -700.0]


def y_args(n):
    out = []
    for i in range(n):
        k = i % 6
        if k < 2:
            y = 1 + random.uniform(0, 50)
        elif k == 2:
            c = 1 + 2.0 ** random.uniform(-52, -4)      # c -> 1+: y = c + sqrt(c^2 - 1) just above 1
            y = c + math.sqrt(c * c - 1)
        elif k == 3:
            y = math.exp(random.uniform(0, 15))
        elif k == 4:
            y = 1 + 2.0 ** random.uniform(-30, 0)
        else:
            y = 10.0 ** random.uniform(0, 290)
        if y > 1.0:
            out.append(y)
    return out + [math.nextafter(1.0, 2), 2.0, math.e, 1e300 * 0.99]


sh, lg = [], []
for f in f_args(1600):
    x = mp.mpf(f)
    sh.append([f.hex(), (f if f == 0.0 else cr(mp.sinh(x))).hex(), cr(mp.cosh(x)).hex()])  # sinh(-0) = -0
for y in y_args(1200):
    lg.append([y.hex(), cr(mp.log(mp.mpf(y))).hex()])
# hard cases by search
hard_sh, hard_lg = [], []
tries = 0
while len(hard_sh) < 60 and tries < 4000000:
    tries += 1
    f = random.uniform(-15, 15) if tries % 3 else random.uniform(-0.3, 0.3)
    x = mp.mpf(f)
    if min(hardness(mp.sinh(x)), hardness(mp.cosh(x))) < 2.0 ** -12:
        hard_sh.append([f.hex(), cr(mp.sinh(x)).hex(), cr(mp.cosh(x)).hex()])
tries = 0
while len(hard_lg) < 40 and tries < 4000000:
    tries += 1
    y = 1 + random.uniform(0, 30) if tries % 2 else 1 + 2.0 ** random.uniform(-26, -2)
    if hardness(mp.log(mp.mpf(y))) < 2.0 ** -12:
        hard_lg.append([y.hex(), cr(mp.log(mp.mpf(y))).hex()])
# atan2: all quadrants, near the axes, tiny and huge ratios, unit-vector components (as orb_elem.rs feeds it)
at, hard_at = [], []
for i in range(1500):
    k = i % 6
    if k < 2:
        th = random.uniform(-math.pi, math.pi)
        r = 10.0 ** random.uniform(-3, 3)
        y, x = r * math.sin(th), r * math.cos(th)
    elif k == 2:
        y, x = random.uniform(-1, 1) * 2.0 ** random.uniform(-60, 0), random.choice([-1, 1]) * random.uniform(0.5, 2)
    elif k == 3:
        x, y = random.uniform(-1, 1) * 2.0 ** random.uniform(-60, 0), random.choice([-1, 1]) * random.uniform(0.5, 2)
    elif k == 4:
        y, x = random.uniform(-1, 1), random.uniform(-1, 1)
    else:
        y, x = random.uniform(-1, 1) * 10.0 ** random.uniform(-100, 100), random.uniform(-1, 1) * 10.0 ** random.uniform(-100, 100)
    if y != 0.0 and x != 0.0:
        at.append([y.hex(), x.hex(), cr(mp.atan2(mp.mpf(y), mp.mpf(x))).hex()])
tries = 0
while len(hard_at) < 40 and tries < 4000000:
    tries += 1
    y, x = random.uniform(-1, 1), random.uniform(-1, 1)
    v = mp.atan2(mp.mpf(y), mp.mpf(x))
    if hardness(v) < 2.0 ** -12:
        hard_at.append([y.hex(), x.hex(), cr(v).hex()])
out = {"atan2": at, "atan2_hard": hard_at, "source": "tools/gen_crmath_hyp_vectors.py (mpmath 400-bit, rounded to nearest)",
       "sinhcosh": sh, "log": lg, "sinhcosh_hard": hard_sh, "log_hard": hard_lg}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "crmath_hyp_vectors.json")
json.dump(out, open(path, "w"))
print("wrote", path, len(sh), len(lg), len(hard_sh), len(hard_lg), len(at), len(hard_at))
