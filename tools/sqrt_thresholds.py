#!/usr/bin/env python3
"""The two constants of outfit_amd/csrc/device/kepler.hpp that replace a square root by a compare on its operand.

sqrt is correctly rounded and monotone, so for a positive double c the set {x >= 0 : RN(sqrt(x)) <= c} is an initial
segment [0, T(c)] of the doubles. RN(sqrt(x)) <= c  <=>  sqrt(x) < m with m the midpoint between c and its successor
(a tie is impossible: m has 54 significant bits, m^2 more than 106, and x is a double), i.e. x < m^2 — evaluated here
with exact rational arithmetic.

    kHnormSqMax = T(1e6 * 2^-52)   |r x v| <= 1e6 eps   (kepler/velocity.rs:167-178)
    kFrobSqMax  = T(2^-52)         |P|_F  <= eps        (gauss.rs:1392)

    python tools/sqrt_thresholds.py        # prints the constants as C hex-float literals
tests/test_device_math_host.py::test_decisions_on_operands_equal_the_literal_expressions checks the header against this."""
import math
from fractions import Fraction


def threshold(c):
    up = math.nextafter(c, math.inf)
    m2 = ((Fraction(c) + Fraction(up)) / 2) ** 2
    t = float(m2)
    while Fraction(t) >= m2:
        t = math.nextafter(t, 0.0)
    while Fraction(math.nextafter(t, math.inf)) < m2:
        t = math.nextafter(t, math.inf)
    assert math.sqrt(t) <= c < math.sqrt(math.nextafter(t, math.inf))      # math.sqrt is correctly rounded
    return t


EPS = 2.0 ** -52
CONSTANTS = {"kHnormSqMax": threshold(1e6 * EPS), "kFrobSqMax": threshold(EPS)}

if __name__ == "__main__":
    for name, v in CONSTANTS.items():
        print(f"constexpr double {name} = {v.hex()};   // {v!r}")
