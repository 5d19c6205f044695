"""Error bounds of the screening arithmetic's primitives (device/screen_math.hpp), on the CPU.

K2c in the glibc-faithful mode is a SCREEN: its sums only choose which candidates K3b re-evaluates exactly, and every
candidate carries an error bound (device/screen.hpp: screen_level / screen_slack_rad, >= 1e-13 rad). That bound is
analytic for the one path-dependent source (the reference's Newton stopping rule); the screen's own roundings — its
atan2, its reciprocal and square-root refinements, the drift of a (sin, cos) pair carried through small rotations —
were until round 4 covered by population measurements only (tests/test_gpu_glibc_mode.py). Here the very header is
compiled for the host (tests/host_harness: the two hardware seed instructions emulated at the worst accuracy their ISA
documentation allows, relative error 2^-23 of either sign) and each primitive is held against long-double references
on dense and adversarial arguments. What is asserted is what screen.hpp's comment claims: all of it together is below
1 % of the smallest slack.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "host_harness")
BUILD = os.path.join(ROOT, "oracle", "_build")
SMALLEST_SLACK = 1.0e-13          # screen_slack_rad(0)
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def H():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libscreen_math_host.so")
    src = os.path.join(HARNESS, "screen_math_host.cpp")
    hdr = os.path.join(ROOT, "outfit_amd", "csrc", "device", "screen_math.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + HARNESS,
                        "-I" + os.path.join(ROOT, "outfit_amd", "csrc"), src, "-o", so], check=True)
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(_dp)


SEED_ERRORS = [0.0, 2.0 ** -23, -2.0 ** -23]


@pytest.mark.parametrize("seed_err", SEED_ERRORS)
def test_reciprocal_and_square_root_refinements(H, seed_err):
    """rcp_nr2 and sqrt_rsqrt are within 2 ulp of the exact value for ANY seed within 2^-23 (two Newton rounds:
    2^-23 -> 2^-46 -> 2^-92, then the roundings of the last step); rcp_nr1 is within 2^-45."""
    H.set_seed_error(C.c_double(seed_err))
    rng = np.random.default_rng(1)
    d = np.concatenate([rng.uniform(0.5, 2.0, 400000), 10.0 ** rng.uniform(-8, 8, 400000),
                        -10.0 ** rng.uniform(-3, 3, 100000)])
    r1, r2 = np.empty_like(d), np.empty_like(d)
    H.rcp_array(C.c_uint64(d.size), _p(d), _p(r1), _p(r2))
    exact = 1.0 / d.astype(np.longdouble)
    assert float(np.max(np.abs((r2 - exact) / exact))) < 2 * 2.0 ** -52
    assert float(np.max(np.abs((r1 - exact) / exact))) < 2.0 ** -45
    x = np.concatenate([rng.uniform(0.25, 4.0, 400000), 10.0 ** rng.uniform(-12, 12, 400000)])
    root, inv = np.empty_like(x), np.empty_like(x)
    H.sqrt_rsqrt_array(C.c_uint64(x.size), _p(x), _p(root), _p(inv))
    ex = np.sqrt(x.astype(np.longdouble))
    assert float(np.max(np.abs((root - ex) / ex))) < 2 * 2.0 ** -52
    assert float(np.max(np.abs((inv * ex) - 1.0))) < 3 * 2.0 ** -52
    z = np.array([0.0, -1.0])
    H.sqrt_rsqrt_array(C.c_uint64(2), _p(z), _p(root), _p(inv))
    assert root[0] == 0.0 and inv[0] == 0.0 and root[1] == 0.0 and inv[1] == 0.0    # x <= 0 gives 0, 0


@pytest.mark.parametrize("seed_err", SEED_ERRORS)
def test_atan2_of_the_screen(H, seed_err):
    """|atan2_screen - atan2| <= 6e-16 rad everywhere (largest seen 4.7e-16, near +-pi, where the rounding of the
    result and of `pi - r` alone are 2 x 2.2e-16; ~2e-16 in the first octant): 6 x 10^6 directions — uniform angles, every octant boundary and the tan(pi/8) switch of the argument
    reduction approached to within 1e-9, both axes, magnitudes from 1e-6 to 1e3 (the residual's components are
    AU-sized). Quadrants and signs as atan2; (0, 0) -> 0."""
    H.set_seed_error(C.c_double(seed_err))
    rng = np.random.default_rng(2)
    th = np.concatenate([rng.uniform(-np.pi, np.pi, 4_000_000)] +
                        [k * np.pi / 8 + s * 10.0 ** rng.uniform(-9, -1, 60000) for k in range(-8, 9) for s in (-1.0, 1.0)])
    th = th[(th > -np.pi) & (th < np.pi)]
    rho = 10.0 ** rng.uniform(-6, 3, th.size)
    y, x = np.sin(th.astype(np.longdouble)) * rho, np.cos(th.astype(np.longdouble)) * rho
    y, x = np.ascontiguousarray(y.astype(np.float64)), np.ascontiguousarray(x.astype(np.float64))
    out = np.empty_like(x)
    H.atan2_screen_array(C.c_uint64(x.size), _p(y), _p(x), _p(out))
    ref = np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))
    err = np.abs(out - ref)
    assert float(err.max()) <= 6.0e-16, float(err.max())
    first_octant = (np.abs(ref) < np.pi / 4)
    assert float(err[first_octant].max()) <= 2.5e-16, float(err[first_octant].max())
    assert float(err.max()) < 0.01 * SMALLEST_SLACK
    sp_y = np.array([0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 1e-300, -3.0])
    sp_x = np.array([0.0, 1.0, 0.0, 0.0, -1.0, -1.0, 1e-300, -3.0])
    got = np.empty_like(sp_x)
    H.atan2_screen_array(C.c_uint64(sp_x.size), _p(sp_y), _p(sp_x), _p(got))
    want = np.arctan2(sp_y, sp_x)
    assert got[0] == 0.0 and np.allclose(got[1:], want[1:], rtol=0, atol=6e-16)
    assert np.signbit(got[5]) == np.signbit(want[5])       # atan2(-0, -1) = -pi


def test_carried_sine_and_cosine_drift(H):
    """A (sin F, cos F) pair carried through `steps` rotations by |d| <= 1/16 (screen_kepler: one prediction and a few
    Newton steps per observation, tens per candidate, hundreds on a 300-observation arc) against the exact angle sum:
    the drift is a random walk of ~1e-16 per rotation; 400 rotations stay below 1e-14 rad in both components and in
    the pair's norm — 10 % of the smallest slack for the longest chains, ~1 % for the bench's 20 observations."""
    H.set_seed_error(C.c_double(0.0))
    rng = np.random.default_rng(3)
    for steps, bound in ((60, 2.0e-15), (400, 1.0e-14)):
        n = 20000
        x0 = rng.uniform(-np.pi, np.pi, n)
        d = np.ascontiguousarray(rng.uniform(-1.0, 1.0, (n, steps)) * 10.0 ** rng.uniform(-6, np.log10(0.0625), (n, steps)))
        s, c = np.empty(n), np.empty(n)
        H.rotate_chain(C.c_uint64(n), C.c_uint(steps), _p(x0), _p(d), _p(s), _p(c))
        # the exact angle: x0 + sum of the steps, in long double (the steps are what the chain was given, exactly)
        ang = x0.astype(np.longdouble) + d.astype(np.longdouble).sum(axis=1)
        es, ec = np.abs(s - np.sin(ang)), np.abs(c - np.cos(ang))
        assert float(max(es.max(), ec.max())) < bound, (steps, float(es.max()), float(ec.max()))
        assert float(np.abs(np.hypot(s, c) - 1.0).max()) < bound
