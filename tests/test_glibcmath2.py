"""glibc 2.35's atan2 / tan / hypot / acos / log and its STANDALONE sin / cos, restated (oracle/glibcmath2.c, glibcmath3.c,
oracle/glibcmath.c `orc_glibc_sin1 / cos1`; device/glibcmath2.hpp, glibcmath3.hpp, glm::sin1 / cos1).

Pins: on a glibc / x86-64 / FMA host the restatements equal the platform libm bit for bit (2 x 10^6 arguments per
function here; tools/glibc_pin2.c and glibc_pin3.c ran 2 x 10^9). The finding of round 3 is asserted too: glibc's sin(x) / cos(x) and
its sincos(x) are DIFFERENT functions at the last bit (0.07 % of arguments) — the first two are what an isolated
`x.sin()` of the reference calls, the third what a sin/cos pair on one operand calls (LLVM's FSINCOS combine).
On the device (-m gpu) the same functions equal the restatement bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from test_glibcmath import host_is_glibc_x86_64, pinned_libm_reason

dp = C.POINTER(C.c_double)
N = 2_000_000


@pytest.fixture(scope="module")
def L():
    lib = O.lib()
    for name, nargs in (("orc_glibc_sin", 1), ("orc_glibc_cos", 1), ("orc_glibc_sin1", 1), ("orc_glibc_cos1", 1),
                        ("orc_glibc_tan", 1), ("orc_glibc_atan2", 2), ("orc_glibc_hypot", 2), ("orc_glibc_acos", 1), ("orc_glibc_log", 1)):
        f = getattr(lib, name)
        f.restype, f.argtypes = C.c_double, [C.c_double] * nargs
    return lib


def args(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "angle":
        return np.concatenate([rng.uniform(-40, 40, n // 2), rng.uniform(-7, 7, n // 4), rng.uniform(-1e5, 1e5, n // 8),
                               rng.uniform(-1, 1, n // 8) * 2.0 ** rng.integers(-30, 2, n // 8)])
    if kind == "acos":   # the whole domain, dense near +-1 and around every interval boundary of e_asin.c, tiny, out of range
        edges = np.array([0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 1.0])
        near = np.nextafter(np.repeat(edges, 41), np.inf) * 0 + (np.repeat(edges, 41).view(np.int64) + np.tile(np.arange(-20, 21), 8)).view(np.float64)
        parts = [rng.uniform(-1, 1, n // 2), 1.0 - rng.uniform(0, 1, n // 4) * 2.0 ** -rng.integers(0, 54, n // 4).astype(float),
                 rng.uniform(-1, 1, n // 8) * 2.0 ** rng.integers(-70, 0, n // 8), near, -near,
                 np.array([0.0, -0.0, 1.0, -1.0, 1.5, -1.5, np.inf, -np.inf, np.nan])]
        x = np.concatenate(parts)
        x[n // 2: n // 2 + n // 8] *= -1.0
        return x
    if kind == "log":    # the hyperbolic cold start passes c + sqrt(c^2 - 1) > 1; the polynomial branch around 1; every exponent
        return np.concatenate([1.0 + rng.uniform(0, 30, n // 2), np.exp(rng.uniform(-700, 700, n // 4)),
                               1.0 + rng.uniform(-0.07, 0.07, n // 8), 1.0 + rng.uniform(-1, 1, n // 8) * 2.0 ** -rng.integers(4, 53, n // 8).astype(float),
                               np.array([1.0, 0.9375, 1.064697265625, np.nextafter(0.9375, 0), np.nextafter(1.064697265625, 2), 5e-324, 1e-310, np.inf])])
    if kind == "tan":
        return np.concatenate([rng.uniform(-1.6, 1.6, n // 2), rng.uniform(-25, 25, n // 4),
                               1.5707963267948966 + rng.uniform(-1, 1, n // 8) * 2.0 ** rng.integers(-40, -3, n // 8),
                               rng.uniform(-1, 1, n // 8) * 2.0 ** rng.integers(-30, 0, n // 8)])
    t = rng.uniform(-np.pi, np.pi, n)
    r = np.exp(rng.uniform(-10, 10, n))
    y, x = r * np.sin(t), r * np.cos(t)
    k = n // 4
    x[:k] = rng.uniform(-2, 2, k) * 2.0 ** rng.integers(-60, 60, k)          # every exponent gap incl. |y/x| ~ 2^+-57
    y[:k] = x[:k] * rng.uniform(-2, 2, k) * 2.0 ** rng.integers(-62, 62, k)
    return y, x


def vec(f, *a):
    return np.array([f(*(float(v) for v in row)) for row in zip(*a)])


@pytest.mark.skipif(not host_is_glibc_x86_64(), reason=pinned_libm_reason())
def test_restatements_equal_the_platform_libm(L):
    libm = C.CDLL("libm.so.6")   # the entry points themselves (numpy may route through its own SIMD kernels)
    for name, nargs in (("sin", 1), ("cos", 1), ("tan", 1), ("atan2", 2), ("hypot", 2), ("acos", 1), ("log", 1)):
        f = getattr(libm, name)
        f.restype, f.argtypes = C.c_double, [C.c_double] * nargs
    n = N // 10
    x = args("angle", n, 1)
    assert np.array_equal(vec(L.orc_glibc_sin1, x), vec(libm.sin, x))
    assert np.array_equal(vec(L.orc_glibc_cos1, x), vec(libm.cos, x))
    t = args("tan", n, 2)
    assert np.array_equal(vec(L.orc_glibc_tan, t), vec(libm.tan, t))
    y, xx = args("pair", n, 3)
    assert np.array_equal(vec(L.orc_glibc_atan2, y, xx), vec(libm.atan2, y, xx))
    assert np.array_equal(vec(L.orc_glibc_hypot, y, xx), vec(libm.hypot, y, xx))
    a = args("acos", n, 5)
    got, want = vec(L.orc_glibc_acos, a), vec(libm.acos, a)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)].view(np.uint64), want[~np.isnan(want)].view(np.uint64))
    v = args("log", n, 6)
    assert np.array_equal(vec(L.orc_glibc_log, v).view(np.uint64), vec(libm.log, v).view(np.uint64))


@pytest.mark.skipif(not host_is_glibc_x86_64(), reason=pinned_libm_reason())
def test_glibc_sin_and_sincos_are_different_functions(L):
    """sincos(x) (one SSE2 variant) and sin(x) / cos(x) (FMA variants) of glibc 2.35 disagree in the last bit for
    0.03 - 0.2 % of arguments; the two restatements reproduce their respective originals exactly."""
    libm = C.CDLL("libm.so.6")
    libm.sincos.argtypes = [C.c_double, dp, dp]
    libm.sin.restype = libm.cos.restype = C.c_double
    libm.sin.argtypes = libm.cos.argtypes = [C.c_double]
    x = args("angle", 400_000, 4)
    s, c = C.c_double(), C.c_double()
    differ = 0
    for v in x:
        v = float(v)
        libm.sincos(v, C.byref(s), C.byref(c))
        a, b = libm.sin(v), libm.cos(v)
        assert (s.value, c.value) == (L.orc_glibc_sin(v), L.orc_glibc_cos(v))      # the sincos-equivalent restatement
        assert (a, b) == (L.orc_glibc_sin1(v), L.orc_glibc_cos1(v))                # the standalone restatement
        differ += (a != s.value) + (b != c.value)
    assert 0.0002 * 2 * len(x) < differ < 0.003 * 2 * len(x), differ


@pytest.mark.gpu
def test_device_functions_equal_the_restatement(L):
    from outfit_amd import _lib
    H = _lib.lib()
    n = 400_000

    def dev(fn, x, y=None):
        x = np.ascontiguousarray(x)
        out = np.empty_like(x)
        yy = None if y is None else np.ascontiguousarray(y)
        rc = H.outfit_hip_glibc_eval(fn, len(x), x.ctypes.data_as(dp), None if yy is None else yy.ctypes.data_as(dp),
                                     out.ctypes.data_as(dp))
        assert rc == 0, _lib.last_error()
        return out
    x = args("angle", n, 11)
    assert np.array_equal(dev(0, x).view(np.uint64), vec(L.orc_glibc_sin1, x).view(np.uint64))
    assert np.array_equal(dev(1, x).view(np.uint64), vec(L.orc_glibc_cos1, x).view(np.uint64))
    t = args("tan", n, 12)
    assert np.array_equal(dev(2, t).view(np.uint64), vec(L.orc_glibc_tan, t).view(np.uint64))
    y, xx = args("pair", n, 13)
    assert np.array_equal(dev(3, y, xx).view(np.uint64), vec(L.orc_glibc_atan2, y, xx).view(np.uint64))
    assert np.array_equal(dev(4, y, xx).view(np.uint64), vec(L.orc_glibc_hypot, y, xx).view(np.uint64))
    a = args("acos", n, 14)
    got, want = dev(5, a), vec(L.orc_glibc_acos, a)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)].view(np.uint64), want[~np.isnan(want)].view(np.uint64))
    v = args("log", n, 15)
    assert np.array_equal(dev(6, v).view(np.uint64), vec(L.orc_glibc_log, v).view(np.uint64))
    # special values of atan2 (e_atan2.c's prologue) and hypot
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-320, 1e308])
    yy, xs = np.meshgrid(sp, sp)
    yy, xs = yy.ravel(), xs.ravel()
    a, b = dev(3, yy, xs), vec(L.orc_glibc_atan2, yy, xs)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))
    a, b = dev(4, yy, xs), vec(L.orc_glibc_hypot, yy, xs)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))
