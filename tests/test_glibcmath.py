"""OUTFIT_MATH_GLIBC: sin / cos / sinh / cosh with the bits of glibc 2.35 (oracle/glibcmath.c, device/glibcmath.hpp).

CPU part: the restatement against the host libm (when that is glibc on x86-64), and the reference's one
libm-dependent KAT (kepler/params.rs:689-725, assert_eq!) in the oracle's restated mode. GPU part: the device's
functions against the oracle's, bit for bit, and the same KAT through the C ABI."""
import ctypes as C
import json
import os
import platform

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def host_is_glibc_x86_64():
    """True only on the ONE libm build the glibc-faithful mode restates (glibc 2.35, x86-64, FMA/AVX2 ifunc variants,
    the libm.so.6 whose sha256 tools/extract_glibc_tables.py knows): on any other libc the "== platform libm" tests
    skip instead of failing (outfit_amd/libm_pin.py)."""
    from outfit_amd.libm_pin import host_has_pinned_libm
    return host_has_pinned_libm()


def pinned_libm_reason():
    from outfit_amd.libm_pin import host_libm_identity
    return "platform libm is not the build OUTFIT_MATH_GLIBC restates: " + str(host_libm_identity()["why_not"])


@pytest.fixture(scope="module")
def L():
    return O.lib()


@pytest.mark.skipif(not host_is_glibc_x86_64(), reason=pinned_libm_reason())
def test_restatement_returns_the_host_libm_bits(L):
    """Zero differences on 8 x 10^6 arguments per function (tools/glibc_pin.c ran 1.5e9 / 1e8 when this was written)."""
    counts = (C.c_uint64 * 6)()
    L.orc_glibc_compare_with_host(C.c_uint64(8_000_000), C.c_uint64(20260102), counts)
    assert list(counts) == [0] * 6, dict(zip(("sin", "cos", "expm1", "exp", "sinh", "cosh"), counts))


def test_restatement_is_faithful_and_differs_from_correct_rounding_where_glibc_does(L):
    """Platform-independent sanity: within 2 ulp of the correctly-rounded value everywhere, and NOT identical to it
    (glibc is not correctly rounded: ~0.15 % of sin/cos and 13-20 % of sinh/cosh arguments differ)."""
    rng = np.random.default_rng(5)
    for name, cr, lo, hi, frac in (("sin", "orc_sin", -40.0, 40.0, (2e-4, 6e-3)), ("cos", "orc_cos", -40.0, 40.0, (2e-4, 6e-3)),
                                   ("sinh", None, -15.0, 15.0, (0.05, 0.4)), ("cosh", None, -15.0, 15.0, (0.03, 0.4))):
        g = getattr(L, "orc_glibc_" + name)
        g.restype = C.c_double
        g.argtypes = [C.c_double]
        ref = getattr(L, cr if cr else "orc_" + name)
        ref.restype = C.c_double
        ref.argtypes = [C.c_double]
        xs = rng.uniform(lo, hi, 20000)
        a = np.array([g(float(x)) for x in xs])
        b = np.array([ref(float(x)) for x in xs])   # default oracle mode: correctly rounded
        ulp = np.spacing(np.abs(b))
        assert (np.abs(a - b) <= 2 * ulp).all(), name   # glibc documents up to 2 ulp for sinh/cosh
        differ = float((a != b).mean())
        assert frac[0] < differ < frac[1], (name, differ)


def kat_cases():
    return json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))["prelim_kepuni"]["cases"]


def test_hyperbolic_cold_start_kat_is_bit_exact_in_the_restated_mode(L):
    """kepler/params.rs:689-725 asserts equality with values recorded on glibc; the restated mode reproduces all three
    branches bit for bit WITHOUT calling the platform's sin/cos/sinh/cosh."""
    with O.math_flags(libm=0, glibc=1):
        for c in kat_cases():
            kp = O.KeplerParams(dt=c["dt"], r0=c["r0"], sig0=c["sig0"], mu=c["mu"], alpha=c["alpha"], e0=c["e0"],
                                convergency=c["convergency"], has_guess=0, psi_guess=0.0, max_iter_prelim_kepuni=20,
                                kind=0)
            assert L.orc_prelim_kepuni(C.byref(kp)) == c["psi"], c


@pytest.mark.gpu
def test_device_functions_equal_the_restatement_bit_for_bit(L):
    import outfit_amd
    from outfit_amd import _lib
    H = _lib.lib()
    rng = np.random.default_rng(11)
    dp = C.POINTER(C.c_double)
    n = 400_000
    for fn, name, xs in ((9, "sin", np.concatenate([rng.uniform(-40, 40, n), rng.uniform(-1e5, 1e5, n), rng.uniform(-1e8, 1e8, n // 4),
                                                      rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-40, 2, n)])),
                         (10, "cos", None), (11, "sinh", np.concatenate([rng.uniform(-21.9, 21.9, n), rng.uniform(-1.2, 1.2, n),
                                                                           rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-27, 1, n)])),
                         (12, "cosh", None)):
        if xs is None:
            xs = prev
        prev = xs
        xs = np.ascontiguousarray(xs)
        out = np.empty_like(xs)
        assert H.outfit_hip_crmath_eval(fn, len(xs), xs.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0, _lib.last_error()
        g = getattr(L, "orc_glibc_" + name)
        g.restype = C.c_double
        g.argtypes = [C.c_double]
        ref = np.array([g(float(x)) for x in xs])
        assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), (name, int((out != ref).sum()))


@pytest.mark.gpu
def test_hyperbolic_cold_start_kat_through_the_abi_in_glibc_mode():
    """The device reproduces the reference's libm-dependent KAT bit for bit in OUTFIT_MATH_GLIBC, and to <= 4 ulp
    (hyperbolic branch only) in the correctly-rounded mode."""
    import math
    import outfit_amd
    from outfit_amd import _lib
    H = _lib.lib()
    cases = [c for c in kat_cases() if c["mu"] == 0.01720209895 ** 2]   # the device's mu is GAUSS_GRAV^2
    assert len(cases) == 3
    prob = np.array([[c["dt"], c["r0"], c["sig0"], c["alpha"], c["e0"], c["convergency"]] for c in cases])
    dp = C.POINTER(C.c_double)
    out = np.empty(len(cases))
    with outfit_amd.math_mode(outfit_amd.MATH_GLIBC):
        assert H.outfit_hip_prelim_kepuni(len(cases), prob.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    assert [float(x) for x in out] == [c["psi"] for c in cases]
    with outfit_amd.math_mode(outfit_amd.MATH_CORRECTLY_ROUNDED):
        assert H.outfit_hip_prelim_kepuni(len(cases), prob.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    for got, c in zip(out, cases):
        if c["alpha"] > 0.0:
            assert abs(got - c["psi"]) <= 4 * math.ulp(c["psi"])
        else:
            assert got == c["psi"]
