"""The device's arithmetic against the CPU oracle, bit for bit, WITHOUT a GPU.

tests/host_harness/device_math_host.cpp compiles the headers the kernels inline (outfit_amd/csrc/device/*.hpp) with g++:
the restated glibc functions, the correctly-rounded functions, the Stumpff series, the universal-Kepler cold starts and
Newton solver, the Lagrange velocity correction and the Aberth iteration in both of its arithmetic forms. The -m gpu
suite compares whole records through the C ABI on the device; this one holds the same code against the oracle function
by function on this machine (a host compiler may schedule differently, but every operation is IEEE and the translation
unit is built with contraction off, exactly as hipcc builds it: same bits)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "host_harness")
BUILD = os.path.join(ROOT, "oracle", "_build")
_dp = C.POINTER(C.c_double)
EPS = 2.220446049250313e-16


@pytest.fixture(scope="module")
def D():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libdevice_math_host.so")
    src = os.path.join(HARNESS, "device_math_host.cpp")
    deps = [src, os.path.join(HARNESS, "hip", "hip_runtime.h")] + \
        [os.path.join(ROOT, "outfit_amd", "csrc", "device", f) for f in os.listdir(os.path.join(ROOT, "outfit_amd", "csrc", "device"))]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-D_GNU_SOURCE", "-shared", "-fPIC", "-I" + HARNESS,
                        "-I" + os.path.join(ROOT, "outfit_amd", "csrc"), src, "-o", so], check=True)
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(_dp)


def same_bits(a, b):
    return a.view(np.uint64) == b.view(np.uint64)


def scalar(L, name, n_args=1):
    f = getattr(L, name)
    f.restype = C.c_double
    f.argtypes = [C.c_double] * n_args
    return f


def test_restated_glibc_functions_device_equals_oracle(D):
    """glibcmath{,2,3}.hpp == oracle/glibcmath{,2,3}.c on the ranges the path uses (and beyond): every bit."""
    L = O.lib()
    rng = np.random.default_rng(11)
    n = 60000
    ang = np.concatenate([rng.uniform(-40, 40, n // 2), rng.uniform(-7, 7, n // 2)])
    one = [("orc_glibc_sin1", 0, ang), ("orc_glibc_cos1", 1, ang), ("orc_glibc_sin", 2, ang), ("orc_glibc_cos", 3, ang),
           ("orc_glibc_tan", 4, rng.uniform(-1.6, 1.6, n)), ("orc_glibc_acos", 7, rng.uniform(-1, 1, n)),
           ("orc_glibc_log", 8, 10.0 ** rng.uniform(-3, 3, n)), ("orc_glibc_sinh", 9, rng.uniform(-15, 15, n)),
           ("orc_glibc_cosh", 10, rng.uniform(-15, 15, n))]
    for name, fn, x in one:
        x = np.ascontiguousarray(x)
        out = np.empty_like(x)
        D.glm_eval(C.c_int(fn), C.c_uint64(x.size), _p(x), _p(x), _p(out))
        f = scalar(L, name)
        ref = np.array([f(float(v)) for v in x])
        assert same_bits(out, ref).all(), (name, int((~same_bits(out, ref)).sum()))
    y, x = np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n)), np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n))
    for name, fn in (("orc_glibc_atan2", 5), ("orc_glibc_hypot", 6)):
        out = np.empty_like(x)
        D.glm_eval(C.c_int(fn), C.c_uint64(n), _p(y), _p(x), _p(out))
        f = scalar(L, name, 2)
        ref = np.array([f(float(a), float(b)) for a, b in zip(y, x)])
        assert same_bits(out, ref).all(), name


def test_branch_free_sincos_and_atan2_at_the_switching_points_of_glibcs_branches(D):
    """Round 5: glm::sincos and glm::atan2 select the operands of glibc's range / quadrant branches per lane instead of
    branching (glibcmath.hpp, glibcmath2.hpp). The oracle keeps glibc's literal branches: same bits at and around every
    switching point — the |x| limits of sin / cos (2^-26, 2^-27, 0.855469, 2.426265, 105414350), |y| = |x|, the
    u = 1/16 limit between polynomial and table in all four quadrants, zeros, infinities, NaN."""
    import math
    L = O.lib()

    def around(v, k=6):
        out = [v]
        lo = hi = v
        for _ in range(k):
            lo, hi = math.nextafter(lo, -math.inf), math.nextafter(hi, math.inf)
            out += [lo, hi]
        return out

    def from_hi_word(w):
        return np.array([np.uint64(w) << np.uint64(32)], dtype=np.uint64).view(np.float64)[0]

    xs = []
    for w in (0x3e400000, 0x3e500000, 0x3feb6000, 0x400368fd, 0x419921FB):
        for v in around(float(from_hi_word(w))):
            xs += [v, -v]
    xs += [0.0, -0.0, 0.126, -0.126, math.pi / 2, math.pi, 3 * math.pi / 2, 2 * math.pi, 1e5, -1e5, 1e8, -1e8, 5e-324]
    xs += list(np.random.default_rng(3).uniform(-7, 7, 20000))
    x = np.ascontiguousarray(xs)
    for name, fn in (("orc_glibc_sin", 2), ("orc_glibc_cos", 3)):
        out = np.empty_like(x)
        D.glm_eval(C.c_int(fn), C.c_uint64(x.size), _p(x), _p(x), _p(out))
        f = scalar(L, name)
        ref = np.array([f(float(v)) for v in x])
        assert same_bits(out, ref).all(), (name, x[~same_bits(out, ref)][:5])
    ys, xq = [], []
    rng = np.random.default_rng(4)
    for base in (1.0, 3.7e-5, 8.1e11):
        for r in around(0.0625, 8) + around(1.0, 8) + [0.03, 0.0624, 0.0626, 0.5, 0.999, 1.001, 2.0, 16.0, 15.999, 16.001, 1e3]:
            for sy in (1.0, -1.0):
                for sx in (1.0, -1.0):
                    ys.append(sy * base * r); xq.append(sx * base)        # |y| / |x| = r
                    ys.append(sy * base); xq.append(sx * base * r)        # |x| / |y| = r
    for a in (0.0, -0.0, math.inf, -math.inf, math.nan, 1e-310, 1e300, 1.0):
        for b in (0.0, -0.0, math.inf, -math.inf, math.nan, 1e-310, 1e300, -1.0):
            ys.append(a); xq.append(b)
    ys += list(rng.normal(size=20000) * 10.0 ** rng.uniform(-2, 2, 20000))
    xq += list(rng.normal(size=20000) * 10.0 ** rng.uniform(-2, 2, 20000))
    y, x = np.ascontiguousarray(ys), np.ascontiguousarray(xq)
    out = np.empty_like(x)
    D.glm_eval(C.c_int(5), C.c_uint64(x.size), _p(y), _p(x), _p(out))
    f = scalar(L, "orc_glibc_atan2", 2)
    ref = np.array([f(float(a), float(b)) for a, b in zip(y, x)])
    bad = ~(same_bits(out, ref) | (np.isnan(out) & np.isnan(ref)))
    assert not bad.any(), (y[bad][:5], x[bad][:5], out[bad][:5], ref[bad][:5])


def test_correctly_rounded_functions_device_equals_oracle(D):
    """crmath.hpp (double-double fast paths + full evaluations) == the oracle's correctly-rounded mode (its own
    double-double sin / cos / acos, binary128 for the rest)."""
    L = O.lib()
    rng = np.random.default_rng(12)
    n = 30000
    with O.math_flags(libm=0, glibc=0):
        for name, fn, x in (("orc_sin", 0, rng.uniform(-40, 40, n)), ("orc_cos", 1, rng.uniform(-40, 40, n)),
                            ("orc_acos", 2, rng.uniform(-1, 1, n)), ("orc_log", 3, 1.0 + 10.0 ** rng.uniform(-6, 3, n)),   # log: y > 1 is the call site's domain (hyperbolic cold start)
                            ("orc_sinh", 4, rng.uniform(-15, 15, n)), ("orc_cosh", 5, rng.uniform(-15, 15, n))):
            x = np.ascontiguousarray(x)
            out = np.empty_like(x)
            D.crm_eval(C.c_int(fn), C.c_uint64(x.size), _p(x), _p(x), _p(out))
            f = scalar(L, name)
            ref = np.array([f(float(v)) for v in x])
            assert same_bits(out, ref).all(), (name, int((~same_bits(out, ref)).sum()))


def _kepler_problems(rng, n):
    a = np.where(rng.uniform(size=n) < 0.85, 10.0 ** rng.uniform(-0.3, 1.7, n), -10.0 ** rng.uniform(-0.3, 1.5, n))   # a < 0: hyperbolic
    alpha = -1.0 / a
    e0 = np.where(a > 0, rng.uniform(0, 0.98, n), 1.0 + 10.0 ** rng.uniform(-3, 0.5, n))
    r0 = np.abs(a) * np.abs(1.0 - e0 * rng.uniform(-1, 1, n)) + 1e-3
    sig0 = rng.normal(size=n) * 0.7
    dt = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-2, 1.6, n)
    return np.ascontiguousarray(np.stack([dt, r0, sig0, alpha, e0, np.full(n, 1e3 * EPS)], axis=1))


@pytest.mark.parametrize("mode", [0, 1], ids=["correctly_rounded", "glibc"])
def test_stumpff_cold_starts_and_newton_device_equals_oracle(D, mode):
    """s_funct (series with the exact constant divisions, halving + duplication), prelim_kepuni (elliptic / hyperbolic
    cold starts with the mode's functions) and the Newton solve from it (quirks incl.): device headers == oracle."""
    L = O.lib()
    rng = np.random.default_rng(13 + mode)
    n = 20000
    psi = np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.uniform(-3, 1.5, n))
    alpha = np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.uniform(-2, 1, n))
    alpha[:50] = 0.0
    out = np.empty(4 * n)
    D.s_funct_array(C.c_uint64(n), _p(psi), _p(alpha), _p(out))
    s4 = O.D4()
    ref = np.empty(4 * n)
    for i in range(n):
        L.orc_s_funct(float(psi[i]), float(alpha[i]), s4)
        ref[4 * i:4 * i + 4] = list(s4)
    assert same_bits(out, ref).all()
    m = 6000
    prob = _kepler_problems(rng, m)
    prelim, sol = np.empty(m), np.empty(6 * m)
    D.kepler_array(C.c_uint64(m), _p(prob), C.c_int(mode), _p(prelim), _p(sol))
    sol = sol.reshape(m, 6)
    with O.math_flags(libm=0, glibc=mode):
        for i in range(m):
            kp = O.KeplerParams(dt=prob[i, 0], r0=prob[i, 1], sig0=prob[i, 2], mu=0.01720209895 ** 2, alpha=prob[i, 3],
                                e0=prob[i, 4], convergency=prob[i, 5], has_guess=0, psi_guess=0.0,
                                max_iter_prelim_kepuni=20, kind=0, parabolic_method=0)
            want = L.orc_prelim_kepuni(C.byref(kp))
            assert np.float64(want).view(np.uint64) == prelim[i:i + 1].view(np.uint64)[0], (i, prob[i], want, prelim[i])
            ks = O.KeplerSolution()
            rc = L.orc_kepler_solve(C.byref(kp), C.byref(ks))
            assert (rc == 1) == (sol[i, 0] == 1.0), (i, prob[i])      # 1 = Some(solution)
            if rc == 1:
                got = sol[i, 1:]
                exp = np.array([ks.psi, ks.s0, ks.s1, ks.s2, ks.s3])
                assert same_bits(np.ascontiguousarray(got), exp).all(), (i, prob[i], got, exp)


@pytest.mark.parametrize("mode", [0, 1], ids=["correctly_rounded", "glibc"])
def test_shared_cold_start_of_the_two_solves_equals_two_cold_starts(D, mode):
    """K2a's fused cold start evaluates the dt-independent part of prelim_kepuni once for the iteration's two solves
    (kepler.hpp prelim_kepuni_pair): same bits as the two separate calls, which the test above holds against the oracle."""
    rng = np.random.default_rng(113 + mode)
    m = 20000
    prob = _kepler_problems(rng, m)
    prob[:40, 3] = 0.0                       # parabolic branch
    prob[40:80, 4] = 1e-20                   # e0 below the convergence threshold
    out = np.empty(4 * m)
    D.kepler_pair_array(C.c_uint64(m), _p(prob), C.c_int(mode), _p(out))
    out = out.reshape(m, 4)
    assert same_bits(np.ascontiguousarray(out[:, 0]), np.ascontiguousarray(out[:, 2])).all()
    assert same_bits(np.ascontiguousarray(out[:, 1]), np.ascontiguousarray(out[:, 3])).all()
    assert (prob[:, 3] > 0).sum() > 100 and (prob[:, 3] < 0).sum() > 100


def test_velocity_correction_device_equals_oracle(D):
    L = O.lib()
    rng = np.random.default_rng(15)
    n = 6000
    x2 = rng.normal(size=(n, 3)) * 2.0
    v2 = rng.normal(size=(n, 3)) * 0.012
    dt = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-2, 1.4, n)
    x1 = x2 + v2 * dt[:, None] + rng.normal(size=(n, 3)) * 1e-4
    has = (rng.uniform(size=n) < 0.6).astype(float)
    chi = rng.normal(size=n) * 0.3
    inp = np.ascontiguousarray(np.concatenate([x1, x2, v2, dt[:, None], has[:, None], chi[:, None]], axis=1))
    out = np.empty(7 * n)
    D.velocity_correction_array(C.c_uint64(n), _p(inp), C.c_double(1e3), C.c_double(5.0), C.c_double(1e3 * EPS), C.c_int(1), _p(out))
    out = out.reshape(n, 7)
    ok_count = 0
    with O.math_flags(libm=0, glibc=1):
        for i in range(n):
            v = O.D3()
            f, g, c = C.c_double(), C.c_double(), C.c_double()
            rc = L.orc_velocity_correction_with_guess(O.D3(*x1[i]), O.D3(*x2[i]), O.D3(*v2[i]), float(dt[i]), 1e3, 5.0,
                                                      int(has[i]), float(chi[i]), 1e3 * EPS, v, C.byref(f), C.byref(g), C.byref(c))
            assert (rc == 0) == (out[i, 0] == 1.0), i
            if rc == 0:
                ok_count += 1
                exp = np.array([v[0], v[1], v[2], f.value, g.value, c.value])
                assert same_bits(np.ascontiguousarray(out[i, 1:]), exp).all(), (i, out[i, 1:], exp)
    assert ok_count > 0.5 * n


def test_aberth_forms_device_equals_oracle(D):
    """aberth_deg8<false> (plain divisions) and <true> (shared reciprocals; restart on the guard) == the oracle's
    restatement of the crate: return code, all eight iterates, their order — incl. the families whose roots reach
    exactly zero imaginary parts, even polynomials and z^8 + c0."""
    L = O.lib()
    rng = np.random.default_rng(16)
    n = 4000
    c = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 2, (n, 3))
    c[: n // 2, 0] = -np.abs(c[: n // 2, 0])      # the physical signs: c0 < 0, c6 < 0
    c[: n // 2, 2] = -np.abs(c[: n // 2, 2])
    c[n // 2: n // 2 + 200, 1] = 0.0
    c[n // 2 + 200: n // 2 + 300, 2] = 0.0
    c[n // 2 + 300: n // 2 + 350, 1:] = 0.0
    coef = np.ascontiguousarray(c)
    got = {}
    for variant in (0, 1):
        code = np.empty(n, dtype=np.int32)
        roots = np.empty(16 * n)
        D.aberth_array(C.c_uint64(n), _p(coef), C.c_uint(50), C.c_double(1e-6), C.c_int(variant),
                       code.ctypes.data_as(C.POINTER(C.c_int)), _p(roots))
        got[variant] = (code, roots.reshape(n, 16))
    assert np.array_equal(got[0][0], got[1][0])
    live = got[0][0] != 2
    assert same_bits(np.ascontiguousarray(got[0][1][live]), np.ascontiguousarray(got[1][1][live])).all()
    re, im = O.D8(), O.D8()
    it = C.c_uint32()
    for i in range(n):
        rc = L.orc_aberth9(O.D9(coef[i, 0], 0, 0, coef[i, 1], 0, 0, coef[i, 2], 0, 1.0), 50, 1e-6, re, im, C.byref(it))
        assert rc == got[0][0][i], (i, coef[i])
        if rc != 2:
            exp = np.array([v for k in range(8) for v in (re[k], im[k])])
            assert same_bits(np.ascontiguousarray(got[0][1][i]), exp).all(), (i, coef[i])


def test_rms_chain_device_equals_oracle_in_the_glibc_mode(D):
    """State -> ccek1 -> to_equinoctial -> ephemeris_error per observation with the restated glibc functions (orbit.hpp
    kFnGlibc: what K3b evaluates for the contenders) == the oracle in its restated-glibc mode: elements, equinoctial
    elements, every residual term — the bits of the `rms` field's ingredients."""
    import outfit_amd  # noqa: F401
    from outfit_amd import synth
    L = O.lib()
    ds = synth.make_dataset(300, seed=41)
    n, n_obs = 300, 20
    truth = ds["truth"]
    rng = np.random.default_rng(17)
    # states: the truth orbit's position / velocity at the middle epoch, perturbed (what a Gauss candidate looks like)
    state = np.empty((n, 7))
    for i in range(n):
        a, e, inc, node, argp, m0 = truth[i]
        el = O.Elements(kind=0, epoch=60000.0, e=(C.c_double * 6)(a, e, inc, node, argp, m0))
        eq = O.Equinoctial()
        with O.math_flags(libm=0, glibc=1):
            assert L.orc_to_equinoctial(C.byref(el), C.byref(eq)) == 0
            r, v = O.D3(), O.D3()
            t_mid = float(ds["mjd_tt"][n_obs * i + n_obs // 2])
            assert L.orc_propagate_twobody(C.byref(eq), 60000.0, t_mid, r, v) == 0
        # ecliptic -> equatorial (orbit.hpp ecl_to_equ), small perturbation
        c, s_ = 9.174820620691818e-1, 3.977771559319137e-1
        rq = np.array([r[0], c * r[1] - s_ * r[2], s_ * r[1] + c * r[2]]) * (1 + 1e-4 * rng.normal(size=3))
        vq = np.array([v[0], c * v[1] - s_ * v[2], s_ * v[1] + c * v[2]]) * (1 + 1e-4 * rng.normal(size=3))
        state[i] = [*rq, *vq, t_mid]
    obs = np.empty((n, n_obs, 8))
    for i in range(n):
        sl = slice(n_obs * i, n_obs * (i + 1))
        obs[i, :, 0] = ds["mjd_tt"][sl]; obs[i, :, 1] = ds["ra"][sl]; obs[i, :, 2] = ds["dec"][sl]
        obs[i, :, 3] = ds["sigma_ra"][sl]; obs[i, :, 4] = ds["sigma_dec"][sl]
        obs[i, :, 5:] = ds["obs_helio_equ"].reshape(-1, 3)[sl]
    state, obs = np.ascontiguousarray(state), np.ascontiguousarray(obs)
    stride = 14 + 2 * n_obs
    out = np.empty(n * stride)
    D.rms_chain_glibc(C.c_uint64(n), C.c_uint(n_obs), _p(state), _p(obs), _p(out))
    out = out.reshape(n, stride)
    checked = 0
    with O.math_flags(libm=0, glibc=1):
        for i in range(n):
            c, s_ = 9.174820620691818e-1, 3.977771559319137e-1
            def to_ecl(x):   # orbit.hpp equ_to_ecl: the reference's rotation matrix, row by row
                return O.D3((1.0 * x[0] + 0.0 * x[1]) + 0.0 * x[2], (0.0 * x[0] + c * x[1]) + s_ * x[2], (0.0 * x[0] + (-s_) * x[1]) + c * x[2])
            el = O.Elements()
            L.orc_ccek1(to_ecl(state[i, :3]), to_ecl(state[i, 3:6]), float(state[i, 6]), C.byref(el))
            assert el.kind == int(out[i, 0])
            assert same_bits(np.ascontiguousarray(out[i, 2:8]), np.array(list(el.e))).all(), (i, out[i, 2:8], list(el.e))
            eq = O.Equinoctial()
            conv = L.orc_to_equinoctial(C.byref(el), C.byref(eq))
            assert conv == int(out[i, 1])
            if conv != 0:
                continue
            assert same_bits(np.ascontiguousarray(out[i, 8:14]), np.array([eq.a, eq.h, eq.k, eq.p, eq.q, eq.lam])).all(), i
            for j in range(n_obs):
                term = C.c_double()
                rc = L.orc_ephemeris_error(C.byref(eq), *[float(x) for x in obs[i, j, :5]], O.D3(*obs[i, j, 5:]), C.byref(term))
                assert (rc == 0) == (out[i, 14 + 2 * j] == 1.0), (i, j)
                if rc == 0:
                    assert np.float64(term.value).view(np.uint64) == out[i, 15 + 2 * j:16 + 2 * j].view(np.uint64)[0], (i, j, term.value, out[i, 15 + 2 * j])
                    checked += 1
    assert checked > 0.9 * n * n_obs


def test_corrector_partials_device_equals_oracle_in_the_glibc_mode(D):
    """compute_obs_and_partials_2body (observation_ephemeris.rs:418-450 -> equinoctial_element.rs:442-582): computed (ra, dec)
    and their twelve element partials, device header (diffcor.hpp, kFnGlibc) == oracle in its restated-glibc mode; and
    angular_diff on wrapped and unwrapped pairs."""
    import outfit_amd  # noqa: F401
    from outfit_amd import synth
    L = O.lib()
    ds = synth.make_dataset(200, seed=43)
    n_obs = 20
    rng = np.random.default_rng(19)
    eqs, obs = [], []
    with O.math_flags(libm=0, glibc=1):
        for i in range(200):
            a, e, inc, node, argp, m0 = ds["truth"][i]
            el = O.Elements(kind=0, epoch=60000.0, e=(C.c_double * 6)(a, e, inc, node, argp, m0))
            eq = O.Equinoctial()
            assert L.orc_to_equinoctial(C.byref(el), C.byref(eq)) == 0
            for j in range(n_obs):
                o = n_obs * i + j
                pert = 1e-5 * rng.normal(size=6)
                eqs.append([60000.0, eq.a * (1 + pert[0]), eq.h + pert[1], eq.k + pert[2], eq.p + pert[3], eq.q + pert[4], eq.lam + pert[5]])
                obs.append([ds["mjd_tt"][o], *ds["obs_helio_equ"][3 * o:3 * o + 3]])
    eqs, obs = np.ascontiguousarray(eqs), np.ascontiguousarray(obs)
    n = len(eqs)
    out = np.empty(15 * n)
    D.obs_partials_glibc(C.c_uint64(n), _p(eqs), _p(obs), _p(out))
    out = out.reshape(n, 15)
    checked = 0
    with O.math_flags(libm=0, glibc=1):
        for i in range(n):
            eq = O.Equinoctial(*eqs[i])
            ra, dec = C.c_double(), C.c_double()
            ga, gd = O.D6(), O.D6()
            rc = L.orc_obs_and_partials_2body(C.byref(eq), float(obs[i, 0]), O.D3(*obs[i, 1:]), C.byref(ra), C.byref(dec), ga, gd)
            assert (rc == 0) == (out[i, 0] == 1.0), i
            if rc == 0:
                exp = np.array([ra.value, dec.value, *ga, *gd])
                assert same_bits(np.ascontiguousarray(out[i, 1:]), exp).all(), (i, out[i, 1:], exp)
                checked += 1
    assert checked > 0.95 * n
    D.angular_diff_host.restype = C.c_double
    D.angular_diff_host.argtypes = [C.c_double, C.c_double]
    for a, b in [(0.1, 6.2), (6.2, 0.1), (3.0, -3.0), (1.0, 1.0 + 1e-9), (100.0, -3.0), (-50.0, 7.0)] + \
            [tuple(rng.uniform(-20, 20, 2)) for _ in range(2000)]:
        assert np.float64(D.angular_diff_host(a, b)).view(np.uint64) == np.float64(L.orc_angular_diff(a, b)).view(np.uint64), (a, b)


@pytest.mark.parametrize("mode", [0, 1], ids=["correctly_rounded", "glibc"])
def test_correction_loop_with_its_early_exits_equals_the_literal_loop(D, mode):
    """gauss.rs:1284-1418 through the DEVICE's code — accept_root, then correction_iteration until it ends, with the exits
    the device adds (an iteration that ends in `continue` and leaves (pos, vel, chi01, chi21) bit for bit unchanged is a
    fixed point: the loop's result is known; eccentricity_control carried from one iteration to the next) — against the
    oracle's literal loop over all newton_max_it iterations: same Some / None, same position, velocity and epoch bits,
    for every root of ~600 triplets of the bench population, several parameter sets."""
    import outfit_amd  # noqa: F401
    from outfit_amd import synth
    from outfit_amd.api import IODParams
    L = O.lib()
    L.orc_accept_root.argtypes = [C.POINTER(O.GaussObs), C.POINTER(O.IODParamsC), C.c_double, C.POINTER(O.GaussPrelim),
                                  O.M33, O.D3, _dp]
    ds = synth.make_dataset(150, seed=47)
    rng = np.random.default_rng(21)
    total = some = none_ = early = guarded_cases = 0
    for kw in (dict(), dict(max_ecc=0.35, max_perihelion_au=3.5), dict(newton_max_it=7), dict(min_rho2_au=0.5)):
        ip = IODParams(n_noise_realizations=0, **kw).as_c()
        with O.math_flags(libm=0, glibc=mode):
            cases, refs = [], []
            for i in range(150):
                idx = sorted(rng.choice(20, 3, replace=False).tolist())
                o = [20 * i + k for k in idx]
                g = O.gauss_obs(idx, [ds["ra"][k] for k in o], [ds["dec"][k] for k in o], [ds["mjd_tt"][k] for k in o],
                                [ds["obs_helio_equ"][3 * k:3 * k + 3] for k in o])
                pre = O.GaussPrelim()
                if L.orc_gauss_prelim_fn(C.byref(g), C.byref(pre)) != 0:
                    continue
                c6, c3, c0 = C.c_double(), C.c_double(), C.c_double()
                L.orc_coeff_eight_poly(C.byref(g), C.byref(pre), C.byref(c6), C.byref(c3), C.byref(c0))
                roots, nr = O.D8(), C.c_int()
                if L.orc_solve_8poly(O.D9(c0.value, 0, 0, c3.value, 0, 0, c6.value, 0, 1.0), 50, 1e-6, 1e-6, roots, C.byref(nr)) != 0:
                    continue
                for r in range(nr.value):
                    root = roots[r]
                    if not (ip.r2_min_au <= root <= ip.r2_max_au):
                        continue
                    pos, vel, ep = O.M33(), O.D3(), C.c_double()
                    acc = L.orc_accept_root(C.byref(g), C.byref(ip), float(root), C.byref(pre), pos, vel, C.byref(ep))
                    ref = {"accepted": acc}
                    if acc:
                        pos2, vel2, ep2 = O.M33(), O.D3(), C.c_double()
                        rc = L.orc_pos_and_vel_correction(C.byref(g), C.byref(ip), pos, vel, pre.S, pre.Sinv, ip.max_perihelion_au,
                                                          ip.max_ecc, ip.newton_eps, ip.newton_max_it, pos2, vel2, C.byref(ep2))
                        ref.update(outcome=rc, state=[pos2[0][1], pos2[1][1], pos2[2][1], vel2[0], vel2[1], vel2[2], ep2.value])
                    refs.append(ref)
                    cases.append([*[g.obs[a][b] for a in range(3) for b in range(3)], *[pre.S[a][b] for a in range(3) for b in range(3)],
                                  g.t[0], g.t[1], g.t[2], root])
        inp = np.ascontiguousarray(cases)
        out = np.empty(10 * len(cases))
        D.correction_loop_array(C.c_uint64(len(cases)), _p(inp), C.c_double(ip.max_ecc), C.c_double(ip.max_perihelion_au),
                                C.c_double(ip.min_rho2_au), C.c_double(ip.kepler_eps), C.c_double(ip.newton_eps),
                                C.c_uint(ip.newton_max_it), C.c_int(mode), C.c_int(0), _p(out))
        out = out.reshape(-1, 10)
        # the fast kernel's form of the iteration: short divisions under the range guard — the same bytes, no guard hit
        fast = np.empty(10 * len(cases))
        D.correction_loop_array(C.c_uint64(len(cases)), _p(inp), C.c_double(ip.max_ecc), C.c_double(ip.max_perihelion_au),
                                C.c_double(ip.min_rho2_au), C.c_double(ip.kepler_eps), C.c_double(ip.newton_eps),
                                C.c_uint(ip.newton_max_it), C.c_int(mode), C.c_int(1), _p(fast))
        fast = fast.reshape(-1, 10)
        assert (fast[:, 9] >= 0).all(), "a guarded division reported an operand outside its window on ordinary data"
        assert same_bits(np.ascontiguousarray(fast), np.ascontiguousarray(out)).all()
        guarded_cases += len(cases)
        for k, ref in enumerate(refs):
            assert (ref["accepted"] != 0) == (out[k, 0] == 1.0), (kw, k)
            if not ref["accepted"]:
                continue
            total += 1
            assert ref["outcome"] == int(out[k, 1]), (kw, k, ref, out[k])
            if ref["outcome"] == 1:
                some += 1
                assert same_bits(np.ascontiguousarray(out[k, 2:9]), np.array(ref["state"])).all(), (kw, k, out[k, 2:9], ref["state"])
                early += out[k, 9] < ip.newton_max_it
            else:
                none_ += 1
    assert total > 600 and some > 500 and none_ > 5 and early > 200 and guarded_cases > 600, (total, some, none_, early)


def test_guarded_divisions_equal_plain_divisions_inside_the_guard(D):
    """fastdiv.hpp: the short forms of a division by a constant (Markstein's correction step from the correctly-rounded
    reciprocal) and of several divisions by one denominator (one refined reciprocal), as the fast correction kernel runs
    them under fdiv::Guard — whenever the guard passes, every quotient is the bits of `/`; operands that are zero, tiny,
    huge or not finite never pass. (Here with an exact reciprocal seed; the device's own seed: outfit_hip_selftest_division.)"""
    rng = np.random.default_rng(2026)
    n = 400000
    ordinary = rng.normal(size=(n, 5)) * 10.0 ** rng.uniform(-6, 6, (n, 5))
    wide = rng.normal(size=(n // 4, 5)) * 10.0 ** rng.uniform(-200, 200, (n // 4, 5))
    special = ordinary[:4000].copy()
    for k, v in enumerate((0.0, -0.0, np.inf, -np.inf, np.nan, 1e-310, 1e-95, 1e95)):
        special[k::8, k % 5] = v
    cases = np.ascontiguousarray(np.concatenate([ordinary, wide, special]))
    out = np.empty(19 * len(cases))
    D.guarded_division_array(C.c_uint64(len(cases)), _p(cases), _p(out))
    out = out.reshape(-1, 19)
    passed = out[:, 0] == 1.0
    assert passed[:n].all()                                             # ordinary operands: the guard never fires
    bad = ~np.isfinite(cases).all(1) | (cases == 0).any(1) | (np.abs(cases) < 4.9e-91).any(1) | (np.abs(cases) > 2.04e90).any(1)
    assert not (passed & bad).any()                                     # an operand outside the window never passes
    assert (~passed[n:]).sum() > 1000 and passed[n:].sum() > 1000
    fast, plain = np.ascontiguousarray(out[passed, 1:10]), np.ascontiguousarray(out[passed, 10:19])
    assert same_bits(fast, plain).all()


def test_decisions_on_operands_equal_the_literal_expressions(D):
    """Round 5 takes four decisions of the correction loop on the operands of a square root or of a quotient instead of on
    its value (kepler.hpp kHnormSqMax / kFrobSqMax / peri_below, gauss.hpp rel_converged). They must be the literal
    decisions for EVERY input: checked here at and around the switching points, where a sloppy bound would show, on
    random operands far from them, and with the pre-tests switched off (NaN bounds)."""
    import math
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sqrt_thresholds
    c = (C.c_double * 2)()
    D.decision_constants(c)
    assert c[0] == sqrt_thresholds.CONSTANTS["kHnormSqMax"] and c[1] == sqrt_thresholds.CONSTANTS["kFrobSqMax"]
    for t, lim in ((c[0], 1e6 * EPS), (c[1], EPS)):
        x = t
        for _ in range(64):                                   # 64 doubles on either side of T
            assert math.sqrt(x) <= lim
            x = math.nextafter(x, 0.0)
        x = math.nextafter(t, math.inf)
        for _ in range(64):
            assert math.sqrt(x) > lim
            x = math.nextafter(x, math.inf)
    rng = np.random.default_rng(5)

    def ulps(x, k):
        return (x.view(np.int64) + k).view(np.float64)

    # ---- perihelion test: RN(h2 / den) < peri_max
    out = (C.c_double * 4)()
    cases = []
    for pm, emax in ((1e3, 5.0), (0.7, 5.0), (3.3e-7, 2.0), (9.1e40, 1e100)):
        D.derive_params(C.c_double(pm), C.c_double(emax), C.c_double(1e-10), out)
        lo, hi = out[0], out[1]
        assert lo < pm < hi and abs(lo / pm - 1) < 1e-14 and abs(hi / pm - 1) < 1e-14
        den = 2.959122082855911e-4 * (1.0 + rng.uniform(0, emax if emax < 10 else 5, 40000))
        # h2 so that the quotient lands within a few ulps of peri_max (the sliver), at the bounds, and anywhere
        near = ulps(den * pm, rng.integers(-40, 41, den.size))
        at_lo = ulps(den * lo, rng.integers(-3, 4, den.size))
        at_hi = ulps(den * hi, rng.integers(-3, 4, den.size))
        far = den * pm * 10.0 ** rng.uniform(-8, 8, den.size)
        for h2 in (near, at_lo, at_hi, far):
            cases.append(np.stack([h2, den, np.full_like(den, pm), np.full_like(den, lo), np.full_like(den, hi)], axis=1))
    # degenerate operands: zero, denormal, huge, infinite, NaN angular momentum; the denominator mu (1 + e) of any
    # FINITE e (an infinite or NaN eccentricity fails `e < max_ecc` before the perihelion test counts: max_ecc <= 1e100
    # whenever the bounds are not NaN) and a NaN one
    D.derive_params(C.c_double(1e3), C.c_double(5.0), C.c_double(1e-10), out)
    odd = [(h2, den, 1e3, out[0], out[1]) for h2 in (0.0, 5e-324, 1e-310, 1e300, math.inf, math.nan)
           for den in (2.96e-4, 1e-3, 2.96e96, math.nan)]
    cases.append(np.array(odd))
    allc = np.ascontiguousarray(np.concatenate(cases))
    a, b = np.zeros(len(allc), np.int32), np.zeros(len(allc), np.int32)
    D.peri_below_array(C.c_ulonglong(len(allc)), _p(allc), a.ctypes.data_as(C.POINTER(C.c_int)), b.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(a, b)
    lit = (allc[:, 0] / allc[:, 1]) < allc[:, 2]
    assert np.array_equal(b.astype(bool), lit) and 0.2 < lit.mean() < 0.8
    # pre-test off outside the analysed parameter range
    for pm, emax in ((1e-200, 5.0), (1e200, 5.0), (1e3, math.inf), (1e3, 1e200)):
        D.derive_params(C.c_double(pm), C.c_double(emax), C.c_double(1e-10), out)
        assert math.isnan(out[0]) and math.isnan(out[1])

    # ---- convergence test: sqrt(d) / sqrt(n) <= newton_eps
    cases = []
    for eps in (1e-10, 1e-3, 1.0, 7.3e-17, 1e-30):
        D.derive_params(C.c_double(1e3), C.c_double(5.0), C.c_double(eps), out)
        lo, hi = out[2], out[3]
        assert lo < eps * eps < hi
        n = 10.0 ** rng.uniform(-28, 12, 40000)               # above kFrobSqMax (the caller's guard), incl. huge
        near = ulps(n * (eps * eps), rng.integers(-60, 61, n.size))
        at_lo = ulps(n * lo, rng.integers(-3, 4, n.size))
        at_hi = ulps(n * hi, rng.integers(-3, 4, n.size))
        far = n * eps * eps * 10.0 ** rng.uniform(-6, 6, n.size)
        for d in (near, at_lo, at_hi, far, np.zeros_like(n), np.full_like(n, math.nan), np.full_like(n, math.inf)):
            cases.append(np.stack([d, n, np.full_like(n, eps), np.full_like(n, lo), np.full_like(n, hi)], axis=1))
    allc = np.ascontiguousarray(np.concatenate(cases))
    a, b = np.zeros(len(allc), np.int32), np.zeros(len(allc), np.int32)
    D.rel_converged_array(C.c_ulonglong(len(allc)), _p(allc), a.ctypes.data_as(C.POINTER(C.c_int)), b.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(a, b)
    with np.errstate(invalid="ignore"):
        lit = np.sqrt(allc[:, 0]) / np.sqrt(allc[:, 1]) <= allc[:, 2]
    assert np.array_equal(b.astype(bool), lit)
    for eps in (1e-31, 1.5, math.inf):
        D.derive_params(C.c_double(1e3), C.c_double(5.0), C.c_double(eps), out)
        assert math.isnan(out[2]) and math.isnan(out[3])

    # ---- f64::clamp(x, -m, m) by magnitude
    x = np.concatenate([rng.normal(0, 10, 5000), [0.0, -0.0, math.inf, -math.inf, math.nan, 3.0, -3.0, 2.9999999999999996]])
    m = np.concatenate([rng.uniform(0.1, 5, 5000), [3.0] * 8])
    got = np.zeros_like(x)
    D.clamp_symmetric_array(C.c_ulonglong(x.size), _p(x), _p(m), _p(got))
    want = np.where(x < -m, -m, np.where(x > m, m, x))        # f64::clamp: NaN passes through
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))

    # ---- the one-word record of the guarded divisions never passes an operand outside fdiv::in_window
    x = np.concatenate([10.0 ** rng.uniform(-320, 308, 20000) * rng.choice([-1.0, 1.0], 20000),
                        2.0 ** np.arange(-310, 311, dtype=np.float64), -(2.0 ** np.arange(-310, 311, dtype=np.float64)),
                        ulps(np.array([2.0 ** -300, 2.0 ** 300]), -1), ulps(np.array([2.0 ** -300, 2.0 ** 300]), 1),
                        [0.0, -0.0, math.inf, -math.inf, math.nan, 5e-324, 2.2250738585072014e-308]])
    x = np.ascontiguousarray(x)
    ok, inw = np.zeros(x.size, np.int32), np.zeros(x.size, np.int32)
    D.guard_array(C.c_ulonglong(x.size), _p(x), ok.ctypes.data_as(C.POINTER(C.c_int)), inw.ctypes.data_as(C.POINTER(C.c_int)))
    assert not (ok.astype(bool) & ~inw.astype(bool)).any()    # passes  =>  inside the window
    with np.errstate(invalid="ignore"):
        inside = (np.abs(x) >= 2.0 ** -300) & (np.abs(x) < 2.0 ** 300)
    assert np.array_equal(ok.astype(bool), inside)             # and it passes everything in [2^-300, 2^300)


def test_screen_direct_route_agrees_with_the_route_through_the_angles(D):
    """K2c's main pass takes the equinoctial elements of a candidate straight from its state (device/screen.hpp
    screen_prologue) where the reference's route through the Keplerian angles decides nothing near a threshold, and starts
    the loop warm from the epoch. Held here against that route in the same screening arithmetic (what the slow-list pass
    and the redo list still take): same classification wherever the direct route classifies at all, elements and mean
    motion to 1e-13, the start (F, sin F, cos F) a solution of Kepler's equation, every observation's residual the same to
    1e-7 of the normalised unit (7e-9 is what 1e-14 rad on the sky gives at 0.3") + 1e-9 of its size."""
    from outfit_amd import synth
    n_obs = 20
    rng = np.random.default_rng(23)
    ds = synth.make_dataset(400, seed=43)
    truth = ds["truth"].copy()
    pops = {"bench": truth.copy()}
    hi = truth.copy(); hi[:, 1] = rng.uniform(0.6, 0.999, len(hi))
    hi[::7, 1] = 1.0 - 10.0 ** rng.uniform(-7, -3.2, len(hi[::7]))   # e^2 >= 0.998: the direct route declines
    pops["eccentric"] = hi
    lo = truth.copy(); lo[:, 1] = 10.0 ** rng.uniform(-9, -3, len(lo)); pops["near circular"] = lo
    eqt = truth.copy(); eqt[:, 2] = 10.0 ** rng.uniform(-12, -4, len(eqt)); pops["near equatorial"] = eqt
    ret = truth.copy(); ret[:, 2] = np.pi - 10.0 ** rng.uniform(-6, -0.5, len(ret)); pops["retrograde"] = ret
    pops["across ra = 0"] = truth.copy()   # the whole geometry turned about the pole: observed ra straddles 0 / 2 pi (quirk Q6)
    pops["irregular angles"] = truth.copy()  # ra given outside [0, 2 pi): the generic residual on both sides
    L = O.lib()
    total_elliptic = 0
    for name, pop in pops.items():
        n = len(pop)
        state = np.empty((n, 7))
        for i in range(n):
            a, e, inc, node, argp, m0 = pop[i]
            el = O.Elements(kind=0, epoch=60000.0, e=(C.c_double * 6)(a, e, inc, node, argp, m0))
            eq = O.Equinoctial()
            assert L.orc_to_equinoctial(C.byref(el), C.byref(eq)) == 0
            r, v = O.D3(), O.D3()
            t_mid = float(ds["mjd_tt"][n_obs * i + n_obs // 2])
            assert L.orc_propagate_twobody(C.byref(eq), 60000.0, t_mid, r, v) == 0
            c, s_ = 9.174820620691818e-1, 3.977771559319137e-1
            rq = np.array([r[0], c * r[1] - s_ * r[2], s_ * r[1] + c * r[2]]) * (1 + 1e-4 * rng.normal(size=3))
            vq = np.array([v[0], c * v[1] - s_ * v[2], s_ * v[1] + c * v[2]]) * (1 + 1e-4 * rng.normal(size=3))
            if name == "eccentric" and i % 3 == 0 and i % 7 != 0:
                vq *= rng.uniform(1.0, 1.6)     # towards and beyond the parabola: divert / hyperbolic
            state[i] = [*rq, *vq, t_mid]
        obs = np.empty((n, n_obs, 8))
        for i in range(n):
            sl = slice(n_obs * i, n_obs * (i + 1))
            obs[i, :, 0] = ds["mjd_tt"][sl]; obs[i, :, 1] = ds["ra"][sl]; obs[i, :, 2] = ds["dec"][sl]
            obs[i, :, 3] = ds["sigma_ra"][sl]; obs[i, :, 4] = ds["sigma_dec"][sl]
            obs[i, :, 5:] = ds["obs_helio_equ"].reshape(-1, 3)[sl]
        if name == "across ra = 0":
            for i in range(n):
                phi = obs[i, n_obs // 2, 1] - (1e-7 if i % 2 else -1e-7) * (i % 5)   # middle observation lands at ra ~ 0 +- k 1e-7
                c_, s__ = np.cos(phi), np.sin(phi)
                rot = np.array([[c_, s__, 0.0], [-s__, c_, 0.0], [0.0, 0.0, 1.0]])
                state[i, :3] = rot @ state[i, :3]
                state[i, 3:6] = rot @ state[i, 3:6]
                obs[i, :, 5:] = obs[i, :, 5:] @ rot.T
                obs[i, :, 1] = np.mod(obs[i, :, 1] - phi, 2 * np.pi)
        if name == "irregular angles":
            obs[::2, :, 1] -= 2 * np.pi
            obs[1::2, :, 1] += 2 * np.pi
        state, obs = np.ascontiguousarray(state), np.ascontiguousarray(obs)
        stride = 19 + 4 * n_obs
        out = np.empty(n * stride)
        D.screen_routes(C.c_uint64(n), C.c_uint(n_obs), _p(state), _p(obs), _p(out))
        out = out.reshape(n, stride)
        cls, kind, conv, hyper = (out[:, k].astype(int) for k in range(4))
        ell, hyp = cls == 0, cls == 1
        assert (kind[ell] == 0).all() and (conv[ell] == 0).all() and (hyper[ell] == 0).all(), name
        assert (kind[hyp] == 1).all() and (conv[hyp] == 0).all() and (hyper[hyp] == 1).all(), name
        if name == "across ra = 0":   # the reference's wrap leaves ra_obs - alpha below -pi alone: residuals of ~2 pi / sigma
            big = np.sqrt(out[:, 19:].reshape(n, n_obs, 4)[:, :, 3]).max(1) > 1e6
            assert big.any() and not big.all()
        if name in ("bench", "near circular", "near equatorial", "across ra = 0", "irregular angles"):
            assert ell.all(), name
        if name == "eccentric":
            assert hyp.any() and (cls == 2).any() and ell.any()
        d, a = out[ell, 4:11], out[ell, 11:18]
        assert (np.abs(d[:, 0] - a[:, 0]) <= 1e-13 * a[:, 0]).all(), name                # a
        assert (np.abs(d[:, 1:3] - a[:, 1:3]) <= 1e-13).all(), (name, np.abs(d[:, 1:3] - a[:, 1:3]).max())   # h, k
        # p, q = tan(i/2) (sin, cos)(node): towards i = 180 deg both routes lose digits (1 + cos i cancels in one, tan runs
        # into its pole in the other); what the frame needs is their error over u = 1 + p^2 + q^2
        u = 1.0 + a[:, 3] ** 2 + a[:, 4] ** 2
        assert (np.abs(d[:, 3:5] - a[:, 3:5]) <= 1e-13 * u[:, None]).all(), (name, (np.abs(d[:, 3:5] - a[:, 3:5]) / u[:, None]).max())
        dl = np.abs((d[:, 5] - a[:, 5] + np.pi) % (2 * np.pi) - np.pi)
        assert (dl <= 2e-13).all(), (name, dl.max())                                          # lambda, modulo 2 pi
        assert (np.abs(d[:, 6] - a[:, 6]) <= 1e-13 * a[:, 6]).all(), name                   # n
        assert (np.abs(out[ell, 18]) <= 1e-15).all(), name                                    # Kepler's equation at the epoch
        t = out[ell, 19:].reshape(-1, n_obs, 4)
        assert (t[:, :, 0] == 1.0).all() and (t[:, :, 2] == 1.0).all(), name
        dr = np.abs(np.sqrt(t[:, :, 1]) - np.sqrt(t[:, :, 3]))
        assert (dr <= 1e-7 + 1e-9 * np.sqrt(t[:, :, 3])).all(), (name, dr.max())
        total_elliptic += int(ell.sum())
    assert total_elliptic > 1400
