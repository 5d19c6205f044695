"""The CPU oracle against the reference's own self-contained known-answer tests.

Every vector in tests/golden/reference_kats.json is transcribed from a #[test]
in /root/reference (file:line in its "source" field) together with the
tolerance that test asserts (0 = assert_eq!, bit-exact). This is what pins the
oracle: the reference itself cannot be built here.
"""
import ctypes as C

import pytest

import oracle_lib as O

EPS = 2.220446049250313e-16


def close(a, b, tol):
    if tol == 0:
        return a == b
    return abs(a - b) <= tol


def all_close(xs, ys, tol):
    return len(xs) == len(ys) and all(close(x, y, tol) for x, y in zip(xs, ys))


def cols(flat9):
    return [flat9[0:3], flat9[3:6], flat9[6:9]]


@pytest.fixture(scope="module")
def L():
    return O.lib()


def _triplet(k):
    return O.gauss_obs([0, 1, 2], k["ra"], k["dec"], k["time"], cols(k["observer_colmajor"]))


def test_gauss_prelim_and_coeffs(L, kats):
    k = kats["triplet_K09R05F"]
    g = _triplet(k)
    p = O.GaussPrelim()
    assert L.orc_gauss_prelim_fn(C.byref(g), C.byref(p)) == 0
    tol = k["tol"]
    assert close(p.tau1, k["tau1"], tol) and close(p.tau3, k["tau3"], tol)
    assert all_close(O.mat33_to_colmajor(p.S), k["unit_matrix_colmajor"], tol)
    assert all_close(O.mat33_to_colmajor(p.Sinv), k["inv_unit_matrix_colmajor"], tol)
    assert all_close(list(p.a), k["vector_a"], tol)
    assert all_close(list(p.b), k["vector_b"], tol)
    c6, c3, c0 = C.c_double(), C.c_double(), C.c_double()
    L.orc_coeff_eight_poly(C.byref(g), C.byref(p), C.byref(c6), C.byref(c3), C.byref(c0))
    assert (c6.value, c3.value, c0.value) == (k["coeff_6"], k["coeff_3"], k["coeff_0"])


@pytest.mark.parametrize("horner_fma", [1, 0])
def test_aberth_roots_and_order(L, kats, horner_fma):
    """aberth 0.4.1 restatement: values AND visiting order (gauss.rs:1543-1570).
    Both Horner variants (fused / unfused) must land on the same roots."""
    k = kats["solve_8poly"]
    flag = C.c_int.in_dll(L, "orc_aberth_horner_fma")
    old = flag.value
    flag.value = horner_fma
    try:
        roots, n = O.D8(), C.c_int()
        rc = L.orc_solve_8poly(O.D9(*k["polynomial"]), k["max_iterations"], k["aberth_epsilon"],
                               k["root_acceptance_epsilon"], roots, C.byref(n))
        assert rc == 0
        assert [roots[i] for i in range(n.value)] == k["roots_in_order"]
    finally:
        flag.value = old


def _vector_c(p, root):
    r2m3 = 1.0 / (root * root * root)
    return O.D3(p.a[0] + p.b[0] * r2m3, -1.0, p.a[2] + p.b[2] * r2m3)


def test_asteroid_position(L, kats):
    k = kats["asteroid_position"]
    g = _triplet(kats["triplet_K09R05F"])
    p = O.GaussPrelim()
    L.orc_gauss_prelim_fn(C.byref(g), C.byref(p))
    ip = O.default_params()
    pos, ep = O.M33(), C.c_double()
    assert L.orc_position_vector_and_reference_epoch(
        C.byref(g), C.byref(ip), p.S, p.Sinv, _vector_c(p, k["rejected_root"]), pos, C.byref(ep)) != 0
    assert L.orc_position_vector_and_reference_epoch(
        C.byref(g), C.byref(ip), p.S, p.Sinv, _vector_c(p, k["accepted_root"]), pos, C.byref(ep)) == 0
    assert O.mat33_to_colmajor(pos) == k["positions_colmajor"]
    assert ep.value == k["reference_epoch"]


def test_gibbs_correction(L, kats):
    k = kats["gibbs_correction"]
    g = _triplet(kats["triplet_K09R05F"])
    p = O.GaussPrelim()
    L.orc_gauss_prelim_fn(C.byref(g), C.byref(p))
    v = O.D3()
    L.orc_gibbs_correction(O.mat33_from_colmajor(k["positions_colmajor"]), p.tau1, p.tau3, v)
    assert list(v) == k["velocity"]


def test_orbit_correction(L, kats):
    k = kats["orbit_correction"]
    g = _triplet(kats["triplet_K09R05F"])
    ip = O.default_params()
    po, vo, eo = O.M33(), O.D3(), C.c_double()
    rc = L.orc_pos_and_vel_correction(
        C.byref(g), C.byref(ip), O.mat33_from_colmajor(k["ast_pos_colmajor"]), O.D3(*k["ast_vel"]),
        O.mat33_from_colmajor(k["unit_matrix_colmajor"]),
        O.mat33_from_colmajor(k["inv_unit_matrix_colmajor"]),
        k["peri_max"], k["ecc_max"], k["err_max"], k["itmax"], po, vo, C.byref(eo))
    assert rc == 1
    assert O.mat33_to_colmajor(po) == k["new_pos_colmajor"]
    assert list(vo) == k["new_vel"]
    assert eo.value == k["corrected_epoch"]


def test_solve_orbit_orbfit(L, kats):
    k = kats["solve_orbit"]
    ip = O.default_params()
    for case in k["cases"]:
        g = O.gauss_obs([0, 1, 2], case["ra"], case["dec"], case["time"], cols(case["observer_colmajor"]))
        res = O.GaussResult()
        assert L.orc_prelim_orbit(C.byref(g), C.byref(ip), C.byref(res)) == 0
        assert res.el.kind == 0
        assert close(res.el.epoch, case["epoch"], k["tol"])
        assert all_close(list(res.el.e), case["elements"], k["tol"])


def test_velocity_correction(L, kats):
    k = kats["velocity_correction"]
    f, g, chi, v = C.c_double(), C.c_double(), C.c_double(), O.D3()
    rc = L.orc_velocity_correction_with_guess(O.D3(*k["x1"]), O.D3(*k["x2"]), O.D3(*k["v2"]), k["dt"],
                                              k["peri_max"], k["ecc_max"], 0, 0.0, k["eps"], v,
                                              C.byref(f), C.byref(g), C.byref(chi))
    assert rc == 0
    assert (f.value, g.value) == (k["f"], k["g"])
    assert list(v) == k["v"]


def test_s_funct(L, kats):
    k = kats["s_funct"]
    s = O.D4()
    L.orc_s_funct(k["psi"], k["alpha"], s)
    assert list(s) == k["s"]


def test_s_funct_invariants(L):
    """stumpff.rs:305-402: s0 = 1 + a*s2, s1 = psi + a*s3; both branches; parity in psi."""
    for psi, alpha in [(0.01, 0.1), (10.0, 5.0), (2.0, 0.0), (2.5, 1.0), (1.0, 0.5), (-30.0, -0.4)]:
        s = O.D4()
        L.orc_s_funct(psi, alpha, s)
        rel = 1e-7 if abs(alpha * psi * psi) >= 100 else 1e-12
        assert abs(s[0] - (1 + alpha * s[2])) <= rel * max(1, abs(s[0]))
        assert abs(s[1] - (psi + alpha * s[3])) <= rel * max(1, abs(s[1]))
        sn = O.D4()
        L.orc_s_funct(-psi, alpha, sn)
        assert abs(s[0] - sn[0]) < 1e-12 * max(1, abs(s[0])) and abs(s[1] + sn[1]) < 1e-12 * max(1, abs(s[1]))
    s = O.D4()
    L.orc_s_funct(0.0, 2.0, s)
    assert list(s) == [1.0, 0.0, 0.0, 0.0]


def test_prelim_kepuni(L, kats):
    """The reference's values are bit-exact with the platform libm it ran on (glibc: `orc_use_libm = 1`).
    The default, correctly-rounded mode differs from them by at most the last places in the HYPERBOLIC case only:
    glibc's sinh/cosh are off by an ulp on ~20 % of arguments (tests/test_crmath.py), its sin/cos/acos almost never."""
    import math
    flag = C.c_int.in_dll(L, "orc_use_libm")
    for c in kats["prelim_kepuni"]["cases"]:
        kp = O.KeplerParams(dt=c["dt"], r0=c["r0"], sig0=c["sig0"], mu=c["mu"], alpha=c["alpha"],
                            e0=c["e0"], convergency=c["convergency"], has_guess=0, psi_guess=0.0,
                            max_iter_prelim_kepuni=20, kind=0)
        flag.value = 1
        try:
            assert L.orc_prelim_kepuni(C.byref(kp)) == c["psi"]
        finally:
            flag.value = 0
        got = L.orc_prelim_kepuni(C.byref(kp))
        if c["alpha"] > 0.0:
            assert abs(got - c["psi"]) <= 4 * math.ulp(c["psi"])
        else:
            assert got == c["psi"]


def test_newton_equals_brent(L):
    """kepler/params.rs:270-458: Newton and Brent-Dekker land on the same psi, residual < 1e-10."""
    mu = 2.959122082855911e-4
    for a in (0.7, 1.0, 2.5, 30.0):
        for dt in (-400.0, -3.0, 0.5, 40.0, 900.0):
            sols = []
            for kind in (0, 1, 2):
                kp = O.KeplerParams(dt=dt, r0=a, sig0=0.0, mu=mu, alpha=-1.0 / a, e0=0.0,
                                    convergency=100 * EPS, has_guess=0, psi_guess=0.0,
                                    max_iter_prelim_kepuni=20, kind=kind)
                sol = O.KeplerSolution()
                assert L.orc_kepler_solve(C.byref(kp), C.byref(sol)) == 1
                s = O.D4()
                L.orc_s_funct(sol.psi, kp.alpha, s)
                assert abs(a * s[1] + s[3] - mu ** 0.5 * dt) < 1e-10
                sols.append(sol.psi)
            assert abs(sols[0] - sols[1]) < 1e-9 * max(1.0, abs(sols[0]))


def test_ccek1(L, kats):
    k = kats["ccek1"]
    el = O.Elements()
    L.orc_ccek1(O.D3(*k["position"]), O.D3(*k["velocity"]), 0.0, C.byref(el))
    assert el.kind == 0
    assert all_close(list(el.e), k["elements"], k["tol"])


def test_eccentricity_control(L, kats):
    k = kats["eccentricity_control"]
    acc, e, q, en = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    assert L.orc_eccentricity_control(O.D3(*k["position"]), O.D3(*k["velocity"]), k["peri_max"],
                                      k["ecc_max"], C.byref(acc), C.byref(e), C.byref(q), C.byref(en)) == 1
    assert bool(acc.value) == k["accepted"]
    assert (e.value, q.value, en.value) == (k["ecc"], k["peri"], k["energy"])
    # orb_elem.rs:514-522: parallel r, v -> None
    assert L.orc_eccentricity_control(O.D3(1, 0, 0), O.D3(1, 0, 0), 1e3, 2.0, C.byref(acc), C.byref(e),
                                      C.byref(q), C.byref(en)) == 0


def _equinoctial(d):
    return O.Equinoctial(d["epoch"], d["a"], d["h"], d["k"], d["p"], d["q"], d["lambda"])


def test_equinoctial_kepler_equation_and_two_body(L, kats):
    """roots 0.0.8 Newton restatement (equinoctial_element.rs:1267-1315), bit-exact."""
    k = kats["equinoctial"]
    eq = _equinoctial(k["elements"])
    F = C.c_double()
    ke = k["kepler_equation"]
    assert L.orc_solve_kepler_equation(C.byref(eq), ke["mean_longitude_t1"],
                                       ke["longitude_of_periastre"], C.byref(F)) == 0
    assert F.value == ke["F"]
    tb = k["two_body"]
    pos, vel = O.D3(), O.D3()
    assert L.orc_propagate_twobody(C.byref(eq), tb["t0"], tb["t1"], pos, vel) == 0
    assert list(pos) == tb["pos"]
    assert list(vel) == tb["vel"]


def test_topocentric_radec(L, kats):
    k = kats["topocentric_radec"]
    eq = _equinoctial(k["elements"])
    pos, vel = O.D3(), O.D3()
    assert L.orc_propagate_twobody(C.byref(eq), 0.0, 0.0, pos, vel) == 0
    c, s = 9.174820620691818e-1, 3.977771559319137e-1

    def ecl2equ(v):
        return O.D3(v[0], c * v[1] - s * v[2], s * v[1] + c * v[2])
    ra, dec = C.c_double(), C.c_double()
    L.orc_topocentric_radec(ecl2equ(pos), ecl2equ(vel), O.D3(0, 0, 0), C.byref(ra), C.byref(dec))
    assert close(ra.value, k["ra"], k["tol"]) and close(dec.value, k["dec"], k["tol"])


def test_propagate_universal(L, kats):
    k = kats["propagate_universal"]
    for case in k["cases"]:
        out = O.D11()
        assert L.orc_propagate_universal(O.D3(*case["r"]), O.D3(*case["v"]), case["t0"], case["t1"],
                                         k["convergency"], 2, out) == 0
        dr = sum((out[i] - case["r1"][i]) ** 2 for i in range(3)) ** 0.5
        dv = sum((out[3 + i] - case["v1"][i]) ** 2 for i in range(3)) ** 0.5
        assert dr < k["tol"] and dv < k["tol"]


def test_horizon_block_index(L, kats):
    """horizon_data.rs:1007-1013: block number and tau are pure arithmetic on the header."""
    import math
    k = kats["horizon_block_index"]
    et = k["mjd_tt"]
    et_jd = 2400000.5 + math.trunc(et)
    nr = math.floor((et_jd - k["jd_start"]) / k["block_days"])
    tau = ((et_jd - (nr * k["block_days"] + k["jd_start"])) + (et - math.trunc(et))) / k["block_days"]
    assert (nr, tau) == (k["nr"], k["tau"])


def test_downsample(L, kats):
    k = kats["downsample_uniform_with_edges"]
    for c in k["cases"]:
        buf = (C.c_size_t * (c["n"] + 3))()
        n = L.orc_downsample_uniform_with_edges(c["n"], c["max_keep"], buf)
        assert [buf[i] for i in range(n)] == c["keep"]
    for c in k["properties"]:
        buf = (C.c_size_t * (c["n"] + 3))()
        n = L.orc_downsample_uniform_with_edges(c["n"], c["max_keep"], buf)
        idx = [buf[i] for i in range(n)]
        assert n == c["max_keep"] and idx[0] == 0 and idx[-1] == c["n"] - 1
        assert all(b > a for a, b in zip(idx, idx[1:]))


def test_params_default_and_validation(L):
    """IODParams::default (mod.rs:308-344) and the builder's rules (mod.rs:544-624)."""
    p = O.default_params()
    assert (p.n_noise_realizations, p.max_triplets, p.max_obs_for_triplets) == (20, 10, 100)
    assert p.kepler_eps == 1e3 * EPS and p.gap_max == 8.0 / 24.0
    assert L.orc_iod_params_validate(C.byref(p)) == 0
    for field, bad in [("noise_scale", -1.0), ("dt_min", float("nan")), ("max_ecc", -0.1),
                       ("min_rho2_au", 0.0), ("aberth_eps", 0.0), ("newton_max_it", 0),
                       ("aberth_max_iter", 0), ("max_tested_solutions", 0), ("r2_min_au", 300.0)]:
        q = O.default_params()
        setattr(q, field, bad)
        assert L.orc_iod_params_validate(C.byref(q)) == -1, field
