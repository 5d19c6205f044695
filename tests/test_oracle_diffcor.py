"""The CPU oracle's differential-correction restatement (oracle/diffcor.c) against the reference's own
self-contained known-answer tests (tests/golden/reference_kats_diffcor.json, file:line in "source")."""
import ctypes as C
import json
import os

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    return O.lib()


@pytest.fixture(scope="module")
def K():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_diffcor.json")) as f:
        return json.load(f)


def _eq(e):
    return O.Equinoctial(e["epoch"], e["a"], e["h"], e["k"], e["p"], e["q"], e["lambda"])


def _equation(e, active=True):
    q = O.ObsEquation()
    for j in range(6):
        q.d_ra[j] = e["g_ra"][j]
        q.d_dec[j] = e["g_dec"][j]
    q.residual_ra, q.residual_dec = e["res_ra"], e["res_dec"]
    q.weight_ra = 1.0 / (e["sigma_ra"] * e["sigma_ra"])
    q.weight_dec = 1.0 / (e["sigma_dec"] * e["sigma_dec"])
    q.weight_cross = 0.0
    q.active = 1 if active else 0
    return q


def test_compute_derivative_bit_exact(L, K):
    k = K["compute_derivative"]
    eq = _eq(k["elements"])
    dpos, dvel = O.M63(), O.M63()
    L.orc_compute_derivative(C.byref(eq), k["t0"], k["t1"], k["mean_motion"], k["mean_longitude_t1"],
                             k["eccentric_anomaly"], k["inv_u"], k["beta"], k["sin_ecc_anom"],
                             k["cos_ecc_anom"], k["xe"], k["ye"], k["v_xe"], k["v_ye"],
                             O.D3(*k["f_vector"]), O.D3(*k["g_vector"]), O.D3(*k["position"]),
                             O.D3(*k["velocity"]), dpos, dvel)
    assert k["tol"] == 0
    for c in range(3):
        assert [dpos[j][c] for j in range(6)] == k["dpos_by_component"][c]
        assert [dvel[j][c] for j in range(6)] == k["dvel_by_component"][c]


def test_propagate_with_partials_reproduces_the_same_jacobian(L, K, kats):
    """propagate_twobody(t0, t1, true) on the KAT elements recomputes every intermediate the KAT feeds
    to compute_derivative (equinoctial_element.rs:1288-1315 pins pos/vel bit-exactly)."""
    k = K["compute_derivative"]
    eq = _eq(k["elements"])
    pos, vel, dpos, dvel = O.D3(), O.D3(), O.M63(), O.M63()
    assert L.orc_propagate_twobody_partials(C.byref(eq), k["t0"], k["t1"], pos, vel, dpos, dvel) == 0
    assert list(pos) == k["position"] and list(vel) == k["velocity"]
    for c in range(3):
        for j in range(6):
            assert abs(dpos[j][c] - k["dpos_by_component"][c][j]) <= 1e-13
            assert abs(dvel[j][c] - k["dvel_by_component"][c][j]) <= 1e-15


def test_topocentric_radec_and_partials(L, K):
    k = K["topocentric_partials"]
    eq = _eq(k["elements"])
    pos, vel = O.D3(), O.D3()
    assert L.orc_propagate_twobody(C.byref(eq), 0.0, 0.0, pos, vel) == 0
    c, s = 9.174820620691818e-1, 3.977771559319137e-1

    def ecl2equ(v):
        return O.D3(v[0], c * v[1] - s * v[2], s * v[1] + c * v[2])
    ra, dec, d_ra, d_dec = C.c_double(), C.c_double(), O.D3(), O.D3()
    L.orc_topocentric_radec_and_partials(ecl2equ(pos), ecl2equ(vel), O.D3(0, 0, 0), C.byref(ra),
                                         C.byref(dec), d_ra, d_dec)
    tol = k["tol"]
    assert abs(ra.value - k["ra"]) <= tol and abs(dec.value - k["dec"]) <= tol
    for i in range(3):
        assert abs(d_ra[i] - k["d_ra"][i]) <= tol and abs(d_dec[i] - k["d_dec"][i]) <= tol


def test_partials_match_finite_differences(L, K):
    """observation_ephemeris.rs:1510-1600 (test_element_partials_match_finite_differences), observer at a
    fixed heliocentric point instead of the F51 cache entry."""
    e = dict(K["topocentric_partials"]["elements"])
    obs = O.D3(0.3, -0.9, -0.1)
    names = ["a", "h", "k", "p", "q", "lambda"]

    def radec(el):
        ra, dec, ga, gd = C.c_double(), C.c_double(), O.D6(), O.D6()
        assert L.orc_obs_and_partials_2body(C.byref(_eq(el)), el["epoch"] + 37.5, obs, C.byref(ra),
                                            C.byref(dec), ga, gd) == 0
        return ra.value, dec.value, list(ga), list(gd)
    _, _, ga, gd = radec(e)
    for j, nm in enumerate(names):
        h = 1e-6
        ep, em = dict(e), dict(e)
        ep[nm] += h
        em[nm] -= h
        rp, dp, _, _ = radec(ep)
        rm, dm, _, _ = radec(em)
        # 1e-3 absolute is the reference's own tolerance: the analytic Jacobian neglects the
        # light-time x d(velocity)/d(element) term of the aberration correction
        assert abs((rp - rm) / (2 * h) - ga[j]) < 1e-3
        assert abs((dp - dm) / (2 * h) - gd[j]) < 1e-3


def test_hyperbolic_elements_are_an_error(L):
    eq = O.Equinoctial(59000.0, 1.8, 0.9, 0.6, 0.01, 0.08, 1.5)  # e = 1.08
    ra, dec = C.c_double(), C.c_double()
    assert L.orc_obs_and_partials_2body(C.byref(eq), 59010.0, O.D3(1, 0, 0), C.byref(ra), C.byref(dec),
                                        O.D6(), O.D6()) == 1


def test_lsq_min_sol(L, K):
    k = K["lsq_min_sol"]
    eqs = (O.ObsEquation * 4)(*[_equation(e) for e in k["equations"]])
    sol = O.LsqSolution()
    L.orc_solve_weighted_least_squares(eqs, 4, (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1), C.byref(sol))
    assert sol.inversion_succeeded == 1 and sol.used_qr == 0 and sol.num_measurements == 8
    for j in range(6):
        assert abs(sol.dx[j] - k["dx"][j]) <= k["tol"]
    assert abs(sol.normalised_rms - k["normalised_rms"]) <= k["tol"]


def _identity_equations(n, sigma, res=0.0):
    out = []
    for i in range(n):
        g_ra, g_dec = [0.0] * 6, [0.0] * 6
        g_ra[i % 6] = 1.0
        g_dec[(i % 6 + 1) % 6] = 1.0
        out.append(dict(g_ra=g_ra, g_dec=g_dec, res_ra=res, res_dec=res, sigma_ra=sigma, sigma_dec=sigma))
    return out


def test_lsq_unit_cases(L):
    """least_square.rs:401-577: zero residuals, Gamma*N = I, rejected rows, fixed element, magnitude."""
    free = (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1)
    eqs = (O.ObsEquation * 6)(*[_equation(e) for e in _identity_equations(6, 1e-5)])
    sol = O.LsqSolution()
    L.orc_solve_weighted_least_squares(eqs, 6, free, C.byref(sol))
    assert sol.normalised_rms == 0.0 and list(sol.dx) == [0.0] * 6 and sol.num_measurements == 12
    err = 0.0
    for i in range(6):
        for j in range(6):
            p = sum(sol.covariance[i][m] * sol.normal[m][j] for m in range(6))
            err += (p - (1.0 if i == j else 0.0)) ** 2
    assert err ** 0.5 < 1e-10
    # an inactive row contributes nothing
    rows = [_equation(e) for e in _identity_equations(6, 1e-5)]
    rows.append(_equation(dict(g_ra=[1.0] * 6, g_dec=[1.0] * 6, res_ra=1e-4, res_dec=1e-4, sigma_ra=1e-5,
                               sigma_dec=1e-5), active=False))
    eqs7 = (O.ObsEquation * 7)(*rows)
    L.orc_solve_weighted_least_squares(eqs7, 7, free, C.byref(sol))
    assert sol.normalised_rms == 0.0 and sol.num_measurements == 12
    # fixed element
    eqs = (O.ObsEquation * 6)(*[_equation(e) for e in _identity_equations(6, 1e-5, 1e-4)])
    L.orc_solve_weighted_least_squares(eqs, 6, (C.c_uint8 * 6)(0, 1, 1, 1, 1, 1), C.byref(sol))
    assert sol.dx[0] == 0.0
    # correction magnitude = residual
    eqs = (O.ObsEquation * 6)(*[_equation(e) for e in _identity_equations(6, 1e-5, 1e-5)])
    L.orc_solve_weighted_least_squares(eqs, 6, free, C.byref(sol))
    assert all(abs(sol.dx[j] - 1e-5) <= 1e-12 for j in range(6))
    # nothing active: Cholesky and QR both fail (diff_cor.rs:572-611 expects DifferentialCorrectionFailed)
    for r in rows:
        r.active = 0
    eqs7 = (O.ObsEquation * 7)(*rows)
    L.orc_solve_weighted_least_squares(eqs7, 7, free, C.byref(sol))
    assert sol.inversion_succeeded == 0 and sol.num_measurements == 0 and list(sol.dx) == [0.0] * 6


def test_qr_fallback_inverts_an_indefinite_matrix(L):
    """least_square.rs:305-316: a symmetric indefinite normal matrix is not Cholesky-factorisable; the
    Householder QR inverse must still satisfy Gamma*N = I (restatement unpinned by the reference)."""
    rows = [_equation(e) for e in _identity_equations(6, 1.0, 0.5)]
    rows[0].weight_ra = -3.0  # N[0][0] = -3 + 1 < 0
    eqs = (O.ObsEquation * 6)(*rows)
    sol = O.LsqSolution()
    L.orc_solve_weighted_least_squares(eqs, 6, (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1), C.byref(sol))
    assert sol.inversion_succeeded == 1 and sol.used_qr == 1
    for i in range(6):
        for j in range(6):
            p = sum(sol.covariance[i][m] * sol.normal[m][j] for m in range(6))
            assert abs(p - (1.0 if i == j else 0.0)) < 1e-12


def test_is_bizarre(L, K):
    cfg = O.default_diffcor_config()
    for c in K["is_bizarre"]["cases"]:
        assert bool(L.orc_is_bizarre(C.byref(_eq(c["elements"])), C.byref(cfg))) == c["bizarre"]


def test_angular_diff(L, K):
    import math
    for c in K["angular_diff"]["cases"]:
        assert abs(L.orc_angular_diff(c["a"], c["b"]) - c["d"]) <= c["tol"]
    d = L.orc_angular_diff(math.pi, 0.0)
    assert -math.pi < d <= math.pi


def test_rescale_covariance(L, K):
    for c in K["rescale_covariance"]["cases"]:
        cov, nrm = O.M66(), O.M66()
        for i in range(6):
            cov[i][i] = 1.0
            nrm[i][i] = 1.0
        L.orc_rescale_covariance(cov, nrm, c["num_free"], c["num_measurements"], c["rms"])
        for i in range(6):
            for j in range(6):
                want = c["mu2"] if i == j else 0.0
                assert abs(cov[i][j] - want) <= c["tol"]
                assert abs(nrm[i][j] - (1.0 / c["mu2"] if i == j else 0.0)) <= c["tol"]


def test_outlier_rejection_unit_cases(L, K):
    k = K["outlier_rejection"]
    s = k["sigma"]
    for c in k["cases"]:
        fit = O.ObsFit(s, s, 0.0, 0.0, c["res"][0], c["res"][1], 0.0, c["selection"])
        eq = _equation(dict(g_ra=c["g_ra"], g_dec=c["g_dec"], res_ra=c["res"][0], res_dec=c["res"][1],
                            sigma_ra=s, sigma_dec=s), active=c["selection"] == 0)
        cov = O.M66()
        if c["cov"] == "identity":
            for i in range(6):
                cov[i][i] = 1.0
        fits = (O.ObsFit * 1)(fit)
        n = L.orc_update_observation_selection(fits, (O.ObsEquation * 1)(eq), 1, cov, 25.0, 9.0)
        assert (n, fits[0].selection) == (c["changes"], c["new_selection"]), c["name"]
