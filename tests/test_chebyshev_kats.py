"""The Chebyshev evaluator (`HorizonRecord::interpolate`, position path) against the analytic tests the reference
holds for it — src/jpl_ephem/horizon/horizon_records.rs:301-600: explicit-basis evaluation with the literal coefficient
vectors of `interpolate_position_matches_explicit_basis` (:389-409), `tau_edge_cases` (:506-523), linearity in the
coefficients (:527-563) and continuity inside a sub-interval (:567-600). The evaluator is reached the way the path
reaches it: through the Earth position of a DE-layout table (EMB = the test record, Moon = Sun = 0).
CPU: the oracle; GPU: the device's K1 arithmetic through outfit_hip_observer_positions, bit-identical to the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd

AU = 149597870.7
X, Y, Z = [1.0, 0.5, -0.25, 0.125], [0.0, -2.0, 0.0, 1.0], [3.0, 0.0, 0.0, 0.0]   # horizon_records.rs:392-394
MJD0, DAYS = 1000.0, 2.0                                                         # rec(2000.0, 2002.0, ..): a 2-day record


def table(x, y, z, n_sub):
    nc = len(x)
    emb = np.tile(np.concatenate([x, y, z]), n_sub)          # the same polynomial in every sub-interval
    coef = np.concatenate([emb, np.zeros(6), np.zeros(6)])   # Moon, Sun: 2 coefficients, 1 sub-interval, all zero
    return outfit_amd.Ephemeris(2400000.5 + MJD0, 2400000.5 + MJD0 + DAYS, DAYS, 81.3, (nc, 2, 2), (n_sub, 1, 1), 1,
                                np.ascontiguousarray(coef))


def tc_from_tau(tau, n_sub):                                  # horizon_records.rs:349-353
    return 2.0 * (np.mod(n_sub * tau, 1.0) + np.trunc(tau)) - 1.0


def explicit(coeffs, tc):                                     # cheb_basis + eval_pos, :324-347
    t = [1.0, tc]
    for i in range(2, len(coeffs)):
        t.append((tc + tc) * t[i - 1] - t[i - 2])
    return sum(c * b for c, b in zip(coeffs, t))


def oracle_earth(eph, mjd):
    p = O.D3()
    ec = eph.as_c()
    assert O.lib().orc_earth_ephemeris(C.byref(ec), float(mjd), p) == 0
    return np.array(list(p)) * AU


def device_earth(eph, mjd):
    m = np.atleast_1d(np.asarray(mjd, dtype=np.float64))
    geo, helio = outfit_amd.observer_positions(eph, m, m, np.zeros(3 * len(m)))
    assert not geo.any()                                      # no site offset: helio = Earth
    return helio.reshape(-1, 3)[0] * AU


BACKENDS = [pytest.param(oracle_earth, id="oracle"), pytest.param(device_earth, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("earth", BACKENDS)
def test_position_matches_explicit_basis(earth):
    n_sub = 4
    eph = table(X, Y, Z, n_sub)
    for tau in (0.0, 0.1, 0.33, 0.5, 0.73, 0.999999):
        mjd = MJD0 + DAYS * tau
        t = ((np.trunc(mjd) - MJD0) + (mjd - np.trunc(mjd))) / DAYS   # the tau the block arithmetic recovers
        tc = tc_from_tau(t, n_sub)
        got = earth(eph, mjd)
        for g, c in zip(got, (X, Y, Z)):
            assert abs(g - explicit(c, tc)) <= 1e-14    # the reference's epsilon (the /AU * AU round trip costs ~2 ulp)


@pytest.mark.parametrize("earth", BACKENDS)
def test_tau_edges_linearity_and_continuity(earth):
    rng = np.random.default_rng(2)
    # tau_edge_cases (:506-523): finite results at 0, eps, 1 - 1e-15 (tau = 1 belongs to the next record here)
    eph = table([0.1, -0.2, 0.3, -0.4, 0.5], [1.0, 0.0, -1.0, 0.0, 1.0], [2.0, 2.0, 2.0, 2.0, 2.0], 4)
    for tau in (0.0, np.finfo(float).eps, 1.0 - 1e-15):
        assert np.isfinite(earth(eph, MJD0 + DAYS * tau)).all()
    # prop_linear_in_coeffs (:527-563): interpolate(a + s b) = interpolate(a) + s interpolate(b)
    for n_sub in (1, 3, 9):
        a, b = rng.uniform(-1, 1, (3, 6)), rng.uniform(-1, 1, (3, 6))
        s = 0.37
        for tau in (0.05, 0.42, 0.91):
            mjd = MJD0 + DAYS * tau
            pa, pb = earth(table(*a, n_sub), mjd), earth(table(*b, n_sub), mjd)
            pc = earth(table(*(a + s * b), n_sub), mjd)
            assert np.allclose(pc, pa + s * pb, rtol=0, atol=1e-6)        # km; coefficients O(1): 1e-14 relative to AU scale
    # prop_continuity_inside_subinterval (:567-600)
    c = rng.uniform(-1, 1, (3, 8))
    for n_sub in (1, 5, 11):
        eph = table(*c, n_sub)
        for base in (0.013, 0.5, 0.987):
            w = 1.0 / n_sub
            k = np.floor(base / w)
            lo, hi = k * w + 1e-9, (k + 1) * w - 1e-9
            t1, t2 = np.clip(base - 1e-7, lo, hi), np.clip(base + 1e-7, lo, hi)
            p1, p2 = earth(eph, MJD0 + DAYS * t1), earth(eph, MJD0 + DAYS * t2)
            assert np.abs(p2 - p1).max() < 1e-3                            # |dP/dtau| <= sum k^2 |c_k| 2 n_sub ~ 1e3


@pytest.mark.gpu
def test_device_equals_oracle_bit_for_bit_on_the_test_records():
    rng = np.random.default_rng(3)
    for n_sub in (1, 4, 7):
        eph = table(*rng.uniform(-1e8, 1e8, (3, 9)), n_sub)
        mjd = MJD0 + DAYS * rng.uniform(0, 1, 257)
        geo, helio = outfit_amd.observer_positions(eph, mjd, mjd, np.zeros(3 * len(mjd)))
        ref = np.array([oracle_earth(eph, m) / AU for m in mjd])
        assert np.array_equal(helio.reshape(-1, 3), ref)
