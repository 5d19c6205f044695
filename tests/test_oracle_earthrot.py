"""The oracle's observer-position pre-pass (oracle/earthrot.c) against the reference's own
self-contained known-answer tests (tests/golden/reference_kats_earthrot.json)."""
import ctypes as C
import json
import math
import os

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    return O.lib()


@pytest.fixture(scope="module")
def K():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_earthrot.json")) as f:
        return json.load(f)


def _ref(kind_epoch):
    kind, ep = kind_epoch
    return O.RefSystem(kind, 1 if ep is None else 0, 0.0 if ep is None else ep)


def _check(m, columns, tol, relative=False):
    for c in range(3):
        for r in range(3):
            got, want = m[r][c], columns[c][r]
            if relative and abs(want) > 0:
                assert abs(got - want) <= tol * abs(want) or abs(got - want) <= 1e-17, (r, c, got, want)
            else:
                assert abs(got - want) <= tol, (r, c, got, want)


def test_obliquity_and_gmst_are_bit_exact(L, K):
    assert L.orc_obleq(K["obleq"]["tjm"]) == K["obleq"]["value"]
    for tut, want in K["gmst"]["cases"]:
        assert L.orc_gmst(tut) == want


def test_nutation_series_from_the_table(L, K):
    k = K["nutn80"]
    dpsi, deps = C.c_double(), C.c_double()
    L.orc_nutn80(k["tjm"], C.byref(dpsi), C.byref(deps))
    assert abs(dpsi.value - k["dpsi"]) <= k["tol_here"] and abs(deps.value - k["deps"]) <= k["tol_here"]
    m = O.M33()
    L.orc_rnut80(K["rnut80"]["tjm"], m)
    _check(m, K["rnut80"]["columns"], K["rnut80"]["tol_here"])
    # equequ = RADSEC * dpsi * cos(obleq)  (earth_orientation.rs:644-660, 1e-12 there)
    radsec = math.pi / 648000.0
    assert abs(L.orc_equequ(k["tjm"]) - radsec * k["dpsi"] * math.cos(L.orc_obleq(k["tjm"]))) < 1e-18


def test_rotpn_matrices(L, K):
    for case in K["rotpn"]["cases"]:
        m = O.M33()
        assert L.orc_rotpn(C.byref(_ref(case["src"])), C.byref(_ref(case["dst"])), m) == 0
        _check(m, case["columns"], case.get("tol_here", case["tol"]), case.get("relative", False))


def test_rotpn_properties(L):
    """ref_system.rs:594-634,833-899, same systems and tolerances: of-date precession is not the identity,
    Equm<->Eclm at J2000 are inverse (1e-12), 80000 -> J2000 stays orthonormal (1e-12), Equt(60725.5) <-> Equm(60730.5)
    round-trips to 1e-5."""
    def rot(a, b):
        m = O.M33()
        assert L.orc_rotpn(C.byref(_ref(a)), C.byref(_ref(b)), m) == 0
        return [[m[r][c] for c in range(3)] for r in range(3)]

    def mul(a, b):
        return [[sum(a[i][k] * b[k][j] for k in range(3)) for j in range(3)] for i in range(3)]

    def near_identity(p, tol):
        return all(abs(p[i][j] - (1.0 if i == j else 0.0)) < tol for i in range(3) for j in range(3))
    p = rot([0, 60000.0], [0, 61000.0])
    assert abs(1.0 - p[1][1]) > 1e-7 and not near_identity(p, 1e-12)
    assert near_identity(mul(rot([2, None], [0, None]), rot([0, None], [2, None])), 1e-12)
    big = rot([0, 80000.0], [0, None])
    assert near_identity(mul(big, [list(r) for r in zip(*big)]), 1e-12)
    assert near_identity(mul(rot([0, 60730.5], [1, 60725.5]), rot([1, 60725.5], [0, 60730.5])), 1e-5)


def test_pvobs_rotates_the_site_with_sidereal_time(L):
    """No self-contained KAT exists for pvobs (observer_centric_cache.rs:264-357 needs DE440 + EOP): properties.
    The geocentric vector keeps its length, its ecliptic-of-J2000 z oscillates with the sidereal day, and a site on
    the rotation axis moves only by precession-nutation (arcminutes over 20 years)."""
    site = O.D3(2.5e-5, -3.1e-5, 1.9e-5)
    n0 = math.sqrt(sum(x * x for x in site))
    out = O.D3()
    prev = None
    for k in range(48):
        tt = 60000.0 + k / 24.0
        L.orc_pvobs(tt, tt - 69.184 / 86400.0, site, out)
        assert abs(math.sqrt(sum(x * x for x in out)) - n0) < 1e-19
        if prev is not None:
            assert prev != list(out)
        prev = list(out)
    a, b = O.D3(), O.D3()
    L.orc_pvobs(60000.0, 60000.0, site, a)
    L.orc_pvobs(60000.0 + 0.99726957, 60000.0 + 0.99726957, site, b)   # one sidereal day later
    assert max(abs(a[i] - b[i]) for i in range(3)) < 1e-9
    pole = O.D3(0.0, 0.0, 4.2e-5)
    L.orc_pvobs(60000.0, 60000.0, pole, a)
    eps = L.orc_obleq(51544.5)
    want = (0.0, 4.2e-5 * math.sin(eps), 4.2e-5 * math.cos(eps))   # the pole of J2000 in ecliptic coordinates
    assert max(abs(a[i] - want[i]) for i in range(3)) < 4.2e-5 * 3e-3


def test_helio_position_is_earth_plus_rotated_site(L):
    out = O.D3()
    L.orc_helio_position(O.D3(0.5, -0.8, 0.1), O.D3(1e-5, 2e-5, -3e-5), out)
    c, s = 9.174820620691818e-1, 3.977771559319137e-1
    assert list(out) == [0.5 + 1e-5, -0.8 + (c * 2e-5 + (-s) * -3e-5), 0.1 + (s * 2e-5 + c * -3e-5)]
