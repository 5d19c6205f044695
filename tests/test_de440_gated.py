"""Known answers that need the real DE440 file (src/jpl_ephem/horizon/horizon_data.rs:863-1173, transcribed into
tests/golden/reference_kats_de440.json). They run when a little-endian DE440 binary (JPL `linux_p1550p2650.440`,
~100 MB) is reachable — environment variable OUTFIT_DE440_FILE, or one of the conventional places below — and are
skipped otherwise (no network in the build and test boxes). Host-only except the last test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd

HERE = os.path.dirname(os.path.abspath(__file__))
K = json.load(open(os.path.join(HERE, "golden", "reference_kats_de440.json")))
CANDIDATES = [os.environ.get("OUTFIT_DE440_FILE", ""), os.path.join(HERE, "data", "linux_p1550p2650.440"),
              os.path.expanduser("~/.cache/outfit/jpl_ephem/jpl_horizon/DE440.bsp"),
              os.path.expanduser("~/.cache/outfit/linux_p1550p2650.440"), "/data/linux_p1550p2650.440"]
DE440 = next((p for p in CANDIDATES if p and os.path.isfile(p)), None)
pytestmark = pytest.mark.skipif(DE440 is None, reason="no DE440 file (set OUTFIT_DE440_FILE)")
AU = 149597870.7


@pytest.fixture(scope="module")
def ephem():
    return outfit_amd.Ephemeris.from_file(DE440)


def test_header_matches_the_reference(ephem):
    h, want = ephem.header, K["header"]
    assert h["jpl_version"] == want["jpl_version"]
    assert [list(r) for r in h["ipt"]] == want["ipt"]
    assert h["start_period"] == want["start_period"] and h["end_period"] == want["end_period"]
    assert ephem.block_days == want["period_lenght"] and ephem.emrat == want["earth_moon_mass_ratio"]
    assert h["recsize"] == want["recsize"] and ephem.n_blocks == want["n_records"]
    assert tuple(ephem.ncoef) == (13, 13, 11) and tuple(ephem.nsub) == (2, 8, 2)


def test_record_index_and_tau(ephem):
    k = K["record_index"]
    et = np.trunc(k["mjd_tt"])
    nr = int(np.floor((2400000.5 + et - ephem.jd_start) / ephem.block_days))
    tau = ((2400000.5 + et - (nr * ephem.block_days + ephem.jd_start)) + (k["mjd_tt"] - et)) / ephem.block_days
    assert nr == k["index"] and tau == k["tau"]


def _sun_km(ephem, mjd):
    """explicit evaluation of the Sun segment of the loaded table (position path of horizon_records.rs:204-298)"""
    per = sum(3 * c * s for c, s in zip(ephem.ncoef, ephem.nsub))
    et = np.trunc(mjd)
    nr = int(np.floor((2400000.5 + et - ephem.jd_start) / ephem.block_days))
    tau = ((2400000.5 + et - (nr * ephem.block_days + ephem.jd_start)) + (mjd - et)) / ephem.block_days
    nc, ns = ephem.ncoef[2], ephem.nsub[2]
    off = 3 * ephem.ncoef[0] * ephem.nsub[0] + 3 * ephem.ncoef[1] * ephem.nsub[1]
    si = min(int(np.floor(tau * ns)), ns - 1)
    seg = ephem.coef[nr * per + off + si * 3 * nc: nr * per + off + (si + 1) * 3 * nc].reshape(3, nc)
    tc = 2.0 * (np.mod(ns * tau, 1.0) + np.trunc(tau)) - 1.0
    t = [1.0, tc]
    for i in range(2, nc):
        t.append((tc + tc) * t[i - 1] - t[i - 2])
    out = []
    for ax in range(3):
        s = 0.0
        for c, b in zip(seg[ax], t):   # left-to-right iterator sum, as the reference
            s += c * b
        out.append(s)
    return out


def test_sun_positions_bit_exact(ephem):
    for case in K["sun_wrt_ssb_km"]:
        assert _sun_km(ephem, case["mjd_tt"]) == case["position"]


def test_earth_sun_vector_oracle(ephem):
    ec = ephem.as_c()
    for case in K["earth_wrt_sun_au"]:
        p = O.D3()
        assert O.lib().orc_earth_ephemeris(C.byref(ec), case["mjd_tt"], p) == 0
        assert list(p) == case["position"]


@pytest.mark.gpu
def test_earth_sun_vector_device(ephem):
    m = np.array([c["mjd_tt"] for c in K["earth_wrt_sun_au"]])
    geo, helio = outfit_amd.observer_positions(ephem, m, m, np.zeros(3 * len(m)))
    for got, case in zip(helio, K["earth_wrt_sun_au"]):
        assert list(got) == case["position"]
