"""K0's lane form (device/triplets.hpp `select_triplets`, with the exact row skip of round 4) compiled for the host and
held against the oracle's literal generator + BinaryHeap restatement — without a GPU.

The row skip drops an (anchor, middle) row when fl(g12 + 1) >= the heap's top (every weight of the row is >= that, and a
candidate with w >= top never enters the heap): the survivors, their heap order and hence the sorted triplet list must be
those of the literal enumeration, ties included. Populations: the bench cadence (exposures exactly 0.02 d apart: equal
weights straddle the K-th place in every trajectory), long arcs of 100 and 300 observations under the reference's
defaults (down-sampling, quirk Q1, windows that cut), ragged lengths, tiny and large K."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd  # noqa: F401  (ctypes mirrors only)
from outfit_amd import synth
from outfit_amd.api import IODParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "host_harness")
BUILD = os.path.join(ROOT, "oracle", "_build")
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def K0():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libtriplets_host.so")
    src = os.path.join(HARNESS, "triplets_host.cpp")
    hdr = os.path.join(ROOT, "outfit_amd", "csrc", "device", "triplets.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + HARNESS,
                        "-I" + os.path.join(ROOT, "outfit_amd", "csrc"), src, "-o", so], check=True)
    return C.CDLL(so)


def both(K0, mjd, p):
    mjd = np.ascontiguousarray(mjd, dtype=np.float64)
    out = (C.c_uint64 * max(1, p.max_triplets))()
    n = K0.k0_lane_form(mjd.ctypes.data_as(_dp), C.c_uint64(len(mjd)), C.c_double(p.dt_min), C.c_double(p.dt_max_triplet),
                        C.c_double(p.optimal_interval_time), C.c_uint64(p.max_obs_for_triplets), C.c_uint(p.max_triplets), out)
    got = [((out[s] >> 42) & 0x1FFFFF, (out[s] >> 21) & 0x1FFFFF, out[s] & 0x1FFFFF) for s in range(n)]
    idx = (C.c_uint32 * (3 * max(1, p.max_triplets)))()
    L = O.lib()
    L.orc_generate_triplet_indices.restype = C.c_size_t
    m = L.orc_generate_triplet_indices(mjd.ctypes.data_as(_dp), C.c_size_t(len(mjd)), C.byref(p), idx, None)
    ref = [(idx[3 * s], idx[3 * s + 1], idx[3 * s + 2]) for s in range(m)]
    return got, ref


def test_lane_form_with_row_skip_equals_the_oracle(K0):
    cases = 0
    bench = synth.make_dataset(400, seed=11)
    for kw in (dict(max_triplets=100, max_obs_for_triplets=20), dict(), dict(max_triplets=1), dict(max_triplets=700),
               dict(dt_min=2.5, dt_max_triplet=11.0, max_triplets=40), dict(optimal_interval_time=3.0, max_triplets=25)):
        p = IODParams(n_noise_realizations=0, **kw).as_c()
        for t in range(400):
            got, ref = both(K0, bench["mjd_tt"][20 * t:20 * t + 20], p)
            assert got == ref, (kw, t)
            cases += 1
    for n_obs, n in ((100, 40), (300, 25)):
        ds = synth.make_long_arc_dataset(n, n_obs)
        for kw in (dict(), dict(max_triplets=100), dict(max_obs_for_triplets=300, max_triplets=50),
                   dict(max_obs_for_triplets=77, max_triplets=20), dict(dt_max_triplet=40.0, optimal_interval_time=7.0)):
            p = IODParams(n_noise_realizations=0, **kw).as_c()
            for t in range(n):
                got, ref = both(K0, ds["mjd_tt"][n_obs * t:n_obs * (t + 1)], p)
                assert got == ref, (n_obs, kw, t)
                assert len(ref) == p.max_triplets
                cases += 1
    rng = np.random.default_rng(5)
    p = IODParams(n_noise_realizations=0, max_triplets=30).as_c()
    for t in range(300):   # ragged, incl. fewer than three epochs and repeated epochs
        k = int(rng.integers(0, 40))
        mjd = np.sort(60000.0 + np.round(rng.uniform(0, 60, k), int(rng.integers(0, 3))))
        got, ref = both(K0, mjd, p)
        assert got == ref, (t, k)
        cases += 1
    assert cases > 2900
