"""BASELINE configs[3] and configs[4] at full size on ONE device — needs an MI355X.

configs[3]: 1 000 000 synthetic objects x 20 obs, 100 triplets each, sharded 8 ways with a gather of the records;
configs[4]: the same objects through IOD -> differential correction. Here the whole 10^6-object batch runs on
one GPU (two trajectory chunks), and the properties the multi-GPU run relies on are checked:
  * every trajectory is answered;
  * independence: three slices recomputed alone give the same bytes;
  * the 8 shards of `partition_by_observations` (what rank r of an 8-GPU job computes), each run alone and
    concatenated in rank order, are byte-identical to the single run (reference: the Rayon fold/reduce of
    per-trajectory results, obs_dataset_api.rs:191-205);
  * 120 random objects against the oracle, in both math modes (bit-identical elements; glibc mode additionally
    against the platform libm within the north-star tolerance).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import dist as odist
from outfit_amd import synth
from test_glibcmath import host_is_glibc_x86_64

pytestmark = pytest.mark.gpu
N = 1_000_000


@pytest.fixture(scope="module")
def big():
    eph_d = synth.make_ephemeris()
    ds = synth.make_dataset_blocks(N, block=100_000, seed=20261000, ephem=eph_d)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    ephem = outfit_amd.Ephemeris.from_dict(eph_d)
    params = synth.bench_params(outfit_amd.IODParams())
    return batch, ephem, params


def oracle_iod(params, batch, ephem, libm=None, glibc=None):   # None: as conftest set it (= the device's default mode)
    with O.math_flags(libm=libm, glibc=glibc):
        rc, ref = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), n_threads=16, hoist=1)
    assert rc == 0
    return ref


def gather_rows(batch, idx):
    """the trajectories `idx` of `batch` as a batch of their own"""
    parts = [batch.slice(int(i), int(i) + 1) for i in idx]
    n = [p.n_traj and int(p.traj_offset[-1]) for p in parts]
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.uint64)
    cat = lambda k: np.ascontiguousarray(np.concatenate([getattr(p, k) for p in parts]))  # noqa: E731
    return outfit_amd.ObsBatch(off, cat("mjd_tt"), cat("ra"), cat("dec"), cat("sigma_ra"), cat("sigma_dec"),
                               cat("obs_helio_equ"), cat("obs_geo_ecl"), None)


INT_FIELDS = ("status", "cause", "stage", "kind", "attempts", "n_triplets", "best_triplet", "best_realization")


def test_config3_million_objects_iod(big):
    batch, ephem, params = big
    s = outfit_amd.IODSession(batch, ephem, params)
    s.run()
    a = s.results()
    s.close()
    assert len(a) == N and set(np.unique(a["status"])) <= {0, 1, 2, 3, 4}
    ok = a["status"] == 0
    assert ok.mean() > 0.98 and np.isfinite(a["rms"][ok]).all() and (a["n_triplets"] == 100).all()
    # independence of trajectories (incl. across the chunk boundary at 671 088)
    for t0, t1 in ((0, 500), (670_900, 671_300), (N - 500, N)):
        b = outfit_amd.fit_full_iod_records(batch.slice(t0, t1), ephem, params, 1)
        assert b.tobytes() == a[t0:t1].tobytes(), (t0, t1)
    # the 8 shards of the multi-GPU partition, each alone, concatenated == the single run
    cuts = odist.partition_by_observations(batch.traj_offset, 8)
    assert cuts[0] == 0 and cuts[-1] == N and all(cuts[r + 1] - cuts[r] == N // 8 for r in range(8))
    shards = [outfit_amd.fit_full_iod_records(batch.slice(cuts[r], cuts[r + 1]), ephem, params, 1) for r in range(8)]
    assert np.concatenate(shards).tobytes() == a.tobytes()
    # 120 random objects against the oracle: same bytes in every integer field and in the elements
    idx = np.sort(np.random.default_rng(77).choice(N, 120, replace=False))
    sub = gather_rows(batch, idx)
    ref = oracle_iod(params, sub, ephem)
    for k in INT_FIELDS:
        assert np.array_equal(a[k][idx], ref[k]), k
    m = ref["status"] == 0
    assert np.array_equal(a["elem"][idx][m], ref["elem"][m]) and np.array_equal(a["epoch"][idx][m], ref["epoch"][m])
    assert (np.abs(a["rms"][idx][m] - ref["rms"][m]) <= 1e-9 * np.maximum(1.0, ref["rms"][m])).all()
    if host_is_glibc_x86_64():   # the default (glibc-faithful) mode against the platform libm = the reference's arithmetic here
        refl = oracle_iod(params, sub, ephem, libm=1)
        for k in INT_FIELDS:
            assert np.array_equal(a[k][idx], refl[k]), k
        d = np.abs(a["elem"][idx][m] - refl["elem"][m])
        assert d[:, :2].max() <= 1e-10 and d[:, 2:].max() <= 1e-12
    # the same 10^6 objects in the other (correctly-rounded) mode, against the oracle's default mode
    g = outfit_amd.fit_full_iod_records(batch, ephem, params, 1, math_mode=outfit_amd.MATH_CORRECTLY_ROUNDED)
    assert np.array_equal(g["status"], a["status"])
    refg = oracle_iod(params, sub, ephem, libm=0, glibc=0)
    for k in INT_FIELDS:
        assert np.array_equal(g[k][idx], refg[k]), k
    assert np.array_equal(g["elem"][idx][m], refg["elem"][m])


def test_config4_million_objects_iod_then_lsq(big):
    batch, ephem, params = big
    cfg = outfit_amd.DifferentialCorrectionConfig()
    s = outfit_amd.IODSession(batch, ephem, params)
    s.run()
    s.run_lsq(cfg)
    rec, obs = s.lsq_results()
    s.close()
    assert len(rec) == N and set(np.unique(rec["status"])) <= {0, 1, 2, 3}
    assert (rec["status"] == 0).sum() > 0.25 * N and ((rec["status"] == 0) | (rec["status"] == 1)).mean() > 0.98
    conv = rec["status"] == 0
    assert np.isfinite(rec["elem"][conv]).all() and np.isfinite(rec["covariance"][conv]).all()
    for t0, t1 in ((0, 500), (670_900, 671_300), (N - 500, N)):
        b, bo = outfit_amd.fit_lsq_records(batch.slice(t0, t1), ephem, params, cfg, None, 1)
        assert b.tobytes() == rec[t0:t1].tobytes(), (t0, t1)
        assert bo.tobytes() == obs[20 * t0:20 * t1].tobytes()
    cuts = odist.partition_by_observations(batch.traj_offset, 8)
    for r in (0, 5, 7):   # three of the 8 shards (each is a 125 000-object IOD + corrector run)
        b, _ = outfit_amd.fit_lsq_records(batch.slice(cuts[r], cuts[r + 1]), ephem, params, cfg, None, 1)
        assert b.tobytes() == rec[cuts[r]:cuts[r + 1]].tobytes(), r
    # 120 random objects against the oracle's corrector on the device's seeds: the same bytes
    idx = np.sort(np.random.default_rng(78).choice(N, 120, replace=False))
    sub = gather_rows(batch, idx)
    seeds = np.ascontiguousarray(rec["iod"][idx])
    rc, lref, oref = O.fit_lsq(params.as_c(), cfg.as_c(), sub.as_c(), ephem.as_c(), seeds, n_threads=16)
    assert rc == 0
    for k in rec.dtype.names:
        if k == "iod":
            continue
        x, y = rec[k][idx], lref[k]
        same = (x == y) | ((x != x) & (y != y)) if x.dtype.kind == "f" else (x == y)
        assert same.all(), k
