"""Parity of the HIP differential-correction path (`outfit_hip_fit_lsq`, kernel k_diffcor) against the
CPU oracle (oracle/diffcor.c) — needs an MI355X. SURVEY §8f row n1: FitLSQ::fit_lsq
(differential_orbit_correction/obs_dataset_api.rs:57-130).

The bar is BIT-EXACT records and per-observation fit data in both math modes (every test runs once per mode): the
corrector evaluates every elementary function either correctly rounded on both sides (device/crmath.hpp,
oracle/crmath.c) or with glibc 2.35's bits on both sides (device/glibcmath*.hpp, oracle/glibcmath*.c — the default mode
since round 3, in which the oracle equals its own platform-libm run, tests/test_gpu_glibc_mode.py), and the linear algebra
follows nalgebra's operation order. (Before that, the normal equations — condition numbers up to 7e12 — amplified the
last-place differences between glibc and ROCm sin/cos/atan2 to 1e-8 sigma and flipped an iteration count on 2 objects in
10^4.)
"""
import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import _abi, synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("device_math_mode")]

INT_FIELDS = ("status", "dc_error", "conv_error", "total_newton_iterations", "num_measurements",
              "n_rejected", "outer_passes")


@pytest.fixture(scope="module")
def ephem_dict():
    return synth.make_ephemeris()


def make(n, ephem_dict, seed=synth.DEFAULT_SEED, **kw):
    ds = synth.make_dataset(n, seed=seed, ephem=ephem_dict, **kw)
    return ds, outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ephem_dict)


def oracle_iod(params, batch, ephem):
    rc, ref = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c() if ephem is not None else None,
                             n_threads=8, hoist=1)
    assert rc == 0
    return ref


def oracle_lsq(params, cfg, batch, ephem, initial):
    rc, res, obs = O.fit_lsq(params.as_c(), cfg.as_c(), batch.as_c(), ephem.as_c() if ephem is not None else None,
                             initial, n_threads=8)
    assert rc == 0
    return res, obs


def compare_lsq(got, got_obs, ref, ref_obs, batch, ignore=()):
    """Bit-exact: the corrector evaluates every elementary function with the same bits on both sides (correctly rounded, or
    glibc's, by math mode) and follows the same operation order, so records and per-observation fit data must be the same BYTES (`ignore`: trajectories
    left out, e.g. degenerate arcs whose records hold NaN payloads)."""
    assert got.shape == ref.shape and got_obs.shape == ref_obs.shape
    keep = np.ones(len(ref), dtype=bool)
    keep[list(ignore)] = False
    for k in got.dtype.names:
        if k == "iod":
            continue  # compared by the IOD parity tests (its rms field follows the platform functions)
        a, b = got[k][keep], ref[k][keep]
        same = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)
        rows = np.nonzero(~same.reshape(len(a), -1).all(1))[0]
        assert len(rows) == 0, f"{k}: {len(rows)} trajectories differ, first {np.nonzero(keep)[0][rows[:5]]}"
    off = batch.traj_offset.astype(np.int64)
    rows = np.concatenate([np.arange(off[t], off[t + 1]) for t in np.nonzero(keep)[0]] or [np.zeros(0, np.int64)])
    for k in got_obs.dtype.names:
        a, b = got_obs[k][rows], ref_obs[k][rows]
        same = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)
        assert same.all(), f"obs_fit.{k}: {int((~same).sum())} observations differ"
    return 0


def test_lsq_on_given_seeds_1000_objects(ephem_dict):
    """Same seeds on both sides (the oracle's IOD records): isolates k_diffcor."""
    ds, batch, ephem = make(1000, ephem_dict)
    params = synth.bench_params(outfit_amd.IODParams())
    seeds = oracle_iod(params, batch, ephem)
    cfg = outfit_amd.DifferentialCorrectionConfig()
    got, got_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, seeds, n_devices=1)
    ref, ref_obs = oracle_lsq(params, cfg, batch, ephem, seeds)
    assert got["iod"].tobytes() == ref["iod"].tobytes()
    assert (ref["status"] == _abi.LSQ_DIFFCOR).sum() > 300 and (ref["status"] == _abi.LSQ_IOD_KEPT).sum() > 100
    compare_lsq(got, got_obs, ref, ref_obs, batch)


@pytest.mark.parametrize("kw", [
    dict(rms_divergence_ratio=10.0),                       # tests/test_diff_cor.rs:52-55
    dict(rms_divergence_ratio=1e300),                      # long Newton sequences
    dict(enable_outlier_rejection=0, rms_divergence_ratio=1e300),
    dict(max_newton_iterations=1),                         # diff_cor.rs:653-690
    dict(max_newton_iterations=0),                         # the inner loop never runs: Ok(initial elements), no fit data
    dict(max_outlier_rejection_passes=0, convergence_before_rejection_threshold=0.0, rms_divergence_ratio=1e300,
         chi_squared_rejection_threshold=4.0, chi_squared_recovery_threshold=1.0),   # `0..=0`: one pass, selection still updated
    dict(max_stagnation_iterations=1, rms_stagnation_ratio=0.0, rms_divergence_ratio=1e300),   # leaves at the 2nd iteration
    dict(free_elements=[0, 1, 1, 1, 1, 1], rms_divergence_ratio=1e300),   # fixed semi-major axis
    dict(convergence_before_rejection_threshold=0.0, rms_divergence_ratio=1e300,
         chi_squared_rejection_threshold=4.0, chi_squared_recovery_threshold=1.0),   # forces rejection passes
])
def test_lsq_config_variants(ephem_dict, kw):
    ds, batch, ephem = make(400, ephem_dict, seed=5)
    params = synth.bench_params(outfit_amd.IODParams())
    seeds = oracle_iod(params, batch, ephem)
    cfg = outfit_amd.DifferentialCorrectionConfig(**kw)
    got, got_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, seeds, n_devices=1)
    ref, ref_obs = oracle_lsq(params, cfg, batch, ephem, seeds)
    compare_lsq(got, got_obs, ref, ref_obs, batch)
    if "chi_squared_rejection_threshold" in kw:
        assert (ref["n_rejected"] > 0).sum() > 20
        if kw.get("max_outlier_rejection_passes") == 0:
            assert (ref["outer_passes"] <= 1).all()
        else:
            assert (ref["outer_passes"] > 1).sum() > 20
    if kw.get("max_newton_iterations") == 0:
        assert (ref["total_newton_iterations"] == 0).all()


def test_lsq_rejects_injected_outliers(ephem_dict):
    """One observation per trajectory displaced by 30 sigma: the corrector must reject it (default thresholds)."""
    ds, batch, ephem = make(300, ephem_dict, seed=21)
    params = synth.bench_params(outfit_amd.IODParams())
    seeds = oracle_iod(params, batch, ephem)   # seeds from the clean data
    off = batch.traj_offset.astype(np.int64)
    bad = off[:-1] + 7
    batch.dec[bad] += 30.0 * batch.sigma_dec[bad]
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300)
    got, got_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, seeds, n_devices=1)
    ref, ref_obs = oracle_lsq(params, cfg, batch, ephem, seeds)
    compare_lsq(got, got_obs, ref, ref_obs, batch)
    ok = ref["status"] == _abi.LSQ_DIFFCOR
    assert (ref_obs["selection"][bad][ok] == _abi.OBS_REJECTED).mean() > 0.8


def test_lsq_end_to_end_without_seeds(ephem_dict):
    """initial_orbits = None: the IOD runs on the device first (mod.rs:86); seeds then differ from the
    oracle's by the IOD parity tolerance, so a borderline divergence test may flip on a few objects."""
    ds, batch, ephem = make(500, ephem_dict, seed=9, topocentric=True)
    params = synth.bench_params(outfit_amd.IODParams())
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    got, got_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=1)
    ref, ref_obs = oracle_lsq(params, cfg, batch, ephem, None)
    assert np.array_equal(got["iod"]["status"], ref["iod"]["status"])
    compare_lsq(got, got_obs, ref, ref_obs, batch)
    res = outfit_amd.fit_lsq(batch, ephem, params, cfg)
    kinds = {type(v).__name__ for v in res.values()}
    assert "DifferentialCorrection" in kinds and "IODGauss" in kinds


def test_lsq_seed_errors_and_partial_seeds(ephem_dict):
    """Trajectories without a usable seed (status != 0 in `initial_orbits`) get the IOD; trajectories
    whose IOD fails report it (mod.rs:86 `?`)."""
    ds, batch, ephem = make(64, ephem_dict, seed=3)
    # shorten a few trajectories to two observations -> NoFeasibleTriplets
    off = batch.traj_offset.astype(np.int64)
    keep = np.ones(int(off[-1]), dtype=bool)
    for t in (5, 17):
        keep[off[t] + 2:off[t + 1]] = False
    new_off = np.concatenate([[0], np.cumsum([keep[off[t]:off[t + 1]].sum() for t in range(64)])])
    b2 = outfit_amd.ObsBatch(new_off, batch.mjd_tt[keep], batch.ra[keep], batch.dec[keep], batch.sigma_ra[keep],
                             batch.sigma_dec[keep], batch.obs_helio_equ.reshape(-1, 3)[keep],
                             batch.obs_geo_ecl.reshape(-1, 3)[keep] if batch.obs_geo_ecl is not None else None)
    params = synth.bench_params(outfit_amd.IODParams())
    seeds = oracle_iod(params, b2, ephem)
    assert seeds["status"][5] == 1 and seeds["status"][17] == 1
    half = seeds.copy()
    half["status"][::2] = 2   # "Err" entries: the reference reruns the IOD for them
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300)
    got, got_obs = outfit_amd.fit_lsq_records(b2, ephem, params, cfg, half, n_devices=1)
    ref, ref_obs = oracle_lsq(params, cfg, b2, ephem, half)
    assert got["status"][5] == _abi.LSQ_IOD_ERROR and got["iod"]["status"][5] == 1
    assert got["status"][17] == _abi.LSQ_IOD_ERROR
    compare_lsq(got, got_obs, ref, ref_obs, b2)
    res = outfit_amd.fit_lsq(b2, ephem, params, cfg, half)
    assert isinstance(res[b2.traj_ids[5]], outfit_amd.NoFeasibleTriplets)


def test_lsq_session_and_errors(ephem_dict):
    ds, batch, ephem = make(128, ephem_dict, seed=13)
    params = synth.bench_params(outfit_amd.IODParams())
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300)
    s = outfit_amd.IODSession(batch, ephem, params)
    s.run()
    s.run_lsq(cfg)
    a, a_obs = s.lsq_results()
    s.run_lsq(cfg)                      # idempotent on resident data
    b, b_obs = s.lsq_results()
    assert a.tobytes() == b.tobytes() and a_obs.tobytes() == b_obs.tobytes()
    assert np.array_equal(a["iod"], s.results())
    one, one_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=1)
    assert one.tobytes() == a.tobytes() and one_obs.tobytes() == a_obs.tobytes()
    assert "k_diffcor" in _times(s, cfg)
    s.close()
    # a session that never ran the IOD has no seeds
    s2 = outfit_amd.IODSession(batch, ephem, params)
    s2.run_lsq(cfg)
    r2, _ = s2.lsq_results()
    assert (r2["status"] == _abi.LSQ_IOD_ERROR).all()
    s2.run_lsq(cfg, a["iod"])           # ... unless the caller provides them
    r3, r3_obs = s2.lsq_results()
    seeded = a["iod"]["status"] == 0
    assert seeded.sum() > 100 and r3[seeded].tobytes() == a[seeded].tobytes()
    assert (r3["status"][~seeded] == _abi.LSQ_IOD_ERROR).all()
    assert r3_obs.tobytes() == a_obs.tobytes()
    s2.close()
    with pytest.raises(outfit_amd.InvalidIODParameter):
        outfit_amd.fit_lsq_records(batch, ephem, params, outfit_amd.DifferentialCorrectionConfig(propagator=1))
    # an absurd angle no longer rejects the batch: the reference's angular_diff would loop 1.6e8 times on it (and for
    # ever on 1e300); bounded at 1024 steps on both sides, the trajectory gets the oracle's (failed) outcome
    wild = outfit_amd.ObsBatch.from_dict(ds)
    wild.ra[3] = 1e9
    w, w_obs = outfit_amd.fit_lsq_records(wild, ephem, params, cfg, a["iod"])
    ref, ref_obs = oracle_lsq(params, cfg, wild, ephem, a["iod"])
    compare_lsq(w, w_obs, ref, ref_obs, wild)
    assert w[1:].tobytes() == a[1:].tobytes()   # the other trajectories are untouched


def test_lane_group_size_does_not_change_a_bit(ephem_dict, monkeypatch):
    """k_diffcor<kG>: kG lanes share a trajectory and add their terms to the LDS accumulators in observation
    order, so 1, 2, 4 or 8 lanes per trajectory must give byte-identical records (ragged lengths included)."""
    ds, batch, ephem = make(300, ephem_dict, seed=17)
    off = batch.traj_offset.astype(np.int64)
    keep = np.ones(int(off[-1]), dtype=bool)
    for t in range(0, 300, 7):
        keep[off[t] + 3 + (t % 11):off[t + 1]] = False        # arcs of 3 .. 13 observations
    new_off = np.concatenate([[0], np.cumsum([keep[off[t]:off[t + 1]].sum() for t in range(300)])])
    b2 = outfit_amd.ObsBatch(new_off, batch.mjd_tt[keep], batch.ra[keep], batch.dec[keep], batch.sigma_ra[keep],
                             batch.sigma_dec[keep], batch.obs_helio_equ.reshape(-1, 3)[keep],
                             batch.obs_geo_ecl.reshape(-1, 3)[keep])
    params = synth.bench_params(outfit_amd.IODParams())
    seeds = oracle_iod(params, b2, ephem)
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300, convergence_before_rejection_threshold=0.0,
                                                  chi_squared_rejection_threshold=6.0)
    outs = []
    for g in ("1", "2", "4", "8"):
        monkeypatch.setenv("OUTFIT_DIFFCOR_GROUP", g)
        outs.append(outfit_amd.fit_lsq_records(b2, ephem, params, cfg, seeds, n_devices=1))
    for rec, obs in outs[1:]:
        assert rec.tobytes() == outs[0][0].tobytes() and obs.tobytes() == outs[0][1].tobytes()
    monkeypatch.delenv("OUTFIT_DIFFCOR_GROUP")
    ref, ref_obs = oracle_lsq(params, cfg, b2, ephem, seeds)
    compare_lsq(outs[0][0], outs[0][1], ref, ref_obs, b2)
    assert (ref["n_rejected"] > 0).sum() > 10


def test_lsq_degenerate_inputs(ephem_dict):
    """Empty batch; seeded trajectories with 0, 1 and 2 observations (normal matrix singular:
    DifferentialCorrectionFailed -> the IOD orbit is kept, diff_cor.rs:335-341); every element fixed."""
    params = synth.bench_params(outfit_amd.IODParams())
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300)
    ephem = outfit_amd.Ephemeris.from_dict(ephem_dict)
    empty = outfit_amd.ObsBatch(np.zeros(1, dtype=np.uint64), *[np.zeros(0)] * 5, np.zeros(0), np.zeros(0))
    rec, obs = outfit_amd.fit_lsq_records(empty, ephem, params, cfg, None, 1)
    assert len(rec) == 0 and len(obs) == 0
    ds, batch, ephem = make(40, ephem_dict, seed=23)
    seeds = oracle_iod(params, batch, ephem)
    off = batch.traj_offset.astype(np.int64)
    keep = np.ones(int(off[-1]), dtype=bool)
    for t, n_keep in ((3, 0), (9, 1), (21, 2), (30, 5)):
        keep[off[t] + n_keep:off[t + 1]] = False
    new_off = np.concatenate([[0], np.cumsum([keep[off[t]:off[t + 1]].sum() for t in range(40)])])
    b2 = outfit_amd.ObsBatch(new_off, batch.mjd_tt[keep], batch.ra[keep], batch.dec[keep], batch.sigma_ra[keep],
                             batch.sigma_dec[keep], batch.obs_helio_equ.reshape(-1, 3)[keep],
                             batch.obs_geo_ecl.reshape(-1, 3)[keep])
    got, got_obs = outfit_amd.fit_lsq_records(b2, ephem, params, cfg, seeds, 1)   # all seeds given: no IOD
    ref, ref_obs = oracle_lsq(params, cfg, b2, ephem, seeds)
    assert got["status"][3] == _abi.LSQ_IOD_KEPT and got["dc_error"][3] == _abi.DC_FAILED
    # 1 or 2 observations: 2-4 equations for 6 unknowns. Cholesky fails, the Householder-QR fallback
    # "inverts" the rank-deficient matrix through rounding-sized pivots (least_square.rs:305-316), and what comes out
    # is an amplification of last-place noise by ~1e16: it must terminate with a legal status on both sides,
    # nothing more can be compared
    for t in (9, 21):
        assert got["status"][t] in (_abi.LSQ_DIFFCOR, _abi.LSQ_IOD_KEPT) and ref["status"][t] in (_abi.LSQ_DIFFCOR, _abi.LSQ_IOD_KEPT)
    compare_lsq(got, got_obs, ref, ref_obs, b2, ignore=(9, 21))
    fixed = outfit_amd.DifferentialCorrectionConfig(free_elements=[0] * 6, rms_divergence_ratio=1e300)
    got, got_obs = outfit_amd.fit_lsq_records(b2, ephem, params, fixed, seeds, 1)
    ref, ref_obs = oracle_lsq(params, fixed, b2, ephem, seeds)
    compare_lsq(got, got_obs, ref, ref_obs, b2, ignore=(9, 21))
    ok = ref["status"] == _abi.LSQ_DIFFCOR
    assert ok.sum() > 20 and (ref["total_newton_iterations"][ok] == 1).all()   # dx = 0: converged at once


def _times(s, cfg):
    s.set_profiling(True)
    s.run_lsq(cfg)
    t = s.kernel_times()
    s.set_profiling(False)
    return t
