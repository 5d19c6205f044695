"""Long trajectories — needs an MI355X.

`IODParams::default()` caps the triplet search at max_obs_for_triplets = 100 epochs (mod.rs:308-344): up to
C(100,3) = 161 700 scored triplets per trajectory (triplet_generation/mod.rs:365-408, index_generator.rs:133-271), and
with extf = -1 the RMS window is the whole arc. Two things are checked here:

* parity with the oracle at 100 observations (the cap is not hit) and at 300 (down-sampling to 100 epochs and quirk Q1:
  the reduced indices are used as observation indices), for the default parameters with noise and for 100 triplets;
* the two forms of K0 — one lane per trajectory, one wave per trajectory (device/triplets.hpp; the host picks the wave
  form from 48 reduced epochs on, OUTFIT_K0_WAVE_MIN moves the threshold) — give the same bytes, per candidate, on
  populations chosen to hurt: the bench cadence (exposures exactly 0.02 d apart: every trajectory has EQUAL weights
  straddling the K-th place, so which triplets survive is decided by the heap's sift order), ragged and degenerate
  trajectories, down-sampling, windows that cut, a heap larger than a wave.
"""
import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import synth
from test_gpu_parity import compare

pytestmark = pytest.mark.gpu


def oracle(params, batch, ephem):
    rc, ref = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), n_threads=16, hoist=1)
    assert rc == 0
    return ref


def params_default_with_noise():
    return outfit_amd.IODParams()                       # 10 triplets x (1 + 20) realisations, cap 100


def params_100_triplets():
    return outfit_amd.IODParams(n_noise_realizations=0, max_triplets=100)


@pytest.mark.usefixtures("device_math_mode")
@pytest.mark.parametrize("n_obs", [100, 300])
@pytest.mark.parametrize("make_params", [params_default_with_noise, params_100_triplets])
def test_long_trajectories_match_the_oracle(n_obs, make_params):
    n = 96
    ds = synth.make_long_arc_dataset(n, n_obs)
    params = make_params()
    nz = n * params.max_triplets * params.n_noise_realizations * 6
    if nz:
        ds["noise_z"] = synth.SplitMix64(77 + n_obs).normal(nz)
    batch, ephem = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ds["ephem"])
    got = outfit_amd.fit_full_iod_records(batch, ephem, params, 1)
    ref = oracle(params, batch, ephem)
    compare(got, ref)
    assert (ref["n_triplets"] == params.max_triplets).all()
    if outfit_amd.get_math_mode() == outfit_amd.MATH_GLIBC:
        assert got.tobytes() == ref.tobytes()
    if n_obs == 100:
        assert (ref["status"] == 0).sum() > 0.9 * n


def _run(batch, ephem, params):
    s = outfit_amd.IODSession(batch, ephem, params)
    s.run()
    rec = s.results()
    scores = s.candidate_scores()
    s.close()
    return rec, scores


def _ragged(ephem_d):
    ds = synth.make_dataset(40, seed=5, ephem=ephem_d, n_nights=8, n_exp=4)  # 32 obs each
    n = 32
    keep_counts = [32, 20, 3, 2, 0, 1, 7, 32, 11, 25] * 4
    cols = {k: [] for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec")}
    helio, geo, off = [], [], [0]
    for t, kc in enumerate(keep_counts):
        sl = slice(t * n, t * n + kc)
        for k in cols:
            cols[k].append(ds[k][sl])
        helio.append(ds["obs_helio_equ"][3 * t * n:3 * (t * n + kc)])
        geo.append(ds["obs_geo_ecl"][3 * t * n:3 * (t * n + kc)])
        off.append(off[-1] + kc)
    return outfit_amd.ObsBatch(np.array(off, dtype=np.uint64), *[np.concatenate(cols[k]) for k in cols],
                               np.concatenate(helio), np.concatenate(geo))


CASES = [
    ("bench cadence: equal weights at the K-th place", lambda e: outfit_amd.ObsBatch.from_dict(synth.make_dataset(1500, seed=11, ephem=e)),
     [synth.bench_params(outfit_amd.IODParams()), outfit_amd.IODParams(n_noise_realizations=0),
      outfit_amd.IODParams(n_noise_realizations=0, max_triplets=1), outfit_amd.IODParams(n_noise_realizations=0, max_triplets=700),
      # the wave-wide pop + push of the wave form (triplets.hpp heap_replace_wave): the smallest heaps, and both sides of
      # its limit of 128 slots (beyond it lane 0 walks the levels)
      outfit_amd.IODParams(n_noise_realizations=0, max_triplets=2), outfit_amd.IODParams(n_noise_realizations=0, max_triplets=3),
      outfit_amd.IODParams(n_noise_realizations=0, max_triplets=128), outfit_amd.IODParams(n_noise_realizations=0, max_triplets=129)]),
    ("ragged and degenerate", _ragged,
     [synth.bench_params(outfit_amd.IODParams()),
      outfit_amd.IODParams(n_noise_realizations=0, max_obs_for_triplets=12, max_triplets=30),
      outfit_amd.IODParams(n_noise_realizations=0, max_obs_for_triplets=3, max_triplets=30),
      outfit_amd.IODParams(n_noise_realizations=0, dt_min=500.0),
      outfit_amd.IODParams(n_noise_realizations=0, dt_min=2.5, dt_max_triplet=11.0, max_triplets=40)]),
    ("long arcs, 100 observations", lambda e: outfit_amd.ObsBatch.from_dict(synth.make_long_arc_dataset(60, 100)),
     [outfit_amd.IODParams(n_noise_realizations=0), outfit_amd.IODParams(n_noise_realizations=0, max_triplets=300),
      outfit_amd.IODParams(n_noise_realizations=0, max_triplets=100, dt_max_triplet=40.0, optimal_interval_time=7.0)]),
    ("long arcs, 300 observations: down-sampling", lambda e: outfit_amd.ObsBatch.from_dict(synth.make_long_arc_dataset(40, 300)),
     [outfit_amd.IODParams(n_noise_realizations=0), outfit_amd.IODParams(n_noise_realizations=0, max_obs_for_triplets=300, max_triplets=50),
      outfit_amd.IODParams(n_noise_realizations=0, max_obs_for_triplets=77, max_triplets=20)]),
]


@pytest.mark.parametrize("case", range(len(CASES)), ids=[c[0] for c in CASES])
def test_wave_form_of_k0_equals_the_lane_form(case, monkeypatch):
    """OUTFIT_K0_WAVE_MIN=3: every trajectory goes through the wave-per-trajectory K0; =0: none does. Same records and
    the same score, window and flag for every candidate, i.e. the same triplets in the same order."""
    label, make_batch, param_sets = CASES[case]
    ephem_d = synth.make_ephemeris(n_blocks=24)
    batch, ephem = make_batch(ephem_d), outfit_amd.Ephemeris.from_dict(ephem_d)
    for params in param_sets:
        monkeypatch.setenv("OUTFIT_K0_WAVE_MIN", "0")
        rec_lane, sc_lane = _run(batch, ephem, params)
        monkeypatch.setenv("OUTFIT_K0_WAVE_MIN", "3")
        rec_wave, sc_wave = _run(batch, ephem, params)
        assert rec_lane.tobytes() == rec_wave.tobytes(), label
        (sum_l, win_l, fl_l), (sum_w, win_w, fl_w) = sc_lane, sc_wave
        assert np.array_equal(fl_l, fl_w), label         # every candidate's flags (written for every slot)
        # kernels.h: 0x10 a preliminary orbit exists, bits 8-10 conversion error, 0x80 an ephemeris_error failed,
        # 0x80000000 empty slot — the score of the others is never written (stale scratch)
        scored = ((fl_l & 0x10) != 0) & ((fl_l >> 8 & 0x7) == 0) & ((fl_l & 0x80) == 0) & ((fl_l & 0x80000000) == 0)
        assert np.array_equal(win_l[scored], win_w[scored]), label
        assert sum_l[scored].tobytes() == sum_w[scored].tobytes(), label     # NaN-safe: byte comparison
        monkeypatch.delenv("OUTFIT_K0_WAVE_MIN")
        rec_auto, _ = _run(batch, ephem, params)       # the host's own choice (wave form from 48 reduced epochs on)
        assert rec_auto.tobytes() == rec_lane.tobytes(), label
    assert (rec_lane["n_triplets"] > 0).any()


@pytest.mark.parametrize("max_keep", [0, 1, 2])
def test_tiny_max_obs_for_triplets_and_trajectories_without_a_triplet(max_keep, monkeypatch):
    """max_obs_for_triplets of 0, 1 or 2 is legal and behaves like 3 (index_generator.rs:66-75). A trajectory of 1 or 2
    observations then has a REDUCED length of 3 although it has no triplet: the lane kernels used to leave it to the
    wave kernel while the host did not list it there, so n_triplets kept what the cached block held. One
    predicate now (device/triplets.hpp wave_form_owns). The block is polluted first: a run of the same shape in which
    every trajectory has triplets."""
    ephem_d = synth.make_ephemeris(n_blocks=24)
    ephem = outfit_amd.Ephemeris.from_dict(ephem_d)
    ragged = _ragged(ephem_d)                                   # lengths 32, 20, 3, 2, 0, 1, 7, ... (40 trajectories)
    full = outfit_amd.ObsBatch.from_dict(synth.make_dataset(40, seed=5, ephem=ephem_d, n_nights=8, n_exp=4))
    params = outfit_amd.IODParams(n_noise_realizations=0, max_obs_for_triplets=max_keep, max_triplets=30)
    ref = oracle(params, ragged, ephem)
    lens = np.diff(ragged.traj_offset.astype(np.int64))
    assert (ref["status"][lens < 3] == 1).all() and (ref["n_triplets"][lens < 3] == 0).all()   # NoFeasibleTriplets
    for wave_min in ("3", None, "0"):                           # every trajectory by the wave form / the host's choice / none
        if wave_min is None:
            monkeypatch.delenv("OUTFIT_K0_WAVE_MIN", raising=False)
        else:
            monkeypatch.setenv("OUTFIT_K0_WAVE_MIN", wave_min)
        pol, _ = _run(full, ephem, outfit_amd.IODParams(n_noise_realizations=0, max_triplets=30))
        assert (pol["n_triplets"] == 30).all()
        got, _ = _run(ragged, ephem, params)
        assert np.array_equal(got["n_triplets"], ref["n_triplets"]) and np.array_equal(got["status"], ref["status"])
        assert got.tobytes() == ref.tobytes() or compare(got, ref) == 0


def test_mixed_batch_uses_both_forms():
    """Short and long trajectories in one batch: the lane kernel takes the short ones, the wave kernel the long ones
    (the session's list), slice by slice; records == the same batch with the wave form switched off."""
    import os
    e = synth.make_ephemeris(n_blocks=24)
    short = synth.make_dataset(5000, seed=21, ephem=e)
    long_ = synth.make_long_arc_dataset(700, 100, ephem=e)
    cols = ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec")
    # interleave blocks: 1000 short, 140 long, ...
    off, parts, helio, geo = [0], {k: [] for k in cols}, [], []
    for b in range(5):
        for ds, lo, hi, n in ((short, b * 1000, (b + 1) * 1000, 20), (long_, b * 140, (b + 1) * 140, 100)):
            for k in cols:
                parts[k].append(ds[k][lo * n:hi * n])
            helio.append(ds["obs_helio_equ"][3 * lo * n:3 * hi * n])
            geo.append(ds["obs_geo_ecl"][3 * lo * n:3 * hi * n])
            base = off[-1]
            off.extend(base + n * (i + 1) for i in range(hi - lo))
    batch = outfit_amd.ObsBatch(np.array(off, dtype=np.uint64), *[np.concatenate(parts[k]) for k in cols],
                                np.concatenate(helio), np.concatenate(geo))
    ephem = outfit_amd.Ephemeris.from_dict(e)
    params = outfit_amd.IODParams(n_noise_realizations=0, max_triplets=20)
    auto, _ = _run(batch, ephem, params)
    os.environ["OUTFIT_K0_WAVE_MIN"] = "0"
    try:
        lane, _ = _run(batch, ephem, params)
    finally:
        del os.environ["OUTFIT_K0_WAVE_MIN"]
    assert auto.tobytes() == lane.tobytes()
    assert (auto["status"] == 0).sum() > 5000
    ref = oracle(params, batch.slice(900, 1300), ephem)      # a stretch across a short/long boundary, against the oracle
    assert auto[900:1300].tobytes() == ref.tobytes() or compare(auto[900:1300], ref) == 0
