"""Forms of the correction kernels chosen by batch size — needs an MI355X.

Up to 393 216 candidates per chunk the two correction kernels run in the LANE-PAIR form (device/gauss.hpp,
`correction_iteration<.., kPair>`): two lanes hold one candidate, each runs one of the two independent Lagrange
corrections of an iteration (gauss.rs:1325-1344), they exchange (v, f, g, chi) and evaluate the same tail — a shorter
chain per candidate, which is what a small batch waits for. The operations and their operands are those of the one-lane
form, so the two forms must return the same bytes; OUTFIT_PAIR_MODE=0 / 1 (read by the launchers) forces either.
The suite's small parity tests run the pair form by default (they are below the threshold) against the oracle, the
large ones the one-lane form; here the two forms are held against each other on the same batches, on both sides of
the threshold, including parameter sets that send many candidates through the general kernel's root walk.
"""
import numpy as np
import pytest

import outfit_amd
from outfit_amd import synth

pytestmark = pytest.mark.gpu


def _run(batch, ephem, params, monkeypatch, pair):
    if pair is None:
        monkeypatch.delenv("OUTFIT_PAIR_MODE", raising=False)
    else:
        monkeypatch.setenv("OUTFIT_PAIR_MODE", "1" if pair else "0")
    s = outfit_amd.IODSession(batch, ephem, params)
    s.run()
    rec = s.results()
    total, n_win, flags = s.candidate_scores()
    s.close()
    return rec, total, n_win, flags


def _same(a, b, ignore_flags=0):
    rec_a, sum_a, win_a, fl_a = a
    rec_b, sum_b, win_b, fl_b = b
    assert rec_a.tobytes() == rec_b.tobytes()
    assert np.array_equal(fl_a & ~np.uint32(ignore_flags), fl_b & ~np.uint32(ignore_flags))
    scored = ((fl_a & 0x10) != 0) & ((fl_a >> 8 & 0x7) == 0) & ((fl_a & 0x80) == 0) & ((fl_a & 0x80000000) == 0)
    assert np.array_equal(win_a[scored], win_b[scored]) and sum_a[scored].tobytes() == sum_b[scored].tobytes()


PARAM_SETS = [
    {},
    dict(max_tested_solutions=1),
    dict(max_ecc=0.35, max_perihelion_au=3.5),                                   # None inside the loop: hand-over, root walk
    dict(max_ecc=0.35, max_perihelion_au=3.5, max_tested_solutions=2, newton_max_it=7),
]


@pytest.mark.usefixtures("device_math_mode")
@pytest.mark.parametrize("kw", PARAM_SETS, ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()) or "bench")
def test_lane_pair_form_equals_the_one_lane_form(kw, monkeypatch):
    e = synth.make_ephemeris()
    ephem = outfit_amd.Ephemeris.from_dict(e)
    for n, seed, extra in ((1000, 5, {}), (4200, 6, {"topocentric": True})):   # 100 000 (latency forms by default) and 420 000 candidates
        batch = outfit_amd.ObsBatch.from_dict(synth.make_dataset(n, seed=seed, ephem=e, **extra))
        params = synth.bench_params(outfit_amd.IODParams())
        for k, v in kw.items():
            setattr(params, k, v)
        one = _run(batch, ephem, params, monkeypatch, False)
        two = _run(batch, ephem, params, monkeypatch, True)
        auto = _run(batch, ephem, params, monkeypatch, None)
        _same(one, two)
        _same(one, auto)
        assert (one[0]["status"] == 0).sum() > 0.8 * n


def test_lane_pair_form_with_noisy_realisations(monkeypatch):
    """IODParams::default(): 10 triplets x 21 realisations, host-drawn normals (the direction matrix of a noisy candidate
    comes from the per-candidate scratch, not from K1's table)."""
    e = synth.make_ephemeris()
    ds = synth.make_dataset(500, seed=77, ephem=e)
    params = outfit_amd.IODParams()
    ds["noise_z"] = synth.SplitMix64(9).normal(500 * params.max_triplets * params.n_noise_realizations * 6)
    batch, ephem = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(e)
    _same(_run(batch, ephem, params, monkeypatch, False), _run(batch, ephem, params, monkeypatch, True))


def test_fold_kernel_launch_shapes_agree(monkeypatch):
    """K3 (k_select_best) stages each tile of candidates with 1 024 threads per 64 trajectories for small batches and with
    256 for throughput (64: the single-wave form); the fold itself is the first wave's either way: same records, same contender lists (read through
    K3b's result), on a noisy batch (210 candidates per trajectory: 14 tiles) and a bench batch."""
    e = synth.make_ephemeris()
    ephem = outfit_amd.Ephemeris.from_dict(e)
    ds = synth.make_dataset(700, seed=31, ephem=e)
    p_noise = outfit_amd.IODParams()
    ds_noise = dict(ds)
    ds_noise["noise_z"] = synth.SplitMix64(10).normal(700 * p_noise.max_triplets * p_noise.n_noise_realizations * 6)
    for data, params in ((ds, synth.bench_params(outfit_amd.IODParams())), (ds_noise, p_noise)):
        batch = outfit_amd.ObsBatch.from_dict(data)
        got = {}
        for threads in ("64", "256", "1024"):
            monkeypatch.setenv("OUTFIT_K3_THREADS_TEST", threads)
            got[threads] = _run(batch, ephem, params, monkeypatch, None)
        _same(got["64"], got["256"])
        _same(got["64"], got["1024"])


def test_guarded_divisions_hand_over_equals_plain_divisions(monkeypatch):
    """The throughput form of the fast correction kernel runs the iteration's divisions by constants and by shared
    denominators in their short forms under a range guard (device/fastdiv.hpp) and hands a candidate to the general
    kernel — plain divisions, replay from the first root — when the guard saw an operand outside its window (kCorrRedo).
    With OUTFIT_TEST_GUARD_REDO=m the kernel reports that for every (candidate + iteration) divisible by m, so that a
    third / a seventh of all iterations end in a hand-over: every byte must be what the undisturbed run returns, in the
    one-lane form (guarded) and in the lane-pair form (plain divisions, same hand-over)."""
    e = synth.make_ephemeris()
    ephem = outfit_amd.Ephemeris.from_dict(e)
    batch = outfit_amd.ObsBatch.from_dict(synth.make_dataset(4000, seed=41, ephem=e, topocentric=True))   # 400 000 candidates: the screen's direct route is on (> 393 216)
    for kw in ({}, dict(max_ecc=0.35, max_perihelion_au=3.5)):
        params = synth.bench_params(outfit_amd.IODParams())
        for k, v in kw.items():
            setattr(params, k, v)
        monkeypatch.delenv("OUTFIT_TEST_GUARD_REDO", raising=False)
        plain = _run(batch, ephem, params, monkeypatch, False)
        assert (plain[0]["status"] == 0).sum() > 2400
        for mod in ("3", "7"):
            monkeypatch.setenv("OUTFIT_TEST_GUARD_REDO", mod)
            # (a handed-over candidate carries kFlagSlow: which kernel finished it — nothing else may differ)
            _same(plain, _run(batch, ephem, params, monkeypatch, False), ignore_flags=0x100000)
            _same(plain, _run(batch, ephem, params, monkeypatch, True), ignore_flags=0x100000)
        monkeypatch.delenv("OUTFIT_TEST_GUARD_REDO", raising=False)
