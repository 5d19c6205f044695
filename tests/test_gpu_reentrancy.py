"""The boundary is re-entrant in fact — needs an MI355X.

SURVEY §8(b): "blocking, re-entrant, … safe to call from multiple host threads (each call owns its streams)".
The reference's analogue is parallel == sequential (tests/test_gauss_iod.rs:170-190). Here four host threads
call the blocking entry points AT THE SAME TIME with different parameters, batches and math modes (ABI v6: the mode
is a per-call option, not process state); every call must return, byte for byte, what the same call returns alone.
ctypes releases the GIL around the foreign call, so the four calls really overlap on the device.
"""
import threading

import numpy as np
import pytest

import outfit_amd
from outfit_amd import synth

pytestmark = pytest.mark.gpu


def jobs():
    eph_d = synth.make_ephemeris()
    ephem = outfit_amd.Ephemeris.from_dict(eph_d)
    out = []
    # 1. bench shape, glibc mode
    ds = synth.make_dataset(3000, seed=101, ephem=eph_d)
    out.append(("iod", outfit_amd.ObsBatch.from_dict(ds), synth.bench_params(outfit_amd.IODParams()), outfit_amd.MATH_GLIBC, None))
    # 2. the same shape, other objects, correctly rounded, tighter limits (root walk + hand-over paths)
    ds = synth.make_dataset(2500, seed=202, ephem=eph_d, topocentric=True)
    p = synth.bench_params(outfit_amd.IODParams(max_ecc=0.35, max_perihelion_au=3.5, max_tested_solutions=2))
    out.append(("iod", outfit_amd.ObsBatch.from_dict(ds), p, outfit_amd.MATH_CORRECTLY_ROUNDED, None))
    # 3. IODParams::default() with host-drawn noise (streamed upload with the sigma columns in group 1), glibc mode
    ds = synth.make_dataset(1200, seed=303, ephem=eph_d)
    p = outfit_amd.IODParams()
    ds["noise_z"] = synth.SplitMix64(9).normal(1200 * p.max_triplets * p.n_noise_realizations * 6)
    out.append(("iod", outfit_amd.ObsBatch.from_dict(ds), p, outfit_amd.MATH_GLIBC, None))
    # 4. IOD -> differential correction in one call, correctly rounded
    ds = synth.make_dataset(2000, seed=404, ephem=eph_d)
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    out.append(("lsq", outfit_amd.ObsBatch.from_dict(ds), synth.bench_params(outfit_amd.IODParams()),
                outfit_amd.MATH_CORRECTLY_ROUNDED, cfg))
    return ephem, out


def run(job, ephem):
    kind, batch, params, mode, cfg = job
    if kind == "iod":
        return outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1, math_mode=mode).tobytes()
    rec, fit = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=1, math_mode=mode)
    return rec.tobytes() + fit.tobytes()


def test_four_host_threads_at_once_equal_their_serial_results():
    ephem, js = jobs()
    serial = [run(j, ephem) for j in js]
    assert len({s[:4096] for s in serial}) == len(js)   # the four jobs really are different
    for round_ in range(3):
        got, errs = [None] * len(js), []
        barrier = threading.Barrier(len(js))

        def worker(i):
            try:
                barrier.wait()
                got[i] = run(js[i], ephem)
            except Exception as e:  # noqa: BLE001
                errs.append((i, repr(e)))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(js))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(len(js)):
            assert got[i] == serial[i], f"round {round_}: job {i} differs from its serial result"


def test_the_process_default_does_not_leak_into_calls_that_name_their_mode():
    """Two threads flip the process-wide default while two others run calls with explicit modes."""
    ephem, js = jobs()
    a, b = js[0], js[1]
    want_a, want_b = run(a, ephem), run(b, ephem)
    stop = threading.Event()

    def flipper():
        while not stop.is_set():
            outfit_amd.set_math_mode(outfit_amd.MATH_CORRECTLY_ROUNDED)
            outfit_amd.set_math_mode(outfit_amd.MATH_GLIBC)

    prev = outfit_amd.get_math_mode()
    f = threading.Thread(target=flipper)
    f.start()
    try:
        res = [None, None]
        ta = threading.Thread(target=lambda: res.__setitem__(0, run(a, ephem)))
        tb = threading.Thread(target=lambda: res.__setitem__(1, run(b, ephem)))
        ta.start(); tb.start(); ta.join(); tb.join()
    finally:
        stop.set()
        f.join()
        outfit_amd.set_math_mode(prev)
    assert res[0] == want_a and res[1] == want_b


def test_sessions_on_one_device_from_several_threads():
    """One session per thread (a session is owned by one thread at a time), created, run and fetched concurrently."""
    eph_d = synth.make_ephemeris()
    ephem = outfit_amd.Ephemeris.from_dict(eph_d)
    params = synth.bench_params(outfit_amd.IODParams())
    batches = [outfit_amd.ObsBatch.from_dict(synth.make_dataset(1500, seed=900 + i, ephem=eph_d)) for i in range(4)]
    modes = [outfit_amd.MATH_GLIBC, outfit_amd.MATH_CORRECTLY_ROUNDED] * 2
    want = [outfit_amd.fit_full_iod_records(b, ephem, params, 1, math_mode=m).tobytes() for b, m in zip(batches, modes)]
    got = [None] * 4

    def worker(i):
        s = outfit_amd.IODSession(batches[i], ephem, params, math_mode=modes[i])
        for _ in range(3):
            s.run()
        got[i] = s.results().tobytes()
        s.close()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == want
    assert np.frombuffer(got[0], dtype=outfit_amd._abi.RESULT_DTYPE)["status"].tolist().count(0) > 1400
