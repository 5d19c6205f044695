"""noisy realisations with device-side draws (OUTFIT_NOISE_DEVICE, device/noise.hpp).

CPU: the numpy restatement of the generator (oracle/philox.py) against the published Philox4x32-10 known-answer
vectors and basic statistics. GPU: the device's normals equal the restatement's (same counters; the transcendental
functions differ in the last place), a device-noise run equals — byte for byte — a host-noise run fed with the
normals the device reports, both equal the oracle within the IOD tolerances, results do not depend on how the
batch is cut, and the fitted orbits are statistically as good as with independently drawn host normals."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import philox  # noqa: E402


def test_philox_known_answer_vectors():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = philox.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_restated_normals_are_standard_normal():
    z = philox.normals(12345, 400, 10, 20)
    assert z.shape == (400 * 10 * 20 * 6,)
    assert abs(z.mean()) < 4.0 / np.sqrt(z.size) and abs(z.var() - 1.0) < 6.0 * np.sqrt(2.0 / z.size)
    assert abs((z ** 3).mean()) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.05
    # independent of how trajectories are keyed into batches
    a = philox.normals(7, 5, 3, 2, traj_key_base=10)
    b = philox.normals(7, 20, 3, 2)
    assert np.array_equal(a, b.reshape(20, -1)[10:15].reshape(-1))
    c = philox.normals(7, 3, 3, 2, traj_key=[17, 3, 11])
    assert np.array_equal(c.reshape(3, -1), b.reshape(20, -1)[[17, 3, 11]])


@pytest.mark.gpu
def test_device_normals_match_the_restatement():
    keys = np.array([5, 2 ** 40 + 7, 2 ** 63 + 1, 0], dtype=np.uint64)
    for kw in (dict(traj_key=keys), dict(traj_key_base=123456789)):
        got = outfit_amd.device_normals(0xDEADBEEFCAFEF00D, 4, 7, 5, **kw)
        ref = philox.normals(0xDEADBEEFCAFEF00D, 4, 7, 5, **kw)
        assert np.abs(got - ref).max() < 1e-14   # log / sqrt / sincos: ROCm vs numpy, last place
    z = outfit_amd.device_normals(99, 2000, 10, 20)
    assert abs(z.mean()) < 4.0 / np.sqrt(z.size) and abs(z.var() - 1.0) < 6.0 * np.sqrt(2.0 / z.size)


@pytest.mark.gpu
def test_device_noise_run_equals_host_noise_run_with_the_same_normals():
    ed = synth.make_ephemeris()
    ds = synth.make_dataset(300, seed=41, ephem=ed)
    ephem = outfit_amd.Ephemeris.from_dict(ed)
    params = outfit_amd.IODParams()                     # defaults: 10 triplets x 20 noisy realisations
    seed = 0x1234567890ABCDEF
    dev = outfit_amd.ObsBatch.from_dict(ds)
    dev.noise_seed = seed
    got = outfit_amd.fit_full_iod_records(dev, ephem, params, 1)
    z = outfit_amd.device_normals(seed, 300, params.max_triplets, params.n_noise_realizations)
    host = outfit_amd.ObsBatch.from_dict(ds)
    host.noise_z = z
    replay = outfit_amd.fit_full_iod_records(host, ephem, params, 1)
    assert got.tobytes() == replay.tobytes()
    assert (got["best_realization"] > 0).sum() > 100    # the noisy realisations do win
    # the oracle on the same normals
    from test_gpu_parity import compare
    rc, ref = O.fit_full_iod(params.as_c(), host.as_c(), ephem.as_c(), 8, 1)
    assert rc == 0
    compare(got, ref)
    # cutting the batch differently (shards, chunks) changes nothing: draws are keyed by trajectory
    part = outfit_amd.fit_full_iod_records(dev.slice(100, 220), ephem, params, 1)
    assert part.tobytes() == got[100:220].tobytes()
    keyed = outfit_amd.ObsBatch.from_dict(ds)
    keyed.noise_seed = seed
    keyed.traj_key = np.arange(300, dtype=np.uint64)
    assert outfit_amd.fit_full_iod_records(keyed, ephem, params, 1).tobytes() == got.tobytes()


@pytest.mark.gpu
def test_device_noise_is_statistically_equivalent_to_host_noise():
    """No bit parity with the reference's SmallRng stream is possible: the definition is that orbits
    fitted with device draws are as good as with independent host draws of the same distribution."""
    ed = synth.make_ephemeris()
    ds = synth.make_dataset(2000, seed=43, ephem=ed)
    ephem = outfit_amd.Ephemeris.from_dict(ed)
    params = outfit_amd.IODParams()
    dev = outfit_amd.ObsBatch.from_dict(ds)
    dev.noise_seed = 2026
    a = outfit_amd.fit_full_iod_records(dev, ephem, params, 1)
    host = outfit_amd.ObsBatch.from_dict(ds)
    host.noise_z = synth.SplitMix64(77).normal(2000 * params.max_triplets * params.n_noise_realizations * 6)
    b = outfit_amd.fit_full_iod_records(host, ephem, params, 1)
    oka, okb = a["status"] == 0, b["status"] == 0
    assert abs(int(oka.sum()) - int(okb.sum())) <= 20
    assert abs(np.median(a["rms"][oka]) - np.median(b["rms"][okb])) < 0.05 * np.median(b["rms"][okb])
    ea = np.abs(a["elem"][oka, 0] - ds["truth"][oka, 0])
    eb = np.abs(b["elem"][okb, 0] - ds["truth"][okb, 0])
    assert abs(np.median(ea) - np.median(eb)) < 0.15 * np.median(eb)
    assert abs((a["best_realization"][oka] > 0).mean() - (b["best_realization"][okb] > 0).mean()) < 0.05
