"""Row n2: the observer-position pre-pass (what `OutfitCache::build` stores per observation,
reference src/cache/observer_centric_cache.rs + src/observer_extension.rs:180-237) on the device.

The device pre-pass shares the correctly-rounded sin/cos and the operation order with oracle/earthrot.c,
so the bar is bit-exact positions; the IOD fed with (site vector, UT1) must then return the same bytes
as the IOD fed with the oracle's positions.
"""
import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import synth


def sites_dataset(n, **kw):
    return synth.add_sites(synth.make_dataset(n, **kw))


# ---------------------------------------------------------------------------------- host side (no GPU)
def test_add_sites_layout():
    ds = sites_dataset(7)
    n_obs = int(ds["traj_offset"][-1])
    assert ds["obs_body_fixed"].shape == (3 * n_obs,) and ds["mjd_ut1"].shape == (n_obs,)
    r = np.linalg.norm(ds["obs_body_fixed"].reshape(-1, 3), axis=1) / synth.ERAU
    assert (r > 0.99).all() and (r < 1.001).all()  # geocentric distance of a ground site, Earth radii
    dt = (ds["mjd_tt"] - ds["mjd_ut1"]) * 86400.0
    assert np.allclose(dt, 69.164, atol=1e-4)


def test_batch_carries_sites_through_slice_and_abi():
    ds = sites_dataset(6)
    b = outfit_amd.ObsBatch.from_dict(ds, device_geometry=True)
    assert b.obs_helio_equ is None and b.obs_geo_ecl is None
    c = b.as_c()
    assert not c.obs_helio_equ and not c.obs_geo_ecl and c.obs_body_fixed and c.mjd_ut1
    s = b.slice(2, 5)
    o0, o1 = int(b.traj_offset[2]), int(b.traj_offset[5])
    assert np.array_equal(s.obs_body_fixed, b.obs_body_fixed[3 * o0:3 * o1])
    assert np.array_equal(s.mjd_ut1, b.mjd_ut1[o0:o1])
    assert s.obs_helio_equ is None


def test_oracle_batch_prepass_matches_single():
    ds = sites_dataset(3)
    e = outfit_amd.Ephemeris.from_dict(ds["ephem"]).as_c()
    rc, geo, hel = O.observer_positions(e, ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
    assert rc == 0
    fx = ds["obs_body_fixed"].reshape(-1, 3)
    for o in (0, 17, 59):
        g, h = O.D3(), O.D3()
        assert O.lib().orc_observer_prepass(e, ds["mjd_tt"][o], ds["mjd_ut1"][o], O.D3(*fx[o]), g, h) == 0
        assert list(g) == list(geo[o]) and list(h) == list(hel[o])
    # a ground site is one Earth radius from the geocentre and the heliocentric vector ~1 AU
    assert np.allclose(np.linalg.norm(geo, axis=1) / synth.ERAU, 1.0, atol=2e-3)
    assert (np.abs(np.linalg.norm(hel, axis=1) - 1.0) < 0.02).all()


# ---------------------------------------------------------------------------------- device parity
@pytest.mark.gpu
def test_device_positions_bit_exact():
    ds = sites_dataset(400)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    geo, hel = outfit_amd.observer_positions(eph, ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
    rc, rgeo, rhel = O.observer_positions(eph.as_c(), ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
    assert rc == 0
    assert np.array_equal(geo.view(np.uint64), rgeo.view(np.uint64)), np.abs(geo - rgeo).max()
    assert np.array_equal(hel.view(np.uint64), rhel.view(np.uint64)), np.abs(hel - rhel).max()


@pytest.mark.gpu
def test_device_positions_odd_epochs():
    """Epochs spread over the whole table, UT1 offsets of both signs, polar / equatorial / geocentric sites, n not a
    multiple of the block size."""
    ds = synth.make_dataset(4)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    rng = np.random.default_rng(11)
    n = 1237
    tt = rng.uniform(ds["ephem"]["jd_start"] - 2400000.5 + 1, ds["ephem"]["jd_end"] - 2400000.5 - 1, n)
    ut = tt - rng.uniform(-1.0, 80.0, n) / 86400.0
    fx = rng.normal(size=(n, 3))
    fx *= synth.ERAU / np.linalg.norm(fx, axis=1)[:, None]
    fx[0] = 0.0                                    # geocentre
    fx[1] = [0.0, 0.0, synth.ERAU]                 # pole
    fx[2] = [synth.ERAU, 0.0, 0.0]                 # equator, Greenwich meridian
    geo, hel = outfit_amd.observer_positions(eph, tt, ut, fx)
    rc, rgeo, rhel = O.observer_positions(eph.as_c(), tt, ut, fx)
    assert rc == 0
    assert np.array_equal(geo.view(np.uint64), rgeo.view(np.uint64))
    assert np.array_equal(hel.view(np.uint64), rhel.view(np.uint64))
    assert np.array_equal(geo[0], np.zeros(3))


@pytest.mark.gpu
def test_device_positions_empty_and_out_of_range():
    ds = synth.make_dataset(2)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    geo, hel = outfit_amd.observer_positions(eph, np.zeros(0), np.zeros(0), np.zeros(0))
    assert geo.shape == (0, 3) and hel.shape == (0, 3)
    tt = np.array([60010.0, 10.0])  # the second epoch is far outside the table
    with pytest.raises(outfit_amd.OutfitError):
        outfit_amd.observer_positions(eph, tt, tt, np.zeros(6))


@pytest.mark.gpu
def test_iod_from_sites_equals_iod_from_positions():
    """fit_full_iod given (site, UT1) == fit_full_iod given the oracle's positions, byte for byte; and both
    agree with the oracle run on those positions."""
    ds = sites_dataset(300)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    rc, geo, hel = O.observer_positions(eph.as_c(), ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
    assert rc == 0
    params = synth.bench_params(outfit_amd.IODParams())
    with_sites = outfit_amd.ObsBatch.from_dict(ds, device_geometry=True)
    with_pos = outfit_amd.ObsBatch(ds["traj_offset"], ds["mjd_tt"], ds["ra"], ds["dec"], ds["sigma_ra"],
                                   ds["sigma_dec"], hel.reshape(-1), geo.reshape(-1))
    a = outfit_amd.fit_full_iod_records(with_sites, eph, params, n_devices=1)
    b = outfit_amd.fit_full_iod_records(with_pos, eph, params, n_devices=1)
    assert a.tobytes() == b.tobytes()
    assert (a["status"] == 0).sum() > 200
    rc, ref = O.fit_full_iod(params.as_c(), with_pos.as_c(), eph.as_c(), n_threads=8, hoist=1)
    assert rc == 0
    assert np.array_equal(a["status"], ref["status"]) and np.array_equal(a["best_triplet"], ref["best_triplet"])
    ok = ref["status"] == 0
    assert np.abs(a["elem"][ok][:, :2] - ref["elem"][ok][:, :2]).max() <= 1e-10


@pytest.mark.gpu
def test_lsq_from_sites_equals_lsq_from_positions():
    ds = sites_dataset(120)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    rc, geo, hel = O.observer_positions(eph.as_c(), ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
    params = synth.bench_params(outfit_amd.IODParams())
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    with_sites = outfit_amd.ObsBatch.from_dict(ds, device_geometry=True)
    with_pos = outfit_amd.ObsBatch(ds["traj_offset"], ds["mjd_tt"], ds["ra"], ds["dec"], ds["sigma_ra"],
                                   ds["sigma_dec"], hel.reshape(-1), geo.reshape(-1))
    a, fa = outfit_amd.fit_lsq_records(with_sites, eph, params, cfg, None, n_devices=1)
    b, fb = outfit_amd.fit_lsq_records(with_pos, eph, params, cfg, None, n_devices=1)
    assert a.tobytes() == b.tobytes() and fa.tobytes() == fb.tobytes()
    assert (a["status"] == 0).sum() > 60


@pytest.mark.gpu
def test_sites_need_ut1_and_ephemeris():
    ds = sites_dataset(4)
    eph = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    params = synth.bench_params(outfit_amd.IODParams())
    b = outfit_amd.ObsBatch.from_dict(ds, device_geometry=True)
    b.mjd_ut1 = None
    with pytest.raises(outfit_amd.InvalidBatch):
        outfit_amd.fit_full_iod_records(b, eph, params, n_devices=1)
    b = outfit_amd.ObsBatch.from_dict(ds, device_geometry=True)
    with pytest.raises(outfit_amd.InvalidBatch):
        outfit_amd.fit_full_iod_records(b, None, params, n_devices=1)
