"""Input-contract parity — needs an MI355X. Rows that are not in epoch order are sorted inside the boundary
exactly as `fit_single_traj` does (`sort_by(total_cmp)`, stable; obs_dataset_api.rs:222-223 and, for the corrector,
differential_orbit_correction/obs_dataset_api.rs:202-208); per-observation outputs return in the caller's order."""
import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import synth

pytestmark = pytest.mark.gpu


def shuffled(ds, n_traj, rng, dup_epochs=False):
    sh = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ds.items()}
    off = ds["traj_offset"]
    perm_all = np.arange(int(off[n_traj]))
    for t in range(n_traj):
        lo, hi = int(off[t]), int(off[t + 1])
        perm_all[lo:hi] = lo + rng.permutation(hi - lo)
    for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec"):
        sh[k][:len(perm_all)] = ds[k][perm_all]
    for k in ("obs_helio_equ", "obs_geo_ecl"):
        sh[k].reshape(-1, 3)[:len(perm_all)] = ds[k].reshape(-1, 3)[perm_all]
    return sh, perm_all


def test_unsorted_rows_iod_and_lsq():
    eph_d = synth.make_ephemeris()
    ds = synth.make_dataset(500, seed=99, ephem=eph_d, topocentric=True)
    rng = np.random.default_rng(8)
    sh, perm = shuffled(ds, 500, rng)
    ephem = outfit_amd.Ephemeris.from_dict(eph_d)
    params = synth.bench_params(outfit_amd.IODParams())
    b_sorted, b_shuf = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.ObsBatch.from_dict(sh)
    a = outfit_amd.fit_full_iod_records(b_sorted, ephem, params, 1)
    b = outfit_amd.fit_full_iod_records(b_shuf, ephem, params, 1)
    assert a.tobytes() == b.tobytes()
    assert (a["status"] == 0).sum() > 400
    # the oracle restates the same sort: same records as on the sorted rows
    rc, ref = O.fit_full_iod(params.as_c(), b_shuf.as_c(), ephem.as_c(), n_threads=8, hoist=1)
    assert rc == 0 and np.array_equal(ref["elem"], a["elem"]) and np.array_equal(ref["best_triplet"], a["best_triplet"])
    # corrector: records identical, per-observation fit data back in the caller's row order
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    la, fa = outfit_amd.fit_lsq_records(b_sorted, ephem, params, cfg, a, n_devices=1)
    lb, fb = outfit_amd.fit_lsq_records(b_shuf, ephem, params, cfg, a, n_devices=1)
    assert la.tobytes() == lb.tobytes()
    assert fb.tobytes() == fa[perm].tobytes()   # shuffled row i holds sorted row perm[i]


def test_repeated_epochs_keep_the_callers_order():
    """A stable sort keeps equal epochs in the caller's order: swapping two rows that share an epoch changes which
    observation sits at which index, exactly as in the reference."""
    eph_d = synth.make_ephemeris()
    ds = synth.make_dataset(64, seed=5, ephem=eph_d)
    off = ds["traj_offset"]
    for t in range(64):   # make observations 1 and 2 of every trajectory simultaneous
        ds["mjd_tt"][int(off[t]) + 2] = ds["mjd_tt"][int(off[t]) + 1]
    ephem = outfit_amd.Ephemeris.from_dict(eph_d)
    params = synth.bench_params(outfit_amd.IODParams())
    batch = outfit_amd.ObsBatch.from_dict(ds)
    got = outfit_amd.fit_full_iod_records(batch, ephem, params, 1)
    rc, ref = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), n_threads=8, hoist=1)
    assert rc == 0
    for k in ("status", "cause", "attempts", "n_triplets", "best_triplet"):
        assert np.array_equal(got[k], ref[k]), k
    ok = ref["status"] == 0
    assert np.array_equal(got["elem"][ok], ref["elem"][ok])
