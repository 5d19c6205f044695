"""GPU box: parity on a RAGGED batch — 20 000 trajectories with 3..64 observations each (random), several parameter
sets — every integer output and the elements must equal the oracle's; the corrector's records likewise."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ed = synth.make_ephemeris()
ds = synth.make_dataset(n, seed=99, ephem=ed, n_nights=8, n_exp=8)
m = 64
rng = np.random.default_rng(3)
counts = rng.integers(3, m + 1, size=n)
counts[::97] = rng.integers(0, 3, size=len(counts[::97]))      # some hopeless ones
keep = np.concatenate([t * m + np.sort(rng.choice(m, size=c, replace=False)) for t, c in enumerate(counts)]).astype(np.int64)
off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
cols = {k: np.ascontiguousarray(ds[k][keep]) for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec")}
helio = np.ascontiguousarray(ds["obs_helio_equ"].reshape(-1, 3)[keep].reshape(-1))
geo = np.ascontiguousarray(ds["obs_geo_ecl"].reshape(-1, 3)[keep].reshape(-1))
batch = outfit_amd.ObsBatch(off, cols["mjd_tt"], cols["ra"], cols["dec"], cols["sigma_ra"], cols["sigma_dec"], helio, geo)
ephem = outfit_amd.Ephemeris.from_dict(ed)
threads = len(os.sched_getaffinity(0))
cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
for params in (synth.bench_params(outfit_amd.IODParams()),
               outfit_amd.IODParams(n_noise_realizations=0, max_triplets=30, max_obs_for_triplets=40),
               outfit_amd.IODParams(n_noise_realizations=0, max_triplets=10, max_obs_for_triplets=100, dt_min=1.0)):
    t = time.perf_counter()
    got = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
    lsq, lobs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, got, n_devices=1)
    tg = time.perf_counter() - t
    t = time.perf_counter()
    rc, ref = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), n_threads=threads, hoist=1)
    assert rc == 0
    rc, lref, lobs_ref = O.fit_lsq(params.as_c(), cfg.as_c(), batch.as_c(), ephem.as_c(), got, n_threads=threads)
    assert rc == 0
    tc = time.perf_counter() - t
    ints = ("status", "cause", "stage", "kind", "attempts", "n_triplets", "best_triplet", "best_realization")
    diff = sum(int((got[k] != ref[k]).sum()) for k in ints)
    ok = ref["status"] == 0
    same_el = int((got["elem"][ok] == ref["elem"][ok]).all(1).sum())
    lsq_same = all(lsq[k].tobytes() == lref[k].tobytes() for k in lsq.dtype.names if k != "iod") and lobs.tobytes() == lobs_ref.tobytes()
    print(f"max_triplets={params.max_triplets} max_obs={params.max_obs_for_triplets} dt_min={params.dt_min}: "
          f"{int(off[-1])} observations, statuses {np.bincount(ref['status'], minlength=3).tolist()}, "
          f"integer differences {diff}, elements bit-identical {same_el}/{int(ok.sum())}, corrector identical {lsq_same} "
          f"(GPU {tg:.2f} s, oracle {tc:.0f} s)", flush=True)
