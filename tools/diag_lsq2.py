"""Diagnostic (GPU box): the injected-outlier scenario of tests/test_gpu_lsq.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

ed = synth.make_ephemeris()
ds = synth.make_dataset(300, seed=21, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
rc, seeds = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), 8, 1)
off = batch.traj_offset.astype(np.int64)
bad = off[:-1] + 7
batch.dec[bad] += 30.0 * batch.sigma_dec[bad]
cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=1e300)
got, gobs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, seeds, 1)
rc, ref, robs = O.fit_lsq(params.as_c(), cfg.as_c(), batch.as_c(), ephem.as_c(), seeds, 8)
ok = (got["status"] == 0) & (ref["status"] == 0)
g, r = got[ok], ref[ok]
sig = np.sqrt(np.abs(np.diagonal(r["covariance"], axis1=1, axis2=2)))
cond = np.array([np.linalg.cond(c / np.outer(s, s)) for c, s in zip(r["covariance"], sig)])
rel = np.abs(g["normalised_rms"] - r["normalised_rms"]) / np.maximum(1, r["normalised_rms"])
for w in np.argsort(rel)[-8:]:
    print("obj", w, "rms", r["normalised_rms"][w], "rel", rel[w], "cond", cond[w], "iters", r["total_newton_iterations"][w],
          "outer", r["outer_passes"][w], "nrej", r["n_rejected"][w], "d_a/sig", abs(g["elem"][w, 0] - r["elem"][w, 0]) / sig[w, 0])
