"""Diagnostic (GPU box): where does the HIP differential correction differ from the oracle?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import _abi, synth  # noqa: E402

ed = synth.make_ephemeris()
ds = synth.make_dataset(1000, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
rc, seeds = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), 8, 1)
cfg = outfit_amd.DifferentialCorrectionConfig(**eval(sys.argv[1]) if len(sys.argv) > 1 else dict(rms_divergence_ratio=1e300))
got, gobs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, seeds, 1)
rc, ref, robs = O.fit_lsq(params.as_c(), cfg.as_c(), batch.as_c(), ephem.as_c(), seeds, 8)
ok = (got["status"] == 0) & (ref["status"] == 0)
print("ok", ok.sum(), "int mismatches", sum(int((got[k] != ref[k]).sum()) for k in
      ("status", "dc_error", "total_newton_iterations", "num_measurements", "n_rejected", "outer_passes")))
g, r = got[ok], ref[ok]
sig = np.sqrt(np.abs(np.diagonal(r["covariance"], axis1=1, axis2=2)))
scale = sig[:, :, None] * sig[:, None, :]
rel = np.abs(g["covariance"] - r["covariance"]) / scale
print("cov rel diff: max", rel.max(), "median of per-object max", np.median(rel.reshape(len(g), -1).max(1)),
      "p99", np.percentile(rel.reshape(len(g), -1).max(1), 99))
cond = np.array([np.linalg.cond(c / np.outer(s, s)) for c, s in zip(r["covariance"], sig)])
worst = np.argsort(rel.reshape(len(g), -1).max(1))[-5:]
for w in worst:
    print("  obj", w, "rel", rel[w].max(), "cond(corr)", cond[w], "sig_a", sig[w, 0], "iters", r["total_newton_iterations"][w])
d = np.abs(g["elem"] - r["elem"])
print("elem diff max per column", d.max(0), "in sigmas", (d / sig).max(0))
print("rms rel", (np.abs(g["normalised_rms"] - r["normalised_rms"])).max())
print("res diff", np.abs(gobs["residual_ra"] - robs["residual_ra"]).max(), np.abs(gobs["residual_dec"] - robs["residual_dec"]).max())


seq = O.Equinoctial()
dx = np.zeros((len(got), 6))
for i in np.nonzero(ok)[0]:
    el = O.Elements(int(seeds["kind"][i]), float(seeds["epoch"][i]), (O.C.c_double * 6)(*seeds["elem"][i]))
    O.lib().orc_to_equinoctial(O.C.byref(el), O.C.byref(seq))
    dx[i] = ref["elem"][i] - np.array([seq.a, seq.h, seq.k, seq.p, seq.q, seq.lam])
dxo = np.abs(dx[ok])
worst = np.argsort((d / sig).max(1))[-6:]
for w in worst:
    print("  obj", w, "d/sig", (d[w] / sig[w]).max(), "d_a", d[w, 0], "sig_a", sig[w, 0], "cond", cond[w], "|dx_a|", dxo[w, 0], "rms", r["normalised_rms"][w],
          "drms", abs(g["normalised_rms"][w] - r["normalised_rms"][w]))
worst = np.argsort(np.abs(g["normalised_rms"] - r["normalised_rms"]) / np.maximum(1, r["normalised_rms"]))[-4:]
for w in worst:
    print("  rms obj", w, r["normalised_rms"][w], abs(g["normalised_rms"][w] - r["normalised_rms"][w]), "cond", cond[w])
