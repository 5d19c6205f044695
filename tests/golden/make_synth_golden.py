#!/usr/bin/env python3
"""Regenerates tests/golden/synth64_oracle.json: the CPU oracle's results on the first 64 objects of the
deterministic synthetic dataset (outfit_amd/synth.py, seed 20260101), bench parameters
(max_triplets=100, max_obs_for_triplets=20, no noise). It pins (1) the generator — SHA-256 of the
input arrays — and (2) the oracle/GPU outputs as a regression fixture. The reference itself cannot
produce vectors here (no Rust toolchain), so this fixture is the restatement's output,
not the reference's; the reference's own vectors are in reference_kats.json.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

ds = synth.make_dataset(64)
batch = outfit_amd.ObsBatch.from_dict(ds)
ephem = outfit_amd.Ephemeris.from_dict(ds["ephem"])
params = synth.bench_params(O.default_params())


def records(glibc):
    with O.math_flags(libm=0, glibc=glibc):
        rc, res = O.fit_full_iod(params, batch.as_c(), ephem.as_c(), n_threads=8, hoist=1)
    assert rc == 0
    return [
        {"status": int(r["status"]), "cause": int(r["cause"]), "stage": int(r["stage"]), "kind": int(r["kind"]),
         "attempts": int(r["attempts"]), "best_triplet": int(r["best_triplet"]),
         "n_triplets": int(r["n_triplets"]), "epoch": float(r["epoch"]).hex(),
         "elem": [float(x).hex() for x in r["elem"]], "rms": float(r["rms"]).hex()}
        for r in res]


res = records(0)
out = {
    "source": "tests/golden/make_synth_golden.py (CPU oracle; records: correctly-rounded elementary functions, "
              "oracle/crmath.c; records_glibc: sin/cos/sinh/cosh restated from glibc 2.35, oracle/glibcmath.c = the "
              "device's default OUTFIT_MATH_GLIBC)",
    "dataset_sha256": synth.dataset_sha256(ds),
    "n_obj": 64,
    "records": res,
    "records_glibc": records(1),
}
with open(os.path.join(HERE, "synth64_oracle.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(res), "records; sha", out["dataset_sha256"][:16])
