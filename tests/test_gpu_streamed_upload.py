"""The streamed blocking call never reads a column before its upload has landed — needs an MI355X.

`outfit_hip_fit_full_iod` uploads the observation columns in groups while kernels that do not need them yet are
already running (api.hip, `Deferred`). Round 2 shipped a race there: sigma_ra / sigma_dec were in the last group but
noisy realisations read them in the FIRST candidate kernel (gather_angles, gauss.rs:355-364); the recycled device
block usually still held the previous call's sigmas, which hid it. OUTFIT_CACHE_POISON=1 fills every device block
with 0xFF bytes (NaN) before it is handed out, so a read-before-upload cannot go unnoticed; the child process below
runs the default-noise blocking call on fresh batches under that setting and compares with the session path, which
uploads everything before it runs anything.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import outfit_amd
from outfit_amd import synth
eph_d = synth.make_ephemeris()
ephem = outfit_amd.Ephemeris.from_dict(eph_d)
report = []
for k, (n, params) in enumerate([(1000, outfit_amd.IODParams()), (300, outfit_amd.IODParams()),
                                 (1000, synth.bench_params(outfit_amd.IODParams())),
                                 (5000, outfit_amd.IODParams(n_noise_realizations=3, max_triplets=30))]):
    ds = synth.make_dataset(n, seed=700 + k, ephem=eph_d, topocentric=bool(k & 1))
    ds["sigma_ra"] = ds["sigma_ra"] * (1.0 + 0.5 * k)          # another sigma per call: a stale block would show
    nz = n * params.max_triplets * params.n_noise_realizations * 6
    if nz:
        ds["noise_z"] = np.random.default_rng(k).standard_normal(nz)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    blocking = outfit_amd.fit_full_iod_records(batch, ephem, params, 1)       # streamed upload groups
    s = outfit_amd.IODSession(batch, ephem, params)                           # everything resident first
    s.run()
    resident = s.results()
    s.close()
    ok = resident["status"] == 0
    report.append({"case": k, "same_bytes": blocking.tobytes() == resident.tobytes(), "orbits": int(ok.sum()),
                   "noisy_winners": int((resident["best_realization"][ok] > 0).sum()),
                   "finite": bool(np.isfinite(blocking["elem"][blocking["status"] == 0]).all())})
print(json.dumps(report))
"""


@pytest.mark.parametrize("env", [{"OUTFIT_CACHE_POISON": "1"}, {"OUTFIT_NO_CACHE": "1"}], ids=["poisoned-cache", "no-cache"])
def test_blocking_call_with_noise_reads_no_column_before_its_upload(env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    report = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(report) == 4
    for r in report:
        assert r["same_bytes"] and r["finite"], r
        assert r["orbits"] > 0.9 * [1000, 300, 1000, 5000][r["case"]], r
    assert report[0]["noisy_winners"] > 100 and report[3]["noisy_winners"] > 100
