"""GPU box: the differential corrector (row n1) on the device in the glibc-faithful mode against the oracle with the PLATFORM
libm (= the reference's arithmetic on this host) on a LARGE batch: IOD -> correction end to end, every byte of the LSQ records
and of the per-observation fit rows. Beside it the correctly-rounded mode against the same oracle (what rounds 1-2 shipped).
usage: parity_lsq_at_scale.py [n_objects] [seed] [bench|topo]  -> gpurun_out/parity_lsq_<mode>_<n>.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20263500
mode = sys.argv[3] if len(sys.argv) > 3 else "bench"
threads = len(os.sched_getaffinity(0))
ed = synth.make_ephemeris()
ds = synth.make_dataset(n, seed=seed, ephem=ed, topocentric=(mode == "topo"))
batch, ephem = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
cfg = outfit_amd.DifferentialCorrectionConfig()

t = time.perf_counter()
with O.math_flags(libm=1, glibc=0):
    ref = np.empty(n, dtype=outfit_amd._abi.LSQ_DTYPE)
    ref_obs = []
    for i in range(0, n, 10000):
        j = min(n, i + 10000)
        rc, r, o = O.fit_lsq(params.as_c(), cfg.as_c(), batch.slice(i, j).as_c(), ephem.as_c(), None, n_threads=threads)
        assert rc == 0
        ref[i:j] = r
        ref_obs.append(o)
        print(f"oracle (platform libm) {j}/{n} ({time.perf_counter() - t:.0f} s)", flush=True)
    ref_obs = np.concatenate(ref_obs)


def distance(mm):
    got, got_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=1, math_mode=mm)
    same = np.array([got[i].tobytes() == ref[i].tobytes() for i in range(n)])
    conv = (ref["status"] == 0) & (got["status"] == 0)
    d = np.abs(got["elem"][conv] - ref["elem"][conv])
    return {"records": n, "corrected_orbits": int((ref["status"] == 0).sum()),
            "byte_identical_records": int(same.sum()),
            "byte_identical_obs_rows": int(sum(got_obs[i].tobytes() == ref_obs[i].tobytes() for i in range(len(ref_obs)))),
            "obs_rows": int(len(ref_obs)),
            "different_status_or_iterations": int(((got["status"] != ref["status"]) |
                                                   (got["total_newton_iterations"] != ref["total_newton_iterations"])).sum()),
            "max_element_difference": float(d.max()) if d.size else 0.0}


out = {"objects": n, "seed": seed, "mode": mode, "glibc": os.confstr("CS_GNU_LIBC_VERSION"),
       "device_glibc_vs_oracle_libm": distance(outfit_amd.MATH_GLIBC),
       "device_cr_vs_oracle_libm": distance(outfit_amd.MATH_CORRECTLY_ROUNDED)}
print(json.dumps(out, indent=1), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"parity_lsq_{mode}_{n}.json"), "w"), indent=1)
