"""GPU box: the observer pre-pass (row n2) over the bench batch — run under `rocprofv3 --kernel-trace --stats`
for the kernel time; prints the wall time of the whole entry point (H2D + kernel + D2H) and, with `--cpu`,
the oracle's time for the same observations (test infrastructure, timed for comparison only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
ed = synth.make_ephemeris()
ds = synth.add_sites(synth.make_dataset(n, ephem=ed))
ephem = outfit_amd.Ephemeris.from_dict(ed)
n_obs = ds["mjd_tt"].shape[0]
outfit_amd.observer_positions(ephem, ds["mjd_tt"][:64], ds["mjd_ut1"][:64], ds["obs_body_fixed"][:192])
t = time.perf_counter()
for _ in range(5):
    geo, hel = outfit_amd.observer_positions(ephem, ds["mjd_tt"], ds["mjd_ut1"], ds["obs_body_fixed"])
dt = (time.perf_counter() - t) / 5
print(f"{n_obs} observations: outfit_hip_observer_positions {1e3 * dt:.2f} ms wall (host buffers in and out) = "
      f"{n_obs / dt:.3e} obs/s")
if "--cpu" in sys.argv:
    import oracle_lib as O
    m = min(n_obs, 400000)
    t = time.perf_counter()
    rc, rgeo, rhel = O.observer_positions(ephem.as_c(), ds["mjd_tt"][:m], ds["mjd_ut1"][:m], ds["obs_body_fixed"][:3 * m])
    dc = time.perf_counter() - t
    same = np.array_equal(geo[:m].view(np.uint64), rgeo.view(np.uint64)) and np.array_equal(hel[:m].view(np.uint64), rhel.view(np.uint64))
    print(f"oracle, 1 core, {m} observations: {dc:.2f} s = {m / dc:.3e} obs/s; bit-identical: {same}")
