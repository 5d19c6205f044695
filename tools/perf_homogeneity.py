"""GPU box: how much of the persistent correction kernel's time is lanes of DIFFERENT trajectories waiting for each other
(measurement aid). The bench batch against the same number of objects made of `groups` distinct trajectories, each
repeated consecutively (objects / groups copies): with few groups nearly every wave holds candidates of one orbit.
    python tools/perf_homogeneity.py [objects] [groups ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import outfit_amd
from outfit_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
groups = [int(x) for x in sys.argv[2:]] or [n, 10000, 1000, 100]
ed = synth.make_ephemeris(); base = synth.make_dataset(n, ephem=ed)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
for g in groups:
    rep = n // g
    idx = np.repeat(np.arange(g), rep)                       # trajectory i of the batch = base trajectory idx[i]
    obs = (idx[:, None] * 20 + np.arange(20)[None, :]).reshape(-1)
    ds = {"n_traj": len(idx), "traj_offset": np.arange(0, (len(idx) + 1) * 20, 20, dtype=np.uint64)}
    for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec"):
        ds[k] = np.ascontiguousarray(base[k][obs])
    for k in ("obs_helio_equ", "obs_geo_ecl"):
        ds[k] = np.ascontiguousarray(base[k].reshape(-1, 3)[obs].reshape(-1))
    s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), ephem, params)
    for _ in range(3): s.run()
    s.results()
    s.set_profiling(True)
    for _ in range(5): s.run()
    r = s.results()
    kt = s.kernel_times()
    s.close()
    print(f"{g:7d} distinct trajectories x {rep:6d} copies: " + " ".join("%s %.2f" % (k.replace('k_', ''), v[0] / 5) for k, v in sorted(kt.items()) if v[1]),
          f" orbits {(r['status'] == 0).sum()}", flush=True)
