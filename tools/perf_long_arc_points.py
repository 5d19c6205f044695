"""GPU box: bench.py's four long-arc points (10 000 trajectories of 100 / 300 observations, IODParams::default() and
100 triplets without noise) and the headline's kernel times for one build of the library.
    python tools/perf_long_arc_points.py [path of liboutfit_hip.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
if len(sys.argv) > 1: L.LIB_PATH = sys.argv[1]
import numpy as np
import outfit_amd
from outfit_amd import synth
for n_obs in (100, 300):
    ds = synth.make_long_arc_dataset(10000, n_obs)
    ephem = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    batch = outfit_amd.ObsBatch.from_dict(ds)
    for name, p in (("default", outfit_amd.IODParams()), ("K=100", outfit_amd.IODParams(n_noise_realizations=0, max_triplets=100))):
        if p.n_noise_realizations:
            batch.noise_z = np.random.default_rng(5).standard_normal(10000 * p.max_triplets * p.n_noise_realizations * 6)
        s = outfit_amd.IODSession(batch, ephem, p)
        for _ in range(3): s.run()
        s.results()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(5): s.run()
            s.results()
            best = min(best, (time.perf_counter() - t) / 5)
        s.set_profiling(True); s.run(); s.results()
        kt = s.kernel_times(); s.close()
        print(n_obs, name, "step %.2f ms" % (1e3 * best), {k: round(v[0], 2) for k, v in kt.items() if k.startswith("k_orbit") or k.startswith("k_refine")}, flush=True)
