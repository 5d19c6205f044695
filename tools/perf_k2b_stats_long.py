"""GPU box: tools/perf_k2b_stats.py's account for long trajectories (10 000 x 100 observations, max_triplets = 100, no
noise) — is K2b's time there work (correction iterations) or a tail of a few long candidates?
    python tools/perf_k2b_stats_long.py [objects] [n_obs] [path of the stats library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
L.LIB_PATH = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "outfit_amd/csrc/_build/variant_stats/liboutfit_hip.so")
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_obs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ds = synth.make_long_arc_dataset(n, n_obs)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ds["ephem"]),
                          outfit_amd.IODParams(n_noise_realizations=0, max_triplets=100))
s.run(); s.results()
print("second run:", file=sys.stderr, flush=True)
s.set_profiling(True)
s.run(); s.results()
print({k: round(v[0], 3) for k, v in s.kernel_times().items()}, file=sys.stderr)
