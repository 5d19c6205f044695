"""GPU box: what the wave form of K0 does per trajectory on long arcs (diagnostic build,
`make -C outfit_amd/csrc variant NAME=k0stats DEFS=-DOUTFIT_K0_STATS`): rows tested / enumerated, chunks scored,
survivors, heap operations and their levels, cycles — for 100 and 300 observations, K = 10 (the reference's default)
and K = 100.    python tools/perf_k0_stats.py [objects] [library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
L.LIB_PATH = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "outfit_amd/csrc/_build/variant_k0stats/liboutfit_hip.so")
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for n_obs in (100, 300):
    ds = synth.make_long_arc_dataset(n, n_obs)
    ephem = outfit_amd.Ephemeris.from_dict(ds["ephem"])
    for k in (10, 100):
        p = outfit_amd.IODParams(n_noise_realizations=0, max_triplets=k)
        print(f"--- {n} trajectories x {n_obs} observations, max_triplets {k}", file=sys.stderr, flush=True)
        s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), ephem, p)
        s.run(); s.results(); s.close()
