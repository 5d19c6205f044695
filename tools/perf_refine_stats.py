"""GPU box: K3b's contender statistics (diagnostic build: `make -C outfit_amd/csrc variant NAME=rstats
DEFS=-DOUTFIT_REFINE_STATS`) — trajectories, contenders per trajectory, list passes, largest list of a wave — for the
bench workload and for IODParams::default() with noise; the kernel prints them to stderr.
    python tools/perf_refine_stats.py [path of the stats library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
L.LIB_PATH = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "outfit_amd/csrc/_build/variant_rstats/liboutfit_hip.so")
import numpy as np, outfit_amd
from outfit_amd import synth
ed = synth.make_ephemeris(); eph = outfit_amd.Ephemeris.from_dict(ed)
for name, n, params in (("bench", 100000, synth.bench_params(outfit_amd.IODParams())), ("default-noise", 100000, outfit_amd.IODParams())):
    ds = synth.make_dataset(n, ephem=ed)
    nz = n * params.max_triplets * params.n_noise_realizations * 6
    if nz: ds["noise_z"] = np.random.default_rng(1).standard_normal(nz)
    print(name, file=sys.stderr, flush=True)
    s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), eph, params); s.run(); r = s.results(); s.close()
