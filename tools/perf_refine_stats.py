import os, sys
sys.path[:0] = ["/root/repo"]
import outfit_amd._lib as L
L.LIB_PATH = "/root/repo/outfit_amd/csrc/_build/variant_rstats/liboutfit_hip.so"
import numpy as np, outfit_amd
from outfit_amd import synth
ed = synth.make_ephemeris(); eph = outfit_amd.Ephemeris.from_dict(ed)
for name, n, params in (("bench", 100000, synth.bench_params(outfit_amd.IODParams())), ("default-noise", 100000, outfit_amd.IODParams())):
    ds = synth.make_dataset(n, ephem=ed)
    nz = n * params.max_triplets * params.n_noise_realizations * 6
    if nz: ds["noise_z"] = np.random.default_rng(1).standard_normal(nz)
    print(name, file=sys.stderr, flush=True)
    s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), eph, params); s.run(); r = s.results(); s.close()
