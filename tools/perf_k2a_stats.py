"""GPU box: where k_gauss_roots' lanes go (diagnostic build, `make -C outfit_amd/csrc variant NAME=k2astats
DEFS=-DOUTFIT_K2A_STATS`): wave passes and active lanes per pass at the kernel body, per Aberth sweep, after the root filter,
at the cold-start section and per Newton iteration of the elliptic cold start; printed to stderr once per run.
    python tools/perf_k2a_stats.py [objects] [path of the stats library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
L.LIB_PATH = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "outfit_amd/csrc/_build/variant_k2astats/liboutfit_hip.so")
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); s.results()
print("second run:", file=sys.stderr, flush=True)
s.run(); s.results()
