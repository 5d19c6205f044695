"""GPU box: kernel times (HIP events, summed over three back-to-back runs) of the bench workload for one build of the
library — the A/B tool of the K2c work (profiles/round05_k2c_screen.log).
    python tools/perf_kernel_times.py [path of liboutfit_hip.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd._lib as L
if len(sys.argv) > 1: L.LIB_PATH = sys.argv[1]
import outfit_amd
from outfit_amd import synth
ed = synth.make_ephemeris(); ds = synth.make_dataset(100000, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); s.results()
s.set_profiling(True)
for _ in range(3): s.run()
s.results()
print({k: round(v[0], 3) for k, v in s.kernel_times().items()})
