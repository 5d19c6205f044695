"""GPU box: k_diffcor in the shipped build and in an experimental build given by its path (measurement aid:
the experimental library is loaded in place of the shipped one, in its own process)."""
import os, subprocess, sys
code = r'''
import sys, hashlib
sys.path[:0] = [sys.argv[1]]
import outfit_amd._lib as L
if sys.argv[3] != "-":
    L.LIB_PATH = sys.argv[3]
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[2])
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); s.results()
cfg = outfit_amd.DifferentialCorrectionConfig()
s.run_lsq(cfg); r, _ = s.lsq_results()
s.set_profiling(True)
for _ in range(5): s.run_lsq(cfg)
t = s.kernel_times()["k_diffcor"]
print(sys.argv[3], "k_diffcor %.3f ms" % (t[0] / t[1]), hashlib.sha1(r.tobytes()).hexdigest()[:12], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000"
for path in ["-"] + sys.argv[2:]:
    subprocess.run([sys.executable, "-c", code, root, n, path], check=True)
