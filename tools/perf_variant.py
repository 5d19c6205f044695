"""GPU box: per-kernel times and a digest of the IOD records for the shipped build and for experimental builds of the
library given by path (measurement aid: each library is loaded in place of the shipped one, in its own process)."""
import os, subprocess, sys
code = r'''
import sys, hashlib, time
sys.path[:0] = [sys.argv[1]]
import outfit_amd._lib as L
if sys.argv[3] != "-":
    L.LIB_PATH = sys.argv[3]
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[2])
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); r = s.results()
for _ in range(3): s.run()
s.results()
t = time.perf_counter()
for _ in range(10): s.run()
s.results()
dt = (time.perf_counter() - t) / 10
s.set_profiling(True)
for _ in range(3): s.run()
s.results()
kt = s.kernel_times()
print(sys.argv[3], "step %.2f ms" % (1e3 * dt), " ".join("%s %.2f" % (k, v[0] / 3) for k, v in sorted(kt.items()) if v[1]),
      hashlib.sha1(r.tobytes()).hexdigest()[:12], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000"
for path in ["-"] + sys.argv[2:]:
    subprocess.run([sys.executable, "-c", code, root, n, path], check=True)
