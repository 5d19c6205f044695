"""GPU box: the RMS stage in the lane-per-candidate mapping (default) and in the wave-per-candidate mapping that
BASELINE.json's north_star names (OUTFIT_RMS_WAVE=1, k_orbit_rms_wave): same bytes, measured time."""
import os, subprocess, sys
code = r'''
import os, sys, time, hashlib
sys.path[:0] = [sys.argv[1]]
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[2])
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds); ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
s = outfit_amd.IODSession(batch, ephem, params)
s.run(); r = s.results()
s.set_profiling(True)
t = time.perf_counter()
for _ in range(3): s.run()
s.results()
dt = (time.perf_counter() - t) / 3
kt = s.kernel_times()
print("wave-per-candidate" if os.environ.get("OUTFIT_RMS_WAVE") else "lane-per-candidate", n, "objects: step %.2f ms, k_orbit_rms %.2f ms" % (1e3 * dt, kt["k_orbit_rms"][0] / 3), hashlib.sha1(r.tobytes()).hexdigest()[:10], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000"
for env in ({}, {"OUTFIT_RMS_WAVE": "1"}):
    subprocess.run([sys.executable, "-c", code, root, n], env=dict(os.environ, **env), check=True)
