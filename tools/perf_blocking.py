"""GPU box: wall time of the BLOCKING entry point outfit_hip_fit_full_iod (host buffers in, records out; what a
host program without a resident session calls), repeated, against the resident-session step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ed = synth.make_ephemeris()
ds = synth.make_dataset(n, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
for i in range(4):
    t = time.perf_counter()
    r = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
    dt = time.perf_counter() - t
    print(f"blocking call {i}: {1e3 * dt:.1f} ms = {n / dt:.3e} objects/s ({(r['status'] == 0).sum()} orbits)", flush=True)
t = time.perf_counter()
s = outfit_amd.IODSession(batch, ephem, params)
t1 = time.perf_counter()
s.run(); s.results()
t2 = time.perf_counter()
s.run(); s.results()
t3 = time.perf_counter()
print(f"session: build {1e3 * (t1 - t):.1f} ms, first run {1e3 * (t2 - t1):.1f} ms, second run {1e3 * (t3 - t2):.1f} ms")
