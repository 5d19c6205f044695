"""GPU box: IOD with IODParams::default() (10 triplets x (1 + 20 noisy realisations)) and device-side draws."""
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
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = outfit_amd.IODParams()
batch = outfit_amd.ObsBatch.from_dict(ds)
batch.noise_seed = 42
s = outfit_amd.IODSession(batch, ephem, params)
s.run()
r = s.results()
s.set_profiling(True)
t = time.perf_counter()
for _ in range(5):
    s.run()
r = s.results()
dt = (time.perf_counter() - t) / 5
print({k: round(v[0] / 5, 2) for k, v in s.kernel_times().items()})
s.set_profiling(False)
print(f"{n} objects, default params, device noise: {1e3 * dt:.2f} ms/step = {n / dt:.3e} objects/s = "
      f"{n * 210 / dt:.3e} candidates/s; orbits {(r['status'] == 0).sum()}, noisy winner {(r['best_realization'] > 0).mean():.2f}")
