"""GPU box: does splitting a resident batch over k concurrently running sessions (each on its own streams) fill
the tails between the kernels of one session? Same 100 000 objects, k = 1, 2, 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import outfit_amd
from outfit_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
full = outfit_amd.ObsBatch.from_dict(ds)
ref = None
for k in (1, 2, 3, 1, 2, 4):
    cuts = [n * i // k for i in range(k + 1)]
    sessions = [outfit_amd.IODSession(full.slice(cuts[i], cuts[i + 1]), ephem, params) for i in range(k)]
    for _ in range(3):
        for s in sessions: s.run()
    out = np.concatenate([s.results() for s in sessions])
    if ref is None: ref = out
    assert out.tobytes() == ref.tobytes()
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        for s in sessions: s.run()
    for s in sessions: s.results()
    dt = (time.perf_counter() - t) / reps
    print(f"{k} concurrent session(s): {1e3 * dt:.2f} ms per {n} objects = {n / dt:.4g} objects/s", flush=True)
    del sessions
