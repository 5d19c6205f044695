"""GPU box: K0 (triplet selection) event time with one slice, and the isolated run, at two batch sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

ed = synth.make_ephemeris()
for n in (1000, 100000):
    ds = synth.make_dataset(n, ephem=ed)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    ephem = outfit_amd.Ephemeris.from_dict(ed)
    params = synth.bench_params(outfit_amd.IODParams())
    for slices in ("1", "4"):
        os.environ["OUTFIT_K0_SLICES"] = slices
        s = outfit_amd.IODSession(batch, ephem, params)
        s.run(); r = s.results()
        s.set_profiling(True)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            s.run(); s.results()
            ts.append(1e3 * (time.perf_counter() - t))
        kt = s.kernel_times()
        s.close()
        import hashlib
        print(f"n {n} slices {slices}: K0 {kt['k_triplet_select'][0] / 5:.3f} ms, isolated run {min(ts):.2f} ms, "
              f"general {kt['k_gauss_correct'][0] / 5:.2f} ms  {hashlib.sha1(r.tobytes()).hexdigest()[:8]}", flush=True)
