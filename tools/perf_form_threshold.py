"""GPU box: step time by batch size with the latency forms (lane pairs, root per lane, 16-lane re-evaluation) forced on and
off — where the threshold between the two families of kernels belongs (measurement aid; OUTFIT_PAIR_MODE is read by the
launchers at every launch).
    python tools/perf_form_threshold.py [objects ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd
from outfit_amd import synth

sizes = [int(x) for x in sys.argv[1:]] or [250, 500, 1000, 1300, 2000, 3000, 5000, 10000]
ed = synth.make_ephemeris(); ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
for n in sizes:
    batch = outfit_amd.ObsBatch.from_dict(synth.make_dataset(n, ephem=ed))
    row = []
    for mode in ("0", "1"):
        os.environ["OUTFIT_PAIR_MODE"] = mode
        s = outfit_amd.IODSession(batch, ephem, params)
        for _ in range(3): s.run()
        s.results()
        t = time.perf_counter()
        for _ in range(20): s.run()
        s.results()
        row.append(1e3 * (time.perf_counter() - t) / 20)
        s.close()
    print(f"{n:6d} objects ({n * 100:8d} candidates): throughput forms {row[0]:.3f} ms, latency forms {row[1]:.3f} ms", flush=True)
