"""GPU box: where a 1 000-object step goes — per-kernel times for the bench parameters and with the root walk cut
to one tested solution (max_tested_solutions = 1), which tells how much of the general kernel's chain is the
second and third root of a candidate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import outfit_amd
from outfit_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds); ephem = outfit_amd.Ephemeris.from_dict(ed)
for label, kw in (("bench params", {}), ("max_tested_solutions=1", dict(max_tested_solutions=1)),
                  ("newton_max_it=10", dict(newton_max_it=10))):
    params = synth.bench_params(outfit_amd.IODParams())
    for k, v in kw.items():
        setattr(params, k, v)
    s = outfit_amd.IODSession(batch, ephem, params)
    for _ in range(3): s.run()
    s.results()
    t = time.perf_counter()
    for _ in range(20): s.run()
    s.results()
    dt = (time.perf_counter() - t) / 20
    s.set_profiling(True)
    for _ in range(5): s.run()
    s.results()
    kt = s.kernel_times()
    print(f"{label:26s} step {1e3 * dt:.2f} ms  " + " ".join("%s %.2f" % (k.replace('k_', ''), v[0] / 5) for k, v in sorted(kt.items()) if v[1]), flush=True)
