"""GPU box: k_diffcor time per 100k objects for a few configurations (HIP-event kernel times).
    python tools/perf_lsq.py [objects] [library to load instead of the shipped one]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
if len(sys.argv) > 2:
    import outfit_amd._lib as _L  # noqa: E402
    _L.LIB_PATH = sys.argv[2]
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ed = synth.make_ephemeris()
ds = synth.make_dataset(n, ephem=ed)
batch = outfit_amd.ObsBatch.from_dict(ds)
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
s = outfit_amd.IODSession(batch, ephem, params)
s.run()
s.results()
for name, kw in [("default", {}), ("one_iteration", dict(max_newton_iterations=1)),
                 ("ratio10", dict(rms_divergence_ratio=10.0)), ("no_divergence_check", dict(rms_divergence_ratio=1e300))]:
    cfg = outfit_amd.DifferentialCorrectionConfig(**kw)
    s.run_lsq(cfg)
    s.lsq_results()
    s.set_profiling(True)
    for _ in range(3):
        s.run_lsq(cfg)
    t = s.kernel_times()["k_diffcor"]
    s.set_profiling(False)
    r, _ = s.lsq_results()
    it = r["total_newton_iterations"]
    # lane efficiency if a wave = 64 consecutive trajectories runs until its slowest lane
    w = it[: len(it) // 64 * 64].reshape(-1, 64)
    print(f"{name:22s} {t[0] / t[1]:8.3f} ms  mean iters {it.mean():.2f}  max {it.max()}  "
          f"wave-max mean {w.max(1).mean():.2f}  lane efficiency {it.mean() / w.max(1).mean():.2f}  "
          f"converged {(r['status'] == 0).sum()}")
    if name == "default":
        print("   iterations histogram:", np.bincount(it)[:40].tolist())
