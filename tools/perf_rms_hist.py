"""GPU box: distribution of the candidates' screened RMS (bench workload) — how many candidates have residuals large enough
to leave a small-angle path of the screen (measurement aid)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, outfit_amd
from outfit_amd import synth
ed = synth.make_ephemeris(); ds = synth.make_dataset(20000, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); s.results()
tot, nw, fl = s.candidate_scores()
scored = ((fl & 0x10) != 0) & ((fl >> 8 & 0x7) == 0) & ((fl & 0x80) == 0) & ((fl & 0x80000000) == 0)
r = np.sqrt(tot[scored] / (2.0 * nw[scored]))
print("scored", scored.mean(), "rms quantiles", np.quantile(r[np.isfinite(r)], [0.1, 0.5, 0.9, 0.99, 0.999]))
for th in (1e2, 1e3, 1e4, 1e5, 2.6e5, 1e6):
    print("rms >", th, float((r > th).mean()))
