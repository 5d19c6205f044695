"""GPU box: k_diffcor per launch for the lane-group sizes (OUTFIT_DIFFCOR_GROUP: auto, 1, 2, 4, 8), each in its own
process; the digest of the LSQ records must not depend on the group (measurement aid).
    python tools/perf_lsq_group.py [objects]"""
import os, subprocess, sys
code = r'''
import sys, hashlib, os
sys.path[:0] = [sys.argv[1]]
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[2])
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); s.results()
cfg = outfit_amd.DifferentialCorrectionConfig()
s.run_lsq(cfg); r, _ = s.lsq_results()
s.set_profiling(True)
for _ in range(5): s.run_lsq(cfg)
t = s.kernel_times()["k_diffcor"]
print("group", os.environ.get("OUTFIT_DIFFCOR_GROUP", "auto"), "k_diffcor %.3f ms" % (t[0] / t[1]), hashlib.sha1(r.tobytes()).hexdigest()[:12], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000"
for g in ("auto", "1", "2", "4", "8"):
    env = dict(os.environ)
    if g != "auto":
        env["OUTFIT_DIFFCOR_GROUP"] = g
    subprocess.run([sys.executable, "-c", code, root, n], env=env, check=True)
