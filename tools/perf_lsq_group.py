"""GPU box: k_diffcor per launch for the lane-group sizes (OUTFIT_DIFFCOR_GROUP), each in its own process; the digest of the
LSQ records must not depend on the group (measurement aid)."""
import os, subprocess, sys
code = r'''
import sys, time, hashlib
sys.path[:0] = [sys.argv[1]]
import outfit_amd
from outfit_amd import synth
ed = synth.make_ephemeris(); ds = synth.make_dataset(100000, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
cfg = outfit_amd.DifferentialCorrectionConfig()
s.run(); s.run_lsq(cfg); r,_ = s.lsq_results()
s.set_profiling(True)
for _ in range(5): s.run_lsq(cfg)
r,_ = s.lsq_results()
kt = s.kernel_times()
print("group", sys.argv[2], "k_diffcor %.3f ms per launch" % (kt["k_diffcor"][0] / kt["k_diffcor"][1]), hashlib.sha1(r.tobytes()).hexdigest()[:10], flush=True)
'''
for g in ("1","2","4","8"):
    subprocess.run([sys.executable, "-c", code, "/root/repo", g], env=dict(os.environ, OUTFIT_DIFFCOR_GROUP=g), check=True)
