"""GPU box: long trajectories (bench.py's long-arc side workload) under environment switches, each setting in its own
process: step time back to back, K0's isolated time, digest of the records.
    python tools/perf_long_arcs.py [objects] [n_obs] VAR=VALUE[,VAR=VALUE] ...   ("-" = no variable)"""
import os, subprocess, sys
code = r'''
import sys, hashlib, time
sys.path[:0] = [sys.argv[1]]
import os
if os.environ.get("OUTFIT_PERF_LIB"):          # measurement aid: load another build of the library
    import outfit_amd._lib as _L
    _L.LIB_PATH = os.environ["OUTFIT_PERF_LIB"]
import outfit_amd
from outfit_amd import synth
n, n_obs = int(sys.argv[2]), int(sys.argv[3])
ds = synth.make_long_arc_dataset(n, n_obs)
ephem = outfit_amd.Ephemeris.from_dict(ds["ephem"])
batch = outfit_amd.ObsBatch.from_dict(ds)
for k, noise in ((100, 0), (10, 0)):
    p = outfit_amd.IODParams(n_noise_realizations=noise, max_triplets=k)
    s = outfit_amd.IODSession(batch, ephem, p)
    s.run(); r = s.results()
    for _ in range(2): s.run()
    s.results()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(5): s.run()
        s.results()
        best = min(best, (time.perf_counter() - t) / 5)
    s.set_profiling(True)
    s.run(); s.results()
    kt = s.kernel_times()
    s.close()
    print(sys.argv[4], f"K={k}", "step %.2f ms" % (1e3 * best), "K0 isolated %.2f ms" % kt["k_triplet_select"][0],
          hashlib.sha1(r.tobytes()).hexdigest()[:12], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
n = args.pop(0) if args and args[0].isdigit() else "10000"
n_obs = args.pop(0) if args and args[0].isdigit() else "100"
for setting in args or ["-"]:
    env = dict(os.environ)
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            env[k] = v
    subprocess.run([sys.executable, "-c", code, root, n, n_obs, setting], check=True, env=env)
