"""GPU box: A/B of environment-variable switches of the library on the bench workload, each setting in its own process
(the switches are read once). Usage: python tools/perf_env_ab.py [objects] VAR=VALUE [VAR=VALUE ...]   ("-" = no variable)"""
import os, subprocess, sys
code = r'''
import sys, hashlib, time
sys.path[:0] = [sys.argv[1]]
import outfit_amd
from outfit_amd import synth
n = int(sys.argv[2])
ed = synth.make_ephemeris(); ds = synth.make_dataset(n, ephem=ed)
s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed), synth.bench_params(outfit_amd.IODParams()))
s.run(); r = s.results()
for _ in range(3): s.run()
s.results()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(10): s.run()
    s.results()
    best = min(best, (time.perf_counter() - t) / 10)
s.set_profiling(True)
for _ in range(3): s.run()
s.results()
kt = s.kernel_times()
print(sys.argv[3], "step %.2f ms" % (1e3 * best), " ".join("%s %.2f" % (k.replace("k_", ""), v[0] / 3) for k, v in sorted(kt.items()) if v[1]),
      hashlib.sha1(r.tobytes()).hexdigest()[:12], flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
n = args.pop(0) if args and args[0].isdigit() else "100000"
for setting in args or ["-"]:
    env = dict(os.environ)
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            env[k] = v
    subprocess.run([sys.executable, "-c", code, root, n, setting], check=True, env=env)
