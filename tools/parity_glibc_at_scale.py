"""GPU box: the device in OUTFIT_MATH_GLIBC mode against the oracle in platform-libm mode (= the reference's
arithmetic on this host's glibc) on a LARGE batch — statistics for DESIGN.md §4.0. Also the default mode against the
same, and the glibc mode against the oracle's restated mode (must be bit-identical).
usage: parity_glibc_at_scale.py [n_objects] [seed] [bench|topo|noise|long100|long300|long100k100|long300k100]  -> gpurun_out/parity_glibc_<mode>_<n>.json
(noise: IODParams::default() — 10 triplets x 21 realisations, host-drawn normals, SURVEY 8(d)'s second point;
long100 / long300: trajectories of 100 / 300 observations (synth.make_long_arc_dataset) under IODParams::default() with
noise — the wave-per-trajectory K0, down-sampling and quirk Q1, RMS windows over the whole arc; long<N>k100: the same
trajectories with max_triplets = 100 and no noise — K0's heap of 100 with ~1 300 replacements per trajectory)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402
from outfit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
mode = sys.argv[3] if len(sys.argv) > 3 else "bench"
threads = len(os.sched_getaffinity(0))
k100 = mode.endswith("k100")
if mode.startswith("long"):
    ds = synth.make_long_arc_dataset(n, int(mode[4:-4] if k100 else mode[4:]), seed=seed)
    ed = ds["ephem"]
else:
    ed = synth.make_ephemeris()
    ds = synth.make_dataset(n, seed=seed, ephem=ed, topocentric=(mode == "topo"))
batch, ephem = outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed)
noisy = (mode == "noise" or mode.startswith("long")) and not k100
params = outfit_amd.IODParams() if noisy else \
    outfit_amd.IODParams(n_noise_realizations=0, max_triplets=100) if k100 else synth.bench_params(outfit_amd.IODParams())
per_traj = params.max_triplets * params.n_noise_realizations * 6
if noisy:
    batch.noise_z = np.random.default_rng(seed).standard_normal(n * per_traj)
L = O.lib()


def oracle(libm, glibc):
    f1, f2 = C.c_int.in_dll(L, "orc_use_libm"), C.c_int.in_dll(L, "orc_math_glibc")
    f1.value, f2.value = libm, glibc
    ref = np.empty(n, dtype=outfit_amd._abi.RESULT_DTYPE)
    t = time.perf_counter()
    try:
        for i in range(0, n, 10000):
            j = min(n, i + 10000)
            sub = batch.slice(i, j)
            if noisy:
                sub.noise_z = batch.noise_z[i * per_traj: j * per_traj]
            rc, ref[i:j] = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), n_threads=threads, hoist=1)
            assert rc == 0
            print(f"oracle(libm={libm}, glibc={glibc}) {j}/{n} ({time.perf_counter() - t:.0f} s)", flush=True)
    finally:
        f1.value, f2.value = 0, 0
    return ref


def distance(got, ref):
    ints = ("status", "cause", "stage", "kind", "attempts", "n_triplets", "best_triplet", "best_realization")
    same = np.ones(n, dtype=bool)
    for k in ints:
        same &= got[k] == ref[k]
    ok = same & (ref["status"] == 0)
    d = np.abs(got["elem"][ok] - ref["elem"][ok])
    d[:, 2:] = np.minimum(d[:, 2:], np.abs(2 * np.pi - d[:, 2:]))
    dl, da = d[:, :2].max(1), d[:, 2:].max(1)
    return {"orbits": int((ref["status"] == 0).sum()), "different_integer_outputs": int((~same).sum()),
            "byte_identical_records": int(sum(got[i].tobytes() == ref[i].tobytes() for i in range(n))),
            "bit_identical_rms": int((got["rms"][ok] == ref["rms"][ok]).sum()),
            "bit_identical": int((got["elem"][ok] == ref["elem"][ok]).all(1).sum()),
            "beyond_1e-10_AU": int((dl > 1e-10).sum()), "beyond_1e-12_rad": int((da > 1e-12).sum()),
            "max_len": float(dl.max()), "max_ang": float(da.max())}


dev_g = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1, math_mode=outfit_amd.MATH_GLIBC)
dev_c = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1, math_mode=outfit_amd.MATH_CORRECTLY_ROUNDED)
ref_libm = oracle(1, 0)
ref_rest = oracle(0, 1)
from outfit_amd import libm_pin  # noqa: E402
out = {"objects": n, "seed": seed, "mode": mode, "glibc": os.confstr("CS_GNU_LIBC_VERSION"), "host_libm": libm_pin.describe(),
       "device_glibc_vs_oracle_libm": distance(dev_g, ref_libm),
       "device_cr_vs_oracle_libm": distance(dev_c, ref_libm),
       "device_glibc_vs_oracle_restated": distance(dev_g, ref_rest)}
print(json.dumps(out, indent=1), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"parity_glibc_{mode}_{n}.json"), "w"), indent=1)
