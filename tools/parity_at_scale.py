"""GPU box: parity of the HIP path against the CPU oracle on a LARGE batch (default 50 000 objects of the bench
workload, a different seed from the tests) — statistics, not asserts: how many trajectories select a different
candidate, and the distribution of the element differences on the rest. The oracle runs on all host cores.
Writes gpurun_out/parity_at_scale.json."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
mode = sys.argv[3] if len(sys.argv) > 3 else "bench"   # bench | noise (IODParams::default(), host normals) | topo
threads = len(os.sched_getaffinity(0))
ed = synth.make_ephemeris()
ds = synth.make_dataset(n, seed=seed, ephem=ed, topocentric=(mode == "topo"))
ephem = outfit_amd.Ephemeris.from_dict(ed)
params = synth.bench_params(outfit_amd.IODParams())
stride = 0
if mode == "noise":
    params = outfit_amd.IODParams()
    stride = params.max_triplets * params.n_noise_realizations * 6
    ds["noise_z"] = synth.SplitMix64(seed + 1).normal(n * stride)
batch = outfit_amd.ObsBatch.from_dict(ds)


def sliced(i, j):
    sub = batch.slice(i, j)
    if stride:
        sub.noise_z = batch.noise_z[i * stride:j * stride]
    return sub


cfg = outfit_amd.DifferentialCorrectionConfig()

t = time.perf_counter()
got = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
lsq, lsq_obs = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, got, n_devices=1)
t_gpu = time.perf_counter() - t
print(f"GPU: IOD + corrector on {n} objects in {t_gpu:.2f} s (host buffers, session build included)", flush=True)

t = time.perf_counter()
ref = np.empty(n, dtype=got.dtype)
step = 5000
for i in range(0, n, step):  # in slices so that a progress line appears every few seconds
    j = min(n, i + step)
    sub = sliced(i, j)
    rc, ref[i:j] = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), n_threads=threads, hoist=1)
    assert rc == 0
    print(f"oracle IOD {j}/{n} ({time.perf_counter() - t:.0f} s)", flush=True)
t_cpu = time.perf_counter() - t

ints = ("status", "cause", "stage", "kind", "attempts", "n_triplets", "best_triplet", "best_realization")
same = np.ones(n, dtype=bool)
for k in ints:
    same &= got[k] == ref[k]
ok = same & (ref["status"] == 0)
d = np.abs(got["elem"][ok] - ref["elem"][ok])
d[:, 2:] = np.minimum(d[:, 2:], np.abs(2 * np.pi - d[:, 2:]))
rms_rel = np.abs(got["rms"][ok] - ref["rms"][ok]) / np.maximum(1.0, ref["rms"][ok])
out = {
    "objects": n, "seed": seed, "mode": mode, "oracle_threads": threads, "oracle_seconds": t_cpu, "gpu_seconds_incl_upload": t_gpu,
    "iod": {
        "orbits_in_oracle": int((ref["status"] == 0).sum()),
        "different_integer_outputs": int((~same).sum()),
        "different_rows": np.nonzero(~same)[0][:20].tolist(),
        "bit_identical_records": int(sum(got[i].tobytes() == ref[i].tobytes() for i in range(n))),
        "max_abs_diff_a_e": float(d[:, :2].max()), "max_abs_diff_angles": float(d[:, 2:].max()),
        "elements_bit_identical": int((got["elem"][ok] == ref["elem"][ok]).all(1).sum()),
        "max_rel_diff_rms": float(rms_rel.max()), "epoch_identical": bool((got["epoch"][ok] == ref["epoch"][ok]).all()),
    },
}
print(json.dumps(out["iod"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_at_scale_iod.npz"), got=got, ref=ref)

# the corrector on the SAME seeds (the oracle's records where both agree; else parity of the IOD is what differs)
t = time.perf_counter()
lref = np.empty(n, dtype=lsq.dtype)
for i in range(0, n, 4 * step):
    j = min(n, i + 4 * step)
    sub = sliced(i, j)
    rc, lref[i:j], _ = O.fit_lsq(params.as_c(), cfg.as_c(), sub.as_c(), ephem.as_c(), got[i:j], n_threads=threads)
    assert rc == 0
    print(f"oracle LSQ {j}/{n} ({time.perf_counter() - t:.0f} s)", flush=True)
lints = ("status", "dc_error", "total_newton_iterations", "num_measurements", "n_rejected")
lsame = np.ones(n, dtype=bool)
for k in lints:
    lsame &= lsq[k] == lref[k]
conv = lsame & (lref["status"] == 0)
sig = np.sqrt(np.abs(np.diagonal(lref["covariance"][conv], axis1=1, axis2=2)))
dl = np.abs(lsq["elem"][conv] - lref["elem"][conv])
dl[:, 5] = np.minimum(dl[:, 5], np.abs(2 * np.pi - dl[:, 5]))
excess = dl - (np.array([1e-10] * 5 + [1e-12])[None, :] + 1e-7 * sig)
out["lsq"] = {
    "converged_in_oracle": int((lref["status"] == 0).sum()), "kept_iod": int((lref["status"] == 1).sum()),
    "different_integer_outputs": int((~lsame).sum()), "different_rows": np.nonzero(~lsame)[0][:20].tolist(),
    "max_abs_element_diff": float(dl.max()), "max_diff_in_sigma": float((dl / np.maximum(sig, 1e-300)).max()),
    "outside_test_tolerance": int((excess > 0).any(1).sum()),
    "max_rel_diff_normalised_rms": float((np.abs(lsq["normalised_rms"][conv] - lref["normalised_rms"][conv])
                                          / np.maximum(1.0, lref["normalised_rms"][conv])).max()),
}
print(json.dumps(out["lsq"]), flush=True)
bad = np.nonzero(~lsame)[0]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_at_scale_lsq.npz"), rows=bad, got=lsq[bad], ref=lref[bad])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"parity_at_scale_{mode}.json"), "w"), indent=1)
