#!/usr/bin/env python3
"""bench.py — objects/s through the full Gauss IOD path on N MI355X (one process per GPU).

A "step" is one pass of the whole hot path (triplet selection, observer pre-pass,
Gauss candidates, best-orbit selection) over this rank's batch of synthetic
objects, inputs already resident in HBM, followed — for N > 1 — by the path's only
exchange step, an RCCL all_gather of the 96-byte result records.

    python bench.py                      # N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the contract in DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X_MICROARCH.md: fp64 vector, 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz
# the ONE fp64 issue peak used everywhere: a 64-lane fp64 VALU instruction occupies a SIMD's 16 lanes for 4 cycles,
# 1024 SIMDs at the 2.4 GHz peak engine clock -> wave-instructions per second
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0
BYTES_PER_OBS_IN = 64          # t, ra, dec, sigma_ra, sigma_dec, obs_helio_equ[3]   
BYTES_PER_OBS_GEO = 24         # obs_geo_ecl[3] when the Earth is evaluated on device
BYTES_PER_RESULT = 96          # outfit_gauss_result_t


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(batch, ephem, params, n_obj_max, budget_s=15.0, cores=None, hoist=0):
    """The CPU oracle (a restatement of the reference's Rayon path: one task per trajectory,
    dynamic scheduling) timed on the host cores, on a bounded prefix of the same workload.
    hoist=0: Earth re-interpolated per (candidate, observation) as the reference does; hoist=1: once per observation."""
    import ctypes
    import oracle_lib as O
    # as the reference runs: the platform libm, not the oracle's correctly-rounded (binary128) functions
    ctypes.c_int.in_dll(O.lib(), "orc_use_libm").value = 1
    cores = cores or host_cores()
    n0 = min(n_obj_max, 64 * cores)
    sub = batch.slice(0, n0)
    t = time.perf_counter()
    rc, _ = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), n_threads=cores, hoist=hoist)
    dt0 = time.perf_counter() - t
    assert rc == 0
    n1 = int(min(n_obj_max, max(n0, n0 * budget_s / max(dt0, 1e-3))))
    sub = batch.slice(0, n1)
    t = time.perf_counter()
    rc, _ = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), n_threads=cores, hoist=hoist)
    dt1 = time.perf_counter() - t
    assert rc == 0
    how = "Earth re-interpolated per (candidate, observation)" if not hoist else "Earth interpolated once per observation (hoisted)"
    return {"value": n1 / dt1, "unit": "objects/s", "cores": cores, "kind": "port",
            "sample": f"first {n1} objects of the same synthetic batch, {dt1:.1f} s, {how}, platform libm as the "
                      "reference"}


def host_libm():
    """What the oracle's `orc_use_libm = 1` mode evaluates with on this host (the reference's arithmetic when glibc)."""
    import platform
    name, ver = platform.libc_ver()
    return f"{name or 'unknown'} {ver} {platform.machine()}".strip()


def parity_vs_libm(batch, ephem, params, records_by_mode, n_sample):
    """Sample check of the timed batch against the reference-faithful arithmetic: the oracle with the PLATFORM libm
    (orc_use_libm = 1: bit-exact on every reference KAT on glibc) on the first n_sample objects, per device math mode.
    Tolerance = BASELINE.json north_star: 1e-10 AU (a, e), 1e-12 rad (i, Omega, omega, M)."""
    import numpy as np
    import oracle_lib as O
    sub = batch.slice(0, n_sample)
    t = time.perf_counter()
    with O.math_flags(libm=1):
        rc, ref = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), n_threads=host_cores(), hoist=1)
    assert rc == 0
    out = {"sample": f"first {n_sample} objects of the timed batch", "oracle": "CPU restatement, platform libm (orc_use_libm=1)",
           "host_libm": host_libm(), "oracle_seconds": time.perf_counter() - t,
           "tolerance": {"len_AU": 1e-10, "angle_rad": 1e-12}}
    ints = ("status", "cause", "stage", "kind", "attempts", "n_triplets", "best_triplet", "best_realization")
    ok = ref["status"] == 0
    for mode, recs in records_by_mode.items():
        g = recs[:n_sample]
        int_same = all(np.array_equal(g[k], ref[k]) for k in ints)
        dl = np.abs(g["elem"][ok][:, :2] - ref["elem"][ok][:, :2]).max(axis=1)
        da = np.abs(g["elem"][ok][:, 2:] - ref["elem"][ok][:, 2:])
        da = np.minimum(da, np.abs(2 * np.pi - da)).max(axis=1)
        rr = np.abs(g["rms"][ok] - ref["rms"][ok]) / np.maximum(1.0, ref["rms"][ok])
        out[mode] = {"orbits": int(ok.sum()), "integer_fields_identical": bool(int_same),
                     "bit_identical_elements": int((g["elem"][ok] == ref["elem"][ok]).all(axis=1).sum()),
                     "bit_identical_rms": int((g["rms"][ok] == ref["rms"][ok]).sum()),
                     "byte_identical_records": int(sum(g[i].tobytes() == ref[i].tobytes() for i in range(len(ref)))),
                     "beyond_1e-10_AU": int((dl > 1e-10).sum()), "beyond_1e-12_rad": int((da > 1e-12).sum()),
                     "max_d_len": float(dl.max()), "max_d_angle": float(da.max()), "max_rel_d_rms": float(rr.max())}
    return out


def noise_point(label, ds, ephem, params, device, stream, steps, seed):
    """SURVEY §8(d)'s second point: noisy realisations with HOST-drawn standard normals (gauss.rs:323-387,
    trajectory.rs:462-477), inputs and normals resident; one warm-up, `steps` timed steps, per-kernel HIP-event times."""
    import numpy as np
    import torch
    import outfit_amd
    n_obj = int(ds["n_traj"])
    n_z = n_obj * params.max_triplets * params.n_noise_realizations * 6
    z = np.empty(n_z, dtype=np.float64)
    rng = np.random.default_rng(seed)
    step_n = 1 << 26
    for a in range(0, n_z, step_n):      # in place, 512 MiB at a time
        rng.standard_normal(out=z[a:a + step_n])
    d = dict(ds)
    d["noise_z"] = z
    batch = outfit_amd.ObsBatch.from_dict(d)
    t = time.perf_counter()
    s = outfit_amd.IODSession(batch, ephem, params, device=device)
    s.run(stream=stream)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t
    s.set_profiling(True)
    t = time.perf_counter()
    for _ in range(steps):
        s.run(stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    kt = s.kernel_times()
    s.set_profiling(False)
    rec = s.results()
    s.close()
    cand = n_obj * params.max_triplets * (1 + params.n_noise_realizations)
    return {"what": label, "objects": n_obj, "max_triplets": params.max_triplets,
            "n_noise_realizations": params.n_noise_realizations, "candidates_per_step": cand,
            "normals": f"{n_z} host-drawn standard normals ({n_z * 8 / 1e9:.2f} GB, numpy PCG64), resident in HBM",
            "steps": steps, "ms_per_step": 1e3 * dt, "objects_per_s": n_obj / dt, "candidates_per_s": cand / dt,
            "kernel_avg_ms_per_step": {k: v[0] / steps for k, v in kt.items()},
            "orbits_found": int((rec["status"] == 0).sum()),
            "noisy_winners": int(((rec["status"] == 0) & (rec["best_realization"] > 0)).sum()),
            "session_build_and_first_run_s": setup}


def source_hash():
    """Content hash of the kernel sources: ties a committed rocprof/PMC profile to the build it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "outfit_amd", "csrc")
    # device code only: the host layer (api.hip, ephem_file.hip) changes no kernel
    for f in sorted([os.path.join(base, "kernels.hip"), os.path.join(base, "diffcor.hip"), os.path.join(base, "kernels.h")] +
                    glob.glob(os.path.join(base, "device", "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline_lsq(batch, ephem, params, seeds, n_obj_max, budget_s=5.0):
    """The oracle's differential correction (the reference loops over trajectories sequentially,
    obs_dataset_api.rs:76-96; threads here = the same dynamic schedule as the IOD baseline) on given seeds."""
    import ctypes
    import oracle_lib as O
    ctypes.c_int.in_dll(O.lib(), "orc_use_libm").value = 1  # platform libm, as the reference runs
    cores = host_cores()
    cfg = O.default_diffcor_config()
    n0 = min(n_obj_max, 2048 * cores)
    sub = batch.slice(0, n0)
    t = time.perf_counter()
    rc, _, _ = O.fit_lsq(params.as_c(), cfg, sub.as_c(), ephem.as_c(), seeds[:n0], n_threads=cores)
    dt = time.perf_counter() - t
    assert rc == 0
    return {"value": n0 / dt, "unit": "objects/s", "cores": cores, "kind": "port",
            "sample": f"first {n0} objects of the same batch, seeds = this run's IOD records, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--objects", type=int, default=100000, help="objects per GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --objects per GPU (default); strong: --total-objects split over the GPUs (BASELINE configs[3])")
    ap.add_argument("--total-objects", type=int, default=1_000_000, help="objects of the whole job for --scaling strong")
    ap.add_argument("--no-noise", action="store_true", help="skip the noisy-realisation side measurements (SURVEY 8d)")
    ap.add_argument("--no-modes", action="store_true", help="skip timing the other math mode and the parity sample")
    ap.add_argument("--parity-sample", type=int, default=2000, help="objects of the timed batch checked against the libm oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lsq", action="store_true", help="skip the differential-correction side measurement")
    ap.add_argument("--no-side", action="store_true", help="skip the isolated-run / blocking-call side measurements")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real path) or gloo (rehearsal of the N>1 logic on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: create the process group and run the gather path even with one rank "
                         "(exercises the RCCL calls and their stream ordering on a one-GPU box)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import outfit_amd
    from outfit_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run "
                         "--nproc-per-node N for --gpus N (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.single_device:
        assert args.backend == "gloo", "--single-device is a rehearsal mode: use --backend gloo"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.force_dist   # the N > 1 code path (gather, barrier, max over ranks)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its communicator is created; stdout must carry the one JSON
        # line only, so the native stdout is pointed at stderr while the group and its first collective come up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend="gloo")
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # ---- this rank's shard of the synthetic dataset ----
    ephem_d = synth.make_ephemeris()
    if args.scaling == "weak":        # fixed objects per GPU, another seed per rank
        n_obj = args.objects
        counts = [n_obj] * world
        ds = synth.make_dataset(n_obj, seed=synth.DEFAULT_SEED + rank, ephem=ephem_d)
    else:                             # one global dataset of 25 000-object blocks (block b = seed + b), contiguous ranges
        blk = 25000
        n_blocks = max(world, -(-args.total_objects // blk))
        cuts = [n_blocks * r // world for r in range(world + 1)]
        counts = [(cuts[r + 1] - cuts[r]) * blk for r in range(world)]
        n_obj = counts[rank]
        ds = synth.make_dataset_blocks(n_obj, block=blk, seed=synth.DEFAULT_SEED + cuts[rank], ephem=ephem_d)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    # every rank knows everybody's record count: `counts` follows the partition rule of the library and dist.py
    # (contiguous ranges balanced by observation count — all trajectories have 20 observations here)
    from outfit_amd import dist as odist
    uniform = np.arange(0, sum(counts) + 1, dtype=np.uint64) * np.uint64(20)
    if len(set(counts)) == 1:
        assert odist.partition_by_observations(uniform, world) == [counts[0] * r for r in range(world + 1)]
    ephem = outfit_amd.Ephemeris.from_dict(ephem_d)
    params = synth.bench_params(outfit_amd.IODParams())
    n_obs = int(batch.traj_offset[-1])
    n_max = max(counts)

    # the headline runs in the library's default math mode (OUTFIT_MATH_GLIBC: the reference-faithful one, §4.0)
    mode_names = {outfit_amd.MATH_GLIBC: "glibc", outfit_amd.MATH_CORRECTLY_ROUNDED: "cr"}
    head_mode = outfit_amd.get_math_mode()
    session = outfit_amd.IODSession(batch, ephem, params, device=local_rank, math_mode=head_mode)
    # result buffers are double-buffered so that the gather of step i (RCCL, its own stream) overlaps the kernels
    # of step i+1; a buffer is reused only after the gather that read it has been waited for
    # (n_max rows: ranks may own different counts under --scaling strong; the gather needs equal shapes)
    out_local = [torch.zeros((n_max, BYTES_PER_RESULT), dtype=torch.uint8, device="cuda") for _ in range(2)]
    gather_dev = "cuda" if args.backend == "nccl" else "cpu"
    out_all = [torch.empty((world * n_max, BYTES_PER_RESULT), dtype=torch.uint8, device=gather_dev) for _ in range(2)] \
        if multi else None
    # An explicit torch stream: torch's default stream has handle 0, which the C ABI reads as "the session's own
    # stream" — the collective (RCCL syncs with torch's CURRENT stream) would then not wait for the kernels.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    pending = [None, None]
    state = {"step": 0}

    def step(sess=None):
        b = state["step"] & 1
        state["step"] += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        (sess or session).run(stream=stream, d_out=out_local[b].data_ptr())
        if multi:
            if args.backend == "nccl":
                pending[b] = dist.all_gather_into_tensor(out_all[b], out_local[b], async_op=True)
            else:  # rehearsal: gloo gathers host copies
                pending[b] = dist.all_gather_into_tensor(out_all[b], out_local[b].cpu(), async_op=True)

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(sess):
        """W warm-up steps, then exactly K steps between fences; (seconds, {kernel: (ms, launches)})."""
        for _ in range(args.warmup):
            step(sess)
        fence()
        sess.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(sess)
        fence()
        dt = time.perf_counter() - t0
        kt = sess.kernel_times()     # HIP events on the run's stream over the whole timed region
        sess.set_profiling(False)
        sess.check()                 # an epoch outside the ephemeris table is an error here as in the blocking API
        return dt, kt

    elapsed, ktimes = timed(session)
    last_b = (state["step"] - 1) & 1
    res = np.frombuffer(out_local[last_b][:n_obj].cpu().numpy().tobytes(), dtype=outfit_amd._abi.RESULT_DTYPE).copy()
    gathered = out_all[last_b].cpu().numpy().copy() if multi else None

    # ---- the other math mode, timed the same way on the same resident batch (never `value`), and a sample of the
    # timed batch against the reference-faithful arithmetic (the oracle with the platform libm) for both modes
    modes = None
    if rank == 0 and world == 1 and not args.no_modes:
        other = outfit_amd.MATH_CORRECTLY_ROUNDED if head_mode == outfit_amd.MATH_GLIBC else outfit_amd.MATH_GLIBC
        s2 = outfit_amd.IODSession(batch, ephem, params, device=local_rank, math_mode=other)
        el2, kt2 = timed(s2)
        res2 = np.frombuffer(out_local[(state["step"] - 1) & 1][:n_obj].cpu().numpy().tobytes(),
                             dtype=outfit_amd._abi.RESULT_DTYPE).copy()
        s2.close()
        def mode_line(el, kt):
            return {"value": n_obj * args.steps / el, "unit": "objects/s", "ms_per_step": 1e3 * el / args.steps,
                    "kernel_avg_ms_per_step": {k: v[0] / args.steps for k, v in kt.items()}}
        modes = {"headline": mode_names[head_mode],
                 mode_names[head_mode]: mode_line(elapsed, ktimes), mode_names[other]: mode_line(el2, kt2)}
        modes["ratio_other_over_headline"] = modes[mode_names[other]]["value"] / modes[mode_names[head_mode]]["value"]
        if args.parity_sample > 0 and not args.no_cpu_baseline:
            modes["parity_vs_libm"] = parity_vs_libm(batch, ephem, params,
                                                     {mode_names[head_mode]: res, mode_names[other]: res2},
                                                     min(args.parity_sample, n_obj))

    # ---- beside the headline: one isolated run (nothing of a previous run to hide K0 under) and the blocking
    # drop-in call from host buffers (PCIe-inclusive: uploads, kernels, records back). Never `value`.
    side = None
    if rank == 0 and world == 1 and not args.no_side:   # single-GPU properties; never measured inside an N > 1 job
        torch.cuda.synchronize()
        iso = []
        for _ in range(3):
            t = time.perf_counter()
            session.run(stream=stream)
            torch.cuda.synchronize()
            iso.append(time.perf_counter() - t)
        blk = []
        for _ in range(4):
            t = time.perf_counter()
            outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
            blk.append(time.perf_counter() - t)
        # BASELINE configs[1] (1 000 objects): latency bound — 10^5 candidates do not fill 256 CUs
        ds1 = synth.make_dataset(1000, seed=synth.DEFAULT_SEED, ephem=ephem_d)
        b1 = outfit_amd.ObsBatch.from_dict(ds1)
        s1 = outfit_amd.IODSession(b1, ephem, params, device=local_rank)
        s1.run(stream=stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            s1.run(stream=stream)
        torch.cuda.synchronize()
        small_step = (time.perf_counter() - t) / 20
        s1.close()
        small_blk = []
        for _ in range(4):
            t = time.perf_counter()
            outfit_amd.fit_full_iod_records(b1, ephem, params, n_devices=1)
            small_blk.append(time.perf_counter() - t)
        side = {"isolated_run_ms": 1e3 * min(iso), "blocking_call_ms": 1e3 * min(blk[1:]),
                "configs1_1000_objects": {"ms_per_step_back_to_back": 1e3 * small_step,
                                          "objects_per_s": 1000 / small_step,
                                          "blocking_call_ms": 1e3 * min(small_blk[1:])},
                "blocking_call_first_ms": 1e3 * blk[0],
                "blocking_call_objects_per_s": n_obj / min(blk[1:]),
                "what": "isolated_run = one session run on resident inputs with the device idle before it; "
                        "blocking_call = outfit_hip_fit_full_iod from host buffers (upload 1 856 B/object, kernels, "
                        "96-B records back; steady state after the first call, which also creates streams and scratch)"}
        # SURVEY §8(d)'s second reported point: noisy realisations with host-drawn normals, on the same objects
        if not args.no_noise:
            p_def = outfit_amd.IODParams()                 # IODParams::default(): 10 triplets x (1 + 20) realisations
            p_8d = synth.bench_params(outfit_amd.IODParams())
            p_8d.n_noise_realizations = 10                 # the §8(d) shape: 100 triplets x (1 + 10)
            nsteps = max(2, min(args.steps, 5))
            side["noise"] = {
                "iod_params_default": noise_point("IODParams::default(): max_triplets 10, n_noise_realizations 20, "
                                                  "max_obs_for_triplets 100 (mod.rs:308-344)", ds, ephem, p_def,
                                                  local_rank, stream, nsteps, 20261004),
                "survey_8d": noise_point("bench params with n_noise_realizations = 10 (SURVEY 8d: 100 triplets x 11)",
                                         ds, ephem, p_8d, local_rank, stream, nsteps, 20261005)}

    # ---- next row (BASELINE configs[4]): IOD seed -> differential correction ----
    # Measured after the headline region and reported beside it, never mixed into `value`.
    lsq = None
    if not args.no_lsq:
        cfg = outfit_amd.DifferentialCorrectionConfig()
        session.run(stream=stream)            # the seeds: this session's own IOD records
        session.run_lsq(cfg, stream=stream)   # warm-up (allocates the corrector's scratch)
        fence()
        session.set_profiling(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            session.run_lsq(cfg, stream=stream)
        fence()
        lsq_elapsed = time.perf_counter() - t1
        kt = session.kernel_times()
        session.set_profiling(False)
        lrec, _ = session.lsq_results()
        lsq = {"elapsed": lsq_elapsed, "kernel_ms": kt.get("k_diffcor", (0.0, 1)),
               "n_diffcor": int((lrec["status"] == 0).sum()), "n_iod_kept": int((lrec["status"] == 1).sum()),
               "mean_newton_iterations": float(lrec["total_newton_iterations"].mean())}

    gather_check = None
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # EVERY rank checks the WHOLE gathered table of the last timed step: rank r's rows must hash to the checksum
        # rank r computed of its own local records (exchanged with an all_gather), in the partition's order and counts
        import zlib
        mine = zlib.crc32(res.tobytes()) & 0xFFFFFFFF
        sums = torch.zeros(world, dtype=torch.int64, device=gather_dev)
        dist.all_gather_into_tensor(sums, torch.tensor([mine], dtype=torch.int64, device=gather_dev))
        sums = [int(x) for x in sums.cpu()]
        bad = []
        for r in range(world):
            rows = gathered[r * n_max: r * n_max + counts[r]]
            if (zlib.crc32(rows.tobytes()) & 0xFFFFFFFF) != sums[r]:
                bad.append(r)
        assert not bad, f"rank {rank}: gathered records of rank(s) {bad} differ from what those ranks computed"
        assert gathered[rank * n_max: rank * n_max + n_obj].tobytes() == res.tobytes()
        gather_check = {"ranks_verified_by_every_rank": world, "records": int(sum(counts)), "crc32_per_rank": sums}
    n_ok = int((res["status"] == 0).sum())

    if rank == 0:
        total_objects = sum(counts)
        value = total_objects * args.steps / elapsed
        # dominant kernel and its roofline (algorithmic bytes, SURVEY §8d / DESIGN.md)
        # ... among the kernels of the critical path: K0, K1 and the general correction kernel run on side
        # streams in the shadow of these and their event times include waiting for a free CU
        main_stream = [k for k in ktimes if k in ("k_gauss_roots", "k_gauss_correct_fast",
                                                    "k_orbit_rms", "k_orbit_rms_slow", "k_select_best")]
        dom = max(main_stream, key=lambda k: ktimes[k][0]) if main_stream else None
        bytes_per_obj = (n_obs / n_obj) * (BYTES_PER_OBS_IN + BYTES_PER_OBS_GEO) + BYTES_PER_RESULT
        roof = None
        if dom:
            total_ms, launches = ktimes[dom]
            avg_ms = total_ms / launches
            units = n_obj * args.steps / launches          # objects one launch processes (rank 0's shard)
            achieved = units * bytes_per_obj / (avg_ms * 1e-3) / 1e9
            # HBM bytes of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
            # separate runs; profiles/), scaled to one launch; null when no profile matches
            # Counter-derived fields come from the committed rocprofv3 PMC passes (separate runs, as the guide
            # prescribes) and are reported ONLY when that profile was taken from this very build (content hash of the
            # kernel sources); otherwise null — a stale profile says nothing about this code.
            traffic = None
            valu = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
                if pmc.get("source_hash") != source_hash():
                    raise KeyError("profile taken from another build")
                key = dom.replace("k_orbit_rms_slow", "k_orbit_rms<true>").replace("k_orbit_rms", "k_orbit_rms<false>") \
                    if dom.startswith("k_orbit_rms") else dom
                key = {"k_triplet_select": "k_triplet_select_lds"}.get(key, key)
                scale = (n_obj / pmc["objects_per_step"]) * args.steps / launches
                traffic = pmc["kernels"][key]["hbm_bytes_per_step"] * scale
                instr = pmc["kernels"][key]["valu_wave_instr"] * scale
                step_instr = sum(k["valu_wave_instr"] for n, k in pmc["kernels"].items() if not n.startswith("k_diffcor")) \
                    * (n_obj / pmc["objects_per_step"])
                valu = {"wave_instr_per_launch": instr, "issue_frac": instr / VALU_ISSUE_PEAK / (avg_ms * 1e-3),
                        "step_wave_instr": step_instr,
                        "step_issue_frac": step_instr / VALU_ISSUE_PEAK / (elapsed / args.steps),
                        "peak_wave_instr_per_s": VALU_ISSUE_PEAK,
                        "peak_is": "1024 SIMDs x 2.4 GHz / 4 cycles per 64-lane fp64 instruction",
                        "kind": "static_profile", "source": pmc["source"], "source_hash": pmc["source_hash"]}
            except (OSError, KeyError, ValueError):
                pass
            # the counted algorithmic fp64 work (tools/count_flops.py on the 1 000-object fixture) against the fp64 peak
            fp64 = None
            executed = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
                if pmc.get("source_hash") == source_hash():
                    ops = sum(k.get("fp64_lane_ops_upper", 0.0) for n, k in pmc["kernels"].items() if not n.startswith("k_diffcor"))
                    if ops > 0:
                        per_obj = ops / pmc["objects_per_step"]
                        executed = {"fp64_lane_ops_per_object_upper": per_obj,
                                    "achieved_tflops": value / world * per_obj / 1e12,
                                    "frac": value / world * per_obj / 1e12 / FP64_PEAK_TFLOPS, "kind": "static_profile",
                                    "what": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes (FMA = 2; inactive lanes "
                                            "included) of the committed profile of this build"}
            except (OSError, KeyError, ValueError):
                pass
            try:
                fl = json.load(open(os.path.join(ROOT, "profiles", "flops_per_object.json")))
                f_obj = fl["hoisted"]["total_fp64_ops_per_object"]
                fp64 = {"algorithmic_fp64_ops_per_object": f_obj,
                        "of_which_libm_nominal": fl["hoisted"]["libm_fp64_ops_per_object_nominal"],
                        "reference_decomposition_ops_per_object": fl["reference"]["total_fp64_ops_per_object"],
                        "achieved_tflops": value / world * f_obj / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                        "frac": value / world * f_obj / 1e12 / FP64_PEAK_TFLOPS,
                        "executed": executed,
                        "note": "algorithmic = what the reference's arithmetic executes per object (counted on the CPU "
                                "restatement, Earth hoisted); the device SKIPS part of it exactly (fixed-point exit of the "
                                "correction loop: 55 % of its iterations), so this is work retired, not instructions "
                                "issued — and the peak counts an FMA as 2 while the path may not contract a*b+c"}
            except (OSError, KeyError, ValueError):
                pass
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "limited_by": "fp64 VALU issue, not HBM (see valu / fp64): the HBM fraction is ~1e-3 by construction",
                    "valu": valu, "fp64": fp64,
                    "avg_launch_ms": avg_ms, "units_per_launch": units, "algorithmic_bytes_per_object": bytes_per_obj,
                    "all_kernels_avg_ms_per_step": {k: v[0] / args.steps for k, v in ktimes.items()}}
        line = {
            "metric": "objects/s full Gauss IOD (20 obs, 100 triplets/obj)",
            "value": value, "unit": "objects/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "math_mode": mode_names[head_mode],
            "config": {"workload": (f"{n_obj} synthetic objects per GPU x 20 obs, 100 triplets each "
                                    "(BASELINE configs[2] shape; synthetic DE440-layout table resident in HBM)")
                       if args.scaling == "weak" else
                       (f"{total_objects} synthetic objects x 20 obs, 100 triplets each, split over {world} GPU(s) "
                        "(BASELINE configs[3]; synthetic DE440-layout table resident in HBM)"),
                       "objects_per_gpu": counts, "obs_per_object": n_obs / n_obj, "triplets_per_object": 100,
                       "n_noise_realizations": 0, "orbits_found_rank0": n_ok,
                       "math_mode": mode_names[head_mode] + " (OUTFIT_MATH_GLIBC: sin/cos/sinh/cosh with glibc 2.35's bits — "
                                    "the reference-faithful mode and the library's default)" if head_mode == outfit_amd.MATH_GLIBC
                       else mode_names[head_mode],
                       "parallelism": f"objects sharded over {world} GPU(s); RCCL all_gather of 96-B results, overlapped with the next step"},
            "roofline": roof,
        }
        if modes is not None:
            line["math_modes"] = modes
            if "parity_vs_libm" in modes:
                line["parity_vs_libm"] = modes["parity_vs_libm"]
        if gather_check is not None:
            line["gather_check"] = gather_check
        if lsq is not None:
            ms = 1e3 * lsq["elapsed"] / args.steps
            line["fit_lsq"] = {
                "what": "differential correction stage alone (k_diffcor on the resident IOD seeds, default "
                        "DifferentialCorrectionConfig), rank 0; BASELINE configs[4] per-GPU shape",
                "ms_per_step": ms, "objects_per_s_stage": n_obj * args.steps / lsq["elapsed"],
                "objects_per_s_iod_plus_lsq": n_obj / (1e-3 * ms + elapsed / args.steps),
                "kernel_avg_ms": lsq["kernel_ms"][0] / max(1, lsq["kernel_ms"][1]),
                "converged": lsq["n_diffcor"], "kept_iod_orbit": lsq["n_iod_kept"],
                "mean_newton_iterations": lsq["mean_newton_iterations"]}
        if side is not None:
            line["side"] = side
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, ephem, params, n_obj)
            line["cpu_baseline_1thread"] = cpu_baseline(batch, ephem, params, n_obj, budget_s=5.0, cores=1)
            line["cpu_baseline_hoisted"] = cpu_baseline(batch, ephem, params, n_obj, budget_s=5.0, hoist=1)
            if lsq is not None:
                line["fit_lsq"]["cpu_baseline"] = cpu_baseline_lsq(batch, ephem, params, res, n_obj)
        print(json.dumps(line), flush=True)
    session.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
