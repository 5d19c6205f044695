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

    # ---- this rank's shard of the synthetic dataset (weak scaling: fixed objects per GPU) ----
    n_obj = args.objects
    ephem_d = synth.make_ephemeris()
    ds = synth.make_dataset(n_obj, seed=synth.DEFAULT_SEED + rank, ephem=ephem_d)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    ephem = outfit_amd.Ephemeris.from_dict(ephem_d)
    params = synth.bench_params(outfit_amd.IODParams())
    n_obs = int(batch.traj_offset[-1])

    session = outfit_amd.IODSession(batch, ephem, params, device=local_rank)
    # result buffers are double-buffered so that the gather of step i (RCCL, its own stream) overlaps the kernels
    # of step i+1; a buffer is reused only after the gather that read it has been waited for
    out_local = [torch.empty((n_obj, BYTES_PER_RESULT), dtype=torch.uint8, device="cuda") for _ in range(2)]
    gather_dev = "cuda" if args.backend == "nccl" else "cpu"
    out_all = [torch.empty((world * n_obj, BYTES_PER_RESULT), dtype=torch.uint8, device=gather_dev) for _ in range(2)] \
        if multi else None
    # An explicit torch stream: torch's default stream has handle 0, which the C ABI reads as "the session's own
    # stream" — the collective (RCCL syncs with torch's CURRENT stream) would then not wait for the kernels.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    pending = [None, None]
    state = {"step": 0}

    def step():
        b = state["step"] & 1
        state["step"] += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        session.run(stream=stream, d_out=out_local[b].data_ptr())
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

    for _ in range(args.warmup):
        step()
    fence()
    session.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ktimes = session.kernel_times()      # HIP events on the run's stream over the whole timed region
    session.set_profiling(False)
    session.check()                      # an epoch outside the ephemeris table is an error here as in the blocking API

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

    # ---- next row (BASELINE configs[4]): IOD seed -> differential correction ----
    # Measured after the headline region and reported beside it, never mixed into `value`.
    lsq = None
    if not args.no_lsq:
        cfg = outfit_amd.DifferentialCorrectionConfig()
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

    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the timed work produced orbits
    last = out_local[(state["step"] - 1) & 1]
    res = np.frombuffer(last.cpu().numpy().tobytes(), dtype=outfit_amd._abi.RESULT_DTYPE)
    if multi:  # the gathered table holds this rank's records at its offset
        mine = out_all[(state["step"] - 1) & 1][rank * n_obj:(rank + 1) * n_obj]
        assert bytes(mine.cpu().numpy().tobytes()) == res.tobytes(), "gathered records differ from the local ones"
    n_ok = int((res["status"] == 0).sum())

    if rank == 0:
        total_objects = world * n_obj
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
            units = n_obj * args.steps / launches          # objects one launch processes
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
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n_obj} synthetic objects per GPU x 20 obs, 100 triplets each "
                                   "(BASELINE configs[2] shape; synthetic DE440-layout table resident in HBM)",
                       "objects_per_gpu": n_obj, "obs_per_object": n_obs / n_obj, "triplets_per_object": 100,
                       "n_noise_realizations": 0, "orbits_found_rank0": n_ok,
                       "parallelism": f"objects sharded over {world} GPU(s); RCCL all_gather of 96-B results, overlapped with the next step"},
            "roofline": roof,
        }
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
