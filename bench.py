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

T_PROCESS_START = time.perf_counter()
# RCCL between processes needs dmabuf IPC on this pool's host driver (hipIpcGetMemHandle fails with "invalid argument"
# in the legacy mode): must be in the environment BEFORE the first HIP call of the process. The driver's launcher and
# the image export it already; set here so that `python -m torch.distributed.run ... bench.py` works from a bare shell.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X_MICROARCH.md: fp64 vector, 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz
# the ONE fp64 issue peak used everywhere: a 64-lane fp64 VALU instruction occupies a SIMD's 16 lanes for 4 cycles,
# 1024 SIMDs at the 2.4 GHz peak engine clock -> wave-instructions per second
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0
# measured issue costs on gfx950 (tools/issue_cost.hip -> profiles/round04_issue_cost_gfx950.json): fp64 add / mul / fma,
# compares, v_div_*, v_ldexp, conversions 4.0 shader cycles per 64-lane instruction per SIMD; v_rcp / v_rsq / v_sqrt_f64
# (the counters' TRANS class) 16.0; plain 32-bit moves / integer adds 2 - 4 (priced at 4 here: a lower bound on the fraction)
ISSUE_CYCLES_PLAIN, ISSUE_CYCLES_TRANS_F64 = 4.0, 16.0
N_SIMDS, NOMINAL_CLOCK_HZ = 1024, 2.4e9
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
    """What the oracle's `orc_use_libm = 1` mode evaluates with on this host, and whether that is the ONE build the
    glibc-faithful device mode restates (outfit_amd/libm_pin.py): only then is `parity_vs_libm` a parity statement."""
    from outfit_amd import libm_pin
    return libm_pin.describe()


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
    from outfit_amd import libm_pin
    pinned = libm_pin.host_has_pinned_libm()
    out = {"sample": f"first {n_sample} objects of the timed batch", "oracle": "CPU restatement, platform libm (orc_use_libm=1)",
           "host_libm": host_libm(), "host_libm_is_the_pinned_build": pinned,
           "meaning": ("platform libm = the glibc build OUTFIT_MATH_GLIBC restates: byte identity is expected in the glibc mode"
                       if pinned else
                       "platform libm is ANOTHER build than the one OUTFIT_MATH_GLIBC restates: the figures below are "
                       "distances to this host's libm, not a parity statement (" + str(libm_pin.host_libm_identity()["why_not"]) + ")"),
           "oracle_seconds": time.perf_counter() - t,
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


def noise_point(label, ds, ephem, params, device, stream, steps, seed, workload):
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
    if n_z:
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
    # K0 and K1 run on side streams under the previous run's kernels: in back-to-back runs their event times include
    # waiting for issue slots. One ISOLATED run (device idle before it) gives what they cost by themselves.
    torch.cuda.synchronize()
    s.set_profiling(True)
    s.run(stream=stream)
    torch.cuda.synchronize()
    kt_iso = s.kernel_times()
    s.set_profiling(False)
    s.close()
    cand = n_obj * params.max_triplets * (1 + params.n_noise_realizations)
    n_obs = int(batch.traj_offset[-1])
    # algorithmic bytes: the observation columns, the record, and the host-drawn normals (an input of this workload)
    bytes_per_obj = (n_obs / n_obj) * (BYTES_PER_OBS_IN + BYTES_PER_OBS_GEO) + BYTES_PER_RESULT + 8.0 * n_z / n_obj
    return {"what": label, "roofline": roofline_block(kt, n_obj, steps, dt, bytes_per_obj, workload), "objects": n_obj, "max_triplets": params.max_triplets,
            "n_noise_realizations": params.n_noise_realizations, "candidates_per_step": cand,
            "normals": f"{n_z} host-drawn standard normals ({n_z * 8 / 1e9:.2f} GB, numpy PCG64), resident in HBM",
            "obs_per_object": n_obs / n_obj,
            "steps": steps, "ms_per_step": 1e3 * dt, "objects_per_s": n_obj / dt, "candidates_per_s": cand / dt,
            "kernel_avg_ms_per_step": {k: v[0] / steps for k, v in kt.items()},
            "isolated_run_kernel_ms": {k: v[0] for k, v in kt_iso.items()},
            "orbits_found": int((rec["status"] == 0).sum()),
            "noisy_winners": int(((rec["status"] == 0) & (rec["best_realization"] > 0)).sum()),
            "session_build_and_first_run_s": setup}


def long_arc_points(outfit_amd, synth, device, stream, steps, which=(100, 300), n_obj=10000):
    """Long trajectories under the reference's own defaults: `IODParams::default()` caps the
    triplet search at max_obs_for_triplets = 100 epochs, i.e. up to C(100,3) = 161 700 scored triplets per trajectory
    (triplet_generation/mod.rs:365-408, index_generator.rs:133-271), and the RMS window covers the whole arc
    (extf = -1). n_obs = 100: the cap is not hit; n_obs = 300: down-sampling + quirk Q1. Two parameter sets each:
    IODParams::default() (10 triplets x 21 realisations, host-drawn normals) and 100 triplets without noise."""
    out = {}
    for n_obs in which:
        ds = synth.make_long_arc_dataset(n_obj, n_obs)
        ephem = outfit_amd.Ephemeris.from_dict(ds["ephem"])
        p_def = outfit_amd.IODParams()
        p_100 = outfit_amd.IODParams()
        p_100.max_triplets = 100
        p_100.n_noise_realizations = 0
        for tag, what, prm in (("iod_params_default", "IODParams::default()", p_def),
                               ("100_triplets_no_noise", "IODParams::default() except max_triplets = 100, "
                                                         "n_noise_realizations = 0", p_100)):
            r = noise_point(f"{n_obj} objects x {n_obs} observations, {what}", ds, ephem, prm, device, stream, steps,
                            20261100 + n_obs, f"long_arcs_{n_obs}_{tag}")
            k0 = r["kernel_avg_ms_per_step"].get("k_triplet_select", 0.0)
            r["k_triplet_select_share_of_step"] = k0 / r["ms_per_step"] if r["ms_per_step"] > 0 else None
            # (event time back to back, i.e. incl. waiting under the previous run's kernels, over the step time; the
            # kernel's own time is isolated_run_kernel_ms.k_triplet_select)
            r["k_triplet_select_isolated_ms"] = r["isolated_run_kernel_ms"].get("k_triplet_select")
            out[f"{n_obs}_obs_{tag}"] = r
    return out


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


MAIN_STREAM_KERNELS = ("k_gauss_roots", "k_gauss_correct_fast", "k_orbit_rms", "k_orbit_rms_slow", "k_select_best",
                       "k_refine_best", "k_diffcor")
# live HIP-event timers (outfit_hip_session_kernel_times) -> which rocprofv3 kernel names (template instances) they cover
PMC_MATCH = {
    "k_triplet_select": lambda k: k.startswith("k_triplet_select"),
    "k_observer_prepass": lambda k: k.startswith("k_observer_prepass"),
    "k_gauss_roots": lambda k: k.startswith("k_gauss_roots"),
    "k_gauss_correct_fast": lambda k: k.startswith("k_gauss_correct_fast"),
    "k_gauss_correct": lambda k: k.startswith("k_gauss_correct<"),
    "k_orbit_rms": lambda k: k.startswith("k_orbit_rms<false"),
    "k_orbit_rms_slow": lambda k: k.startswith("k_orbit_rms<true"),
    "k_select_best": lambda k: k.startswith("k_select_best"),
    "k_refine_best": lambda k: k.startswith("k_refine_best"),
    "k_diffcor": lambda k: k.startswith("k_diffcor"),
}


def _pmc_workload(workload):
    """The committed PMC profile of `workload`, or None unless it was taken from THIS build (source hash)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
    except (OSError, ValueError):
        return None
    if pmc.get("source_hash") != source_hash():
        return None
    w = pmc.get("workloads", {}).get(workload)
    if w is None and workload == "headline" and "kernels" in pmc:      # layout of rounds 2-3
        w = {"kernels": pmc["kernels"], "objects_per_step": pmc["objects_per_step"]}
    if w is None:
        return None
    w = dict(w)
    w["source"], w["source_hash"] = pmc.get("source"), pmc.get("source_hash")
    return w


def _pmc_rows(w, timer):
    """Profile rows of the kernel(s) behind one live timer (template instances are summed)."""
    match = PMC_MATCH.get(timer, lambda k: k == timer)
    return [v for k, v in w["kernels"].items() if match(k)]


def roofline_block(ktimes, n_obj, steps, step_s, bytes_per_obj, workload, objects_per_s_per_gpu=None,
                   exclude=("k_diffcor",), only=None):
    """The `roofline` object of one timed region: the contract figure (algorithmic bytes of the dominant
    kernel against the HBM peak) and, from the hash-matched PMC profile of the same workload, what actually limits the
    path: VALU issue, executed fp64, lanes per instruction and counter bytes against algorithmic bytes — per kernel
    and for the step. Counter-derived fields are null when no profile of THIS build exists (a stale one says nothing)."""
    pick = [k for k in ktimes if (only is None or k in only) and k not in exclude]
    main = [k for k in pick if k in MAIN_STREAM_KERNELS]
    if not main:
        return None
    dom = max(main, key=lambda k: ktimes[k][0])
    total_ms, launches = ktimes[dom]
    avg_ms = total_ms / launches
    units = n_obj * steps / launches                 # objects one launch processes (this rank's shard)
    achieved = units * bytes_per_obj / (avg_ms * 1e-3) / 1e9
    w = _pmc_workload(workload)
    per_kernel, traffic, issue, executed, step_traffic = [], None, None, None, None
    alg_step = n_obj * bytes_per_obj
    if w is not None:
        sc = n_obj / w["objects_per_step"]
        tot_instr = tot_f64i = tot_ops = tot_bytes = tot_cyc = tot_cyc_s = 0.0
        clocks_known = True
        for k in sorted(pick, key=lambda k: -ktimes[k][0]):
            rows = _pmc_rows(w, k)
            if not rows:
                continue
            instr = sum(r["valu_wave_instr"] for r in rows) * sc
            f64i = sum(r.get("fp64_wave_instr", 0.0) for r in rows) * sc
            ops = sum(r.get("fp64_lane_ops_upper", 0.0) for r in rows) * sc
            byt = sum(r["hbm_bytes_per_step"] for r in rows) * sc
            thr = sum(r.get("thread_cycles_valu", 0.0) for r in rows)
            act = sum(r.get("active_inst_valu", 0.0) for r in rows)
            ms = ktimes[k][0] / steps
            trans = sum(r.get("fp64_trans_wave_instr", 0.0) for r in rows) * sc
            cyc = (instr - trans) * ISSUE_CYCLES_PLAIN + trans * ISSUE_CYCLES_TRANS_F64   # SIMD-cycles of pure issue
            ghz = [r["engine_clock_ghz"] for r in rows if r.get("engine_clock_ghz")]
            # (short kernels that overlap others read above the 2.4 GHz maximum: capped)
            clock_hz = min(1e9 * sum(ghz) / len(ghz), NOMINAL_CLOCK_HZ) if ghz and k in MAIN_STREAM_KERNELS else None
            tot_instr += instr; tot_f64i += f64i; tot_ops += ops; tot_bytes += byt; tot_cyc += cyc
            if clock_hz:
                tot_cyc_s += cyc / N_SIMDS / clock_hz
            elif k in MAIN_STREAM_KERNELS:
                clocks_known = False
            per_kernel.append({
                "kernel": k, "stream": "main" if k in MAIN_STREAM_KERNELS else "side (overlapped; its time includes waiting for CUs)",
                "ms_per_step": ms, "valu_wave_instr_per_step": instr,
                "issue_frac": instr / VALU_ISSUE_PEAK / (ms * 1e-3) if ms > 0 else None,
                # the same with the measured cost of each instruction class, at the nominal and at the measured clock
                "issue_cycle_frac": cyc / N_SIMDS / NOMINAL_CLOCK_HZ / (ms * 1e-3) if ms > 0 and trans > 0 else None,
                "clock_ghz_under_load": clock_hz / 1e9 if clock_hz else None,
                "issue_cycle_frac_at_measured_clock": cyc / N_SIMDS / clock_hz / (ms * 1e-3) if ms > 0 and trans > 0 and clock_hz else None,
                "lanes_per_instruction": thr / act if act > 0 else None,
                "non_fp64_share": 1.0 - f64i / instr if instr > 0 and f64i > 0 else None,
                "counter_bytes_per_step": byt, "counter_over_algorithmic_bytes": byt / alg_step})
            if k == dom:
                traffic = byt * steps / launches
        issue = {"step_wave_instr": tot_instr, "step_issue_frac": tot_instr / VALU_ISSUE_PEAK / step_s,
                 "peak_wave_instr_per_s": VALU_ISSUE_PEAK,
                 "peak_is": "1024 SIMDs x 2.4 GHz / 4 cycles per 64-lane fp64 instruction",
                 # every instruction at its MEASURED issue cost (profiles/round04_issue_cost_gfx950.json): 4 cycles, the
                 # TRANS class (v_rcp / v_rsq / v_sqrt_f64: every division and square root holds one) 16
                 "step_issue_cycle_frac": tot_cyc / N_SIMDS / NOMINAL_CLOCK_HZ / step_s if tot_cyc > tot_instr * ISSUE_CYCLES_PLAIN else None,
                 # ... and each main-stream kernel at the shader clock measured under this load (GRBM_GUI_ACTIVE /
                 # duration, 2.1 - 2.3 GHz, not the 2.4 GHz of the peak): the share of the step in which a SIMD issues
                 "step_issue_cycle_frac_at_measured_clock": tot_cyc_s / step_s if clocks_known and tot_cyc_s > 0 else None,
                 "cycles_per_instruction": {"fp64_add_mul_fma_and_others": ISSUE_CYCLES_PLAIN, "fp64_trans": ISSUE_CYCLES_TRANS_F64,
                                            "source": "profiles/round04_issue_cost_gfx950.json"}}
        if tot_ops > 0:
            tf = tot_ops / step_s / 1e12
            executed = {"fp64_lane_ops_per_object_upper": tot_ops / n_obj, "achieved_tflops": tf,
                        "peak_tflops": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                        "non_fp64_share_of_valu": 1.0 - tot_f64i / tot_instr if tot_f64i > 0 else None,
                        "what": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes (FMA = 2; inactive lanes included)"}
        step_traffic = {"counter_bytes": tot_bytes, "algorithmic_bytes": alg_step, "ratio": tot_bytes / alg_step,
                        "GBps": tot_bytes / step_s / 1e9, "frac_of_hbm_peak": tot_bytes / step_s / 1e9 / HBM_PEAK_GBS}
    ref_equiv = None
    if workload == "headline" and objects_per_s_per_gpu:
        try:   # counted on the CPU restatement (tools/count_flops.py): work RETIRED in the reference's terms, not a utilisation
            fl = json.load(open(os.path.join(ROOT, "profiles", "flops_per_object.json")))
            f_obj = fl["hoisted"]["total_fp64_ops_per_object"]
            ref_equiv = {"reference_equivalent_tflops": objects_per_s_per_gpu * f_obj / 1e12,
                         "algorithmic_fp64_ops_per_object": f_obj,
                         "of_which_libm_nominal": fl["hoisted"]["libm_fp64_ops_per_object_nominal"],
                         "reference_decomposition_ops_per_object": fl["reference"]["total_fp64_ops_per_object"],
                         "note": "what the reference's arithmetic would execute for these objects per second (counted on "
                                 "the CPU restatement, Earth hoisted, libm calls at nominal cost). NOT a utilisation: the "
                                 "device skips part of that work exactly (fixed-point exit of the correction loop: 55 % "
                                 "of its iterations), and the fp64 peak counts an FMA as 2 while the path may not "
                                 "contract a*b+c. The utilisations are `issue.step_issue_frac` and `fp64_executed.frac`."}
        except (OSError, KeyError, ValueError):
            pass
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "limited_by": "fp64 VALU issue, not HBM: `frac` (algorithmic bytes of the dominant kernel against the HBM "
                          "peak, the contract figure) is ~1e-3 by construction; read issue.step_issue_frac and "
                          "fp64_executed.frac",
            "issue": issue, "fp64_executed": executed, "step_traffic": step_traffic, "per_kernel": per_kernel or None,
            "reference_equivalent": ref_equiv,
            "counters": None if w is None else {"kind": "static_profile", "workload": workload, "source": w["source"],
                                                "source_hash": w["source_hash"]},
            "avg_launch_ms": avg_ms, "units_per_launch": units, "algorithmic_bytes_per_object": bytes_per_obj,
            "all_kernels_avg_ms_per_step": {k: v[0] / steps for k, v in ktimes.items()}}


def lsq_configs(outfit_amd):
    """(name, description, config): the reference's default, and two workloads on which the corrector CORRECTS — on the
    bench population the default divergence rule (diff_cor.rs:354-358, ratio 1.5 at :183) ends two thirds of the
    trajectories at Newton iteration 2, which times the exit path; the reference's own integration test raises the ratio
    to 10 for that reason (tests/test_diff_cor.rs:50-56)."""
    D = outfit_amd.DifferentialCorrectionConfig
    return [("default", "DifferentialCorrectionConfig::default() (rms_divergence_ratio 1.5)", D()),
            ("ratio10", "rms_divergence_ratio = 10, the value of the reference's own test (tests/test_diff_cor.rs:52-55)",
             D(rms_divergence_ratio=10.0)),
            ("no_divergence_check", "rms_divergence_ratio = 1e300 (the divergence rule never fires: every trajectory "
                                    "iterates to convergence, stagnation or the iteration cap)", D(rms_divergence_ratio=1e300))]


def lsq_point(session, cfg, what, stream, steps, fence, n_obj, n_obs, iod_step_s, workload):
    """k_diffcor on the session's resident IOD seeds: W = 1 warm-up pass, `steps` timed passes between fences."""
    import numpy as np
    session.run_lsq(cfg, stream=stream)   # warm-up (allocates the corrector's scratch)
    fence()
    session.set_profiling(True)
    t1 = time.perf_counter()
    for _ in range(steps):
        session.run_lsq(cfg, stream=stream)
    fence()
    el = time.perf_counter() - t1
    kt = session.kernel_times()
    session.set_profiling(False)
    lrec, _ = session.lsq_results()
    ms = 1e3 * el / steps
    it = lrec["total_newton_iterations"]
    hist = np.bincount(np.minimum(it, 20).astype(np.int64), minlength=21)
    bytes_per_obj = (n_obs / n_obj) * (BYTES_PER_OBS_IN + 32) + BYTES_PER_RESULT + 480   # inputs, fit rows, seed, record
    kd = kt.get("k_diffcor", (0.0, 1))
    return {"what": "differential correction stage alone (k_diffcor on the resident IOD seeds), rank 0; BASELINE "
                    "configs[4] per-GPU shape; config: " + what,
            "ms_per_step": ms, "objects_per_s_stage": n_obj * steps / el,
            "objects_per_s_iod_plus_lsq": n_obj / (1e-3 * ms + iod_step_s),
            "kernel_avg_ms": kd[0] / max(1, kd[1]),
            "converged": int((lrec["status"] == 0).sum()), "kept_iod_orbit": int((lrec["status"] == 1).sum()),
            "dc_error_counts": {str(k): int(v) for k, v in zip(*np.unique(lrec["dc_error"][lrec["status"] == 1], return_counts=True))},
            "dc_error_names": {"1": "DifferentialCorrectionFailed", "2": "BizarreOrbit", "3": "DifferentialCorrectionDiverged"},
            "mean_newton_iterations": float(it.mean()),
            "newton_iteration_histogram": {str(i) if i < 20 else "20+": int(c) for i, c in enumerate(hist) if c},
            "outlier_passes_mean": float(lrec["outer_passes"].mean()), "rejected_observations": int(lrec["n_rejected"].sum()),
            "roofline": roofline_block(kt, n_obj, steps, el / steps, bytes_per_obj, "lsq_" + workload, exclude=(),
                                       only=("k_diffcor",))}


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


LINE_BYTE_BUDGET = 4096        # the driver keeps an 8 KB tail of stdout: the line must fit it with room to spare


def _num(x, digits=6):
    """Floats of the printed line at `digits` significant digits (the detail file keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None            # the line is strict JSON: no NaN / Infinity tokens
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_roofline(r):
    """The contract keys of a roofline block plus ONE number per derived quantity."""
    if r is None:
        return None
    return {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
            "frac": r["frac"], "traffic": r["traffic"], "avg_launch_ms": r["avg_launch_ms"],
            "units_per_launch": r["units_per_launch"], "algorithmic_bytes_per_object": r["algorithmic_bytes_per_object"],
            "limited_by": "fp64 VALU issue, not HBM",
            "issue": {"step_issue_frac": _get(r, "issue", "step_issue_frac"),
                      "step_issue_cycle_frac_at_measured_clock": _get(r, "issue", "step_issue_cycle_frac_at_measured_clock")},
            "fp64_executed": {"frac": _get(r, "fp64_executed", "frac"), "achieved_tflops": _get(r, "fp64_executed", "achieved_tflops"),
                              "peak_tflops": FP64_PEAK_TFLOPS},
            "step_traffic": {"ratio": _get(r, "step_traffic", "ratio"), "counter_bytes": _get(r, "step_traffic", "counter_bytes")},
            "counters_source_hash": _get(r, "counters", "source_hash"),
            "kernel_ms_per_step": r["all_kernels_avg_ms_per_step"]}


def compact_line(full, detail_path):
    """What rank 0 prints: the contract keys, `roofline`, `cpu_baseline`, a short parity summary and one number per side
    workload. Everything else of `full` (per-kernel tables, mode tables, long arcs, notes) is in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "math_mode")
    line = {k: full[k] for k in keep if k in full}
    c = full["config"]
    line["config"] = {"workload": c["workload"], "objects_per_gpu": c["objects_per_gpu"], "obs_per_object": c["obs_per_object"],
                      "triplets_per_object": c["triplets_per_object"], "n_noise_realizations": c["n_noise_realizations"],
                      "orbits_found_rank0": c["orbits_found_rank0"], "parallelism": c["parallelism"]}
    line["roofline"] = compact_roofline(full.get("roofline"))
    if "cpu_baseline" in full:
        line["cpu_baseline"] = full["cpu_baseline"]
        line["cpu_baseline_1thread_objects_per_s"] = _get(full, "cpu_baseline_1thread", "value")
        line["cpu_baseline_hoisted_objects_per_s"] = _get(full, "cpu_baseline_hoisted", "value")
    p = full.get("parity_vs_libm")
    if p is not None:
        m = p.get(full["math_mode"], {})
        line["parity_vs_libm"] = {"objects": m.get("orbits") and int(p["sample"].split()[1]), "mode": full["math_mode"],
                                  "host_libm_is_the_pinned_build": p["host_libm_is_the_pinned_build"],
                                  "byte_identical_records": m.get("byte_identical_records"),
                                  "beyond_1e-10_AU": m.get("beyond_1e-10_AU"), "beyond_1e-12_rad": m.get("beyond_1e-12_rad")}
    if "math_modes" in full:
        mm = full["math_modes"]
        other = [k for k in ("glibc", "cr") if k in mm and k != mm["headline"]]
        if other:
            line["other_math_mode"] = {"mode": other[0], "objects_per_s": mm[other[0]]["value"]}
    side = full.get("side") or {}
    s = {}
    if side:
        s["isolated_run_ms"] = side.get("isolated_run_ms")
        s["blocking_call_ms"] = side.get("blocking_call_ms")
        s["configs1_1000_objects_ms_per_step"] = _get(side, "configs1_1000_objects", "ms_per_step_back_to_back")
        s["configs1_1000_objects_blocking_call_ms"] = _get(side, "configs1_1000_objects", "blocking_call_ms")
        s["noise_default_objects_per_s"] = _get(side, "noise", "iod_params_default", "objects_per_s")
        s["noise_8d_objects_per_s"] = _get(side, "noise", "survey_8d", "objects_per_s")
        la = side.get("long_arcs") or {}
        if la:
            s["long_arcs_ms_per_10000_objects"] = {k: v["ms_per_step"] for k, v in la.items()}
    if "fit_lsq" in full:
        f = full["fit_lsq"]
        s["fit_lsq"] = {"ms_per_step": f["ms_per_step"], "objects_per_s_stage": f["objects_per_s_stage"],
                        "objects_per_s_iod_plus_lsq": f["objects_per_s_iod_plus_lsq"],
                        "roofline_frac": _get(f, "roofline", "frac"),
                        "cpu_baseline_objects_per_s": _get(f, "cpu_baseline", "value"),
                        "ms_per_step_by_config": {k: v["ms_per_step"] for k, v in (f.get("workloads") or {}).items()}}
    if s:
        line["side"] = s
    g = full.get("gather_check")
    if g is not None:
        line["gather_check"] = {"ranks_verified_by_every_rank": g["ranks_verified_by_every_rank"], "records": g["records"],
                                "ipc_mode": g["ipc_mode"],
                                "ms_per_step_per_rank": g["per_rank"]["ms_per_step"],
                                "gather_wait_ms_per_step_per_rank": g["per_rank"]["gather_wait_ms_per_step"]}
    line["detail"] = detail_path
    exact = {k: line[k] for k in ("value", "ms_per_step") if k in line}     # the driver re-derives these: full precision
    line = _num(line)
    line.update(exact)
    if len(json.dumps(line)) > LINE_BYTE_BUDGET:        # never again a line the driver cannot keep: shed the optional parts
        for k in ("side", "gather_check", "other_math_mode"):
            line.pop(k, None)
            if len(json.dumps(line)) <= LINE_BYTE_BUDGET:
                break
    return line


def write_detail(full):
    """The full record (per-kernel tables, mode tables, side workloads) beside the line: gpurun_out/ is merged back from
    the GPU box; falls back to the working directory, then to stderr only."""
    text = json.dumps(full)
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_detail_n{full.get('n_gpus', 1)}.json")
            with open(path, "w") as f:
                f.write(text + "\n")
            return os.path.relpath(path, ROOT)
        except OSError:
            continue
    return None


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
    ap.add_argument("--no-long-arcs", action="store_true", help="skip the long-trajectory side measurements (100 / 300 observations)")
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
    ap.add_argument("--workload", default="all", choices=["all", "headline", "noise_default", "noise_8d", "lsq_default",
                                                          "lsq_ratio10", "lsq_no_divergence_check", "long_arcs_100",
                                                          "long_arcs_300"],
                    help="profiling aid (tools/profile_gpu.sh): run ONLY that workload's timed region, so that a "
                         "rocprofv3 pass attributes its counters to it; `all` is the bench line")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import outfit_amd
    from outfit_amd import libm_pin, synth

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
    if args.workload == "headline":
        args.no_modes = args.no_side = args.no_lsq = args.no_cpu_baseline = True
    elif args.workload.startswith("lsq_"):
        args.no_modes = args.no_side = args.no_cpu_baseline = True
    elif args.workload != "all":
        # profiling aid: ONLY that side workload's timed region runs in this process (tools/profile_gpu.sh)
        assert world == 1, "--workload is a single-GPU profiling aid"
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        nsteps = max(1, args.steps)
        if args.workload == "noise_default":
            r = noise_point("IODParams::default()", ds, ephem, outfit_amd.IODParams(), local_rank, ts.cuda_stream, nsteps,
                            20261004, "noise_default")
        elif args.workload == "noise_8d":
            p_8d = synth.bench_params(outfit_amd.IODParams())
            p_8d.n_noise_realizations = 10
            r = noise_point("bench params with n_noise_realizations = 10", ds, ephem, p_8d, local_rank, ts.cuda_stream,
                            nsteps, 20261005, "noise_8d")
        else:
            r = long_arc_points(outfit_amd, synth, local_rank, ts.cuda_stream, nsteps,
                                which=(int(args.workload.rsplit("_", 1)[1]),))
        print(json.dumps({"workload": args.workload, "result": r}), flush=True)
        return
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
    state = {"step": 0, "gather_wait_s": 0.0}

    def wait_gather(b):
        if pending[b] is not None:
            t = time.perf_counter()
            pending[b].wait()
            state["gather_wait_s"] += time.perf_counter() - t
            pending[b] = None

    def step(sess=None):
        b = state["step"] & 1
        state["step"] += 1
        wait_gather(b)
        (sess or session).run(stream=stream, d_out=out_local[b].data_ptr())
        if multi:
            if args.backend == "nccl":
                pending[b] = dist.all_gather_into_tensor(out_all[b], out_local[b], async_op=True)
            else:  # rehearsal: gloo gathers host copies
                pending[b] = dist.all_gather_into_tensor(out_all[b], out_local[b].cpu(), async_op=True)

    def fence():
        for b in (0, 1):
            wait_gather(b)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(sess):
        """W warm-up steps, then exactly K steps between fences; (seconds, {kernel: (ms, launches)})."""
        for _ in range(args.warmup):
            step(sess)
        fence()
        sess.set_profiling(True)
        state["gather_wait_s"] = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(sess)
        state["enqueue_s"] = time.perf_counter() - t0   # host time to enqueue the K steps (incl. waits on old gathers)
        fence()
        dt = time.perf_counter() - t0
        kt = sess.kernel_times()     # HIP events on the run's stream over the whole timed region
        sess.set_profiling(False)
        sess.check()                 # an epoch outside the ephemeris table is an error here as in the blocking API
        return dt, kt

    setup_s = time.perf_counter() - T_PROCESS_START      # imports, process group, dataset, upload, session
    elapsed, ktimes = timed(session)
    my_elapsed, my_wait, my_enqueue = elapsed, state["gather_wait_s"], state["enqueue_s"]
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
                                                  local_rank, stream, nsteps, 20261004, "noise_default"),
                "survey_8d": noise_point("bench params with n_noise_realizations = 10 (SURVEY 8d: 100 triplets x 11)",
                                         ds, ephem, p_8d, local_rank, stream, nsteps, 20261005, "noise_8d")}
        if not args.no_long_arcs:
            side["long_arcs"] = long_arc_points(outfit_amd, synth, local_rank, stream, max(2, min(args.steps, 3)))

    # ---- next row (BASELINE configs[4]): IOD seed -> differential correction ----
    # Measured after the headline region and reported beside it, never mixed into `value`.
    lsq = None
    if not args.no_lsq and (args.workload == "all" or args.workload.startswith("lsq_")):
        session.run(stream=stream)            # the seeds: this session's own IOD records
        lsq = {name: lsq_point(session, cfg, what, stream, args.steps, fence, n_obj, n_obs, elapsed / args.steps, name)
               for name, what, cfg in lsq_configs(outfit_amd)
               if args.workload == "all" or args.workload == "lsq_" + name}

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
        # per-rank timing of the timed region (the headline uses the MAX): step time, host time blocked on gathers,
        # host set-up time before the timed region — what to look at when the N-GPU value is not N x the 1-GPU value
        mine_t = torch.tensor([my_elapsed, my_wait, my_enqueue, setup_s,
                               sum(v[0] for k, v in ktimes.items() if k in MAIN_STREAM_KERNELS) * 1e-3],
                              dtype=torch.float64, device=gather_dev)
        all_t = torch.zeros(world * 5, dtype=torch.float64, device=gather_dev)
        dist.all_gather_into_tensor(all_t, mine_t)
        all_t = all_t.cpu().numpy().reshape(world, 5)
        gather_check = {"ranks_verified_by_every_rank": world, "records": int(sum(counts)), "crc32_per_rank": sums,
                        "per_rank": {"ms_per_step": [1e3 * x / args.steps for x in all_t[:, 0]],
                                     "gather_wait_ms_per_step": [1e3 * x / args.steps for x in all_t[:, 1]],
                                     "host_enqueue_ms_per_step": [1e3 * x / args.steps for x in all_t[:, 2]],
                                     "main_stream_kernel_ms_per_step": [1e3 * x / args.steps for x in all_t[:, 4]],
                                     "setup_s_before_timed_region": [float(x) for x in all_t[:, 3]]},
                        "ipc_mode": "HSA_ENABLE_IPC_MODE_LEGACY=" + os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>")}
    n_ok = int((res["status"] == 0).sum())

    if rank == 0:
        total_objects = sum(counts)
        value = total_objects * args.steps / elapsed
        bytes_per_obj = (n_obs / n_obj) * (BYTES_PER_OBS_IN + BYTES_PER_OBS_GEO) + BYTES_PER_RESULT
        roof = roofline_block(ktimes, n_obj, args.steps, elapsed / args.steps, bytes_per_obj, "headline",
                              objects_per_s_per_gpu=value / world)
        line = {
            "metric": "objects/s full Gauss IOD (20 obs, 100 triplets/obj)",
            "value": value, "unit": "objects/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "math_mode": mode_names[head_mode], "host_libm": host_libm(),
            "config": {"workload": (f"{n_obj} synthetic objects per GPU x 20 obs, 100 triplets each "
                                    "(BASELINE configs[2] shape; synthetic DE440-layout table resident in HBM)")
                       if args.scaling == "weak" else
                       (f"{total_objects} synthetic objects x 20 obs, 100 triplets each, split over {world} GPU(s) "
                        "(BASELINE configs[3]; synthetic DE440-layout table resident in HBM)"),
                       "objects_per_gpu": counts, "obs_per_object": n_obs / n_obj, "triplets_per_object": 100,
                       "n_noise_realizations": 0, "orbits_found_rank0": n_ok,
                       "math_mode": mode_names[head_mode] + " (OUTFIT_MATH_GLIBC, the library's default: every elementary "
                                    "function of the path — sin, cos, sincos, tan, atan2, acos, hypot, log, exp / sinh / cosh — "
                                    "with the bits of ONE libm build, " + libm_pin.PINNED_LIBM_WHAT + "; IOD and the "
                                    "differential corrector both follow the mode)" if head_mode == outfit_amd.MATH_GLIBC
                       else mode_names[head_mode] + " (OUTFIT_MATH_CORRECTLY_ROUNDED: platform-independent)",
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
            first = "default" if "default" in lsq else next(iter(lsq))
            line["fit_lsq"] = dict(lsq[first])          # the reference's default config: the keys of rounds 1-3
            line["fit_lsq"]["workloads"] = {k: v for k, v in lsq.items() if k != first}
        if side is not None:
            line["side"] = side
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, ephem, params, n_obj)
            line["cpu_baseline_1thread"] = cpu_baseline(batch, ephem, params, n_obj, budget_s=5.0, cores=1)
            line["cpu_baseline_hoisted"] = cpu_baseline(batch, ephem, params, n_obj, budget_s=5.0, hoist=1)
            if lsq is not None:
                line["fit_lsq"]["cpu_baseline"] = cpu_baseline_lsq(batch, ephem, params, res, n_obj)
        detail = write_detail(line)
        if detail is None:                                       # no writable place: the full record goes to stderr
            print(json.dumps(line), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(line, detail), allow_nan=False), flush=True)   # THE line: last on stdout, compact
    session.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
