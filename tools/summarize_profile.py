#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh) into profiles/<name>_rocprof_summary.md,
profiles/<name>_kernel_stats.csv and profiles/latest_pmc.json (read by bench.py for roofline.traffic).

usage: tools/summarize_profile.py <tag> <name> [objects_per_step]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
n_obj = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")


def bench_runs(passdir):
    """How many times the pass's bench.py process ran the IOD kernels / the corrector: read from the JSON line it
    printed (bench_<pass>.log). Counters are summed per kernel name over the process, so per-step = sum / runs."""
    iod = lsq = 1
    try:
        line = [ln for ln in open(os.path.join(src, f"bench_{passdir}.log")) if ln.startswith("{")][-1]
        j = json.loads(line)
        if "workload" in j:                      # --workload noise_* / long_arcs_*: one set-up run + `steps` timed runs
            r = j["result"]
            r = r if "steps" in r else next(iter(r.values()))
            iod = 1 + int(r["steps"])
        else:
            iod = int(j["steps"]) + int(j["warmup"])
            if "fit_lsq" in j or "fit_lsq" in (j.get("side") or {}):                   # --workload lsq_*: + the seed run; corrector: one warm-up + steps
                iod += 1
                lsq = 1 + int(j["steps"])
    except (OSError, IndexError, KeyError, ValueError):
        pass
    return iod, lsq


def runs_of(k, passdir):
    iod, lsq = bench_runs(passdir)
    return float(lsq if k.startswith("k_diffcor") else iod)


def short(k):
    return k.replace("void ", "").replace("outfit::", "").split("(")[0]


def counters(sub):
    """({(kernel, counter): sum over the process}, {kernel: runs of that kernel's stage in the process})"""
    agg, calls = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        seen, disp = set(), set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("__amd"):
                continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            seen.add(k)
            disp.add((k, r["Dispatch_Id"]))
        for k in seen:
            # the calibration program is not bench.py: there the divisor is the number of launches
            calls[k] = runs_of(k, sub) if not sub.startswith("calib") else sum(1 for q, _ in disp if q == k)
    return agg, calls


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{name}_kernel_stats.csv"))
fetch, cf = counters("pmc_fetch")
write, cw = counters("pmc_write")
sq, cs = counters("pmc_sq")
f64, c64 = counters("pmc_f64")
# calibration of the byte counters on this library's access width (8 B per lane): tools/calib_stream.py streams
# 2^28 doubles in and out per launch; factor = true bytes / reported bytes (reported in KiB)
CAL_BYTES = (1 << 28) * 8.0
cal_f, ccf = counters("calib_fetch")
cal_w, ccw = counters("calib_write")
k_cal = "k_calibration_stream"
fetch_factor = write_factor = None
if ccf.get(k_cal) and cal_f.get((k_cal, "FETCH_SIZE")):
    fetch_factor = CAL_BYTES / (cal_f[(k_cal, "FETCH_SIZE")] * 1024 / ccf[k_cal])
if ccw.get(k_cal) and cal_w.get((k_cal, "WRITE_SIZE")):
    write_factor = CAL_BYTES / (cal_w[(k_cal, "WRITE_SIZE")] * 1024 / ccw[k_cal])
# engine clock under load: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / kernel duration, from the same pass's kernel trace
clk_lines = []
clock_ghz = {}   # kernel -> shader clock under the bench load (headline workload)
try:
    cyc, dur = collections.defaultdict(float), {}
    for f in glob.glob(os.path.join(src, "pmc_clk", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cyc[(short(r["Kernel_Name"]), r["Dispatch_Id"])] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(src, "pmc_clk", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    for (k, d), c in cyc.items():
        if d in dur and not k.startswith("__amd"):
            agg[k][0] += c
            agg[k][1] += dur[d]
    for k, (c, t) in sorted(agg.items()):
        clk_lines.append("%-24s GRBM_GUI_ACTIVE %.3e cycles over %.3e ns -> %.2f GHz per XCD (sum / 8)" % (k, c, t, c / t / 8.0))
        clock_ghz[k] = c / t / 8.0
except (OSError, KeyError, ValueError):
    pass
source_hash = None
try:
    source_hash = open(os.path.join(src, "source_hash.txt")).read().strip()
except OSError:
    pass
kernels = sorted({k for k, _ in sq})
per_step = {}
lines = []
sq_names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"]
hdr = "%-24s%7s" % ("kernel", "steps") + "".join("%16s" % c[:15] for c in sq_names) + "%16s%16s" % ("FETCH_SIZE", "WRITE_SIZE")
lines.append(hdr)
for k in kernels:
    def steps(c):
        return max(1.0, c.get(k, 0))
    row = {c: sq.get((k, c), 0.0) / steps(cs) for c in sq_names}
    row["FETCH_SIZE"] = fetch.get((k, "FETCH_SIZE"), 0.0) / steps(cf)
    row["WRITE_SIZE"] = write.get((k, "WRITE_SIZE"), 0.0) / steps(cw)
    per_step[k] = row
    lines.append("%-24s%7.0f" % (k, steps(cs)) + "".join("%16.4g" % row[c] for c in sq_names)
                 + "%16.4g%16.4g" % (row["FETCH_SIZE"], row["WRITE_SIZE"]))
derived = []
total_valu = 0.0
for k in kernels:
    r = per_step[k]
    fb, wb = r["FETCH_SIZE"] * 1024 * (fetch_factor or 1.0), r["WRITE_SIZE"] * 1024 * (write_factor or 1.0)
    total_valu += r["SQ_INSTS_VALU"] if not k.startswith("k_diffcor") else 0.0
    derived.append("%-24s VALU wave-instr %.3e  per wave %9.0f  VALU-active/wave-cycles %.2f  fetch %8.1f MB  write %8.1f MB"
                   % (k, r["SQ_INSTS_VALU"], r["SQ_INSTS_VALU"] / max(1.0, r["SQ_WAVES"]),
                      r["SQ_ACTIVE_INST_VALU"] / max(1.0, r["SQ_WAVE_CYCLES"]), fb / 1e6, wb / 1e6))
out = os.path.join(ROOT, "profiles", f"{name}_rocprof_summary.md")
with open(out, "w") as f:
    f.write(f"# {name} profile — bench.py ({n_obj} objects/step, 1x MI355X), rocprofv3; commands in tools/profile_gpu.sh\n\n")
    f.write("Kernels on the session's side streams (k_triplet_select_lds, k_observer_prepass, k_gauss_correct) run\n"
            "concurrently with the main-stream kernels named in DESIGN.md §3 (their durations include waiting for free CUs). k_diffcor is the differential-correction stage measured after the\n"
            "headline region (bench.py `fit_lsq`).\n\n")
    f.write("## rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\n```\n")
    f.write(open(stats).read())
    f.write("```\n## PMC passes (separate runs: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*; bench.py --steps 1 --warmup 0)\n"
            "Totals per step (= per launch set over the step's objects; a kernel launched in several slices or twice in\n"
            "the run is divided by its number of steps). FETCH_SIZE / WRITE_SIZE in KiB as reported. The guide calibrates these\n"
            "counters for 16-byte-per-lane streams only (gfx950: FETCH_SIZE reports half the bytes) and asks for a calibration\n"
            "in one's own access pattern otherwise: k_calibration_stream (8 B per lane, 2 GiB each way per launch, far beyond\n"
            f"the Infinity Cache) gives true/reported = {fetch_factor} for FETCH_SIZE and {write_factor} for WRITE_SIZE;\n"
            "the `derived` block and profiles/latest_pmc.json apply these factors.\n```\n")
    f.write("\n".join(lines) + "\n```\n## derived\n```\n" + "\n".join(derived) + "\n")
    f.write("IOD kernels: total VALU wave-instructions per step %.3e; at the fp64 issue peak of 1024 SIMDs x 2.4 GHz / 4 "
            "cycles = 6.14e11 wave-instructions/s this is %.1f ms of pure issue time\n" % (total_valu, total_valu / 6.144e11 * 1e3))
    if f64:
        f.write("\nfp64 instruction mix (wave-instructions per step; SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64):\n")
        tot = collections.Counter()
        for k in kernels:
            st = max(1.0, c64.get(k, 0))
            row = {c: f64.get((k, c), 0.0) / st for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
            f.write("%-24s add %.3e  mul %.3e  fma %.3e  trans %.3e\n" % (k, *row.values()))
            if not k.startswith("k_diffcor"):
                tot.update(row)
        fl = (tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + 2 * tot["SQ_INSTS_VALU_FMA_F64"] + tot["SQ_INSTS_VALU_TRANS_F64"]) * 64
        f.write("IOD kernels: executed fp64 lane-operations per step (add + mul + 2 fma + trans, x 64 lanes, upper bound: "
                "inactive lanes included) %.3e = %.3e per object\n" % (fl, fl / n_obj))
    mix, cmix = counters("pmc_mix")
    if mix:
        f.write("\ndynamic instruction classes per step (separate pass: SQ_INSTS_VALU, _VALU_INT32, _VALU_INT64, _VALU_CVT, _SALU, "
                "_SMEM, _LDS, _BRANCH; `other VALU` = VALU - fp64 (pass above) - int32 - int64 - cvt: moves, selects, compares, "
                "lane reads/writes, v_div_* scaffolding, ldexp, min/max):\n")
        for k in kernels:
            st = max(1.0, cmix.get(k, 0))
            g = lambda c: mix.get((k, c), 0.0) / st  # noqa: E731
            v = g("SQ_INSTS_VALU")
            if v <= 0:
                continue
            st64 = max(1.0, c64.get(k, 0))
            fp = sum(f64.get((k, c), 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                                     "SQ_INSTS_VALU_TRANS_F64")) / st64
            i32, i64, cvt = g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_VALU_INT64"), g("SQ_INSTS_VALU_CVT")
            f.write("%-24s VALU %.3e: fp64 %.3f  int32 %.3f  int64 %.3f  cvt %.3f  other %.3f | per VALU instruction: "
                    "SALU %.2f  SMEM %.3f  LDS %.3f  branch %.3f\n"
                    % (k, v, fp / v, i32 / v, i64 / v, cvt / v, max(0.0, 1.0 - (fp + i32 + i64 + cvt) / v),
                       g("SQ_INSTS_SALU") / v, g("SQ_INSTS_SMEM") / v, g("SQ_INSTS_LDS") / v, g("SQ_INSTS_BRANCH") / v))
    if clk_lines:
        f.write("\nengine clock under this load (separate pass, --pmc GRBM_GUI_ACTIVE; kernels on side streams overlap others and read high):\n")
        f.write("\n".join(clk_lines) + "\n")
    f.write("```\n")
lanes, cl = counters("pmc_lanes")


def kernel_rows(kernel_names, sq_, cs_, fetch_, cf_, write_, cw_, f64_, c64_, lanes_, cl_):
    """Per-step figures of one workload for latest_pmc.json (bench.py roofline_block reads them)."""
    rows = {}
    for k in kernel_names:
        def per(agg, calls, c):
            return agg.get((k, c), 0.0) / max(1.0, calls.get(k, 0))
        fb = per(fetch_, cf_, "FETCH_SIZE") * 1024 * (fetch_factor or 1.0)
        wb = per(write_, cw_, "WRITE_SIZE") * 1024 * (write_factor or 1.0)
        g = lambda c: per(f64_, c64_, c)  # noqa: E731
        rows[k] = {"hbm_bytes_per_step": fb + wb, "fetch_bytes": fb, "write_bytes": wb,
                   "fp64_wave_instr": g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
                                      + g("SQ_INSTS_VALU_TRANS_F64"),
                   "fp64_trans_wave_instr": g("SQ_INSTS_VALU_TRANS_F64"),   # v_rcp / v_rsq / v_sqrt_f64: 16 issue cycles each
                   "fp64_lane_ops_upper": 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")
                                                  + 2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")),
                   "valu_wave_instr": per(sq_, cs_, "SQ_INSTS_VALU"),
                   # lanes per VALU instruction = thread_cycles_valu / active_inst_valu (both in quad-cycles), of 64
                   "thread_cycles_valu": per(lanes_, cl_, "SQ_THREAD_CYCLES_VALU"),
                   "active_inst_valu": per(lanes_, cl_, "SQ_ACTIVE_INST_VALU")}
    return rows


pmc = {"source": f"profiles/{name}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_*, separate passes; "
                 "byte counters calibrated on 8-B-per-lane streams)",
       "source_hash": source_hash, "fetch_factor": fetch_factor, "write_factor": write_factor,
       "objects_per_step": n_obj,
       "kernels": kernel_rows(kernels, sq, cs, fetch, cf, write, cw, f64, c64, lanes, cl), "workloads": {}}
for k_, ghz in clock_ghz.items():
    if k_ in pmc["kernels"]:
        pmc["kernels"][k_]["engine_clock_ghz"] = ghz
pmc["workloads"]["headline"] = {"objects_per_step": n_obj, "kernels": pmc["kernels"]}
# the extra workloads of tools/profile_gpu.sh (bench.py --workload W: only that region ran in the process)
WORKLOAD_OBJECTS = {"long_arcs_100": 10000, "long_arcs_300": 10000}
try:
    extra = open(os.path.join(src, "extra_workloads.txt")).read().split()
except OSError:
    extra = []
with open(out, "a") as f:
    for w in extra:
        wsq, wcs = counters(f"w_{w}_sq")
        wf, wcf = counters(f"w_{w}_fetch")
        ww, wcw = counters(f"w_{w}_write")
        w64, wc64 = counters(f"w_{w}_f64")
        names = sorted({k for k, _ in wsq})
        if not names:
            continue
        rows = kernel_rows(names, wsq, wcs, wf, wcf, ww, wcw, w64, wc64, wsq, wcs)
        pmc["workloads"][w] = {"objects_per_step": WORKLOAD_OBJECTS.get(w, n_obj), "kernels": rows}
        f.write(f"\n## workload `{w}` (bench.py --workload {w} --steps 1 --warmup 0; per step)\n```\n")
        for k in names:
            r = rows[k]
            f.write("%-26s VALU wave-instr %.3e  lanes/instr %5.1f  fp64 share %.2f  fetch %8.1f MB  write %8.1f MB\n"
                    % (k, r["valu_wave_instr"], r["thread_cycles_valu"] / max(1.0, r["active_inst_valu"]),
                       r["fp64_wave_instr"] / max(1.0, r["valu_wave_instr"]), r["fetch_bytes"] / 1e6, r["write_bytes"] / 1e6))
        f.write("```\n")
json.dump(pmc, open(os.path.join(ROOT, "profiles", "latest_pmc.json"), "w"), indent=1)
print(open(out).read())
