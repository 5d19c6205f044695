"""The bench line's `roofline` object, without a GPU: the contract keys, the arithmetic between them, and what the
counter-derived fields do when the committed PMC profile belongs to this build or to another one."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KT = {"k_triplet_select": (80.0, 80), "k_observer_prepass": (70.0, 40), "k_gauss_roots": (140.0, 80),
      "k_gauss_correct_fast": (128.0, 20), "k_gauss_correct": (10.0, 20), "k_orbit_rms": (58.0, 20),
      "k_orbit_rms_slow": (1.4, 20), "k_select_best": (3.6, 20), "k_refine_best": (12.6, 20)}


def test_contract_keys_and_arithmetic():
    r = bench.roofline_block(KT, 100000, 20, 17.4e-3, 1856.0, "headline", objects_per_s_per_gpu=5.7e6)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
    assert r["kernel"] == "k_gauss_roots"                         # dominant MAIN-stream kernel by summed time
    avg_ms = 140.0 / 80
    assert abs(r["avg_launch_ms"] - avg_ms) < 1e-12 and r["units_per_launch"] == 100000 * 20 / 80
    assert abs(r["achieved"] - 25000 * 1856.0 / (avg_ms * 1e-3) / 1e9) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-15 and r["frac"] < 0.01
    # nothing that is not a utilisation is called `frac`
    assert "frac" not in (r["reference_equivalent"] or {})
    assert "reference_equivalent_tflops" in r["reference_equivalent"]


def test_counter_fields_follow_the_source_hash(monkeypatch):
    pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
    matches = pmc.get("source_hash") == bench.source_hash()
    r = bench.roofline_block(KT, 100000, 20, 17.4e-3, 1856.0, "headline")
    if matches:   # the committed profile was taken from this build: the line says what limits the path
        assert r["counters"]["source_hash"] == bench.source_hash() and r["traffic"] > 0
        assert 0.3 < r["issue"]["step_issue_frac"] < 1.0
        # measured issue costs (4 cycles, TRANS 16) and the measured clock raise the fraction, never above 1
        assert r["issue"]["step_issue_frac"] < r["issue"]["step_issue_cycle_frac"] < 1.0
        assert r["issue"]["step_issue_cycle_frac"] < r["issue"]["step_issue_cycle_frac_at_measured_clock"] < 1.0
        assert 0.1 < r["fp64_executed"]["frac"] < 1.0
        assert r["step_traffic"]["ratio"] > 1.0
        rows = {row["kernel"]: row for row in r["per_kernel"]}
        for k in ("k_gauss_roots", "k_gauss_correct_fast", "k_orbit_rms"):
            assert 30.0 < rows[k]["lanes_per_instruction"] <= 64.0 and 0.0 < rows[k]["issue_frac"] < 1.2
            assert rows[k]["issue_frac"] < rows[k]["issue_cycle_frac"] < rows[k]["issue_cycle_frac_at_measured_clock"] < 1.2
            assert 1.8 < rows[k]["clock_ghz_under_load"] <= 2.4
        for w in ("noise_default", "noise_8d", "lsq_default", "lsq_ratio10", "lsq_no_divergence_check"):
            assert w in pmc["workloads"], w
    else:
        assert r["counters"] is None and r["traffic"] is None and r["issue"] is None
    # a profile of another build says nothing about this code
    monkeypatch.setattr(bench, "source_hash", lambda: "0000000000000000")
    r = bench.roofline_block(KT, 100000, 20, 17.4e-3, 1856.0, "headline")
    assert r["counters"] is None and r["traffic"] is None and r["issue"] is None and r["per_kernel"] is None
    assert r["frac"] > 0                                         # the contract figure needs no profile


def test_corrector_block_is_about_the_corrector():
    r = bench.roofline_block({"k_diffcor": (45.0, 20), "k_gauss_roots": (140.0, 80)}, 100000, 20, 2.27e-3, 2496.0,
                             "lsq_default", exclude=(), only=("k_diffcor",))
    assert r["kernel"] == "k_diffcor" and set(r["all_kernels_avg_ms_per_step"]) >= {"k_diffcor"}


def _full_record():
    """A full record as main() assembles it, from canned inputs (no GPU): the largest shape the line is built from."""
    roof = bench.roofline_block(KT, 100000, 20, 17.4e-3, 1856.0, "headline", objects_per_s_per_gpu=5.7e6)
    lsq_roof = bench.roofline_block({"k_diffcor": (45.0, 20)}, 100000, 20, 2.27e-3, 2496.0, "lsq_default", exclude=(),
                                    only=("k_diffcor",))
    mode = {"orbits": 1986, "integer_fields_identical": True, "bit_identical_elements": 1986, "bit_identical_rms": 1986,
            "byte_identical_records": 2000, "beyond_1e-10_AU": 0, "beyond_1e-12_rad": 0, "max_d_len": 0.0,
            "max_d_angle": 0.0, "max_rel_d_rms": float("nan")}
    cpu = {"value": 2404.4, "unit": "objects/s", "cores": 16, "kind": "port", "sample": "first 34913 objects " + "x" * 150}
    noise = {"what": "y" * 200, "roofline": roof, "objects_per_s": 3.07e6, "ms_per_step": 32.5,
             "kernel_avg_ms_per_step": {k: 1.0 for k in KT}, "isolated_run_kernel_ms": {k: 1.0 for k in KT}}
    lsq = {"what": "z" * 200, "ms_per_step": 2.27, "objects_per_s_stage": 4.4e7, "objects_per_s_iod_plus_lsq": 5.1e6,
           "roofline": lsq_roof, "newton_iteration_histogram": {str(i): i for i in range(21)}}
    return {
        "metric": "objects/s full Gauss IOD (20 obs, 100 triplets/obj)", "value": 5860410.123456789, "unit": "objects/s",
        "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 17.063712345678, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "math_mode": "glibc", "host_libm": "glibc 2.35 " + "h" * 60,
        "config": {"workload": "100000 synthetic objects per GPU x 20 obs, 100 triplets each " + "w" * 100,
                   "objects_per_gpu": [100000] * 8, "obs_per_object": 20.0, "triplets_per_object": 100,
                   "n_noise_realizations": 0, "orbits_found_rank0": 99485, "math_mode": "m" * 400,
                   "parallelism": "objects sharded over 8 GPU(s); RCCL all_gather of 96-B results, overlapped with the next step"},
        "roofline": roof,
        "math_modes": {"headline": "glibc", "glibc": {"value": 5.86e6, "kernel_avg_ms_per_step": dict.fromkeys(KT, 1.0)},
                       "cr": {"value": 5.97e6, "kernel_avg_ms_per_step": dict.fromkeys(KT, 1.0)}},
        "parity_vs_libm": {"sample": "first 2000 objects of the timed batch", "host_libm_is_the_pinned_build": True,
                           "meaning": "p" * 300, "glibc": mode, "cr": mode},
        "gather_check": {"ranks_verified_by_every_rank": 8, "records": 800000, "crc32_per_rank": [2 ** 32 - 1] * 8,
                         "per_rank": {k: [17.123456789] * 8 for k in ("ms_per_step", "gather_wait_ms_per_step",
                                                                       "host_enqueue_ms_per_step",
                                                                       "main_stream_kernel_ms_per_step",
                                                                       "setup_s_before_timed_region")},
                         "ipc_mode": "HSA_ENABLE_IPC_MODE_LEGACY=0"},
        "fit_lsq": dict(lsq, workloads={"ratio10": lsq, "no_divergence_check": lsq}, cpu_baseline=cpu),
        "side": {"isolated_run_ms": 18.7, "blocking_call_ms": 21.2, "what": "s" * 400,
                 "configs1_1000_objects": {"ms_per_step_back_to_back": 1.0, "objects_per_s": 1e6, "blocking_call_ms": 1.47},
                 "noise": {"iod_params_default": noise, "survey_8d": noise},
                 "long_arcs": {f"{n}_obs_{t}": noise for n in (100, 300) for t in ("iod_params_default", "100_triplets_no_noise")}},
        "cpu_baseline": cpu, "cpu_baseline_1thread": cpu, "cpu_baseline_hoisted": cpu,
    }


def test_the_printed_line_is_one_short_strict_json_line():
    """BENCH_r04.json.parsed was null: the line had grown to 47 KB and the driver keeps an 8 KB tail of stdout. The
    printed line is built by compact_line() from the full record; the rest goes to the detail file it names."""
    full = _full_record()
    assert len(json.dumps(full)) > 8192                      # the canned record is as large as the real one was
    line = bench.compact_line(full, "gpurun_out/bench_detail_n8.json")
    text = json.dumps(line, allow_nan=False)                 # strict JSON: a NaN in the record becomes null
    assert len(text) <= bench.LINE_BYTE_BUDGET < 8192 and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]      # not rounded
    assert set(back["config"]) >= {"workload", "objects_per_gpu", "parallelism"} and "model" not in back["config"]
    r = back["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "units_per_launch",
              "algorithmic_bytes_per_object"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * r["frac"]
    assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert back["parity_vs_libm"]["byte_identical_records"] == 2000
    assert back["side"]["fit_lsq"]["ms_per_step"] == 2.27 and back["side"]["noise_default_objects_per_s"] == 3.07e6
    assert len(back["side"]["long_arcs_ms_per_10000_objects"]) == 4


def test_an_oversized_line_sheds_its_optional_parts_not_the_contract(monkeypatch):
    full = _full_record()
    monkeypatch.setattr(bench, "LINE_BYTE_BUDGET", 2600)
    line = bench.compact_line(full, "d.json")
    assert len(json.dumps(line)) <= 2600 and "side" not in line
    assert "roofline" in line and "cpu_baseline" in line and "value" in line


def test_detail_file_holds_the_full_record(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _full_record()
    full = json.loads(json.dumps(full).replace("NaN", "null"))
    rel = bench.write_detail(full)
    assert rel == os.path.join("gpurun_out", "bench_detail_n8.json")
    assert json.load(open(tmp_path / rel)) == full
