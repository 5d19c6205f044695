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
