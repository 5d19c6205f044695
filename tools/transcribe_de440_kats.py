#!/usr/bin/env python3
"""Development-time transcription of the known answers the reference's tests hold for the legacy-DE reader and the
Chebyshev evaluation on the REAL DE440 file (src/jpl_ephem/horizon/horizon_data.rs:863-1173: test_jpl_reader_from_horizon,
test_get_record_from_horizon, test_get_record_index, test_interpolation_from_horizon, test_target_center_interpolation)
into tests/golden/reference_kats_de440.json — DATA only (inputs and expected values, every one an `assert_eq!`), read
from the test bodies as text. They run through tests/test_de440_gated.py when a DE440 file is present.

Usage (needs /root/reference, i.e. the build container): python tools/transcribe_de440_kats.py
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/jpl_ephem/horizon/horizon_data.rs"
NUM = r"[-+]?[0-9][0-9_]*\.?[0-9_]*(?:[eE][-+]?[0-9]+)?"


def nums(s):
    return [float(x.replace("_", "")) for x in re.findall(NUM, s)]


def body_of(text, name):
    m = re.search(rf"fn {name}\(\) \{{", text)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i]


def record(block):
    """HorizonRecord { start_jd, end_jd, x: vec![..], y: vec![..], z: vec![..] }"""
    out = {"start_jd": nums(re.search(r"start_jd:\s*([^,]+),", block).group(1))[0],
           "end_jd": nums(re.search(r"end_jd:\s*([^,]+),", block).group(1))[0]}
    for ax in "xyz":
        out[ax] = nums(re.search(rf"\b{ax}: vec!\[(.*?)\]", block, re.S).group(1))
    return out


def interp(block):
    """InterpResult { position: [[..]], velocity: Some([[..]]), acceleration: Some([[..]]) }"""
    return {k: nums(re.search(rf"{k}:\s*(?:Some\(\s*)?\[\[(.*?)\]\]", block, re.S).group(1))
            for k in ("position", "velocity", "acceleration")}


def main():
    text = open(SRC).read()
    out = {"source": "src/jpl_ephem/horizon/horizon_data.rs:863-1173 (test_jpl_reader_from_horizon, "
                     "test_get_record_from_horizon, test_get_record_index, test_interpolation_from_horizon, "
                     "test_target_center_interpolation); every value is an assert_eq! on what the reference reads from / "
                     "interpolates in the real DE440 file (linux_p1550p2650.440); transcribed by tools/transcribe_de440_kats.py"}
    t = body_of(text, "test_jpl_reader_from_horizon")
    ipt = nums(re.search(r"ipt: \[(.*?)\],\s*start_period", t, re.S).group(1))
    hdr = {"jpl_version": re.search(r'jpl_version: "(\w+)"', t).group(1),
           "ipt": [[int(v) for v in ipt[3 * i:3 * i + 3]] for i in range(15)]}
    for k in ("start_period", "end_period", "period_lenght", "earth_moon_mass_ratio"):
        hdr[k] = nums(re.search(rf"{k}: ([^,]+),", t).group(1))[0]
    hdr["recsize"] = int(nums(re.search(r"recsize: (\d+)", t).group(1))[0])
    hdr["n_records"] = int(re.search(r"records\.len\(\), (\d+)\)", t).group(1))
    hdr["sum_of_bodies_per_record"] = int(re.search(r"x\.len\(\)\),\s*(\d+)\s*\)", t).group(1))
    out["header"] = hdr
    out["first_record"] = {"block": 0, "ipt_row": 0, "sub_interval": 0, "what": "records[0].get(&0)[0] (Mercury)",
                           "record": record(t[t.index("records[0].get(&0)"):])}
    t = body_of(text, "test_get_record_from_horizon")
    out["record_at_epoch"] = {"ipt_row": int(re.search(r"get_record_horizon\((\d+),", t).group(1)),
                              "mjd_tt": nums(re.search(r"get_record_horizon\(\d+, ([^)]+)\)", t).group(1))[0],
                              "tau": nums(re.search(r"assert_eq!\(tau, ([^)]+)\)", t).group(1))[0],
                              "what": "get_record_horizon(4, ..) (Jupiter)", "record": record(t)}
    t = body_of(text, "test_get_record_index")
    out["record_index"] = {"mjd_tt": nums(re.search(r"get_record_index\(([^)]+)\)", t).group(1))[0],
                           "index": int(re.search(r"assert_eq!\(index, (\d+)\)", t).group(1)),
                           "tau": nums(re.search(r"assert_eq!\(tau, ([^)]+)\)", t).group(1))[0]}
    t = body_of(text, "test_interpolation_from_horizon")
    out["sun_wrt_ssb_km"] = []
    for m in re.finditer(r"get_record_horizon\((\d+), ([^)]+)\)(.*?)(?=let horizon_data|\Z)", t, re.S):
        assert m.group(1) == "10"
        out["sun_wrt_ssb_km"].append(dict(mjd_tt=nums(m.group(2))[0], n_subintervals=2, **interp(m.group(3))))
    t = body_of(text, "test_target_center_interpolation")
    rows = {"Earth": 2, "Mars": 3, "Sun": 10}
    out["target_wrt_center_au"] = []
    for m in re.finditer(r"HorizonID::(\w+),\s*HorizonID::(\w+),\s*([^,]+),\s*true,\s*true,\s*\);(.*?)(?=let horizon_data|\Z)", t, re.S):
        out["target_wrt_center_au"].append(dict(target=m.group(1), center=m.group(2), target_row=rows[m.group(1)],
                                                center_row=rows[m.group(2)], mjd_tt=nums(m.group(3))[0], **interp(m.group(4))))
    # the two Earth-Sun cases are what the hot path evaluates (positions only): kept under their round-2 key too
    out["earth_wrt_sun_au"] = [{"mjd_tt": c["mjd_tt"], "position": c["position"]} for c in out["target_wrt_center_au"]
                               if c["target"] == "Earth"]
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_de440.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (len(v) if isinstance(v, list) else "ok") for k, v in out.items() if k != "source"})


if __name__ == "__main__":
    main()
