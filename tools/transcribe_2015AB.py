#!/usr/bin/env python3
"""Development-time transcription of the observation file the reference's own tests hold
(/root/reference/tests/data/2015AB.obs, MPC 80-column) into tests/golden/reference_2015AB.json: the 37 epochs
(TT) and positions of the trajectory `DATASET_2015AB` materialises as "K09R05F" (src/lib.rs:465-468 — the file is
one linked object, 2009 RF5 = 2015 AB), plus the known answers of src/trajectory.rs:566-625
(`test_select_rms_interval`): the RMS windows of the best triplet for three (extf, dtmax) settings. Those answers
depend only on the epochs, the triplet generator/weights/heap (a3) and `select_rms_interval` (a15).

UTC -> TT: + 32.184 s + (TAI - UTC), 34 s from 2009-01-01, 35 s from 2012-07-01 (the check in SURVEY Appendix B:
2015-01-06.47852 -> 57028.479297592596, cf. src/initial_orbit_determination/gauss.rs:1438).
"""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/tests/data/2015AB.obs"


def mjd_of_date(y, m, d):
    a = (14 - m) // 12
    yy, mm = y + 4800 - a, m + 12 * a - 3
    jdn = d + (153 * mm + 2) // 5 + 365 * yy + yy // 4 - yy // 100 + yy // 400 - 32045
    return jdn - 2400001  # JDN at noon -> MJD at 0h


def main():
    rows = []
    for ln in open(SRC):
        if len(ln) < 56:
            continue
        date = ln[15:32]
        y, m, dfrac = int(date[0:4]), int(date[5:7]), float(date[8:].strip())
        day = int(dfrac)
        mjd_utc = mjd_of_date(y, m, day) + (dfrac - day)
        tai_utc = 34.0 if (y, m) < (2012, 7) else 35.0
        mjd_tt = mjd_utc + (32.184 + tai_utc) / 86400.0
        ra_s, dec_s = ln[32:44], ln[44:56]
        h, mi, s = float(ra_s[0:2]), float(ra_s[3:5]), float(ra_s[6:].strip())
        ra = math.radians(15.0 * (h + mi / 60.0 + s / 3600.0))
        sign = -1.0 if dec_s[0] == "-" else 1.0
        dd, dm, ds = float(dec_s[1:3]), float(dec_s[4:6]), float(dec_s[7:].strip())
        dec = sign * math.radians(dd + dm / 60.0 + ds / 3600.0)
        rows.append({"designation": ln[5:12], "mjd_tt": mjd_tt, "ra": ra, "dec": dec, "observatory": ln[77:80]})
    assert len(rows) == 37
    assert abs(rows[23]["mjd_tt"] - 57028.479297592596) < 1e-9, rows[23]["mjd_tt"]   # gauss.rs:1438
    out = {"source": "tests/data/2015AB.obs (MPC 80-column data file of the reference's tests) + src/trajectory.rs:566-625",
           "note": "mjd_tt here is reproduced by hand (UTC + leap seconds); the reference goes through hifitime, "
                   "ns-quantised — equal to <= 1e-9 d, far below the day-scale thresholds the known answers test",
           "observations": rows,
           "select_rms_interval": {
               "params": {"dt_min": 0.03, "dt_max_triplet": 150.0, "optimal_interval_time": 20.0,
                          "max_obs_for_triplets": 37, "max_triplets": 10},
               "cases": [{"extf": -1.0, "dtmax": 30.0, "window": [0, 36]},
                         {"extf": 10.0, "dtmax": 30.0, "window": [14, 36]},
                         {"extf": 0.001, "dtmax": 3.0, "window": [17, 33]}]}}
    path = os.path.join(ROOT, "tests", "golden", "reference_2015AB.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
