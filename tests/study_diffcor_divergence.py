#!/usr/bin/env python3
"""Why two thirds of the bench population leave the differential corrector at Newton iteration 2 (CPU study, oracle only).

Not a test (pytest does not collect it): `python tests/study_diffcor_divergence.py [objects]` writes
profiles/round04_diffcor_divergence_study.json. The reference's rule (differential_orbit_correction/diff_cor.rs:354-358,
default ratio 1.5 at :183): an iteration whose normalised RMS — evaluated at the elements the PREVIOUS iteration
produced — is >= 1.5 x the previous iteration's RMS ends the fit with DifferentialCorrectionDiverged, and `fit_lsq`
keeps the IOD orbit (mod.rs:109). The RMS of iteration 2 is therefore the RMS after ONE Gauss-Newton step from the
IOD seed. For every object this script takes, from the oracle's corrector (platform-independent restated-glibc mode):

  rms_1   normalised RMS at the IOD seed                         (run with max_newton_iterations = 1)
  rms_2   normalised RMS after the first step                    (2 iterations, divergence check off)
  cond    2-norm condition number of the 6x6 normal matrix G^T W G at the seed (from its inverse, the covariance)
  weak    the eigenvector of the smallest eigenvalue (the direction the arc does not constrain), in element space
          scaled by the seed's magnitude (a, h, k, p, q, lambda)
  step    the first correction in units of the formal uncertainty: sqrt(dx^T N dx) = the reference's correction norm
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import outfit_amd  # noqa: E402,F401  (the ctypes mirrors of the ABI structs; no device is touched)
from outfit_amd import synth  # noqa: E402
from outfit_amd.api import Ephemeris, IODParams, ObsBatch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    ds = synth.make_dataset(n)
    batch, ephem = ObsBatch.from_dict(ds), Ephemeris.from_dict(ds["ephem"])
    params = synth.bench_params(IODParams())
    threads = len(os.sched_getaffinity(0))
    with O.math_flags(libm=0, glibc=1):
        rc, iod = O.fit_full_iod(params.as_c(), batch.as_c(), ephem.as_c(), n_threads=threads, hoist=1)
        assert rc == 0

        def lsq(**kw):
            cfg = O.default_diffcor_config()
            for k, v in kw.items():
                setattr(cfg, k, v)
            rc2, rec, _ = O.fit_lsq(params.as_c(), cfg, batch.as_c(), ephem.as_c(), iod, n_threads=threads)
            assert rc2 == 0
            return rec
        default = lsq()
        one = lsq(max_newton_iterations=1, enable_outlier_rejection=0)
        two = lsq(max_newton_iterations=2, rms_divergence_ratio=1e300, enable_outlier_rejection=0)
        free = lsq(rms_divergence_ratio=1e300)
    seeded = iod["status"] == 0
    ok = seeded & (one["status"] == 0) & (two["status"] == 0) & np.isfinite(one["normalised_rms"]) & np.isfinite(two["normalised_rms"])
    rms1, rms2 = one["normalised_rms"][ok], two["normalised_rms"][ok]
    ratio = rms2 / rms1
    cov = one["covariance"][ok]
    cond = np.array([np.linalg.cond(c) if np.isfinite(c).all() else np.nan for c in cov])
    # weakest direction of the normal matrix = largest-eigenvalue direction of the covariance, in relative element units
    scale = np.abs(np.concatenate([one["elem"][ok][:, :1], np.ones((ok.sum(), 5))], axis=1))   # a in AU, the rest O(1)
    weak = []
    for c, s in zip(cov, scale):
        if not np.isfinite(c).all():
            weak.append(np.full(6, np.nan))
            continue
        w, v = np.linalg.eigh(c / np.outer(s, s))
        d = np.abs(v[:, -1])
        weak.append(d / d.max())
    weak = np.array(weak)
    diverged_default = (default["status"][ok] == 1) & (default["dc_error"][ok] == 3)
    # the first step in the elements: a after one step against the seed's a, in units of the seed's sigma_a
    da = one["elem"][ok][:, 0] - iod["elem"][ok][:, 0]
    hyper = iod["kind"][ok] != 0
    q = lambda x, p: float(np.nanquantile(x, p))  # noqa: E731
    out = {
        "what": "bench population (synth.make_dataset, 5 nights over 20 days, 0.3 arcsec), oracle in the restated-glibc mode",
        "objects": int(n), "with_iod_seed": int(seeded.sum()), "studied": int(ok.sum()),
        "default_config": {"converged": int((default["status"] == 0).sum()), "kept_iod": int((default["status"] == 1).sum()),
                           "diverged": int(((default["status"] == 1) & (default["dc_error"] == 3)).sum()),
                           "iterations_histogram": np.bincount(default["total_newton_iterations"]).tolist()},
        "no_divergence_check": {"converged": int((free["status"] == 0).sum()),
                                "mean_iterations": float(free["total_newton_iterations"].mean())},
        "rms_at_seed": {"median": q(rms1, 0.5), "p10": q(rms1, 0.1), "p90": q(rms1, 0.9)},
        "rms_after_first_step_over_rms_at_seed": {"median": q(ratio, 0.5), "p10": q(ratio, 0.1), "p90": q(ratio, 0.9),
                                                  "share_ge_1.5": float((ratio >= 1.5).mean()),
                                                  "share_ge_10": float((ratio >= 10).mean())},
        "the_rule_is_exactly_that_ratio": bool(((ratio >= 1.5) == diverged_default).mean() > 0.999),
        "normal_matrix_condition_number": {"median": q(cond, 0.5), "p10": q(cond, 0.1), "p90": q(cond, 0.9), "max": q(cond, 1.0)},
        "condition_number_by_outcome": {"diverged_median": q(cond[diverged_default], 0.5),
                                        "not_diverged_median": q(cond[~diverged_default], 0.5)},
        "ratio_by_condition_number_tercile": [
            {"cond_range": [q(cond, lo), q(cond, hi)],
             "share_ratio_ge_1.5": float((ratio[(cond >= q(cond, lo)) & (cond <= q(cond, hi))] >= 1.5).mean())}
            for lo, hi in ((0.0, 1 / 3), (1 / 3, 2 / 3), (2 / 3, 1.0))],
        "weakest_direction_mean_abs_components": dict(zip(("a/|a|", "h", "k", "p", "q", "lambda"),
                                                          np.nanmean(weak, axis=0).round(3).tolist())),
        "first_step_in_a_AU": {"median_abs": q(np.abs(da), 0.5), "p90_abs": q(np.abs(da), 0.9)},
        "seed_rms_by_outcome": {"diverged_median": q(rms1[diverged_default], 0.5),
                                "not_diverged_median": q(rms1[~diverged_default], 0.5)},
        "seed_was_hyperbolic_or_parabolic": int(hyper.sum()),
    }
    path = os.path.join(ROOT, "profiles", "round04_diffcor_divergence_study.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
