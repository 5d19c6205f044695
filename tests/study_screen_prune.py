"""Study (CPU, oracle): how much of K2c's residual loop a two-stage screen could skip on the bench workload.

For every candidate of `n` bench trajectories the per-observation terms of rms_orbit_error (trajectory.rs:352-427) are
evaluated with the oracle; then, per trajectory: stage 1 = the terms of `k` observations spread over the window, the
pilot = the candidate with the smallest stage-1 sum, bound = the pilot's full RMS; a candidate survives (needs its other
observations) when its stage-1 RMS (sum over the subset / full denominator) is below the bound.
    python tests/study_screen_prune.py [n] [k ...]
Test infrastructure: loads the oracle; nothing of the product is involved.

Result (150 trajectories, 14 381 candidates): nothing to skip. The candidates of a bench trajectory are all nearly as good
as its best one — full RMS / trajectory minimum: 1.12 / 1.23 / 1.40 / 1.61 / 1.96 at the 10 / 25 / 50 / 75 / 90 % quantiles —
so a bound from a pilot (median 1.06 - 1.30 x the minimum) prunes almost nobody: survivors 0.987 / 0.979 / 0.952 / 0.918 /
0.889 of the candidates after k = 2 / 3 / 4 / 5 / 6 observations. (The same holds for the reference's own
`prune_if_rms_ge`: its partial sums rarely cross the cutoff before the last observations.) K2c stays one full pass.
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tests.oracle_lib as O  # noqa: E402
from outfit_amd import synth  # noqa: E402
from outfit_amd._abi import IODParamsC  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ks = [int(x) for x in sys.argv[2:]] or [2, 3, 4, 5]
    L = O.lib()
    ds = synth.make_dataset(n)
    n_obs = 20
    p = IODParamsC()
    L.orc_iod_params_default(C.byref(p))
    synth.bench_params(p)
    helio = ds["obs_helio_equ"].reshape(-1, 3)
    terms_all = []  # per trajectory: (n_cand_ok, 20) terms
    for i in range(n):
        sl = slice(n_obs * i, n_obs * (i + 1))
        mjd, ra, dec = ds["mjd_tt"][sl], ds["ra"][sl], ds["dec"][sl]
        sra, sdec, hp = ds["sigma_ra"][sl], ds["sigma_dec"][sl], helio[sl]
        idx = (C.c_uint32 * 300)()
        w = (C.c_double * 100)()
        nt = L.orc_generate_triplet_indices(np.ascontiguousarray(mjd).ctypes.data_as(O._dp), n_obs, C.byref(p), idx, w)
        rows = []
        for t in range(nt):
            tri = [idx[3 * t + k] for k in range(3)]
            g = O.gauss_obs(tri, [ra[j] for j in tri], [dec[j] for j in tri], [mjd[j] for j in tri],
                            [list(hp[j]) for j in tri])
            res = O.GaussResult()
            if L.orc_prelim_orbit(C.byref(g), C.byref(p), C.byref(res)) != 0:
                continue
            eq = O.Equinoctial()
            if L.orc_to_equinoctial(C.byref(res.el), C.byref(eq)) != 0:
                continue
            row = np.full(n_obs, np.nan)
            for j in range(n_obs):
                term = C.c_double()
                if L.orc_ephemeris_error(C.byref(eq), float(mjd[j]), float(ra[j]), float(dec[j]), float(sra[j]),
                                         float(sdec[j]), O.D3(*hp[j]), C.byref(term)) != 0:
                    break
                row[j] = term.value
            if np.isfinite(row).all():
                rows.append(row)
        if rows:
            terms_all.append(np.array(rows))
    cands = sum(len(t) for t in terms_all)
    print(f"{len(terms_all)} trajectories, {cands} candidates with a finite RMS")
    full = np.concatenate([np.sqrt(t.sum(1) / 40.0) for t in terms_all])
    best = np.concatenate([np.full(len(t), np.sqrt(t.sum(1).min() / 40.0)) for t in terms_all])
    q = np.quantile(full / best, [0.1, 0.25, 0.5, 0.75, 0.9])
    print("full RMS / trajectory minimum, quantiles 10/25/50/75/90 %:", np.round(q, 2))
    for k in ks:
        sub = np.unique(np.round(np.linspace(0, n_obs - 1, k)).astype(int))
        surv = tot = 0
        pilot_ratio = []
        for t in terms_all:
            part = t[:, sub].sum(1)
            pilot = int(np.argmin(part))
            bound = t[pilot].sum()           # the pilot's full sum (same denominator for every candidate here)
            surv += int((part < bound).sum())
            tot += len(t)
            pilot_ratio.append(np.sqrt(bound / t.sum(1).min()))
        print(f"k = {k} (observations {list(sub)}): survivors {surv / tot:.3f} of the candidates, "
              f"pilot RMS / minimum: median {np.median(pilot_ratio):.3f}, 90 % {np.quantile(pilot_ratio, 0.9):.3f}")


if __name__ == "__main__":
    main()
