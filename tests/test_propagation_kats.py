"""`kepler::propagate_universal` against the known answers of the reference's own tests
(tests/golden/reference_kats_propagation.json, transcribed by tools/transcribe_propagation_kats.py from
src/kepler/propagation.rs:217-880): 4 real fink-fat states + 10 edge regimes with independent ground truth
(scipy DOP853 / mpmath) and the property tests (near-zero dt, continuity across a gap, degenerate state, warm vs
cold start, agreement of the three solver kinds). CPU: the oracle. GPU: the device through the C ABI, which must
also return the oracle's bits (same operation order, correctly-rounded functions)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import SolverType

HERE = os.path.dirname(os.path.abspath(__file__))
K = json.load(open(os.path.join(HERE, "golden", "reference_kats_propagation.json")))
CONV = 2.220446049250313e-14
KIND_NAMES = ["NewtonRaphson", "BrentDecker", "Auto"]


def oracle_propagate(pos, vel, t0, t1, solver):
    out = O.D11()
    sc = solver.as_c()
    st = O.lib().orc_propagate_universal_ex(O.D3(*pos), O.D3(*vel), t0, t1, C.byref(sc), out)
    return st, np.array(list(out))


def device_propagate(pos, vel, t0, t1, solver):
    out, st = outfit_amd.propagate_universal([pos], [vel], t0, t1, solver=solver)
    return int(st[0]), out[0]


def default_solver(**kw):
    return SolverType(kind="Auto", convergency=CONV, **kw)


BACKENDS = [pytest.param(oracle_propagate, id="oracle"),
            pytest.param(device_propagate, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("run", BACKENDS)
@pytest.mark.parametrize("case", K["cases"], ids=[c["name"] for c in K["cases"]])
def test_ground_truth_states(run, case):
    st, out = run(case["position"], case["velocity"], case["t0"], case["t1"], default_solver(psi_guess=case["psi_guess"]))
    assert st == 0
    assert np.linalg.norm(out[:3] - case["expected_r1"]) < case["tol"]
    assert np.linalg.norm(out[3:6] - case["expected_v1"]) < case["tol"]
    # every solver kind lands on the same state (kepler/params.rs:270-458 asserts this on a grid)
    for kind in KIND_NAMES:
        st2, o2 = run(case["position"], case["velocity"], case["t0"], case["t1"],
                      SolverType(kind=kind, convergency=CONV, psi_guess=case["psi_guess"]))
        assert st2 == 0, kind
        assert np.linalg.norm(o2[:3] - case["expected_r1"]) < case["tol"], kind
        assert np.linalg.norm(o2[3:6] - case["expected_v1"]) < case["tol"], kind
    # Lagrange identity f gdot - g fdot = 1 (propagation.rs doc, :100-103)
    assert abs(out[6] * out[9] - out[7] * out[8] - 1.0) < 1e-11


@pytest.mark.parametrize("run", BACKENDS)
def test_special_cases(run):
    sp = K["special"]
    c = sp["test_dt_near_zero_returns_initial_state"]
    st, out = run(c["position"], c["velocity"], *c["calls"][0], default_solver())
    assert st == 0 and np.linalg.norm(out[:3] - c["position"]) < 1e-8 and np.linalg.norm(out[3:6] - c["velocity"]) < 1e-8
    c = sp["test_continuity_across_lunation_gap"]
    (sa, a), (sb, b) = (run(c["position"], c["velocity"], *c["calls"][k], default_solver()) for k in (0, 1))
    assert sa == 0 and sb == 0
    assert np.linalg.norm(a[:3] - c["inline_expected"][0]) < 1e-9 and np.linalg.norm(b[:3] - c["inline_expected"][1]) < 1e-9
    assert np.linalg.norm(b[:3] - a[:3]) < 10.0 * np.linalg.norm(a[3:6]) * 0.2
    c = sp["test_degenerate_position_returns_err"]
    st, _ = run(c["position"], c["velocity"], *c["calls"][0], default_solver())
    assert st == 1   # OutfitError::DegenerateState (initial radius)
    c = sp["test_warm_start_consistent_with_cold_start"]
    st, base = run(c["position"], c["velocity"], *c["calls"][0], default_solver())
    assert st == 0
    _, warm = run(c["position"], c["velocity"], *c["calls"][1], default_solver(psi_guess=float(base[10])))
    _, cold = run(c["position"], c["velocity"], *c["calls"][2], default_solver())
    assert np.linalg.norm(warm[:3] - cold[:3]) < 1e-9 and np.linalg.norm(warm[3:6] - cold[3:6]) < 1e-9
    c = sp["test_solver_kind_agreement"]
    res = [run(c["position"], c["velocity"], *c["calls"][0], SolverType(kind=k, convergency=CONV))[1] for k in KIND_NAMES]
    for o in res[1:]:
        assert np.linalg.norm(o[:3] - res[0][:3]) < 1e-9 and np.linalg.norm(o[3:6] - res[0][3:6]) < 1e-9


def test_parabolic_newton_prelim_matches_cardano_on_monotonic_cubics():
    """prelim_parabolic.rs:519-560: both methods agree to 1e-9 on the monotonic cubic; Newton delegates to Cardano
    otherwise (sig0^2 > 2 r0)."""
    L = O.lib()
    mu = 0.01720209895 ** 2
    for r0, sig0, dt in ((1.0, 0.1, 3.0), (2.5, -0.4, -20.0), (0.7, 0.0, 50.0), (1.2, 3.0, 4.0)):
        vals = []
        for method in (0, 1):
            kp = O.KeplerParams(dt=dt, r0=r0, sig0=sig0, mu=mu, alpha=0.0, e0=1.0, convergency=1e-13, has_guess=0,
                                psi_guess=0.0, max_iter_prelim_kepuni=100, kind=0, parabolic_method=method)
            vals.append(L.orc_prelim_kepuni(C.byref(kp)))
        assert abs(vals[0] - vals[1]) < 1e-9 * max(1.0, abs(vals[0])), (r0, sig0, dt, vals)


@pytest.mark.gpu
def test_device_returns_the_oracle_bits_for_every_solver_kind():
    """2 000 random bound and unbound states x 3 solver kinds x cold / warm start: status and all 11 outputs identical
    to the oracle's (Brent-Dekker and the parabolic Newton start are plain arithmetic in the reference's order)."""
    rng = np.random.default_rng(12)
    n = 2000
    r = rng.normal(size=(n, 3)) * rng.uniform(0.3, 4.0, (n, 1))
    v = rng.normal(size=(n, 3)) * rng.uniform(0.002, 0.02, (n, 1))
    r[:3] = 0.0                                    # degenerate states
    dt = rng.uniform(-400.0, 400.0, n)
    for kind in KIND_NAMES:
        for guess in (None, 0.37):
            sv = SolverType(kind=kind, convergency=CONV, psi_guess=guess)
            out, st = outfit_amd.propagate_universal(r, v, 60000.0, 60000.0 + dt, solver=sv)
            for i in range(n):
                so, oo = oracle_propagate(r[i], v[i], 60000.0, 60000.0 + dt[i], sv)
                assert so == st[i], (kind, guess, i)
                if so == 0:
                    assert np.array_equal(oo.view(np.uint64), out[i].view(np.uint64)), (kind, guess, i)
