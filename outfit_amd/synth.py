"""Deterministic synthetic many-object datasets for the Gauss-IOD hot path.

Not part of the product path: this is the input generator used by tests/ and
bench.py. The real inputs (DE440, EOP, MPC
observations) are not available offline, so the workload is reproduced with

* Keplerian truth orbits  a~U[1.8,3.6] AU, e~U[0,0.35], i~Rayleigh(8 deg)<=40 deg,
  Omega, omega, M ~ U[0,2pi), epoch MJD 60000;
* N = 20 epochs per object = 5 nights x 4 exposures (nights at +0,3,7,12,20 d
  +U[0,0.2] d, exposures 0.02 d apart), first night at 60000 + U[0, start_spread);
* astrometric noise sigma_ra = sigma_dec = 0.3 arcsec;
* Earth from a synthetic Chebyshev table with the DE440 record layout
  (32-day blocks; EMB 13x2, Moon 13x8, Sun 11x2 coefficients — the IPT at
  /root/reference/src/jpl_ephem/horizon/horizon_data.rs:869-885) fitted to a
  Keplerian Earth–Moon model.

All randomness comes from one SplitMix64 stream (seed 20260101), implemented
here in numpy so the datasets are reproducible bit for bit.
"""
import hashlib
import os

import numpy as np

AU_KM = 149597870.7
GAUSS_GRAV = 0.01720209895
MU = GAUSS_GRAV * GAUSS_GRAV
VLIGHT_AU = 2.99792458e5 / AU_KM * 86400.0
COS_EPS = 9.174820620691818e-1
SIN_EPS = 3.977771559319137e-1
EMRAT = 81.30056822149722
ARCSEC = np.pi / 648000.0
DEFAULT_SEED = 20260101

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class SplitMix64:
    """Vectorised SplitMix64: draw k = state + (i+1)*gamma, then the usual mix."""

    def __init__(self, seed):
        self.state = np.uint64(seed)

    def next_u64(self, n):
        gamma = np.uint64(0x9E3779B97F4A7C15)
        with np.errstate(over="ignore"):
            i = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + i * gamma
            self.state = np.uint64(self.state + np.uint64(n) * gamma)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, n):
        """U[0,1) with 53 random bits."""
        return (self.next_u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        """Box–Muller on two uniform streams."""
        u1 = 1.0 - self.uniform(n)
        u2 = self.uniform(n)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ---------------------------------------------------------------------------
# synthetic DE-layout Chebyshev table
# ---------------------------------------------------------------------------
NCOEF = (13, 13, 11)   # EMB, Moon, Sun   (DE440 IPT rows 2, 9, 10)
NSUB = (2, 8, 2)
BLOCK_DAYS = 32.0
PER_BLOCK = sum(3 * c * s for c, s in zip(NCOEF, NSUB))  # 456 doubles = 3648 B


def _kepler_E(M, e, iters=12):
    E = M + e * np.sin(M)
    for _ in range(iters):
        E = E - (E - e * np.sin(E) - M) / (1.0 - e * np.cos(E))
    return E


def _ecl_to_equ(v):
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    return np.stack([x, COS_EPS * y - SIN_EPS * z, SIN_EPS * y + COS_EPS * z], axis=-1)


def _model_bodies_km(jd):
    """Keplerian Earth–Moon model, equatorial J2000, km. jd: array of TT Julian dates.
    Returns EMB wrt SSB, Moon wrt Earth, Sun wrt SSB."""
    t = jd - 2451545.0
    # EMB heliocentric ellipse (ecliptic plane)
    a, e = 1.00000261, 0.01671123
    n = GAUSS_GRAV / a ** 1.5
    varpi = np.deg2rad(102.93768193)
    M = np.deg2rad(100.46457166 - 102.93768193) + n * t
    E = _kepler_E(np.mod(M, 2 * np.pi), e)
    xp = a * (np.cos(E) - e)
    yp = a * np.sqrt(1 - e * e) * np.sin(E)
    emb_h = np.stack([xp * np.cos(varpi) - yp * np.sin(varpi),
                      xp * np.sin(varpi) + yp * np.cos(varpi), np.zeros_like(xp)], axis=-1)
    # Sun wobble about the barycentre (a Jupiter-like and a Saturn-like term)
    sun = np.zeros_like(emb_h)
    for amp, period, ph in ((4.96e-3, 4332.589, 0.6), (2.7e-3, 10759.22, 2.1)):
        ang = 2 * np.pi * t / period + ph
        sun[..., 0] += -amp * np.cos(ang)
        sun[..., 1] += -amp * np.sin(ang)
    emb = emb_h + sun
    # Moon geocentric: inclined circle with a small elliptic term
    am = 384400.0 / AU_KM
    lm = np.deg2rad(218.316) + 2 * np.pi * t / 27.321661
    mm = np.deg2rad(134.963) + 2 * np.pi * t / 27.554550
    lon = lm + 0.1098 * np.sin(mm)
    rr = am * (1 - 0.0549 * np.cos(mm))
    lat = np.deg2rad(5.145) * np.sin(np.deg2rad(93.272) + 2 * np.pi * t / 27.212221)
    moon = np.stack([rr * np.cos(lat) * np.cos(lon), rr * np.cos(lat) * np.sin(lon), rr * np.sin(lat)], -1)
    return _ecl_to_equ(emb) * AU_KM, _ecl_to_equ(moon) * AU_KM, _ecl_to_equ(sun) * AU_KM


def make_ephemeris(mjd_start=59984.0, n_blocks=12):
    """Chebyshev table in the layout of include/outfit_hip.h (outfit_ephem_t).

    Returns a dict with jd_start, jd_end, block_days, emrat, ncoef, nsub,
    n_blocks and `coef` (float64, n_blocks*456)."""
    jd_start = 2400000.5 + mjd_start
    coef = np.empty((n_blocks, PER_BLOCK), dtype=np.float64)
    for nb in range(n_blocks):
        b0 = jd_start + nb * BLOCK_DAYS
        off = 0
        for body in range(3):
            nc, ns = NCOEF[body], NSUB[body]
            sub_len = BLOCK_DAYS / ns
            k = np.arange(nc)
            nodes = np.cos(np.pi * (k + 0.5) / nc)  # Chebyshev nodes: interpolation of degree nc-1
            for s in range(ns):
                jd = b0 + (s + 0.5 * (nodes + 1.0)) * sub_len
                vals = _model_bodies_km(jd)[body]  # (nc, 3)
                for ax in range(3):
                    c = np.polynomial.chebyshev.chebfit(nodes, vals[:, ax], nc - 1)
                    coef[nb, off:off + nc] = c
                    off += nc
    return {
        "jd_start": jd_start, "jd_end": jd_start + n_blocks * BLOCK_DAYS,
        "block_days": BLOCK_DAYS, "emrat": EMRAT, "ncoef": NCOEF, "nsub": NSUB,
        "n_blocks": n_blocks, "coef": np.ascontiguousarray(coef.reshape(-1)),
    }


def earth_equ_au(ephem, mjd_tt):
    """numpy evaluation of the table: Earth wrt Sun, equatorial J2000, AU (generator side).
    Mirrors the block / sub-interval arithmetic of horizon_data.rs:711-777."""
    mjd_tt = np.asarray(mjd_tt, dtype=np.float64)
    et_trunc = np.trunc(mjd_tt)
    et_jd = 2400000.5 + et_trunc
    nr = np.floor((et_jd - ephem["jd_start"]) / ephem["block_days"]).astype(np.int64)
    if np.any(nr < 0) or np.any(nr >= ephem["n_blocks"]):
        raise ValueError("epoch outside the synthetic ephemeris")
    tau = ((et_jd - (nr * ephem["block_days"] + ephem["jd_start"])) + (mjd_tt - et_trunc)) / ephem["block_days"]
    coef = ephem["coef"].reshape(ephem["n_blocks"], PER_BLOCK)
    bodies = []
    off = 0
    for body in range(3):
        nc, ns = ephem["ncoef"][body], ephem["nsub"][body]
        si = np.minimum(np.floor(tau * ns), ns - 1).astype(np.int64)
        tc = 2.0 * np.mod(ns * tau, 1.0) - 1.0
        T = np.empty((nc,) + tc.shape)
        T[0] = 1.0
        T[1] = tc
        for i in range(2, nc):
            T[i] = 2.0 * tc * T[i - 1] - T[i - 2]
        pos = np.empty(tc.shape + (3,))
        for ax in range(3):
            idx = off + si * 3 * nc + ax * nc
            c = coef[nr[..., None], idx[..., None] + np.arange(nc)]  # (..., nc)
            pos[..., ax] = np.einsum("...k,k...->...", c, T)
        bodies.append(pos)
        off += 3 * nc * ns
    earth = bodies[0] - bodies[1] / (1.0 + ephem["emrat"]) - bodies[2]
    return earth / AU_KM


# ---------------------------------------------------------------------------
# objects and observations
# ---------------------------------------------------------------------------
def _propagate_kepler_ecl(a, e, inc, node, argp, m0, dt):
    n = GAUSS_GRAV / a ** 1.5
    M = np.mod(m0 + n * dt, 2 * np.pi)
    E = _kepler_E(M, e, 20)
    xp = a * (np.cos(E) - e)
    yp = a * np.sqrt(1 - e * e) * np.sin(E)
    cw, sw, cn, sn, ci, si = np.cos(argp), np.sin(argp), np.cos(node), np.sin(node), np.cos(inc), np.sin(inc)
    x = (cn * cw - sn * sw * ci) * xp + (-cn * sw - sn * cw * ci) * yp
    y = (sn * cw + cn * sw * ci) * xp + (-sn * sw + cn * cw * ci) * yp
    z = (sw * si) * xp + (cw * si) * yp
    return np.stack([x, y, z], axis=-1)


def night_offsets(n_nights):
    """Day offsets of the observing nights: the bench cadence {0, 3, 7, 12, 20, 27, 35, 44}, continued for long arcs
    with gaps cycling through 3, 5, 4, 8 days (25 nights span 129 days, 50 nights 254 days)."""
    off = [0.0, 3.0, 7.0, 12.0, 20.0, 27.0, 35.0, 44.0]
    gaps = (3.0, 5.0, 4.0, 8.0)
    while len(off) < n_nights:
        off.append(off[-1] + gaps[(len(off) - 8) % 4])
    return np.array(off[:n_nights])


def make_long_arc_dataset(n_obj, n_obs, seed=DEFAULT_SEED + 1000, **kw):
    """Long trajectories: n_obs = 100 -> 25 nights x 4 exposures (129-day arcs, exactly
    `IODParams::default().max_obs_for_triplets`: the cap is not hit), n_obs = 300 -> 50 nights x 6 exposures (254-day
    arcs: down-sampling to 100 epochs, quirk Q1, and the dt_max_triplet = 150 d window really cuts). The Chebyshev
    table is extended to 24 blocks (768 days) so that every epoch is inside it."""
    n_exp = 4 if n_obs <= 128 else 6
    assert n_obs % n_exp == 0, "n_obs must be a multiple of the exposures per night"
    ephem = kw.pop("ephem", None) or make_ephemeris(n_blocks=24)
    return make_dataset(n_obj, seed=seed, ephem=ephem, n_nights=n_obs // n_exp, n_exp=n_exp, **kw)


def make_dataset(n_obj, seed=DEFAULT_SEED, ephem=None, topocentric=False, n_nights=5, n_exp=4,
                 start_spread=300.0, sigma_arcsec=0.3, night_scale=1.0, a_min=1.8, a_span=1.8, e_max=0.35):
    """Synthetic batch in the SoA layout of outfit_obs_batch_t.

    Returns dict(traj_offset, mjd_tt, ra, dec, sigma_ra, sigma_dec, obs_helio_equ,
    obs_geo_ecl, truth) — all float64/uint64 numpy arrays, observations time-sorted
    per trajectory. `night_scale` stretches the spacing of the nights (long arcs), `a_min / a_span / e_max` move
    the population of truth orbits (fast movers, eccentric orbits); the defaults are the bench population."""
    if ephem is None:
        ephem = make_ephemeris()
    rng = SplitMix64(seed)
    a = a_min + a_span * rng.uniform(n_obj)
    e = e_max * rng.uniform(n_obj)
    inc = np.minimum(np.deg2rad(8.0) * np.sqrt(-2.0 * np.log(1.0 - rng.uniform(n_obj))), np.deg2rad(40.0))
    node = 2 * np.pi * rng.uniform(n_obj)
    argp = 2 * np.pi * rng.uniform(n_obj)
    m0 = 2 * np.pi * rng.uniform(n_obj)
    start = 60000.0 + start_spread * rng.uniform(n_obj)
    night_off = night_scale * night_offsets(n_nights)
    jitter = (0.2 * night_scale) * rng.uniform(n_obj * n_nights).reshape(n_obj, n_nights)
    nights = start[:, None] + night_off[None, :] + jitter
    mjd = (nights[:, :, None] + 0.02 * np.arange(n_exp)[None, None, :]).reshape(n_obj, n_nights * n_exp)
    n_obs = n_nights * n_exp

    earth = earth_equ_au(ephem, mjd)  # (n_obj, n_obs, 3)
    if topocentric:
        # fixed body-like offset of one Earth radius, direction per object (exercises obs_geo_ecl)
        th = 2 * np.pi * rng.uniform(n_obj)
        ph = np.arcsin(2 * rng.uniform(n_obj) - 1)
        r_e = 6378.137 / AU_KM
        g = r_e * np.stack([np.cos(ph) * np.cos(th), np.cos(ph) * np.sin(th), np.sin(ph)], axis=-1)
        geo_ecl = np.repeat(g[:, None, :], n_obs, axis=1)
    else:
        geo_ecl = np.zeros((n_obj, n_obs, 3))
    obs_helio = earth + _ecl_to_equ(geo_ecl)

    el = [x[:, None] for x in (a, e, inc, node, argp, m0)]
    dt = mjd - 60000.0
    lt = np.zeros_like(dt)
    for _ in range(3):  # light-time iteration
        ast = _ecl_to_equ(_propagate_kepler_ecl(*el, dt - lt))
        rho = ast - obs_helio
        lt = np.linalg.norm(rho, axis=-1) / VLIGHT_AU
    sigma = sigma_arcsec * ARCSEC
    dec = np.arctan2(rho[..., 2], np.hypot(rho[..., 0], rho[..., 1]))
    ra = np.mod(np.arctan2(rho[..., 1], rho[..., 0]), 2 * np.pi)
    z1 = rng.normal(n_obj * n_obs).reshape(n_obj, n_obs)
    z2 = rng.normal(n_obj * n_obs).reshape(n_obj, n_obs)
    ra = np.mod(ra + sigma * z1 / np.cos(dec), 2 * np.pi)
    dec = dec + sigma * z2

    flat = lambda x: np.ascontiguousarray(x.reshape(-1))  # noqa: E731
    return {
        "n_traj": n_obj,
        "traj_offset": np.arange(0, (n_obj + 1) * n_obs, n_obs, dtype=np.uint64),
        "mjd_tt": flat(mjd), "ra": flat(ra), "dec": flat(dec),
        "sigma_ra": np.full(n_obj * n_obs, sigma), "sigma_dec": np.full(n_obj * n_obs, sigma),
        "obs_helio_equ": flat(obs_helio), "obs_geo_ecl": flat(geo_ecl),
        "truth": np.stack([a, e, inc, node, argp, m0], axis=-1),
        "ephem": ephem,
    }


# (longitude deg east, rho cos phi', rho sin phi') — MPC-style parallax constants of a few real sites
SITES = np.array([
    [203.74409, 0.936241, 0.351543],   # Haleakala
    [289.26345, 0.864981, -0.500958],  # Cerro Pachon
    [243.14022, 0.836325, 0.546877],   # Palomar
    [6.92124, 0.723655, 0.687873],     # Caussols
    [149.06119, 0.855632, -0.516198],  # Siding Spring
])
ERAU = 6378.137 / AU_KM  # Earth radius in AU (reference constants.rs:61-67)


def add_sites(ds, seed=DEFAULT_SEED + 7, tt_minus_ut1=69.184 - 0.02):
    """Give every observation of `ds` an observing site and a UT1 epoch — the inputs of the device pre-pass
    (outfit_obs_batch_t.obs_body_fixed / .mjd_ut1) instead of precomputed observer positions.

    One site per trajectory; the body-fixed vector is [rho cos phi' cos lon, rho cos phi' sin lon, rho sin phi']
    x Earth radius in AU — `Observer::earth_fixed_position` of the reference's observer dependency; its value for
    Haleakala is pinned by src/cache/observer_centric_cache.rs:241-261 and reproduced by SITES[0] to 4 digits. UT1 = TT - const
    (the reference goes through hifitime's UT1 provider; any epoch pair is a valid input here)."""
    n_traj = int(ds["n_traj"])
    rng = SplitMix64(seed)
    site = np.minimum((rng.uniform(n_traj) * len(SITES)).astype(np.int64), len(SITES) - 1)
    lon = np.deg2rad(SITES[site, 0])
    bf = ERAU * np.stack([SITES[site, 1] * np.cos(lon), SITES[site, 1] * np.sin(lon), SITES[site, 2]], axis=-1)
    counts = np.diff(ds["traj_offset"]).astype(np.int64)
    out = dict(ds)
    out["obs_body_fixed"] = np.ascontiguousarray(np.repeat(bf, counts, axis=0).reshape(-1))
    out["mjd_ut1"] = ds["mjd_tt"] - tt_minus_ut1 / 86400.0
    return out


def dataset_sha256(ds):
    h = hashlib.sha256()
    for k in ("traj_offset", "mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec", "obs_helio_equ", "obs_geo_ecl"):
        h.update(np.ascontiguousarray(ds[k]).tobytes())
    return h.hexdigest()


def bench_params(params):
    """defaults except max_triplets=100, max_obs_for_triplets=20, no noise."""
    params.max_triplets = 100
    params.max_obs_for_triplets = 20
    params.n_noise_realizations = 0
    return params


def make_dataset_blocks(n_obj, block=100000, seed=DEFAULT_SEED, ephem=None, workers=None, **kw):
    """A large batch as the concatenation of `make_dataset(block, seed + b)` for b = 0, 1, ... — block b is
    byte-identical to that call, so any block can be regenerated alone. Blocks are generated by child
    interpreters (numpy only; started with subprocess, so this is safe in a process that has already
    initialised the GPU and does not depend on what the parent's __main__ is)."""
    import pickle
    import subprocess
    import sys
    import tempfile
    if ephem is None:
        ephem = make_ephemeris()
    sizes = [min(block, n_obj - b) for b in range(0, n_obj, block)]
    if workers is None:
        workers = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, pickle, numpy as np; sys.path.insert(0, sys.argv[1]); from outfit_amd import synth; "
            "a = pickle.load(open(sys.argv[2], 'rb')); d = synth.make_dataset(a['n'], seed=a['seed'], ephem=a['ephem'], **a['kw']); "
            "d.pop('ephem'); np.savez(sys.argv[3], **{k: v for k, v in d.items() if k != 'n_traj'})")
    parts = [None] * len(sizes)
    with tempfile.TemporaryDirectory() as tmp:
        pending = list(enumerate(sizes))
        running = []
        while pending or running:
            while pending and len(running) < workers:
                i, n = pending.pop(0)
                arg, out = os.path.join(tmp, f"a{i}.pkl"), os.path.join(tmp, f"b{i}.npz")
                with open(arg, "wb") as f:
                    pickle.dump({"n": n, "seed": seed + i, "ephem": ephem, "kw": kw}, f)
                running.append((i, out, subprocess.Popen([sys.executable, "-c", code, root, arg, out])))
            i, out, proc = running.pop(0)
            if proc.wait() != 0:
                for _, _, p in running:
                    p.kill()
                raise RuntimeError(f"dataset block {i} failed")
            with np.load(out) as z:
                parts[i] = {k: z[k] for k in z.files}
            os.remove(out)
    out = {"n_traj": n_obj, "ephem": ephem}
    for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec", "obs_helio_equ", "obs_geo_ecl", "truth"):
        out[k] = np.ascontiguousarray(np.concatenate([p[k] for p in parts], axis=0))
    offs, base = [np.zeros(1, dtype=np.uint64)], 0
    for p in parts:
        offs.append(p["traj_offset"][1:] + np.uint64(base))
        base += int(p["traj_offset"][-1])
    out["traj_offset"] = np.concatenate(offs)
    return out
