"""CPU restatement (numpy) of the reference's legacy-DE binary reader — TEST INFRASTRUCTURE ONLY.

Follows src/jpl_ephem/horizon/horizon_data.rs: `read_horizon_file` :598-684 (TTL 14*3 char6, CNAM 400 char6,
SS 3 f64, NCON i32, 8 skipped bytes, EMRAT f64; IPT at byte 2696 :646), `parse_ipt` :284-305 (36 u32 row-major,
NUMDE, LPT -> row 12), `read_ipt_13_14` :323-347 (rows 13/14 after the 400th constant name, DE440+ and ncon > 400),
`compute_recsize` :239-251 with `dimension` :204-213, `parse_all_blocks` :422-460 (data from 2*recsize),
`extract_body_records` :468-491 + `HorizonRecord::new` horizon_records.rs:132-171.
Pin: the DE440 header KAT of horizon_data.rs:863-896 (IPT -> recsize 8144) — the file itself is absent offline.
"""
import numpy as np

DIMENSION = [3] * 11 + [2, 3, 3, 1]
BODIES = (2, 9, 10)  # EMB, Moon, Sun: what HorizonData::ephemeris(Earth, Sun) reads (:810-849)


def compute_recsize(ipt):
    return 4 * (4 + sum(2 * int(ipt[i][1]) * int(ipt[i][2]) * DIMENSION[i] for i in range(15)))


def read(path):
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 4096:
        raise ValueError("shorter than the 4096-byte header")
    b = raw.tobytes()
    off = 14 * 3 * 6 + 400 * 6
    ss = np.frombuffer(b, dtype="<f8", count=3, offset=off)
    ncon = int(np.frombuffer(b, dtype="<i4", count=1, offset=off + 24)[0])
    emrat = float(np.frombuffer(b, dtype="<f8", count=1, offset=off + 36)[0])
    words = np.frombuffer(b, dtype="<u4", count=40, offset=2696)
    ipt = np.zeros((15, 3), dtype=np.uint32)
    ipt[:12] = words[:36].reshape(12, 3)
    numde = int(words[36])
    ipt[12] = words[37:40]
    if numde >= 440 and ncon > 400:
        ipt[13:15] = np.frombuffer(b, dtype="<u4", count=6, offset=2856 + (ncon - 400) * 6).reshape(2, 3)
    recsize = compute_recsize(ipt)
    ncoeff = recsize // 8
    n_blocks = (raw.size - 2 * recsize) // recsize
    blocks = np.frombuffer(b, dtype="<f8", count=n_blocks * ncoeff, offset=2 * recsize).reshape(n_blocks, ncoeff)
    segs = []
    for body in BODIES:
        o, nc, ns = (int(x) for x in ipt[body])
        coeffs = blocks[:, 2:]  # `&block[2..]`
        seg = np.empty((n_blocks, ns, 3, nc))
        for s in range(ns):
            base = o - 3 + s * nc * 3
            for ax in range(3):
                seg[:, s, ax, :] = coeffs[:, base + ax * nc: base + ax * nc + nc]
        segs.append(seg.reshape(n_blocks, -1))
    return {
        "jd_start": float(ss[0]), "jd_end": float(ss[1]), "block_days": float(ss[2]), "emrat": emrat,
        "ncoef": tuple(int(ipt[b][1]) for b in BODIES), "nsub": tuple(int(ipt[b][2]) for b in BODIES),
        "n_blocks": int(n_blocks), "coef": np.ascontiguousarray(np.concatenate(segs, axis=1).reshape(-1)),
        "ipt": ipt, "numde": numde, "ncon": ncon, "recsize": recsize,
        "block_jd": blocks[:, :2].copy(),
    }


def body_records(path, ipt_row, block):
    """`extract_body_records` (:468-491) for any IPT row and one data block: ([n_sub, axes, n_coeffs], (start_jd, end_jd))."""
    hdr = read(path)
    o, nc, ns = (int(x) for x in hdr["ipt"][ipt_row])
    dim = DIMENSION[ipt_row]
    ncoeff = hdr["recsize"] // 8
    blk = np.fromfile(path, dtype="<f8", count=ncoeff, offset=(2 + block) * hdr["recsize"])
    seg = blk[o - 1: o - 1 + dim * nc * ns].reshape(ns, dim, nc).copy()
    return seg, (float(blk[0]), float(blk[1]))


def record_index(hdr, mjd):
    """`get_record_index` (:711-735): (nr, tau)."""
    et = np.trunc(mjd)
    et_jd = 2400000.5 + et
    nr = int(np.floor((et_jd - hdr["jd_start"]) / hdr["block_days"]))
    if abs(et_jd - hdr["jd_end"]) < 1e-10:
        nr -= 1
    return nr, ((et_jd - (nr * hdr["block_days"] + hdr["jd_start"])) + (mjd - et)) / hdr["block_days"]


def interpolate(rec, start_jd, end_jd, tau, n_sub):
    """`HorizonRecord::interpolate(tau, true, true, n_sub)` (horizon_records.rs:204-298) for one record
    rec[axes, n_coeffs]: (position, velocity, acceleration), every sum left to right as the iterator `.sum()`."""
    n = rec.shape[1]
    dt1 = float(int(tau))
    temp = n_sub * tau
    tc = 2.0 * (np.mod(temp, 1.0) + dt1) - 1.0
    t = [0.0] * n
    t[0] = 1.0
    twot = 0.0
    if tc != t[1]:
        t[1] = tc
        twot = tc + tc
    for i in range(2, n):
        t[i] = twot * t[i - 1] - t[i - 2]
    vfac = (2.0 * n_sub) / (end_jd - start_jd)
    afac = vfac * vfac
    d = [0.0] * n
    d[1] = 1.0
    d[2] = twot + twot
    for i in range(3, n):
        d[i] = twot * d[i - 1] + 2.0 * t[i - 1] - d[i - 2]
    a = [0.0] * n
    for i in range(2, n):
        a[i] = 2.0 * t[1] * a[i - 1] + 4.0 * d[i - 1] - a[i - 2]

    def ev(c, basis):
        s = 0.0
        for x, b in zip(c, basis):
            s += float(x) * b
        return s
    pos = [ev(rec[ax], t) for ax in range(rec.shape[0])]
    vel = [vfac * ev(rec[ax], d) for ax in range(rec.shape[0])]
    acc = [afac * ev(rec[ax], a) for ax in range(rec.shape[0])]
    return pos, vel, acc


def body_state(path, hdr, ipt_row, mjd):
    """`get_record_horizon` (:763-777) + interpolate: (pos, vel, acc) of one body in km, km/day, km/day^2."""
    nr, tau = record_index(hdr, mjd)
    recs, (j0, j1) = body_records(path, ipt_row, nr)
    ns = recs.shape[0]
    sub = int(min(np.floor(tau * ns), ns - 1.0))
    # every HorizonRecord of a block carries the BLOCK's [start_jd, end_jd] (horizon_data.rs:476-489; the Mercury
    # known answer at :901-905 is sub-interval 0 of 4 with a 32-day span), hence vfac = 2 n_sub / block_days
    return interpolate(recs[sub], j0, j1, tau, ns)


def ephemeris(path, hdr, target, center, mjd, au=149597870.7):
    """`HorizonData::ephemeris(target, center, et, true, true).to_au()` (:810-849, interpolation_result.rs:82-84);
    target / center are IPT rows (Earth = 2 with the Moon (row 9) removed by EMRAT, Sun = 10, Mars = 3)."""
    tp = [np.array(x) for x in body_state(path, hdr, target, mjd)]
    if target == 2:
        mp = [np.array(x) for x in body_state(path, hdr, 9, mjd)]
        tp = [x - m / (1.0 + hdr["emrat"]) for x, m in zip(tp, mp)]
    cp = [np.array(x) for x in body_state(path, hdr, center, mjd)]
    return [list((x - c) / au) for x, c in zip(tp, cp)]
