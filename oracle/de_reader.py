"""CPU restatement (numpy) of the reference's legacy-DE binary reader — TEST INFRASTRUCTURE ONLY.

Follows src/jpl_ephem/horizon/horizon_data.rs: `read_horizon_file` :533-642 (TTL 14*3 char6, CNAM 400 char6,
SS 3 f64, NCON i32, 8 skipped bytes, EMRAT f64; IPT at byte 2696 :585), `parse_ipt` :284-305 (36 u32 row-major,
NUMDE, LPT -> row 12), `read_ipt_13_14` :323-347 (rows 13/14 after the 400th constant name, DE440+ and ncon > 400),
`compute_recsize` :239-251 with `dimension` :204-213, `parse_all_blocks` :422-460 (data from 2*recsize),
`extract_body_records` :362-392 + `HorizonRecord::new` horizon_records.rs:132-171.
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
