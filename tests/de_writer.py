"""Writes a legacy-DE ("Horizons" binary, little-endian) file around a synthetic EMB/Moon/Sun table, for the
reader tests. Layout as parsed by the reference (src/jpl_ephem/horizon/horizon_data.rs:284-405,598-684):
record 1 = TTL, CNAM, SS, NCON, AU, EMRAT, IPT(12 rows), NUMDE, LPT[, extra constant names, IPT rows 13/14];
record 2 = constant values; then one record per 32-day block: JD0, JD1, Chebyshev coefficients per body."""
import struct

import numpy as np

# IPT of DE440 (horizon_data.rs:868-884): [offset, n_coeffs, n_subintervals]
IPT_DE440 = [[3, 14, 4], [171, 10, 2], [231, 13, 2], [309, 11, 1], [342, 8, 1], [366, 7, 1], [387, 6, 1],
             [405, 6, 1], [423, 6, 1], [441, 13, 8], [753, 11, 2], [819, 10, 4], [899, 10, 4], [1019, 0, 0],
             [1019, 0, 0]]
DIMENSION = [3] * 11 + [2, 3, 3, 1]


def recsize_of(ipt):
    return 4 * (4 + sum(2 * r[1] * r[2] * d for r, d in zip(ipt, DIMENSION)))


def write_de(path, table, ipt=IPT_DE440, numde=440, ncon=645, filler_seed=1):
    """`table` = outfit_amd.synth.make_ephemeris() dict; its three bodies go to IPT rows 2, 9, 10 (their
    (n_coeffs, n_subintervals) must match those rows); every other coefficient is pseudo-random filler."""
    rows = (2, 9, 10)
    for b, r in enumerate(rows):
        assert (ipt[r][1], ipt[r][2]) == (table["ncoef"][b], table["nsub"][b])
    recsize = recsize_of(ipt)
    ncoeff = recsize // 8
    head = bytearray(max(recsize, 4096))
    ttl = b"JPL Planetary Ephemeris DE440/LE440".ljust(84)[:84] * 3
    head[0:252] = ttl
    for i in range(400):
        head[252 + 6 * i:258 + 6 * i] = f"C{i:05d}".encode()[:6]
    off = 252 + 2400
    struct.pack_into("<3d", head, off, table["jd_start"], table["jd_end"], table["block_days"])
    struct.pack_into("<i", head, off + 24, ncon)
    struct.pack_into("<d", head, off + 28, 149597870.7)
    struct.pack_into("<d", head, off + 36, table["emrat"])
    flat = [v for r in ipt[:12] for v in r]
    struct.pack_into("<36I", head, 2696, *flat)
    struct.pack_into("<I", head, 2696 + 144, numde)
    struct.pack_into("<3I", head, 2696 + 148, *ipt[12])
    if numde >= 440 and ncon > 400:
        p = 2856
        for i in range(400, ncon):
            head[p:p + 6] = f"X{i:05d}".encode()[:6]
            p += 6
        struct.pack_into("<6I", head, p, *(ipt[13] + ipt[14]))
    rng = np.random.default_rng(filler_seed)
    per_block = sum(3 * table["ncoef"][b] * table["nsub"][b] for b in range(3))
    coef = np.asarray(table["coef"]).reshape(table["n_blocks"], per_block)
    with open(path, "wb") as f:
        f.write(bytes(head[:recsize]) if recsize >= 4096 else bytes(head))
        if recsize < 4096:  # keep the data at 2*recsize as the reference assumes
            raise ValueError("record size below the 4096-byte header is not representable")
        f.write(rng.standard_normal(ncoeff).astype("<f8").tobytes())  # record 2: constant values
        for nb in range(table["n_blocks"]):
            blk = rng.standard_normal(ncoeff)
            blk[0] = table["jd_start"] + nb * table["block_days"]
            blk[1] = blk[0] + table["block_days"]
            o = 0
            for b, r in enumerate(rows):
                n = 3 * table["ncoef"][b] * table["nsub"][b]
                blk[ipt[r][0] - 1: ipt[r][0] - 1 + n] = coef[nb, o:o + n]
                o += n
            f.write(blk.astype("<f8").tobytes())
    return recsize
