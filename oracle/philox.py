"""CPU restatement (numpy) of the device noise generator (outfit_amd/csrc/device/noise.hpp) — TEST INFRASTRUCTURE ONLY.

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), pinned by the
Random123 known-answer vectors (tests/test_noise.py); then two 53-bit uniforms per block and Box-Muller. There is
no reference counterpart: Outfit draws from rand's SmallRng + rand_distr::StandardNormal (gauss.rs:341-386),
which cannot be pinned offline — parity of the draws is statistical by definition."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays (uint64 holding 32-bit values) -> four uint64 arrays of 32-bit outputs."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK, p1 & MASK, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def normals(seed, n_traj, max_triplets, n_noise, traj_key=None, traj_key_base=0):
    """The noise_z layout: [traj][triplet][realisation 1..n_noise][ra0 ra1 ra2 dec0 dec1 dec2]."""
    key = (np.arange(n_traj, dtype=np.uint64) + np.uint64(traj_key_base)) if traj_key is None \
        else np.asarray(traj_key, dtype=np.uint64)
    k, t, r, j = np.meshgrid(key, np.arange(max_triplets, dtype=np.uint64), np.arange(1, n_noise + 1, dtype=np.uint64),
                             np.arange(3, dtype=np.uint64), indexing="ij")
    b = philox4x32_10(k & MASK, k >> np.uint64(32), t, np.uint64(4) * r + j, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a1 = ((b[0] << np.uint64(32)) | b[1]) >> np.uint64(11)
    a2 = ((b[2] << np.uint64(32)) | b[3]) >> np.uint64(11)
    u1 = (a1.astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (a2.astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 6.283185307179586 * u2
    z = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)   # [..., j, 2]
    return np.ascontiguousarray(z.reshape(n_traj, max_triplets, n_noise, 6).reshape(-1))
