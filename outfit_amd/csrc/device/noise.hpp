// noise.hpp — device-side standard normals for the noisy realisations.
//
// The reference perturbs (ra, dec) of a triplet with 6 draws of rand_distr::StandardNormal from a
// per-trajectory SmallRng seeded with `base_seed ^ traj_id.stable_hash()` (gauss.rs:341-386,
// obs_dataset_api.rs:285-286). That stream (Xoshiro256++ + ziggurat) cannot be pinned offline, so the
// ABI's primary mode takes host-drawn normals (noise_z). This is the other mode: a counter-based
// generator on the device, so that default-parameter runs (20 realisations x 10 triplets) do not ship
// 9.6 kB of normals per trajectory over PCIe. Parity is STATISTICAL by definition (different generator);
// everything downstream of the draws is unchanged and is checked bit for bit by feeding the very same
// normals (outfit_hip_device_normals) back through the host-normal mode and through the oracle.
//
// Generator: Philox4x32-10 (Salmon et al. 2011; pinned by the Random123 known-answer vectors in
// tests/test_noise.py), key = noise_seed, counter = (trajectory key lo, hi, triplet, 4 * realisation + j),
// j = 0..2; each block gives two 53-bit uniforms in (0, 1) and, by Box–Muller, two normals:
// (ra0, ra1), (ra2, dec0), (dec1, dec2) — the reference's draw order for one realisation.
#pragma once
#include <hip/hip_runtime.h>

namespace outfit {
namespace noise {

struct U4 {
  unsigned x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    c = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// the six normals of (trajectory key, triplet t, realisation r >= 1), reference draw order
__device__ __forceinline__ void normals6(unsigned long long seed, unsigned long long key, unsigned t, unsigned r,
                                         double (&z)[6]) {
#pragma unroll
  for (unsigned j = 0; j < 3; ++j) {
    const U4 b = philox4x32_10({(unsigned)key, (unsigned)(key >> 32), t, 4u * r + j}, (unsigned)seed,
                               (unsigned)(seed >> 32));
    const unsigned long long a1 = (((unsigned long long)b.x << 32) | b.y) >> 11;
    const unsigned long long a2 = (((unsigned long long)b.z << 32) | b.w) >> 11;
    const double u1 = ((double)a1 + 0.5) * 0x1p-53, u2 = ((double)a2 + 0.5) * 0x1p-53;  // (0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586 * u2, &sn, &cs);
    z[2 * j] = rad * cs;
    z[2 * j + 1] = rad * sn;
  }
}

}  // namespace noise
}  // namespace outfit
