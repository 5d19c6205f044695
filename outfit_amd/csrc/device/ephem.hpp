// ephem.hpp — device-side legacy-DE Chebyshev interpolation of the Earth.
//
// Replaces JPLEphem::earth_ephemeris -> HorizonData::ephemeris(Earth, Sun, ..)
// (reference src/jpl_ephem/mod.rs:145-162, horizon/horizon_data.rs:711-849) and
// HorizonRecord::interpolate (horizon/horizon_records.rs:204-298), plus the
// observer geometry of resolve_observer_geometry
// (src/ephemeris/observation_ephemeris.rs:303-318).
//
// The reference re-evaluates this for every (candidate, observation) pair; it
// depends on the observation only, so here it is a pre-pass with one lane per
// observation. Coefficient blocks are staged through LDS with coalesced loads
// when the workgroup's observations fall into a window of blocks that fits.
#pragma once
#include "orbit.hpp"

namespace outfit {

struct EphemTable {
  double jd_start, jd_end, block_days, emrat;
  unsigned ncoef[3], nsub[3];
  unsigned off[3];       // offset of each body inside a block (doubles)
  unsigned per_block;    // doubles per block
  unsigned long long n_blocks;
  const double* coef;    // device pointer
};

// horizon_data.rs:711-735 ; returns false when the epoch is outside the table
// (the reference panics there: horizon_data.rs:722-724)
__device__ __forceinline__ bool record_index(const EphemTable& T, double mjd_tt, unsigned long long& nr, double& tau) {
  const double et_trunc = trunc(mjd_tt);
  const double et_jd = 2400000.5 + et_trunc;
  if (!(et_jd >= T.jd_start && et_jd <= T.jd_end)) return false;
  nr = (unsigned long long)floor((et_jd - T.jd_start) / T.block_days);
  if (fabs(et_jd - T.jd_end) < 1e-10) nr -= 1;
  if (nr >= T.n_blocks) return false;
  const double interval_start = (double)nr * T.block_days + T.jd_start;
  tau = ((et_jd - interval_start) + (mjd_tt - et_trunc)) / T.block_days;
  return true;
}

// horizon_records.rs:204-298, position only. `c` points at x[nc] y[nc] z[nc]
// (LDS or global). The basis is built on the fly; the sum runs left to right
// from 0.0 exactly like the reference's iterator sum.
__device__ __forceinline__ Vec3 cheb_position(const double* c, unsigned nc, double tau, unsigned n_sub) {
  const double dt1 = (double)(long long)tau;
  double fr = fmod((double)n_sub * tau, 1.0);
  if (fr < 0.0) fr += 1.0;
  const double tc = 2.0 * (fr + dt1) - 1.0;
  double t1 = 0.0, twot = 0.0;
  if (tc != 0.0) { t1 = tc; twot = tc + tc; }
  double tm2 = 1.0, tm1 = t1;
  double sx = 0.0 + c[0] * 1.0, sy = 0.0 + c[nc] * 1.0, sz = 0.0 + c[2 * nc] * 1.0;
  sx += c[1] * t1; sy += c[nc + 1] * t1; sz += c[2 * nc + 1] * t1;
  for (unsigned i = 2; i < nc; ++i) {
    const double ti = twot * tm1 - tm2;
    sx += c[i] * ti; sy += c[nc + i] * ti; sz += c[2 * nc + i] * ti;
    tm2 = tm1; tm1 = ti;
  }
  return {sx, sy, sz};
}

// Earth wrt Sun, equatorial J2000, AU, from one block (blk = LDS or global pointer)
__device__ __forceinline__ Vec3 earth_from_block(const EphemTable& T, const double* blk, double tau) {
  Vec3 b[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double ns = (double)T.nsub[k];
    double sif = floor(tau * ns);
    if (sif > ns - 1.0) sif = ns - 1.0;
    const unsigned si = (unsigned)sif;
    b[k] = cheb_position(blk + T.off[k] + si * 3u * T.ncoef[k], T.ncoef[k], tau, T.nsub[k]);
  }
  const double k = 1.0 + T.emrat;
  return {((b[0].x - b[1].x / k) - b[2].x) / kAU, ((b[0].y - b[1].y / k) - b[2].y) / kAU,
          ((b[0].z - b[1].z / k) - b[2].z) / kAU};
}

// observation_ephemeris.rs:309-317: R_ecl->equ * (geo_ecl + R_equ->ecl * earth_equ)
__device__ __forceinline__ Vec3 observer_equ(const Vec3& earth_equ, const Vec3& geo_ecl) {
  const Vec3 e = equ_to_ecl(earth_equ);
  return ecl_to_equ({geo_ecl.x + e.x, geo_ecl.y + e.y, geo_ecl.z + e.z});
}

}  // namespace outfit
