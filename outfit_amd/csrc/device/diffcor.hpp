// diffcor.hpp — device-side differential orbit correction, one lane per trajectory.
//
// Replaces run_differential_correction (differential_orbit_correction/diff_cor.rs:282-436)
// with single_iteration (single_iteration.rs:158-328), solve_weighted_least_squares /
// rescale_covariance (least_square.rs:211-362), update_observation_selection
// (outlier_rejection.rs:132-235) and the two-body partials
// compute_obs_and_partials_2body (observation_ephemeris.rs:418-450) ->
// propagate_twobody(.., true) / compute_derivative (equinoctial_element.rs:442-582,639-759)
// -> topocentric_radec_and_partials (observation_ephemeris.rs:222-275).
//
// Operation order is the reference's (nalgebra 0.34.2 for Cholesky / QR / gemv / dot);
// everything that depends on the elements only is evaluated once per Newton iteration
// instead of once per observation (same expressions, same bits).
#pragma once
#include "orbit.hpp"

namespace outfit {

// outfit_diffcor_config_t with the counters saturated to 32 bits
struct DiffCorConfig {
  unsigned max_newton_iterations, max_outlier_rejection_passes, max_stagnation_iterations;
  unsigned enable_outlier_rejection;
  double convergence_threshold, convergence_before_rejection_threshold;
  double rms_stagnation_ratio, rms_divergence_ratio;
  double chi2_reject, chi2_recover;
  double eccentricity_limit, min_a, max_a, min_peri, max_apo;
  unsigned char free_el[6];
  unsigned char n_free, pad;
};

// equinoctial_element.rs:258-268
__device__ __forceinline__ bool is_bizarre(const Equinoctial& q, const DiffCorConfig& c) {
  const double e = sqrt(q.h * q.h + q.k * q.k);
  const double peri = q.a * (1.0 - e), apo = q.a * (1.0 + e);
  return e > c.eccentricity_limit || q.a < c.min_a || q.a > c.max_a || peri < c.min_peri || apo > c.max_apo;
}

// Element-only part of propagate_twobody(.., true) + compute_derivative.
struct PartialsFrame {
  TwoBodyFrame F;
  double inv_u, beta3_i1b, tmp2, tmp3, tmp6, a2, two_a;
  Vec3 w;
  bool elliptic;
};

// kFn (orbit.hpp): kFnCr — correctly-rounded functions throughout the corrector (the correctly-rounded math mode: its
// ill-conditioned normal equations amplify a last place, so that mode pins every function); kFnGlibc — glibc's bits
// throughout, i.e. what the reference computes on a glibc host (round 3: the CPU restatement in that mode equals its own
// platform-libm run on every byte of 3 000 + 3 000 corrector records; the correctly-rounded one on 80 % of them).
template <int kFn>
__device__ __forceinline__ PartialsFrame partials_frame(const Equinoctial& q) {
  PartialsFrame P;
  P.F = twobody_frame<kFn>(q);
  P.elliptic = !(sqrt(P.F.e2) >= 1.0);  // check_elliptical_orbit (observation_ephemeris.rs:288-296)
  const double u = 1.0 + q.p * q.p + q.q * q.q;
  P.inv_u = 1.0 / u;
  P.w = {2.0 * q.p * P.inv_u, -2.0 * q.q * P.inv_u, (1.0 - q.p * q.p - q.q * q.q) * P.inv_u};
  const double beta = P.F.beta;
  const double inv_1_beta = 1.0 / (1.0 - beta);
  const double beta3 = beta * beta * beta;
  P.tmp2 = beta + (q.h * q.h) * beta3 * inv_1_beta;
  P.tmp3 = q.h * q.k * beta3 * inv_1_beta;
  P.tmp6 = beta + (q.k * q.k) * beta3 * inv_1_beta;
  P.a2 = q.a * q.a;
  P.two_a = 2.0 * q.a;
  return P;
}

struct ObsPartials {
  double ra, dec;
  double ga[6], gd[6];
};

// compute_obs_and_partials_2body for one observation; false = Err (the observation is skipped)
template <int kFn>
__device__ __forceinline__ bool obs_and_partials(const Equinoctial& q, const PartialsFrame& P, double mjd,
                                                 const Vec3& obs_pos, ObsPartials& o) {
  if (!P.elliptic) return false;
  const TwoBodyFrame& F = P.F;
  const double dt = mjd - q.epoch;
  double lam = principal_angle_fast(q.lambda + F.n * (dt - 0.0));
  if (lam < F.varpi) lam += kTwoPi;
  double x, sx, cx;
  if (!solve_kepler_equation<kFn>(q, F, lam, x, sx, cx)) return false;
  const double a = q.a;
  const double xe = a * (F.one_bh2 * cx + F.bet * sx - q.k);
  const double ye = a * (F.one_bk2 * sx + F.bet * cx - q.h);
  const Vec3 pos = {xe * F.f.x + ye * F.g.x, xe * F.f.y + ye * F.g.y, xe * F.f.z + ye * F.g.z};
  const double r = sqrt(xe * xe + ye * ye);
  const double vc = F.n * P.a2 / r;
  const double vxe = vc * (F.bet * cx - F.one_bh2 * sx);
  const double vye = vc * (F.one_bk2 * cx - F.bet * sx);
  const Vec3 vel = {vxe * F.f.x + vye * F.g.x, vxe * F.f.y + vye * F.g.y, vxe * F.f.z + vye * F.g.z};

  // compute_derivative: position rows only (the velocity Jacobian is not used by the two-body path)
  const double inv_r = 1.0 / r;
  const double tmp1 = lam - x;
  const double tmp4 = F.beta * q.h - sx;
  const double tmp5 = F.beta * q.k - cx;
  Vec3 dp[6];
  dp[0] = {(pos.x - 3.0 * vel.x * dt / 2.0) / a, (pos.y - 3.0 * vel.y * dt / 2.0) / a,
           (pos.z - 3.0 * vel.z * dt / 2.0) / a};
  const double dx1de2 = -a * (tmp1 * P.tmp2 + a * cx * tmp4 * inv_r);
  const double dx2de2 = a * (tmp1 * P.tmp3 - 1.0 + a * cx * tmp5 * inv_r);
  dp[1] = {dx1de2 * F.f.x + dx2de2 * F.g.x, dx1de2 * F.f.y + dx2de2 * F.g.y, dx1de2 * F.f.z + dx2de2 * F.g.z};
  const double dx1de3 = -a * (tmp1 * P.tmp3 + 1.0 - a * sx * tmp4 * inv_r);
  const double dx2de3 = a * (tmp1 * P.tmp6 - a * sx * tmp5 * inv_r);
  dp[2] = {dx1de3 * F.f.x + dx2de3 * F.g.x, dx1de3 * F.f.y + dx2de3 * F.g.y, dx1de3 * F.f.z + dx2de3 * F.g.z};
  dp[3] = {2.0 * (q.q * (ye * F.f.x - xe * F.g.x) - xe * P.w.x) * P.inv_u,
           2.0 * (q.q * (ye * F.f.y - xe * F.g.y) - xe * P.w.y) * P.inv_u,
           2.0 * (q.q * (ye * F.f.z - xe * F.g.z) - xe * P.w.z) * P.inv_u};
  dp[4] = {2.0 * (q.p * (-ye * F.f.x + xe * F.g.x) + ye * P.w.x) * P.inv_u,
           2.0 * (q.p * (-ye * F.f.y + xe * F.g.y) + ye * P.w.y) * P.inv_u,
           2.0 * (q.p * (-ye * F.f.z + xe * F.g.z) + ye * P.w.z) * P.inv_u};
  dp[5] = {vel.x / F.n, vel.y / F.n, vel.z / F.n};

  // topocentric_radec_and_partials (equatorial frame)
  const Vec3 pq = ecl_to_equ_finite(pos), vq = ecl_to_equ_finite(vel);
  const Vec3 rel = {pq.x - obs_pos.x, pq.y - obs_pos.y, pq.z - obs_pos.z};
  const double rel_norm = norm(rel);
  const double lt = rel_norm / kVlightAU;
  const Vec3 cor = {rel.x - lt * vq.x, rel.y - lt * vq.y, rel.z - lt * vq.z};
  const double rho = norm(cor);
  const double rho_xy = hypot_sel<kFn>(cor.x, cor.y);
  const double rho_xy_sq = rho_xy * rho_xy;
  o.dec = atan2_sel<kFn>(cor.z, rho_xy);
  double alpha = atan2_sel<kFn>(cor.y, cor.x);
  if (alpha < 0.0) alpha += kTwoPi;  // rem_euclid(TAU) of a value in [-pi, pi]
  o.ra = alpha;
  const double rho_sq = rho * rho;
  const Vec3 grad_ra = {-cor.y / rho_xy_sq, cor.x / rho_xy_sq, 0.0};
  const Vec3 grad_dec = {-cor.z * cor.x / (rho_xy * rho_sq), -cor.z * cor.y / (rho_xy * rho_sq),
                         rho_xy / rho_sq};
  const double aberr = 1.0 / (rel_norm * kVlightAU);
  const double ca = dot(grad_ra, vq) * aberr;
  const double cd = dot(grad_dec, vq) * aberr;
  const Vec3 d_ra = {grad_ra.x - ca * rel.x, grad_ra.y - ca * rel.y, grad_ra.z - ca * rel.z};
  const Vec3 d_dec = {grad_dec.x - cd * rel.x, grad_dec.y - cd * rel.y, grad_dec.z - cd * rel.z};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const Vec3 dq = ecl_to_equ_finite(dp[j]);
    o.ga[j] = dot(d_ra, dq);
    o.gd[j] = dot(d_dec, dq);
  }
  return true;
}

// least_square.rs:188-198
__device__ __forceinline__ double angular_diff(double a, double b) {
  // the reference's `while` loops never end for an infinite or absurdly large difference (an observed angle of
  // 1e300): bounded here and in the oracle alike — |a - b| < 6400 rad reduces exactly as the reference, beyond: NaN
  double d = a - b;
  for (int i = 0; i < 1024 && d > kPi; ++i) d -= kTwoPi;
  for (int i = 0; i < 1024 && d < -kPi; ++i) d += kTwoPi;
  if (d > kPi || d < -kPi) return nan("");
  return d;
}

// nalgebra Cholesky::new + inverse() (linalg/cholesky.rs, column-oriented; forward then adjoint
// substitution on the identity), operation for operation; false = not positive definite.
// Reads the matrix from a strided column n[i * ns] (LDS) and writes the
// inverse column by column to out[(6*r + c) * os] (global scratch): only L (21 values) and one
// right-hand side live in registers.
__device__ __forceinline__ bool cholesky_inverse6_lds(const double* n, unsigned ns, double* out,
                                                      unsigned long long os) {
  double m[36];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) m[6 * i + j] = n[(6 * i + j) * ns];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int k = 0; k < j; ++k) {
      const double factor = -m[6 * j + k];
#pragma unroll
      for (int i = j; i < 6; ++i) m[6 * i + j] = factor * m[6 * i + k] + m[6 * i + j];
    }
    const double diag = m[6 * j + j];
    if (diag == 0.0 || !(diag >= 0.0)) ok = false;
    const double denom = sqrt(diag);
    m[6 * j + j] = denom;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) m[6 * i + j] /= denom;
  }
  if (!ok) return false;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = i == c ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double coeff = b[i] / m[6 * i + i];
      b[i] = coeff;
#pragma unroll
      for (int r = i + 1; r < 6; ++r) b[r] = (-coeff) * m[6 * r + i] + b[r];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double dt = 0.0;
#pragma unroll
      for (int r = i + 1; r < 6; ++r) dt += m[6 * r + i] * b[r];
      b[i] = (b[i] - dt) / m[6 * i + i];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) out[(unsigned long long)(6 * r + c) * os] = b[r];
  }
  return true;
}

__device__ __forceinline__ double signum(double x) { return (x > 0.0 || (x == 0.0 && !signbit(x))) ? 1.0 : -1.0; }

// nalgebra QR::new(m).try_inverse() (Householder), the fallback of least_square.rs:305-316.
// Rare (normal matrix not positive definite), so the factorisation runs with rolled loops on a
// per-lane scratch column in global memory, w[idx * stride], idx < 42 (36 matrix + 6 diagonal):
// dynamic indexing of a register array would otherwise put the whole kernel on the scratch stack.
// Reads the matrix from n[i * ns], writes the inverse to out[(6*r + c) * os].
__device__ __forceinline__ bool qr_inverse6(double* w, unsigned long long stride, const double* n, unsigned ns,
                                            double* out, unsigned long long os) {
#pragma unroll 1
  for (int i = 0; i < 36; ++i) w[i * stride] = n[i * ns];
#pragma unroll 1
  for (int c = 0; c < 6; ++c) {
    double sq = 0.0;
#pragma unroll 1
    for (int r = c; r < 6; ++r) sq += w[(6 * r + c) * stride] * w[(6 * r + c) * stride];
    const double nrm = sqrt(sq);
    const double x0 = w[(6 * c + c) * stride];
    const double modulus = x0 >= 0.0 ? x0 : -x0, sign = x0 >= 0.0 ? 1.0 : -1.0;
    const double signed_norm = sign * nrm;
    const double factor = (sq + modulus * nrm) * 2.0;
    w[(6 * c + c) * stride] = x0 + signed_norm;
    double dg;
    if (factor != 0.0) {
      const double sf = sqrt(factor);
#pragma unroll 1
      for (int r = c; r < 6; ++r) w[(6 * r + c) * stride] /= sf;
      double n2 = 0.0;
#pragma unroll 1
      for (int r = c; r < 6; ++r) n2 += w[(6 * r + c) * stride] * w[(6 * r + c) * stride];
      const double nn = sqrt(n2);
#pragma unroll 1
      for (int r = c; r < 6; ++r) w[(6 * r + c) * stride] /= nn;
      dg = -signed_norm;
      const double sg = signum(dg), m_two = sg * -2.0;
#pragma unroll 1
      for (int j = c + 1; j < 6; ++j) {
        double dt = 0.0;
#pragma unroll 1
        for (int r = c; r < 6; ++r) dt += w[(6 * r + c) * stride] * w[(6 * r + j) * stride];
        const double fct = (dt - 0.0) * m_two;
#pragma unroll 1
        for (int r = c; r < 6; ++r)
          w[(6 * r + j) * stride] = fct * w[(6 * r + c) * stride] + sg * w[(6 * r + j) * stride];
      }
    } else {
      dg = signed_norm;
    }
    w[(36 + c) * stride] = dg;
  }
  bool ok = true;
#pragma unroll 1
  for (int col = 0; col < 6; ++col) {
    double b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = i == col ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {  // q_tr_mul: reflect_with_sign(rows c.., signum(diag[c]))
      const double sg = signum(w[(36 + c) * stride]);
      double dt = 0.0;
#pragma unroll
      for (int r = c; r < 6; ++r) dt += w[(6 * r + c) * stride] * b[r];
      const double fct = (dt - 0.0) * (sg * -2.0);
#pragma unroll
      for (int r = c; r < 6; ++r) b[r] = fct * w[(6 * r + c) * stride] + sg * b[r];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {  // solve_upper_triangular_mut
      const double d = fabs(w[(36 + i) * stride]);
      if (d == 0.0) ok = false;
      const double coeff = b[i] / d;
      b[i] = coeff;
#pragma unroll
      for (int r = 0; r < i; ++r) b[r] = (-coeff) * w[(6 * r + i) * stride] + b[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) out[(unsigned long long)(6 * r + col) * os] = b[r];
  }
  return ok;
}

// nalgebra Matrix6 * Vector6 (gemv = column axpy) and Vector6::dot (sequential)
__device__ __forceinline__ void gemv6(const double* A, const double* x, double* y) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double acc = A[6 * r] * x[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) acc = A[6 * r + j] * x[j] + acc;
    y[r] = acc;
  }
}
__device__ __forceinline__ double dot6(const double* a, const double* b) {
  double res = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) res += a[i] * b[i];
  return res;
}

}  // namespace outfit
