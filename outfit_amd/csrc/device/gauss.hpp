// gauss.hpp — device-side Gauss preliminary orbit for one (object, triplet,
// realisation) candidate, one lane per candidate.
//
// Replaces GaussObs::prelim_orbit (reference
// src/initial_orbit_determination/gauss.rs:1119-1247) and what it calls:
// gauss_prelim :532, coeff_eight_poly :585, descartes_upper_bound_deg8_sparse
// :214, aberth::aberth (crate aberth 0.4.1, call site :971), accept_root :816,
// position_vector_and_reference_epoch :702, gibbs_correction :754,
// pos_and_vel_correction :1284.
//
// Register/LDS plan: the 3x3 direction matrix, its inverse, the observer
// matrix and the running position matrix live in VGPRs; the 8 complex Aberth
// iterates (two generations, 256 B per lane) live in LDS, laid out
// [generation][root][re|im][lane] so that a wave reads 64 consecutive doubles
// (conflict-free ds_read_b64) and the root loops stay rolled.
#pragma once
#include "kepler.hpp"

namespace outfit {

struct Mat3 {
  // m[row][col]; for the observer/position matrices the columns are the three epochs
  double m[3][3];
};

// nalgebra gemv: ((A[:,0]*x0) + A[:,1]*x1) + A[:,2]*x2
__device__ __forceinline__ Vec3 gemv(const Mat3& A, const Vec3& x) {
  return {(A.m[0][0] * x.x + A.m[0][1] * x.y) + A.m[0][2] * x.z,
          (A.m[1][0] * x.x + A.m[1][1] * x.y) + A.m[1][2] * x.z,
          (A.m[2][0] * x.x + A.m[2][1] * x.y) + A.m[2][2] * x.z};
}
__device__ __forceinline__ Vec3 column(const Mat3& A, int c) { return {A.m[0][c], A.m[1][c], A.m[2][c]}; }

// The scalar tunables of IODParams the candidate kernel needs (mod.rs:225-266).
struct GaussParams {
  double max_ecc, max_perihelion_au, min_rho2_au;
  double aberth_eps, kepler_eps, r2_min_au, r2_max_au, newton_eps, root_imag_eps;
  unsigned aberth_max_iter;
  unsigned max_tested_solutions;  // saturated to 2^32-1 on the host
  unsigned newton_max_it;         // saturated likewise
};

struct GaussTriplet {
  double ra[3], dec[3], t[3];
  Mat3 obs;  // heliocentric observer positions, equatorial J2000, columns = epochs
};

// error codes = OUTFIT_CAUSE_* of include/outfit_hip.h
constexpr int kCauseNoRoots = 1, kCauseSingular = 2, kCauseRootFinding = 3;

struct StateSolution {
  int cause;       // 0 ok
  bool corrected;  // GaussResult::CorrectedOrbit vs PrelimOrbit
  Vec3 r2, v2;     // equatorial J2000 state at the light-time corrected epoch
  double epoch;
};

// nalgebra 0.34 Matrix3::try_inverse (adjugate / determinant), gauss.rs:544-546
__device__ __forceinline__ bool try_inverse(const Mat3& A, Mat3& o) {
  const double m11 = A.m[0][0], m12 = A.m[0][1], m13 = A.m[0][2];
  const double m21 = A.m[1][0], m22 = A.m[1][1], m23 = A.m[1][2];
  const double m31 = A.m[2][0], m32 = A.m[2][1], m33 = A.m[2][2];
  const double minor_m12_m23 = m22 * m33 - m32 * m23;
  const double minor_m11_m23 = m21 * m33 - m31 * m23;
  const double minor_m11_m22 = m21 * m32 - m31 * m22;
  const double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
  if (det == 0.0) return false;
  o.m[0][0] = minor_m12_m23 / det;
  o.m[0][1] = (m13 * m32 - m33 * m12) / det;
  o.m[0][2] = (m12 * m23 - m22 * m13) / det;
  o.m[1][0] = -minor_m11_m23 / det;
  o.m[1][1] = (m11 * m33 - m31 * m13) / det;
  o.m[1][2] = (m13 * m21 - m23 * m11) / det;
  o.m[2][0] = minor_m11_m22 / det;
  o.m[2][1] = (m12 * m31 - m32 * m11) / det;
  o.m[2][2] = (m11 * m22 - m21 * m12) / det;
  return true;
}

// ---- complex helpers with num-complex 0.4 operation order ---------------------
struct Cx {
  double re, im;
};
__device__ __forceinline__ Cx cadd(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cx csub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cx cmul(Cx a, Cx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ Cx cdiv(Cx a, Cx b) {
  const double n = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n};
}
// Complex::mul_add with a real addend: (a*b) + c, fused where num-complex fuses
__device__ __forceinline__ Cx cmul_add_real(Cx a, Cx b, double c) {
  return {fma(a.re, b.re, c) - (a.im * b.im), fma(a.re, b.im, fma(a.im, b.re, 0.0))};
}

// LDS view of the Aberth iterates of one lane: element (gen, k, part) at
// base[((gen*8 + k)*2 + part) * stride]. stride = threads per block.
struct RootStore {
  double* base;
  unsigned stride;
  __device__ __forceinline__ Cx get(int gen, int k) const {
    const unsigned o = (unsigned)((gen * 8 + k) * 2) * stride;
    return {base[o], base[o + stride]};
  }
  __device__ __forceinline__ void set(int gen, int k, Cx z) const {
    const unsigned o = (unsigned)((gen * 8 + k) * 2) * stride;
    base[o] = z.re;
    base[o + stride] = z.im;
  }
};

// aberth::aberth::<9, f64>(&[c0,0,0,c3,0,0,c6,0,1], max_iter, eps): Aberth–Ehrlich
// sweeps from Aberth's initial circle; every root of a sweep is updated from the
// previous sweep's iterates. Returns 0 converged / 1 max-iteration / 2 failed and
// the generation index holding the roots.
__device__ __forceinline__ int aberth_deg8(double c0, double c3, double c6, unsigned max_iter, double eps,
                                  const RootStore& st, int& gen_out) {
  // initial guesses: the polynomial is already monic with c7 = 0, so the centre is
  // a = -0/8 = -0.0 and the Pascal shift is the identity; r0 = first positive
  // integer with S(r0) = r0^8 - |c6| r0^6 - |c3| r0^3 - |c0| > 0 (dense Horner).
  const double a6 = -fabs(c6), a3 = -fabs(c3), a0 = -fabs(c0);
  double r0 = 1.0;
  for (int guard = 0; guard < 4096; ++guard) {
    // Horner over [a0,0,0,a3,0,0,a6,0,1] at the real point r0 with num-complex mul_add
    double re = 1.0;                 // c8
    re = fma(re, r0, 0.0);           // c7 = 0
    re = fma(re, r0, a6);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a3);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, 0.0);
    re = fma(re, r0, a0);
    if (re > 0.0) break;
    r0 += 1.0;
  }
  const double neg_zero = -0.0;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    double s, c;  // cos/sin of theta_k = fma(2pi/8, k, (pi/2)/8), correctly rounded
    crm::aberth_dir(k, c, s);
    st.set(0, k, {fma(r0, c, neg_zero), r0 * s});
  }
  int cur = 0;
  for (unsigned it = 1; it <= max_iter; ++it) {
    bool converged = true;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
      const Cx z = st.get(cur, i);
      // p(z): dense Horner over the 9 coefficients (zeros included, as the crate does)
      Cx p = {1.0, 0.0};
      p = cmul_add_real(p, z, 0.0);
      p = cmul_add_real(p, z, c6);
      p = cmul_add_real(p, z, 0.0);
      p = cmul_add_real(p, z, 0.0);
      p = cmul_add_real(p, z, c3);
      p = cmul_add_real(p, z, 0.0);
      p = cmul_add_real(p, z, 0.0);
      p = cmul_add_real(p, z, c0);
      // p'(z): coefficients [0,0,3c3,0,0,6c6,0,8]
      Cx dp = {8.0, 0.0};
      dp = cmul_add_real(dp, z, 0.0);
      dp = cmul_add_real(dp, z, 6.0 * c6);
      dp = cmul_add_real(dp, z, 0.0);
      dp = cmul_add_real(dp, z, 0.0);
      dp = cmul_add_real(dp, z, 3.0 * c3);
      dp = cmul_add_real(dp, z, 0.0);
      dp = cmul_add_real(dp, z, 0.0);
      Cx sum = {0.0, 0.0};
#pragma unroll 1
      for (int k = 0; k < 8; ++k) {
        if (k == i) continue;
        sum = cadd(sum, cdiv({1.0, 0.0}, csub(z, st.get(cur, k))));
      }
      const Cx zn = cadd(z, cdiv(p, csub(cmul(p, sum), dp)));
      st.set(cur ^ 1, i, zn);
      if (isnan(zn.re) || isnan(zn.im) || isinf(zn.re) || isinf(zn.im)) {
        gen_out = cur ^ 1;
        return 2;
      }
      if (!(fabs(zn.re - z.re) < eps && fabs(zn.im - z.im) < eps)) converged = false;
    }
    cur ^= 1;
    if (converged) { gen_out = cur; return 0; }
  }
  gen_out = cur;
  return 1;
}

struct GaussGeometry {
  double tau1, tau3;
  Mat3 S, Sinv;
  double a0, a2, b0, b2;  // vector_a = (a0,-1,a2), vector_b = (b0,0,b2)
};

// gauss.rs:702-724 ; false = SpuriousRootDetected
__device__ __forceinline__ bool position_and_epoch(const GaussTriplet& g, const GaussGeometry& q, const Vec3& c,
                                          double min_rho2, Mat3& pos, double& epoch) {
  const Vec3 gcap = gemv(g.obs, c);
  const Vec3 crhom = gemv(q.Sinv, gcap);
  const double rho0 = -(crhom.x / c.x), rho1 = -(crhom.y / c.y), rho2 = -(crhom.z / c.z);
  if (rho1 < min_rho2) return false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pos.m[r][0] = g.obs.m[r][0] + rho0 * q.S.m[r][0];
    pos.m[r][1] = g.obs.m[r][1] + rho1 * q.S.m[r][1];
    pos.m[r][2] = g.obs.m[r][2] + rho2 * q.S.m[r][2];
  }
  epoch = g.t[1] - rho1 / kVlightAU;
  return true;
}

// gauss.rs:754-781
__device__ __forceinline__ Vec3 gibbs_correction(const Mat3& pos, double tau1, double tau3) {
  const double tau13 = tau3 - tau1;
  const double n1 = norm(column(pos, 0)), n2 = norm(column(pos, 1)), n3 = norm(column(pos, 2));
  const double r1m3 = 1.0 / (n1 * n1 * n1), r2m3 = 1.0 / (n2 * n2 * n2), r3m3 = 1.0 / (n3 * n3 * n3);
  const double d1 = tau3 * (r1m3 / 12.0 - 1.0 / (tau1 * tau13));
  const double d2 = (tau1 + tau3) * (r2m3 / 12.0 - 1.0 / (tau1 * tau3));
  const double d3 = -tau1 * (r3m3 / 12.0 + 1.0 / (tau3 * tau13));
  const double e0 = -d1, e1 = d2, e2 = d3;
  return {kGaussGrav * (((0.0 + pos.m[0][0] * e0) + pos.m[0][1] * e1) + pos.m[0][2] * e2),
          kGaussGrav * (((0.0 + pos.m[1][0] * e0) + pos.m[1][1] * e1) + pos.m[1][2] * e2),
          kGaussGrav * (((0.0 + pos.m[2][0] * e0) + pos.m[2][1] * e1) + pos.m[2][2] * e2)};
}

// Matrix3::norm(): sum over columns of (x^2+y^2)+z^2 accumulated from 0, then sqrt
__device__ __forceinline__ double frob(const Mat3& A) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) s += (A.m[0][k] * A.m[0][k] + A.m[1][k] * A.m[1][k]) + A.m[2][k] * A.m[2][k];
  return sqrt(s);
}

// gauss.rs:1284-1418 ; returns false for None. Quirk Q4: Some even when never converged;
// epoch stays 0.0 if every iteration `continue`d.
__device__ __forceinline__ bool pos_and_vel_correction(const GaussTriplet& g, const GaussGeometry& q,
                                              const GaussParams& P, Mat3& pos, Vec3& vel, double& epoch) {
  epoch = 0.0;
  bool has01 = false, has21 = false;
  double chi01 = 0.0, chi21 = 0.0;
  const double dt01 = g.t[0] - g.t[1], dt21 = g.t[2] - g.t[1];
  if (fabs(dt01) <= kEps || fabs(dt21) <= kEps) return false;
  for (unsigned it = 0; it < P.newton_max_it; ++it) {
    const Vec3 r1 = column(pos, 0), r2 = column(pos, 1), r3 = column(pos, 2);
    // the two Lagrange corrections (t1<-t2 and t3<-t2) are independent; a rolled loop keeps ONE
    // copy of the universal-Kepler solver in the instruction stream
    VelCorr a, b;
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
      const VelCorr vc = velocity_correction(side ? r3 : r1, r2, vel, side ? dt21 : dt01,
                                             P.max_perihelion_au, P.max_ecc, side ? has21 : has01,
                                             side ? chi21 : chi01, P.kepler_eps);
      if (side) b = vc; else a = vc;
    }
    if (!(a.ok && b.ok)) continue;
    has01 = true; chi01 = a.chi;
    has21 = true; chi21 = b.chi;
    if (!isfinite(a.g) || !isfinite(b.g)) continue;
    const Vec3 nv = {(a.v.x + b.v.x) * 0.5, (a.v.y + b.v.y) * 0.5, (a.v.z + b.v.z) * 0.5};
    const double fl = a.f * b.g - b.f * a.g;
    if (!isfinite(fl) || fabs(fl) < kEps) continue;
    const double inv_f = 1.0 / fl;
    const Vec3 c = {b.g * inv_f, -1.0, -a.g * inv_f};
    Mat3 np;
    double ne;
    if (!position_and_epoch(g, q, c, P.min_rho2_au, np, ne)) continue;
    const EccControl ec = eccentricity_control(column(np, 1), nv, P.max_perihelion_au, P.max_ecc);
    if (!ec.some || !ec.accepted) return false;
    const double denom = frob(np);
    if (!isfinite(denom) || denom <= kEps) continue;
    Mat3 d;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) d.m[r][k] = np.m[r][k] - pos.m[r][k];
    const double rel = frob(d) / denom;
    pos = np;
    vel = nv;
    epoch = ne;
    if (rel <= P.newton_eps) break;
  }
  return true;
}

// gauss.rs:532-614 (gauss_prelim + coeff_eight_poly); false = SingularDirectionMatrix
__device__ __forceinline__ bool gauss_geometry(const GaussTriplet& g, GaussGeometry& q, double& c6, double& c3,
                                      double& c0) {
  const double tau1 = kGaussGrav * (g.t[0] - g.t[1]);
  const double tau3 = kGaussGrav * (g.t[2] - g.t[1]);
  const double tau13 = tau3 - tau1;
  q.tau1 = tau1; q.tau3 = tau3;
  q.a0 = tau3 / tau13;
  q.a2 = -(tau1 / tau13);
  q.b0 = q.a0 * (tau13 * tau13 - tau3 * tau3) / 6.0;
  q.b2 = q.a2 * (tau13 * tau13 - tau1 * tau1) / 6.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double sr, cr, sd, cd;
    crm::sincos(g.ra[c], sr, cr);  // correctly rounded: see crmath.hpp
    crm::sincos(g.dec[c], sd, cd);
    q.S.m[0][c] = cr * cd;
    q.S.m[1][c] = sr * cd;
    q.S.m[2][c] = sd;
  }
  if (!try_inverse(q.S, q.Sinv)) return false;
  const Vec3 ra = gemv(g.obs, {q.a0, -1.0, q.a2});
  const Vec3 rb = gemv(g.obs, {q.b0, 0.0, q.b2});
  const Vec3 row1 = {q.Sinv.m[1][0], q.Sinv.m[1][1], q.Sinv.m[1][2]};
  const double a2s = dot(row1, ra), b2s = dot(row1, rb);
  const Vec3 R2 = column(g.obs, 1);
  const double r22 = ((0.0 + R2.x * R2.x) + R2.y * R2.y) + R2.z * R2.z;
  const double s2r2 = ((0.0 + q.S.m[0][1] * R2.x) + q.S.m[1][1] * R2.y) + q.S.m[2][1] * R2.z;
  c6 = -(a2s * a2s) - r22 - (2.0 * a2s * s2r2);
  c3 = -(2.0 * b2s * (a2s + s2r2));
  c0 = -(b2s * b2s);
  return true;
}

// gauss.rs:214-240 with zero_eps = 0.0 (is_sign_positive tests the sign bit: quirk Q11)
__device__ __forceinline__ unsigned descartes_upper_bound(double c0, double c3, double c6) {
  int last = 1;
  unsigned count = 0;
  const double v[3] = {c6, c3, c0};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (fabs(v[i]) <= 0.0) continue;
    const int s = signbit(v[i]) ? -1 : 1;
    if (s != last) ++count;
    last = s;
  }
  return count;
}

// GaussObs::prelim_orbit (gauss.rs:1119-1247) up to the Cartesian state.
// prelim_orbit returns the first CorrectedOrbit among the first
// max_tested_solutions accepted roots, else the first PrelimOrbit; roots are
// visited in solver order, so the walk can stop at the first corrected one.
__device__ __forceinline__ StateSolution gauss_prelim_state(const GaussTriplet& g, const GaussParams& P,
                                                   const RootStore& st) {
  StateSolution out;
  out.cause = 0;
  out.corrected = false;
  out.epoch = 0.0;
  out.r2 = {0, 0, 0};
  out.v2 = {0, 0, 0};
  GaussGeometry q;
  double c6, c3, c0;
  if (!gauss_geometry(g, q, c6, c3, c0)) { out.cause = kCauseSingular; return out; }
  if (descartes_upper_bound(c0, c3, c6) == 0) { out.cause = kCauseNoRoots; return out; }
  int gen;
  if (aberth_deg8(c0, c3, c6, P.aberth_max_iter, P.aberth_eps, st, gen) == 2) {
    out.cause = kCauseRootFinding;
    return out;
  }
  unsigned n_solutions = 0;
  bool have_prelim = false;
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const Cx z = st.get(gen, i);
    if (!(z.re > 0.0 && fabs(z.im) < P.root_imag_eps)) continue;
    const double r2 = z.re;
    if (!(r2 >= P.r2_min_au && r2 <= P.r2_max_au)) continue;
    // accept_root (gauss.rs:816-870)
    const double r2m3 = 1.0 / (r2 * r2 * r2);
    const Vec3 c = {q.a0 + q.b0 * r2m3, -1.0, q.a2 + q.b2 * r2m3};
    Mat3 pos;
    double epoch;
    if (!position_and_epoch(g, q, c, P.min_rho2_au, pos, epoch)) continue;
    Vec3 vel = gibbs_correction(pos, q.tau1, q.tau3);
    const EccControl ec = eccentricity_control(column(pos, 1), vel, P.max_perihelion_au, P.max_ecc);
    if (!ec.some || !ec.accepted) continue;
    const Vec3 r_pre = column(pos, 1);
    const Vec3 v_pre = vel;
    const double e_pre = epoch;
    double e_cor;
    if (pos_and_vel_correction(g, q, P, pos, vel, e_cor)) {
      out.corrected = true;
      out.r2 = column(pos, 1);
      out.v2 = vel;
      out.epoch = e_cor;
      return out;
    }
    if (!have_prelim) {
      have_prelim = true;
      out.r2 = r_pre; out.v2 = v_pre; out.epoch = e_pre;
    }
    if (++n_solutions >= P.max_tested_solutions) break;
  }
  if (!have_prelim) out.cause = kCauseNoRoots;
  return out;
}

}  // namespace outfit
