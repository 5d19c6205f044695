// screen.hpp — the SCREENING evaluation of a candidate's residuals (K2c in OUTFIT_MATH_GLIBC).
//
// In the glibc-faithful math mode the RMS that decides estimate_best_orbit's fold (trajectory.rs:455-544) is computed
// twice: K2c screens all candidates, K3b (k_refine_best) re-evaluates the contenders — every candidate that can still
// be the minimum — with the glibc-faithful functions, and only K3b's bits reach the result. The screen therefore owes
// nobody a last place: it has to come with an error BOUND (screen_level / screen_slack_rad below; measured: median
// 5e-11 in RMS units, every one of 1.5 x 10^7 candidates within a quarter of its bound, tests/test_gpu_glibc_mode.py,
// which also holds the selected records byte for byte against the exhaustive exact evaluation), and it has to agree
// with the exact evaluation on which candidates fail. Within that it is free to be cheap, and the reference's
// per-observation work
// (observation_ephemeris.rs:369-416, equinoctial_element.rs:326-348,809-867) is mostly spent in places where it can:
//   * Kepler's equation: the reference restarts Newton from F0 = pi + varpi for every observation (5-7 iterations,
//     a sin/cos each). Consecutive observations of one candidate are close in mean longitude, so the screen carries
//     the solution forward: a first-order prediction from the previous observation, ONE sin/cos there, then Newton
//     steps whose sin/cos come from the angle-addition formulas with a 4-term series of the (tiny) step. The first
//     observation, a jump of more than 0.5 rad, and any step that does not shrink as expected go through the
//     reference's own iteration (solve_kepler_equation), which is also what decides Err(RootFindingError).
//   * atan2 (twice per observation): one division, an odd polynomial of degree 21 on |t| <= tan(pi/8) (1.4e-16
//     relative), constants in scalar registers — 45 instructions against the platform's 105.
//   * sqrt and the divisions: v_rsq_f64 / v_rcp_f64 with Newton refinements, no scaling for subnormals.
//   * explicit fma() throughout (the exact paths are compiled with contraction off and spell out every rounding).
//   * the constants are compile-time literals: passed as kernel arguments (50 doubles) they overflowed the scalar
//     register file and every use became two v_readlane (690 per loop body; K2c 3.95 ms instead of 2.99).
// Result: 877 -> ~300 VALU instructions per (candidate, observation); K2c 5.66 -> 2.99 ms per 10^7 candidates.
#pragma once
#include "orbit.hpp"   // (brings screen_math.hpp: ScreenConsts, rcp / rsqrt, atan2_screen)

namespace outfit {

// The solution of F - k sin F + h cos F = lam carried from one observation of a candidate to the next.
struct ScreenKepler {
  bool warm;
  double x, sx, cx;  // F and its sine / cosine at the previous observation
  double lam;        // the right-hand side it solved
  double off;        // lam - (lambda + n dt): the multiple of 2 pi the reference's principal_angle removed
  double dmin;       // smallest dlam/dF = 1 - k cos F - h sin F met so far (-> screen_level)
  double inv_rho_max;  // largest 1 / topocentric distance met so far (-> screen_level)
};
// The reference's Newton iteration stops as soon as |f| < 100 eps = 2.2e-14 (equinoctial_element.rs:339), i.e. with an
// error of up to 2.2e-14 / (dlam/dF) in F — part of its result, reproduced by the exact evaluation only because that
// follows the same iterates. Any other solver, however accurate, differs from it by that much; dlam/dF = 1 - e cos E
// gets arbitrarily small for a near-parabolic candidate at perihelion, and on the sky the error is multiplied by
// |dr/dF| / rho <= a / rho, large for a close approach (measured against the exact evaluation: 3 candidates in 10^7 off
// by 0.1 - 1.4 % of their RMS, 3 in 10^4 by more than 1e-8). The screen therefore reports its own bound with every sum:
//   bound_rad = 8 x 2.2e-14 x a x max_i(1 / rho_i) / min_i(dlam/dF)_i        (8 = safety; |dr/dF| <= a)
// rounded up to 1e-13 x 2^E rad, E in five bits of the candidate's flag word (E = 31: unbounded), and k_refine_best
// widens that candidate's window by it (screen_slack_rad). Typical main-belt candidate: E = 2 - 3, i.e. 4 - 8e-13 rad
// = 3 - 6e-7 of an RMS normalised by 0.3"; the screen's own roundings (Newton to 1e-16, ~1e-15 rad of drift in the
// carried sine / cosine, <= 6e-16 in atan2: bounded per primitive by tests/test_screen_math.py on the host) are ~1 % of that.
constexpr unsigned kFlagLevelShift = 23, kFlagLevelMask = 0x1Fu << kFlagLevelShift;
__device__ __forceinline__ unsigned screen_level(double a, double inv_rho_max, double dmin) {
  const double x = 1.76 * a * inv_rho_max / dmin;   // bound_rad / 1e-13
  if (!(x < 1.0e9) || !(dmin > 0.0)) return 31u;
  if (x <= 1.0) return 0u;
  return (unsigned)__builtin_amdgcn_frexp_exp(x);   // x = m 2^e, 0.5 <= m < 1: x < 2^e, e <= 30
}
__device__ __forceinline__ double screen_slack_rad(unsigned level) {
  if (level >= 31u) return INFINITY;
  return ldexp(1.0e-13, (int)level);
}

// false = Err(RootFindingError), decided by the reference's own iteration
__device__ __forceinline__ bool screen_kepler(const Equinoctial& q, const TwoBodyFrame& F, double dt, ScreenKepler& S,
                                              const SinCosConsts& K, const ScreenConsts& C) {
  const double lamu = fma(F.n, dt, q.lambda);
  bool full = !S.warm;
  if (S.warm) {
    const double lam = lamu + S.off;
    const double d0 = fma(-q.h, S.sx, fma(-q.k, S.cx, 1.0));  // dlam/dF at the previous solution
    const double dx = (lam - S.lam) * rcp_nr1(d0);
    double x = S.x + dx;
    if (!(fabs(dx) < 0.5)) {
      full = true;
    } else {
      // sin / cos at the prediction: by angle addition from the previous solution for a small step (most observations:
      // 0.02 d within a night is 1e-4 rad, a few days 1e-2), a fresh evaluation otherwise. Chained rotations drift by a
      // random walk of ~1e-16 rad each (tens per candidate: 1e-15, against a slack of 1.6e-10).
      double sx = S.sx, cx = S.cx;
      if (fabs(dx) < 0.0625) rotate_small(sx, cx, dx, C);
      else sincos_k(x, K, sx, cx);
      bool ok = false;
      for (int it = 0; it < 6; ++it) {
        const double f = fma(q.h, cx, fma(-q.k, sx, x - lam));
        const double d = fma(-q.h, sx, fma(-q.k, cx, 1.0));
        S.dmin = fmin(S.dmin, d);
        const double dl = -f * rcp_nr1(d);
        if (!(fabs(dl) < 0.0625)) break;
        x += dl;
        rotate_small(sx, cx, dl, C);
        if (fabs(dl) < 1e-9) { ok = true; break; }  // quadratic: the step after this one is below 1e-17
      }
      if (ok) { S.x = x; S.sx = sx; S.cx = cx; S.lam = lam; }
      else full = true;
    }
  }
  if (full) {
    double lam = principal_angle_fast(lamu);
    if (lam < F.varpi) lam += kTwoPi;
    if (!solve_kepler_equation<kFnPlatform>(q, F, lam, S.x, S.sx, S.cx, &K)) return false;
    S.dmin = fmin(S.dmin, fma(-q.h, S.sx, fma(-q.k, S.cx, 1.0)));
    S.lam = lam;
    S.off = lam - lamu;
    S.warm = true;
  }
  return true;
}

// Candidate-constant part of the screen's residual: the orbital-plane axes in the equatorial frame.
struct ScreenFrame {
  Vec3 f, g;
  double na2;
};
__device__ __forceinline__ ScreenFrame screen_frame(const Equinoctial& q, const TwoBodyFrame& F) {
  return {ecl_to_equ_finite(F.f), ecl_to_equ_finite(F.g), F.n * (q.a * q.a)};
}

// The squared normalised residual of one observation from (sin F, cos F): equinoctial_element.rs:850-867,
// observation_ephemeris.rs:222-275,388-416 (incl. the sign-keeping `%` of quirk Q6), in the screen's arithmetic.
// w_ra = cos(dec_obs) / sigma_ra and w_dec = 1 / sigma_dec depend on the observation only: K1 evaluates them once per
// observation (kernels.h kObsRecDoubles) instead of the loop once per (candidate, observation).
__device__ __forceinline__ double screen_term(const Equinoctial& q, const TwoBodyFrame& F, const ScreenFrame& G,
                                              double sx, double cx, double ra, double dec, double w_ra, double w_dec,
                                              const Vec3& obs_pos_equ, const ScreenConsts& C, double& inv_rho_max) {
  const double xe = q.a * fma(F.one_bh2, cx, fma(F.bet, sx, -q.k));
  const double ye = q.a * fma(F.one_bk2, sx, fma(F.bet, cx, -q.h));
  double r, inv_r;
  sqrt_rsqrt(fma(xe, xe, ye * ye), r, inv_r);
  const double vc = G.na2 * inv_r;
  const double vxe = vc * fma(F.bet, cx, -(F.one_bh2 * sx));
  const double vye = vc * fma(F.one_bk2, cx, -(F.bet * sx));
  const Vec3 rel = {fma(xe, G.f.x, fma(ye, G.g.x, -obs_pos_equ.x)), fma(xe, G.f.y, fma(ye, G.g.y, -obs_pos_equ.y)),
                    fma(xe, G.f.z, fma(ye, G.g.z, -obs_pos_equ.z))};
  const Vec3 vq = {fma(vxe, G.f.x, vye * G.g.x), fma(vxe, G.f.y, vye * G.g.y), fma(vxe, G.f.z, vye * G.g.z)};
  double dist, inv_dist;
  sqrt_rsqrt(fma(rel.x, rel.x, fma(rel.y, rel.y, rel.z * rel.z)), dist, inv_dist);
  inv_rho_max = fmax(inv_rho_max, inv_dist);
  const double lt = dist * C.inv_vlight;
  const Vec3 cor = {fma(-lt, vq.x, rel.x), fma(-lt, vq.y, rel.y), fma(-lt, vq.z, rel.z)};
  double rho, inv_rho;
  sqrt_rsqrt(fma(cor.x, cor.x, cor.y * cor.y), rho, inv_rho);
  const double delta = atan2_screen(cor.z, rho, C);
  double alpha = atan2_screen(cor.y, cor.x, C);
  if (alpha < 0.0) alpha += C.two_pi;
  double da = ra - alpha;
  if (!(fabs(da) < C.two_pi)) da = fmod_2pi(da);
  if (da > C.pi) da -= C.two_pi;
  const double t1 = da * w_ra;
  const double t2 = (dec - delta) * w_dec;
  return fma(t1, t1, t2 * t2);
}

}  // namespace outfit
