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
//     relative), constants in scalar registers — 45 instructions against the platform's 105. Round 5: the residual is
//     the DIFFERENCE of two angles, and K1 stores the sine and cosine of the observed ones with each observation: the
//     difference comes from one small-argument arctangent per coordinate (screen_term_fast: no octant logic, a short
//     series), the generic form remains for directions more than 22.5 degrees apart.
//   * the elements: straight from the state vector for short trajectories (screen_prologue), the loop warm from the epoch.
//   * sqrt and the divisions: v_rsq_f64 / v_rcp_f64 with Newton refinements, no scaling for subnormals.
//   * explicit fma() throughout (the exact paths are compiled with contraction off and spell out every rounding).
//   * the constants are compile-time literals: passed as kernel arguments (50 doubles) they overflowed the scalar
//     register file and every use became two v_readlane (690 per loop body; K2c 3.95 ms instead of 2.99).
// Result: 877 -> ~300 VALU instructions per (candidate, observation); K2c 5.66 -> 2.99 ms per 10^7 candidates (round 3),
// 2.69 -> 2.06 ms with the direct conversion and the difference form of the residual (round 5).
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
// lazy_start: the frame comes from screen_prologue (below), which leaves the Newton start of the reference's iteration
// (varpi, pi + varpi and its sine / cosine: an atan2 and a sincos) to the rare call that needs it
__device__ __forceinline__ bool screen_kepler(const Equinoctial& q, const TwoBodyFrame& F, double dt, ScreenKepler& S,
                                              const SinCosConsts& K, const ScreenConsts& C, bool lazy_start = false) {
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
    TwoBodyFrame Fs = F;  // (only the four start fields may differ; the rest is not read by the solver)
    if (lazy_start) {
      Fs.varpi = 0.0;
      if (F.e2 > kEps * 1e2) Fs.varpi = principal_angle(atan2_screen(q.h, q.k, C));
      Fs.x0 = kPi + Fs.varpi;
      sincos_k(Fs.x0, K, Fs.sx0, Fs.cx0);
    }
    if (lam < Fs.varpi) lam += kTwoPi;
    if (!solve_kepler_equation<kFnPlatform>(q, Fs, lam, S.x, S.sx, S.cx, &K)) return false;
    S.dmin = fmin(S.dmin, fma(-q.h, S.sx, fma(-q.k, S.cx, 1.0)));
    S.lam = lam;
    S.off = lam - lamu;
    S.warm = true;
  }
  return true;
}

// ---- candidate -> equinoctial elements for the SCREEN, straight from the state vector ---------------------------------
// The reference goes state -> Keplerian elements -> equinoctial elements through the angles (ccek1 orb_elem.rs:58-226,
// from_kepler equinoctial_element.rs:285-313, then the frame of propagate_twobody): five atan2, six sincos, a tan,
// eight square roots and fifteen divisions per candidate — 0.55 of K2c's 2.7 ms (measured with a one-observation loop).
// The exact evaluation (K3b, and this kernel's other modes) has to follow that route rounding by rounding; the screen
// only needs elements that describe the same orbit to ~1e-15, and the equinoctial set is a rational function of the
// state: W = r x v / |r x v|, p = Wx / (1 + Wz), q = -Wy / (1 + Wz), f and g from (p, q), (k, h) = the eccentricity
// vector ((v^2 - mu/r) r - (r.v) v) / mu in that frame, 1/a = 2/r - v^2/mu, and the eccentric longitude F AT THE EPOCH from
// the position in the frame — the 2 x 2 system xe = a (one_bh2 cos F + bet sin F - k), ye = a (one_bk2 sin F + bet cos F - h)
// has determinant sqrt(1 - e^2) — one atan2 in all. lambda = F - k sin F + h cos F needs no wrap, and (F, sin F, cos F)
// is a solved Kepler equation: the loop starts WARM from the epoch (a candidate's first observation is ~10 d away:
// a rotation by ~0.04 rad) instead of with the reference's iteration from pi + varpi.
// What the conversion DECIDES must be what the reference's route decides (kind, conversion error, e >= 1): the direct
// formulas are used where those decisions are far from their thresholds —
//   kScreenElliptic    1/a > 1e-6 /AU, e^2 < 0.998, 1 + Wz > 1e-4: Keplerian, no conversion error, e < 1;
//   kScreenHyperbolic  1/a < -1e-6, e^2 > 1 + 1e-6: Cometary with e > 1 (ccek1's third branch, |e - 1| far above 5e-15
//                      and 1e-12), to_equinoctial succeeds, check_elliptical_orbit fails: flag + Err, no residuals;
//   kScreenDivert      anything else (2 % of the bench candidates have e >= 0.9 or are hyperbolic; < 0.1 % land here),
//                      non-finite input included: the candidate takes the reference's route in the slow-list pass.
constexpr int kScreenElliptic = 0, kScreenHyperbolic = 1, kScreenDivert = 2;
__device__ __forceinline__ int screen_prologue(const Vec3& pos, const Vec3& vel, double epoch, Equinoctial& q,
                                               TwoBodyFrame& F, ScreenKepler& S, const ScreenConsts& C) {
  constexpr double inv_mu = 1.0 / kMu;
  const Vec3 hv = cross(pos, vel);
  double hn, inv_hn, r, inv_r;
  sqrt_rsqrt(dot(hv, hv), hn, inv_hn);
  sqrt_rsqrt(dot(pos, pos), r, inv_r);
  const double v2 = dot(vel, vel), rv = dot(pos, vel);
  const double inv_a = fma(-v2, inv_mu, 2.0 * inv_r);
  // eccentricity vector
  const double c_r = fma(v2, inv_mu, -inv_r), c_v = -rv * inv_mu;
  const Vec3 ev = {fma(c_r, pos.x, c_v * vel.x), fma(c_r, pos.y, c_v * vel.y), fma(c_r, pos.z, c_v * vel.z)};
  const double e2v = dot(ev, ev);
  const double wz1 = fma(hv.z, inv_hn, 1.0);  // 1 + Wz
  if (inv_a < -1e-6 && e2v > 1.0 + 1e-6 && e2v < 1e300 && hn > 0.0) return kScreenHyperbolic;
  if (!(inv_a > 1e-6 && inv_a < 1e6 && e2v < 0.998 && wz1 > 1e-4 && hn > 0.0)) return kScreenDivert;
  const double d = rcp_nr2(wz1);
  const double pp = (hv.x * inv_hn) * d, qq = -(hv.y * inv_hn) * d;
  const double p2 = pp * pp, q2 = qq * qq;
  const double inv_u = rcp_nr2(1.0 + p2 + q2);
  const double cc = 2.0 * pp * qq * inv_u;
  F.f = {(1.0 - p2 + q2) * inv_u, cc, -2.0 * pp * inv_u};
  F.g = {cc, (1.0 + p2 - q2) * inv_u, 2.0 * qq * inv_u};
  const double k = dot(ev, F.f), h = dot(ev, F.g);
  const double a = rcp_nr2(inv_a);
  F.e2 = fma(h, h, k * k);
  double sq, inv_sq;
  sqrt_rsqrt(1.0 - F.e2, sq, inv_sq);     // sqrt(1 - e^2) > 0.044
  F.beta = rcp_nr2(1.0 + sq);
  F.bet = F.beta * h * k;
  F.one_bh2 = fma(-F.beta, h * h, 1.0);
  F.one_bk2 = fma(-F.beta, k * k, 1.0);
  double sa, inv_sa;
  sqrt_rsqrt(inv_a, sa, inv_sa);
  F.n = kGaussGrav * inv_a * sa;          // sqrt(mu / a^3)
  // eccentric longitude at the epoch
  const double xa = fma(dot(pos, F.f), inv_a, k), ya = fma(dot(pos, F.g), inv_a, h);
  const double cf = fma(F.one_bk2, xa, -(F.bet * ya)) * inv_sq, sf = fma(F.one_bh2, ya, -(F.bet * xa)) * inv_sq;
  const double x = atan2_screen(sf, cf, C);
  q.epoch = epoch; q.a = a; q.h = h; q.k = k; q.p = pp; q.q = qq;
  q.lambda = fma(h, cf, fma(-k, sf, x));  // the right-hand side (x, sf, cf) solves; no wrap needed (screen_kepler adds S.off)
  F.varpi = F.x0 = F.sx0 = F.cx0 = 0.0;   // screen_kepler<true> evaluates them when it needs them
  S.warm = true;
  S.x = x; S.sx = sf; S.cx = cf;
  S.lam = q.lambda;
  S.off = 0.0;
  S.dmin = fmin(1.0, fma(-h, sf, fma(-k, cf, 1.0)));
  S.inv_rho_max = 0.0;
  return kScreenElliptic;
}

// Candidate-constant part of the screen's residual: the orbital-plane axes in the equatorial frame.
struct ScreenFrame {
  Vec3 f, g;
  double na2;
};
__device__ __forceinline__ ScreenFrame screen_frame(const Equinoctial& q, const TwoBodyFrame& F) {
  return {ecl_to_equ_finite(F.f), ecl_to_equ_finite(F.g), F.n * (q.a * q.a)};
}

// One observation as the screening loop reads it: K1 packs it once per observation (kernels.h kObsRecDoubles = 12: six
// 16-byte loads from one address) — w_ra = cos(dec_obs) / sigma_ra and w_dec = 1 / sigma_dec, and the sines and cosines of
// the observed angles, all of which depend on the observation only, instead of being evaluated per (candidate, observation).
struct ScreenObs {
  double mjd, ra, cra, sra, cdec, sdec, w_ra, w_dec;
  Vec3 pos;     // observer, equatorial
  double dec;
};
// cra = NaN marks an observation whose angles are outside 0 <= ra < 2 pi, |dec| <= pi / 2 (or not finite): the residual of
// such an observation takes the generic expressions (the reference subtracts the angles as they are given)
__device__ __forceinline__ void pack_screen_obs(double* rec, double mjd, double ra, double dec, double w_ra, double w_dec) {
  double sr, cr, sd, cd;
  ::sincos(ra, &sr, &cr);
  ::sincos(dec, &sd, &cd);
  const bool regular = ra >= 0.0 && ra < kTwoPi && fabs(dec) <= 0.5 * kPi;
  rec[0] = mjd; rec[1] = ra; rec[2] = regular ? cr : nan(""); rec[3] = sr; rec[4] = cd; rec[5] = sd;
  rec[6] = w_ra; rec[7] = w_dec; rec[11] = dec;   // [8..10]: the observer's position, written by its own stage
}
__device__ __forceinline__ ScreenObs load_screen_obs(const double* rec) {
  const double2* r = reinterpret_cast<const double2*>(rec);
  const double2 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
  return {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, {e.x, e.y, f.x}, f.y};
}

// The squared normalised residual of one observation from (sin F, cos F): equinoctial_element.rs:850-867,
// observation_ephemeris.rs:222-275,388-416 (incl. the sign-keeping `%` of quirk Q6), in the screen's arithmetic —
// the generic form: both angles of the computed direction by atan2, then the reference's subtractions.
__device__ __forceinline__ double screen_term_generic(const Equinoctial& q, const TwoBodyFrame& F, const ScreenFrame& G,
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

// atan(num / den) for den > 0, |num| <= tan(pi/8) den: the kernel of atan2_screen without its octant logic.
// kShortSeries: four terms of the series for |num / den| < 1e-2 (z^4 / 9 < 1.2e-17), and three of sqrt(1 + z) in
// screen_term_fast — every residual of a candidate worth its name is that small, and the bench workload's K2c takes
// 2.05 instead of 2.19 ms; but the candidates of long noisy arcs (residuals of degrees) make their waves run both forms,
// so the callers enable it for short trajectories only (kernels.hip rms_candidate, kDirect).
template <bool kShortSeries>
__device__ __forceinline__ double atan_small(double num, double den, const ScreenConsts& C, double& z) {
  const double rd = rcp_nr1(den);                 // 2^-46; the correction below squares that
  double t = num * rd;
  t = fma(fma(-den, t, num), rd, t);
  z = t * t;
  if (kShortSeries && z < 1e-4) return fma(t * z, fma(z, fma(z, -1.0 / 7.0, 0.2), -1.0 / 3.0), t);
  double p = C.at[9];
#pragma unroll
  for (int j = 8; j >= 0; --j) p = fma(p, z, C.at[j]);
  return fma(t * z, p, t);
}

// The same residual where the observation is regular (ScreenObs) and the computed direction lies within 22.5 degrees of
// the observed one in both coordinates — anything else goes through screen_term_generic. The differences of the angles
// are evaluated directly: rotating the computed direction by -ra_obs about z gives rho (cos D, sin D), D = alpha - ra_obs,
// and (dec_obs - delta) is the angle of (cos dec rho + sin dec z, sin dec rho - cos dec z); each is ONE small-argument
// arctangent (no octant logic, a four-term series) instead of a full atan2 followed by a subtraction, rho comes from
// the first as u sqrt(1 + tan^2 D). The reference's wrap of ra_obs - alpha (alpha in [0, 2 pi), `%` keeping the sign, then
// - 2 pi above pi: quirk Q6) reduces to: -D, minus 2 pi when ra_obs + D < 0.
// Also cheaper than the generic form, to the accuracy the light-time correction needs (1e-12 relative): r = a (1 - k cos F
// - h sin F) instead of a square root, one Newton round for |rel|.
// `applies` = false: not within 22.5 degrees in ra (or an irregular observation): nothing but inv_rho_max has been touched,
// the caller takes screen_term_generic for this observation (the main pass leaves its hot loop for a generic one then).
template <bool kShortSeries>
__device__ __forceinline__ double screen_term_fast(const Equinoctial& q, const TwoBodyFrame& F, const ScreenFrame& G,
                                                   double sx, double cx, const ScreenObs& ob, const ScreenConsts& C,
                                                   double& inv_rho_max, bool& applies) {
  const double xe = q.a * fma(F.one_bh2, cx, fma(F.bet, sx, -q.k));
  const double ye = q.a * fma(F.one_bk2, sx, fma(F.bet, cx, -q.h));
  const double vc = G.na2 * rcp_nr1(q.a * fma(-q.h, sx, fma(-q.k, cx, 1.0)));
  const double vxe = vc * fma(F.bet, cx, -(F.one_bh2 * sx));
  const double vye = vc * fma(F.one_bk2, cx, -(F.bet * sx));
  const Vec3 rel = {fma(xe, G.f.x, fma(ye, G.g.x, -ob.pos.x)), fma(xe, G.f.y, fma(ye, G.g.y, -ob.pos.y)),
                    fma(xe, G.f.z, fma(ye, G.g.z, -ob.pos.z))};
  const Vec3 vq = {fma(vxe, G.f.x, vye * G.g.x), fma(vxe, G.f.y, vye * G.g.y), fma(vxe, G.f.z, vye * G.g.z)};
  const double d2 = fma(rel.x, rel.x, fma(rel.y, rel.y, rel.z * rel.z));
  double dist, inv_dist;
  {  // one coupled Newton round on v_rsq_f64: 2^-46
    const double y = __builtin_amdgcn_rsq(d2);
    double g = d2 * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    dist = g;
    inv_dist = 2.0 * h;
  }
  inv_rho_max = fmax(inv_rho_max, inv_dist);
  const double lt = dist * C.inv_vlight;
  const Vec3 cor = {fma(-lt, vq.x, rel.x), fma(-lt, vq.y, rel.y), fma(-lt, vq.z, rel.z)};
  const double u = fma(cor.x, ob.cra, cor.y * ob.sra), v = fma(cor.y, ob.cra, -(cor.x * ob.sra));
  applies = u > 0.0 && fabs(v) <= C.tan_pi_8 * u;   // (false also for cra = NaN, d2 = 0 or not finite)
  if (!applies) return 0.0;
  double z;
  const double dal = atan_small<kShortSeries>(v, u, C, z);       // alpha - ra_obs
  double s1;                                       // sqrt(1 + z)
  if (kShortSeries && z < 1e-4) s1 = fma(z, fma(z, fma(z, 0.0625, -0.125), 0.5), 1.0);
  else { double inv; sqrt_rsqrt(1.0 + z, s1, inv); }
  const double rho = u * s1;
  const double p = fma(ob.sdec, rho, -(ob.cdec * cor.z)), qq = fma(ob.cdec, rho, ob.sdec * cor.z);
  double dd;                                       // dec_obs - delta
  if (qq > 0.0 && fabs(p) <= C.tan_pi_8 * qq) { double z2; dd = atan_small<kShortSeries>(p, qq, C, z2); }
  else dd = ob.dec - atan2_screen(cor.z, rho, C);
  double da = -dal;
  if (ob.ra + dal < 0.0) da -= C.two_pi;
  const double t1 = da * ob.w_ra;
  const double t2 = dd * ob.w_dec;
  return fma(t1, t1, t2 * t2);
}
template <bool kShortSeries = false>
__device__ __forceinline__ double screen_term(const Equinoctial& q, const TwoBodyFrame& F, const ScreenFrame& G,
                                              double sx, double cx, const ScreenObs& ob, const ScreenConsts& C,
                                              double& inv_rho_max) {
  bool applies;
  const double t = screen_term_fast<kShortSeries>(q, F, G, sx, cx, ob, C, inv_rho_max, applies);
  if (applies) return t;
  return screen_term_generic(q, F, G, sx, cx, ob.ra, ob.dec, ob.w_ra, ob.w_dec, ob.pos, C, inv_rho_max);
}

}  // namespace outfit
