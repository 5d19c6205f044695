// Host build of the device's arithmetic (tests/test_device_math_host.py): the headers under outfit_amd/csrc/device/ compiled
// with g++ against the stand-in hip_runtime.h, plain C entry points over arrays. What runs here is the code the kernels
// inline — restated glibc functions, correctly-rounded functions, Stumpff series, universal-Kepler Newton solver, cold
// starts, Lagrange velocity correction, the Aberth iteration in its three forms of arithmetic — so that it can be held
// against the CPU oracle bit for bit WITHOUT a GPU. (Lane-cooperative code — wave form of K0, pair / root-per-lane forms —
// needs the device: the -m gpu suite.) v_rcp_f64 is emulated by the correctly rounded reciprocal (seed error 0).
#include "hip/hip_runtime.h"
#include "device/gauss.hpp"
#include "device/orbit.hpp"
#include "device/screen.hpp"
#include "device/diffcor.hpp"
extern "C" {
double g_seed_rel_error = 0.0;
using namespace outfit;
// fn: 0 sin1 1 cos1 2 sincos->sin 3 sincos->cos 4 tan 5 atan2(a,b) 6 hypot(a,b) 7 acos 8 log 9 sinh 10 cosh
void glm_eval(int fn, unsigned long long n, const double* a, const double* b, double* out) {
  for (unsigned long long i = 0; i < n; ++i) {
    double s, c;
    switch (fn) {
      case 0: out[i] = glm::sin1(a[i]); break;
      case 1: out[i] = glm::cos1(a[i]); break;
      case 2: glm::sincos(a[i], s, c); out[i] = s; break;
      case 3: glm::sincos(a[i], s, c); out[i] = c; break;
      case 4: out[i] = glm::tan(a[i]); break;
      case 5: out[i] = glm::atan2(a[i], b[i]); break;
      case 6: out[i] = glm::hypot(a[i], b[i]); break;
      case 7: out[i] = glm::acos(a[i]); break;
      case 8: out[i] = glm::log(a[i]); break;
      case 9: glm::sinhcosh(a[i], s, c); out[i] = s; break;
      default: glm::sinhcosh(a[i], s, c); out[i] = c; break;
    }
  }
}
// fn: 0 sin 1 cos 2 acos 3 log 4 sinh 5 cosh 6 atan2(a,b) 7 tan 8 hypot(a,b)   (correctly rounded, crmath.hpp)
void crm_eval(int fn, unsigned long long n, const double* a, const double* b, double* out) {
  for (unsigned long long i = 0; i < n; ++i) {
    double s, c;
    switch (fn) {
      case 0: crm::sincos(a[i], s, c); out[i] = s; break;
      case 1: crm::sincos(a[i], s, c); out[i] = c; break;
      case 2: out[i] = crm::acos(a[i]); break;
      case 3: out[i] = crm::log(a[i]); break;
      case 4: crm::sinhcosh(a[i], s, c); out[i] = s; break;
      case 5: crm::sinhcosh(a[i], s, c); out[i] = c; break;
      case 6: out[i] = crm::atan2(a[i], b[i]); break;
      case 7: out[i] = crm::tan(a[i]); break;
      default: out[i] = crm::hypot(a[i], b[i]); break;
    }
  }
}
// Round 5: decisions taken on operands instead of on a square root / a quotient (kepler.hpp peri_below, kHnormSqMax,
// kFrobSqMax; gauss.hpp rel_converged, GaussParams::derive), the symmetric clamp of the Newton step, and the one-word
// operand record of the guarded divisions.
void decision_constants(double* out /* 2 */) { out[0] = kHnormSqMax; out[1] = kFrobSqMax; }
void derive_params(double max_peri, double max_ecc, double newton_eps, double* out /* 4 */) {
  GaussParams P{};
  P.max_perihelion_au = max_peri; P.max_ecc = max_ecc; P.newton_eps = newton_eps;
  P.derive();
  out[0] = P.peri_lo; out[1] = P.peri_hi; out[2] = P.eps2_lo; out[3] = P.eps2_hi;
}
// in: (h2, den, peri_max, lo, hi) x n; out: decision with the bounds, decision with NaN bounds (the quotient alone)
void peri_below_array(unsigned long long n, const double* in, int* with_bounds, int* literal) {
  const double nan = __builtin_nan("");
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = in + 5 * i;
    with_bounds[i] = peri_below(q[0], q[1], q[2], q[3], q[4]);
    literal[i] = peri_below(q[0], q[1], q[2], nan, nan);
  }
}
// in: (diff_sq, denom_sq, newton_eps, lo, hi) x n
void rel_converged_array(unsigned long long n, const double* in, int* with_bounds, int* literal) {
  const double nan = __builtin_nan("");
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = in + 5 * i;
    with_bounds[i] = rel_converged(q[0], q[1], q[2], q[3], q[4]);
    literal[i] = rel_converged(q[0], q[1], q[2], nan, nan);
  }
}
void clamp_symmetric_array(unsigned long long n, const double* x, const double* m, double* out) {
  for (unsigned long long i = 0; i < n; ++i) out[i] = clamp_symmetric(x[i], m[i]);
}
// per operand: does a guard that has seen only this operand pass, and is the operand inside fdiv::in_window?
void guard_array(unsigned long long n, const double* x, int* passes, int* in_window) {
  for (unsigned long long i = 0; i < n; ++i) {
    fdiv::Guard g;
    g.reset();
    g.see(1.0);
    g.see(x[i]);
    passes[i] = g.ok();
    in_window[i] = fdiv::in_window(x[i]);
  }
}
void s_funct_array(unsigned long long n, const double* psi, const double* alpha, double* out /* 4 n */) {
  for (unsigned long long i = 0; i < n; ++i) {
    const Stumpff s = s_funct(psi[i], alpha[i]);
    out[4 * i] = s.s0; out[4 * i + 1] = s.s1; out[4 * i + 2] = s.s2; out[4 * i + 3] = s.s3;
  }
}
// problems: (dt, r0, sig0, alpha, e0, convergency) x n; out: prelim_kepuni value; then the Newton solve from it:
// sol[6 i ..] = ok, psi, s0, s1, s2, s3
void kepler_array(unsigned long long n, const double* prob, int math_mode, double* prelim, double* sol) {
  for (unsigned long long i = 0; i < n; ++i) {
    KeplerProblem p;
    p.dt = prob[6 * i]; p.r0 = prob[6 * i + 1]; p.sig0 = prob[6 * i + 2]; p.alpha = prob[6 * i + 3];
    p.e0 = prob[6 * i + 4]; p.convergency = prob[6 * i + 5]; p.math_mode = math_mode;
    prelim[i] = prelim_kepuni(p);
    const KeplerSolution s = solve_kepuni(p, false, 0.0);
    sol[6 * i] = s.ok ? 1.0 : 0.0; sol[6 * i + 1] = s.psi;
    sol[6 * i + 2] = s.s.s0; sol[6 * i + 3] = s.s.s1; sol[6 * i + 4] = s.s.s2; sol[6 * i + 5] = s.s.s3;
  }
}
// the shared cold start of an iteration's two solves: out[4 i ..] = pair(dt, -0.6 dt), prelim_kepuni(dt), prelim_kepuni(-0.6 dt)
void kepler_pair_array(unsigned long long n, const double* prob, int math_mode, double* out) {
  for (unsigned long long i = 0; i < n; ++i) {
    KeplerProblem p;
    p.dt = prob[6 * i]; p.r0 = prob[6 * i + 1]; p.sig0 = prob[6 * i + 2]; p.alpha = prob[6 * i + 3];
    p.e0 = prob[6 * i + 4]; p.convergency = prob[6 * i + 5]; p.math_mode = math_mode;
    const double dt_b = -0.6 * p.dt;
    prelim_kepuni_pair(p, p.dt, dt_b, out[4 * i], out[4 * i + 1]);
    out[4 * i + 2] = prelim_kepuni(p);
    p.dt = dt_b;
    out[4 * i + 3] = prelim_kepuni(p);
  }
}
// in: x1[3] x2[3] v2[3] dt has_guess chi (12 per case); out: ok v[3] f g chi (7 per case)
void velocity_correction_array(unsigned long long n, const double* in, double peri_max, double ecc_max, double eps,
                               int math_mode, double* out) {
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = in + 12 * i;
    const VelCorr v = velocity_correction({q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}, q[9], peri_max,
                                          ecc_max, q[10] != 0.0, q[11], eps, math_mode);
    double* o = out + 7 * i;
    o[0] = v.ok ? 1.0 : 0.0;
    o[1] = v.ok ? v.v.x : 0.0; o[2] = v.ok ? v.v.y : 0.0; o[3] = v.ok ? v.v.z : 0.0;
    o[4] = v.ok ? v.f : 0.0; o[5] = v.ok ? v.g : 0.0; o[6] = v.ok ? v.chi : 0.0;
  }
}
// coef: (c0, c3, c6) x n; variant 0 = plain divisions, 1 = shared reciprocal (restart with the plain form on 3, as K2a does)
void aberth_array(unsigned long long n, const double* coef, unsigned max_iter, double eps, int variant, int* code,
                  double* roots /* 16 n: re, im x 8 */) {
  for (unsigned long long i = 0; i < n; ++i) {
    Cx z[8];
    int rc = variant ? aberth_deg8<true>(coef[3 * i], coef[3 * i + 1], coef[3 * i + 2], max_iter, eps, z)
                     : aberth_deg8<false>(coef[3 * i], coef[3 * i + 1], coef[3 * i + 2], max_iter, eps, z);
    if (rc == 3) rc = aberth_deg8<false>(coef[3 * i], coef[3 * i + 1], coef[3 * i + 2], max_iter, eps, z);
    code[i] = rc;
    for (int k = 0; k < 8; ++k) { roots[16 * i + 2 * k] = z[k].re; roots[16 * i + 2 * k + 1] = z[k].im; }
  }
}
// The RMS chain of one candidate in the glibc-faithful mode (what K3b evaluates exactly): state (equatorial r, v, epoch)
// -> ccek1 -> to_equinoctial -> for every observation ephemeris_error. in: r[3] v[3] epoch (7); obs: mjd ra dec sra sdec
// obs_pos[3] (8 per observation, n_obs per candidate); out per candidate: kind, conv, elem[6], eq (a h k p q lambda),
// then n_obs x (ok, term).
void rms_chain_glibc(unsigned long long n, unsigned n_obs, const double* state, const double* obs, double* out) {
  const unsigned stride = 2 + 6 + 6 + 2 * n_obs;
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = state + 7 * i;
    double* o = out + stride * i;
    const Elements el = ccek1<kFnGlibc>(equ_to_ecl({q[0], q[1], q[2]}), equ_to_ecl({q[3], q[4], q[5]}), q[6]);
    Equinoctial eq{};
    const int conv = to_equinoctial<kFnGlibc>(el, eq);
    o[0] = el.kind; o[1] = conv;
    for (int k = 0; k < 6; ++k) o[2 + k] = el.e[k];
    o[8] = eq.a; o[9] = eq.h; o[10] = eq.k; o[11] = eq.p; o[12] = eq.q; o[13] = eq.lambda;
    if (conv != 0 || !(sqrt(eq.h * eq.h + eq.k * eq.k) < 1.0)) {
      for (unsigned j = 0; j < 2 * n_obs; ++j) o[14 + j] = -1.0;
      continue;
    }
    const TwoBodyFrame F = twobody_frame<kFnGlibc>(eq);
    for (unsigned j = 0; j < n_obs; ++j) {
      const double* b = obs + 8 * ((unsigned long long)n_obs * i + j);
      double term = 0.0;
      const bool ok = ephemeris_error<kFnGlibc>(eq, F, b[0], b[1], b[2], glm::cos1(b[2]), b[3], b[4], {b[5], b[6], b[7]}, term);
      o[14 + 2 * j] = ok ? 1.0 : 0.0;
      o[15 + 2 * j] = ok ? term : 0.0;
    }
  }
}
// The screen's two routes from a state to the loop's inputs (device/screen.hpp): `direct` = screen_prologue, `angles` =
// the reference's route in the screen's arithmetic (ccek1 / to_equinoctial / twobody_frame with kFnScreen), and the
// per-observation squared residuals of each (screen_kepler + screen_term), the direct route warm from the epoch.
// state: 7 per case (equatorial r, v, epoch); obs as in rms_chain_glibc (8 per observation; weights from sra / sdec);
// out per case: cls, kind, conv, e>=1, then direct (a h k p q lambda n), angles (a h k p q lambda n), kepler residual of
// the direct start, then n_obs x (ok_direct, term_direct, ok_angles, term_angles).
void screen_routes(unsigned long long n, unsigned n_obs, const double* state, const double* obs, double* out) {
  const unsigned stride = 4 + 7 + 7 + 1 + 4 * n_obs;
  const ScreenConsts C = screen_consts();
  const SinCosConsts K = sincos_consts();
  for (unsigned long long i = 0; i < n; ++i) {
    const double* s = state + 7 * i;
    double* o = out + stride * i;
    for (unsigned j = 0; j < stride; ++j) o[j] = 0.0;
    const Vec3 pe = equ_to_ecl({s[0], s[1], s[2]}), ve = equ_to_ecl({s[3], s[4], s[5]});
    Equinoctial qd{}, qa{};
    TwoBodyFrame Fd{}, Fa{};
    ScreenKepler Sd{}, Sa{};
    const int cls = screen_prologue(pe, ve, s[6], qd, Fd, Sd, C);
    const Elements el = ccek1<kFnScreen>(pe, ve, s[6]);
    const int conv = to_equinoctial<kFnScreen>(el, qa);
    o[0] = cls; o[1] = el.kind; o[2] = conv;
    const bool hyper = conv == 0 && sqrt(qa.h * qa.h + qa.k * qa.k) >= 1.0;
    o[3] = hyper ? 1.0 : 0.0;
    if (conv == 0 && !hyper) {
      Fa = twobody_frame<kFnScreen>(qa);
      o[11] = qa.a; o[12] = qa.h; o[13] = qa.k; o[14] = qa.p; o[15] = qa.q; o[16] = qa.lambda; o[17] = Fa.n;
    }
    if (cls == kScreenElliptic) {
      o[4] = qd.a; o[5] = qd.h; o[6] = qd.k; o[7] = qd.p; o[8] = qd.q; o[9] = qd.lambda; o[10] = Fd.n;
      o[18] = Sd.x - qd.k * Sd.sx + qd.h * Sd.cx - qd.lambda;
    }
    if (cls != kScreenElliptic || conv != 0 || hyper) continue;
    const ScreenFrame Gd = screen_frame(qd, Fd), Ga = screen_frame(qa, Fa);
    Sa.warm = false; Sa.dmin = 1.0; Sa.inv_rho_max = 0.0;
    bool okd = true, oka = true;
    for (unsigned j = 0; j < n_obs; ++j) {
      const double* b = obs + 8 * ((unsigned long long)n_obs * i + j);
      const double w_ra = cos(b[2]) / b[3], w_dec = 1.0 / b[4];
      double rec[12];
      pack_screen_obs(rec, b[0], b[1], b[2], w_ra, w_dec);
      rec[8] = b[5]; rec[9] = b[6]; rec[10] = b[7];
      const ScreenObs ob = load_screen_obs(rec);
      double* t = o + 19 + 4 * j;
      if (okd) okd = screen_kepler(qd, Fd, b[0] - qd.epoch, Sd, K, C, true);
      if (okd) { t[0] = 1.0; t[1] = (j & 1) ? screen_term<true>(qd, Fd, Gd, Sd.sx, Sd.cx, ob, C, Sd.inv_rho_max) : screen_term<false>(qd, Fd, Gd, Sd.sx, Sd.cx, ob, C, Sd.inv_rho_max); }
      // the route through the angles with the GENERIC residual (two atan2, the reference's subtractions)
      if (oka) oka = screen_kepler(qa, Fa, b[0] - qa.epoch, Sa, K, C, false);
      if (oka) { t[2] = 1.0; t[3] = screen_term_generic(qa, Fa, Ga, Sa.sx, Sa.cx, b[1], b[2], w_ra, w_dec, {b[5], b[6], b[7]}, C, Sa.inv_rho_max); }
    }
  }
}
// compute_obs_and_partials_2body of the corrector in the glibc-faithful mode (diffcor.hpp kFnGlibc): eq = epoch a h k p q
// lambda (7 per case), obs = mjd obs_pos[3] (4 per case); out per case: ok, ra, dec, ga[6], gd[6] (15)
void obs_partials_glibc(unsigned long long n, const double* eqs, const double* obs, double* out) {
  for (unsigned long long i = 0; i < n; ++i) {
    const double* e = eqs + 7 * i;
    Equinoctial q{};
    q.epoch = e[0]; q.a = e[1]; q.h = e[2]; q.k = e[3]; q.p = e[4]; q.q = e[5]; q.lambda = e[6];
    const PartialsFrame P = partials_frame<kFnGlibc>(q);
    ObsPartials ob;
    const bool ok = obs_and_partials<kFnGlibc>(q, P, obs[4 * i], {obs[4 * i + 1], obs[4 * i + 2], obs[4 * i + 3]}, ob);
    double* o = out + 15 * i;
    o[0] = ok ? 1.0 : 0.0;
    o[1] = ok ? ob.ra : 0.0; o[2] = ok ? ob.dec : 0.0;
    for (int j = 0; j < 6; ++j) { o[3 + j] = ok ? ob.ga[j] : 0.0; o[9 + j] = ok ? ob.gd[j] : 0.0; }
  }
}
double angular_diff_host(double a, double b) { return angular_diff(a, b); }
// The guarded divisions of the fast correction kernel (fastdiv.hpp): per case a, b, n1, n2, n3 (5 doubles);
// out per case: guard passed (1 / 0), then a / sqrt(mu), a / mu, a / c, a / 3, n1 / b, n2 / b, n3 / b, 1 / b, mu / b in
// the short forms (9), then the same nine with `/` (9)
void guarded_division_array(unsigned long long n, const double* in, double* out) {
  const double sqrt_mu = sqrt(kMu);
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = in + 5 * i;
    double* o = out + 19 * i;
    fdiv::Guard g;
    g.reset();
    o[1] = div_by_sqrt_mu(q[0], &g); o[2] = div_by_mu(q[0], &g); o[3] = div_by_vlight(q[0], &g);
    o[4] = fdiv::div_const(q[0], 3.0, 1.0 / 3.0, &g);
    const double num[3] = {q[2], q[3], q[4]};
    double r[3];
    fdiv::divn_shared<3>(num, q[1], r, &g);
    o[5] = r[0]; o[6] = r[1]; o[7] = r[2];
    const double one_mu[2] = {1.0, kMu};
    double w[2];
    fdiv::divn_shared<2>(one_mu, q[1], w, &g);
    o[8] = w[0]; o[9] = w[1];
    o[0] = g.ok() ? 1.0 : 0.0;
    o[10] = q[0] / sqrt_mu; o[11] = q[0] / kMu; o[12] = q[0] / kVlightAU; o[13] = q[0] / 3.0;
    o[14] = q[2] / q[1]; o[15] = q[3] / q[1]; o[16] = q[4] / q[1]; o[17] = 1.0 / q[1]; o[18] = kMu / q[1];
  }
}
// One root of one candidate through the device's accept_root and correction loop (gauss.hpp), as the general kernel runs
// it: correction_iteration<false, false> until it ends — WITH the early exits of the device (fixed point of the
// (pos, vel, chi01, chi21) tuple, carried eccentricity_control) that the reference's 50-iteration loop does not have.
// in per case: obs[9] (m[r][k] row-major), S[9], t[3], root (22); out: accepted, outcome (1 Some / 0 None), pos column 1 [3],
// vel[3], epoch, iterations consumed (10)
// `guarded` != 0: the iterations in the fast kernel's form (correction_iteration<true, false>: divisions by constants
// and by shared denominators in their short forms under fdiv::Guard); out[9] = -1 when an iteration reported kCorrRedo
// (a guard hit: the device would replay the candidate with plain divisions), kCorrSlow counts as None.
void correction_loop_array(unsigned long long n, const double* in, double max_ecc, double max_peri, double min_rho2,
                           double kepler_eps, double newton_eps, unsigned newton_max_it, int math_mode, int guarded,
                           double* out) {
  GaussParams P{};
  P.max_ecc = max_ecc; P.max_perihelion_au = max_peri; P.min_rho2_au = min_rho2; P.kepler_eps = kepler_eps;
  P.newton_eps = newton_eps; P.newton_max_it = newton_max_it; P.math_mode = math_mode;
  P.derive();
  for (unsigned long long i = 0; i < n; ++i) {
    const double* q = in + 22 * i;
    double* o = out + 10 * i;
    Mat3 obs, S, Sinv;
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) { obs.m[r][k] = q[3 * r + k]; S.m[r][k] = q[9 + 3 * r + k]; }
    const double t[3] = {q[18], q[19], q[20]};
    for (int k = 0; k < 10; ++k) o[k] = 0.0;
    if (!try_inverse(S, Sinv)) { o[0] = -1.0; continue; }
    const GaussGeometry g = gauss_times(t);
    CorrectionState st{};
    double e_pre;
    if (!accept_root(obs, S, Sinv, t[1], g, P, q[21], st.pos, st.vel, e_pre, st.vc)) continue;
    o[0] = 1.0;
    const double dt01 = t[0] - t[1], dt21 = t[2] - t[1];
    if (fabs(dt01) <= kEps || fabs(dt21) <= kEps) continue;  // gauss.rs:1311-1313: None
    st.epoch = 0.0; st.has01 = false; st.has21 = false; st.chi01 = st.chi21 = 0.0; st.it = 0;
    if (guarded) prepare_guarded(st);
    CorrStep step = kCorrContinue;
    while (step == kCorrContinue)
      step = guarded ? correction_iteration<true, false>(obs, S, Sinv, t[1], dt01, dt21, P, st)
                     : correction_iteration<false, false>(obs, S, Sinv, t[1], dt01, dt21, P, st);
    o[1] = step == kCorrSome ? 1.0 : 0.0;
    o[2] = st.pos.m[0][1]; o[3] = st.pos.m[1][1]; o[4] = st.pos.m[2][1];
    o[5] = st.vel.x; o[6] = st.vel.y; o[7] = st.vel.z; o[8] = st.epoch; o[9] = step == kCorrRedo ? -1.0 : st.it;
  }
}
}
