// screen_math.hpp — the primitives of the screening arithmetic (device/screen.hpp explains what the screen is and why it may
// be inexact): reciprocal / reciprocal square root from the hardware seeds with Newton steps, and an atan2 of <= 6e-16
// absolute error, the small-angle rotation of a (sin, cos) pair. Separate from screen.hpp because orbit.hpp's function
// selectors (kFnScreen) need them — and so that tests/test_screen_math.py can compile exactly this file for the host
// (hardware seeds emulated at their documented worst) and bound every primitive's error against long-double references.
#pragma once
#include <hip/hip_runtime.h>

namespace outfit {

constexpr double kScreenInvVlight = 1.0 / 173.14463267424034;  // 1 / kVlightAU (orbit.hpp)

struct ScreenConsts {
  double at[10];               // atan t = t + t z P(z), z = t^2, |t| <= tan(pi/8)
  double s3, s5, s7, s9;       // sin d = d + d z (s3 + z (s5 + z (s7 + z s9))), |d| <= 1/16: error < 2e-21
  double c2, c4, c6, c8;       // cos d = 1 + z (c2 + z (c4 + z (c6 + z c8))): error < 3e-19
  double tan_pi_8, pi_4, pi_2, pi, two_pi, inv_vlight;
};
__host__ __device__ inline ScreenConsts screen_consts() {
  return {{-0.3333333333333325, 0.19999999999898407, -0.14285714266096589, 0.11111109636532501, -0.09090852557115886,
           0.07691055157306188, -0.06649613683809145, 0.057363320954987404, -0.04483334390288593, 0.022750523818742745},
          -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0,
          -0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0,
          0.41421356237309503, 0.7853981633974483, 1.5707963267948966, 3.141592653589793, 6.283185307179586,
          kScreenInvVlight};
}

// 1/d to ~2^-28 (one Newton step on v_rcp_f64) and to the last place or two (two steps)
__device__ __forceinline__ double rcp_nr1(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ double rcp_nr2(double d) {
  const double r = rcp_nr1(d);
  return fma(fma(-d, r, 1.0), r, r);
}
// sqrt(x) and 1/sqrt(x) for a positive, normal x (coupled Newton on v_rsq_f64, two rounds); x <= 0 gives 0, 0
__device__ __forceinline__ void sqrt_rsqrt(double x, double& root, double& inv_root) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const bool ok = x > 0.0;
  root = ok ? g : 0.0;
  inv_root = ok ? 2.0 * h : 0.0;
}

// (sin, cos)(x) -> (sin, cos)(x + d), |d| <= 1/16
__device__ __forceinline__ void rotate_small(double& sx, double& cx, double d, const ScreenConsts& C) {
  const double z = d * d;
  const double sd = fma(d * z, fma(z, fma(z, fma(z, C.s9, C.s7), C.s5), C.s3), d);
  const double cd = fma(z, fma(z, fma(z, fma(z, C.c8, C.c6), C.c4), C.c2), 1.0);
  const double ns = fma(sx, cd, cx * sd), nc = fma(cx, cd, -(sx * sd));
  sx = ns;
  cx = nc;
}

// atan2(y, x) to 6e-16 absolute (2.5e-16 in the first octant; tests/test_screen_math.py); (0, 0) -> 0
__device__ __forceinline__ double atan2_screen(double y, double x, const ScreenConsts& C) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const bool big = mn > C.tan_pi_8 * mx;           // atan t = pi/4 + atan((t - 1) / (t + 1))
  const double num = big ? mn - mx : mn, den = big ? mn + mx : mx;
  const double rd = rcp_nr2(den);
  double t = num * rd;
  t = fma(fma(-den, t, num), rd, t);
  t = (mx > 0.0) ? t : 0.0;
  const double z = t * t;
  double p = C.at[9];
#pragma unroll
  for (int j = 8; j >= 0; --j) p = fma(p, z, C.at[j]);
  double r = fma(t * z, p, t);
  r = big ? r + C.pi_4 : r;
  r = (ay > ax) ? C.pi_2 - r : r;
  r = (x < 0.0) ? C.pi - r : r;
  return copysign(r, y);
}

}  // namespace outfit
