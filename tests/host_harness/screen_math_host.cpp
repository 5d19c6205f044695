// Host build of the screen's primitives (tests/test_screen_math.py): plain C entry points around the device header.
#include "hip/hip_runtime.h"
#include "device/screen_math.hpp"
extern "C" {
double g_seed_rel_error = 0.0;
void set_seed_error(double e) { g_seed_rel_error = e; }
void atan2_screen_array(unsigned long long n, const double* y, const double* x, double* out) {
  const outfit::ScreenConsts C = outfit::screen_consts();
  for (unsigned long long i = 0; i < n; ++i) out[i] = outfit::atan2_screen(y[i], x[i], C);
}
void rcp_array(unsigned long long n, const double* d, double* r1, double* r2) {
  for (unsigned long long i = 0; i < n; ++i) { r1[i] = outfit::rcp_nr1(d[i]); r2[i] = outfit::rcp_nr2(d[i]); }
}
void sqrt_rsqrt_array(unsigned long long n, const double* x, double* root, double* inv) {
  for (unsigned long long i = 0; i < n; ++i) outfit::sqrt_rsqrt(x[i], root[i], inv[i]);
}
// chains of `steps` small rotations from (sin x0, cos x0): returns the final pair per chain
void rotate_chain(unsigned long long n, unsigned steps, const double* x0, const double* d /* n x steps */, double* s, double* c) {
  const outfit::ScreenConsts C = outfit::screen_consts();
  for (unsigned long long i = 0; i < n; ++i) {
    double sx = std::sin(x0[i]), cx = std::cos(x0[i]);
    for (unsigned k = 0; k < steps; ++k) outfit::rotate_small(sx, cx, d[i * steps + k], C);
    s[i] = sx; c[i] = cx;
  }
}
}
