// fastdiv.hpp — IEEE-exact fp64 divisions with less work than the generic expansion.
//
// hipcc expands `a / b` (f64) into
//   div_scale x2, rcp, 4 fma (two Newton steps on the reciprocal), mul, fma, div_fmas, div_fixup
// which is correctly rounded. Two situations on the hot path allow the SAME bits for less:
//
//  * div_const: the denominators of the Stumpff series are the integer constants
//    (2j+1)(2j+2) and (2j+2)(2j+3) (kepler/stumpff.rs:153-184). With rc = RN(1/c) known
//    at compile time, q0 = a*rc; r = fma(-q0, c, a); q = fma(r, rc, q0) is the correctly
//    rounded quotient (Markstein's theorem: rc correctly rounded and q0 faithful), 3
//    operations instead of ~11.
//  * div2_shared: num-complex's reciprocal 1/(x+iy) divides two numerators by the same
//    |z|^2 (gauss.rs:971 via aberth). The refined reciprocal depends on the denominator
//    only, so it is computed once; every other operation is the compiler's own sequence.
//
// Both fall back to `/` outside a conservative exponent window, where div_scale /
// div_fixup would intervene. tests/test_gpu_parity.py::test_fast_division_is_exact
// checks bit-equality with `/` on 2^27 operand pairs on the GPU.
#pragma once
#include <hip/hip_runtime.h>

namespace outfit {
namespace fdiv {

// |x| in [2^-300, 2^300] (or x == 0 for numerators): no scaling, no fix-up needed
__device__ __forceinline__ bool in_window(double x) {
  const double ax = fabs(x);
  return ax >= 4.909093465297727e-91 && ax <= 2.037035976334486e+90;
}

// a / c for a compile-time constant c with rc == RN(1/c)
__device__ __forceinline__ double div_const(double a, double c, double rc) {
  if (!(in_window(a))) return a / c;
  const double q0 = a * rc;
  const double r = fma(-q0, c, a);
  return fma(r, rc, q0);
}

// refined reciprocal exactly as in the compiler's f64 division (no scaling case)
__device__ __forceinline__ double refined_rcp(double b) {
  const double y0 = __builtin_amdgcn_rcp(b);
  const double e0 = fma(-b, y0, 1.0);
  const double y1 = fma(y0, e0, y0);
  const double e1 = fma(-b, y1, 1.0);
  return fma(y1, e1, y1);
}
__device__ __forceinline__ double div_with_rcp(double a, double b, double y) {
  const double q = a * y;
  const double r = fma(-b, q, a);
  return fma(r, y, q);
}

// (a1 / b, a2 / b), bit-identical to two `/`
__device__ __forceinline__ void div2_shared(double a1, double a2, double b, double& q1, double& q2) {
  const bool safe = in_window(b) && (a1 == 0.0 || in_window(a1)) && (a2 == 0.0 || in_window(a2));
  if (!safe) {
    q1 = a1 / b;
    q2 = a2 / b;
    return;
  }
  const double y = refined_rcp(b);
  q1 = div_with_rcp(a1, b, y);
  q2 = div_with_rcp(a2, b, y);
}

// ---- guarded divisions: the fast forms WITHOUT their per-division window test --------------------------------------
// The window test above is a divergent branch per division (two compares, exec bookkeeping, the plain sequence as the
// other arm): measured, it costs what the short sequence saves (K2b 6.45 -> 6.58 ms). A Guard instead RECORDS the
// magnitude of every operand that goes through a fast division — smallest one, and the sum (an upper bound of the
// largest that also turns NaN / Inf when an operand is) — two instructions per operand, and the caller tests it ONCE
// for a whole stretch of work: if any operand was zero, outside [2^-300, 2^300] or not finite, nothing computed in that
// stretch is used and the work is redone with plain divisions (K2b: the candidate is handed to the general kernel,
// which is built without a guard). Inside the window the fast forms return the bits of `/` (selftest on 2^27 operands,
// tests/test_gpu_parity.py::test_fast_division_is_exact), so results never depend on which path ran.
// A null guard = plain divisions: every other caller of the functions that take one.
// Round 5: the record is ONE 32-bit word. With e the biased exponent of an operand, |x| lies in [2^-300, 2^300) iff
// 723 <= e <= 1322, i.e. iff (e - 723) as an unsigned number is < 600; zero and denormals (e = 0) wrap to a huge value,
// Inf / NaN (e = 2047) give 1324. The high word shifted left by one (drops the sign) minus 723 << 21 is that number with
// mantissa bits below it — one v_lshl_add_u32 — and the running maximum one v_max_u32 (v_max3_u32 for two operands):
// two 32-bit instructions per operand instead of four fp64-rate ones (|x| added to a sum, and fmin, which the compiler
// brackets with two canonicalising v_max_f64), one VGPR instead of four.
struct Guard {
  unsigned worst;
  static constexpr unsigned kBias = 723u << 21, kSpan = 600u << 21;
  __device__ __forceinline__ void reset() { worst = 0u; }
  __device__ __forceinline__ void see(double a) {
    unsigned u;
#if defined(__HIP_DEVICE_COMPILE__)
    // (written out: from `(hi << 1) - bias` the compiler builds v_alignbit_b32 + v_and_b32 + v_add_u32)
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(u) : "v"(__double2hiint(a)), "s"(0u - kBias));
#else
    u = ((unsigned)__double2hiint(a) << 1) - kBias;
#endif
    worst = u > worst ? u : worst;
  }
  __device__ __forceinline__ bool ok() const { return worst < kSpan; }
};

// A compile-time constant held in a scalar register pair where it is used (device build). `fma(-q0, c, a)` with a
// constant c whose low word is zero (3.0, the series' integer denominators) otherwise becomes v_fmac_f64 with a
// 32-bit literal — a VOP2 whose destination is its addend, so the addend (a value that stays live: beta, the
// numerator) is copied first with a v_mov_b64 on the vector ALU. From a scalar register the same fma is ONE VOP3
// instruction; the two s_mov_b32 that load the constant issue on the scalar unit beside other waves' vector work.
__device__ __forceinline__ double scalar_const(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(c));
#endif
  return c;
}
// a / c, c a compile-time constant of moderate magnitude, rc == RN(1 / c)
__device__ __forceinline__ double div_const(double a, double c, double rc, Guard* g) {
  if (!g) return a / c;
  g->see(a);
  rc = scalar_const(rc);
  const double q0 = a * rc;
  return fma(fma(-q0, scalar_const(c), a), rc, q0);
}
// (a[0] / b, .., a[N-1] / b): velocity.rs:155-157 divides the three components of the corrected velocity by the same g,
// eccentricity_control 1 and mu by the same |r|
// Returns the refined reciprocal of b it used (0 without a guard) for callers that divide by the same b again later.
template <int N>
__device__ __forceinline__ double divn_shared(const double (&a)[N], double b, double (&q)[N], Guard* g) {
  if (!g) {
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = a[k] / b;
    return 0.0;
  }
  g->see(b);
  const double y = refined_rcp(b);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    g->see(a[k]);
    q[k] = div_with_rcp(a[k], b, y);
  }
  return y;
}

// Stumpff-series denominators and their correctly-rounded reciprocals, term j = 1..70:
//   s2 series: (2j+1)(2j+2)      s3 series: (2j+2)(2j+3)
struct SeriesTable {
  double d2[71], r2[71], d3[71], r3[71];
};
constexpr SeriesTable make_series_table() {
  SeriesTable t{};
  for (int j = 0; j <= 70; ++j) {
    t.d2[j] = (double)((2 * j + 1) * (2 * j + 2));
    t.d3[j] = (double)((2 * j + 2) * (2 * j + 3));
    t.r2[j] = 1.0 / t.d2[j];  // constant-evaluated: correctly rounded
    t.r3[j] = 1.0 / t.d3[j];
  }
  return t;
}
__device__ constexpr SeriesTable kSeries = make_series_table();

}  // namespace fdiv
}  // namespace outfit
