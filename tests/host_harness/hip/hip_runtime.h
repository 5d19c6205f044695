// Host stand-in for <hip/hip_runtime.h>, used ONLY by tests/test_screen_math.py to compile
// outfit_amd/csrc/device/screen_math.hpp with g++: the qualifiers vanish, the two hardware seed instructions are
// emulated at the WORST accuracy their ISA documentation allows (V_RCP_F64 / V_RSQ_F64: 2^29 ulp, i.e. a relative error
// of 2^-23; the sign of the error is a test parameter), everything else is the header's own arithmetic.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
using std::fabs; using std::fma; using std::fmax; using std::fmin; using std::copysign;
#include <cstring>
using std::isfinite; using std::isnan; using std::isinf; using std::signbit; using std::sqrt; using std::ldexp;
using std::fmod; using std::floor; using std::trunc; using std::sin; using std::cos; using std::tan; using std::atan2;
using std::hypot; using std::acos; using std::log; using std::exp; using std::sinh; using std::cosh; using std::cbrt;
#ifndef INFINITY
#define INFINITY __builtin_inf()
#endif
struct double2 { double x, y; };
// bit casts and word access of the HIP device library
static inline long long __double_as_longlong(double x) { long long r; std::memcpy(&r, &x, 8); return r; }
static inline double __longlong_as_double(long long x) { double r; std::memcpy(&r, &x, 8); return r; }
static inline int __double2hiint(double x) { return (int)(__double_as_longlong(x) >> 32); }
static inline int __double2loint(double x) { return (int)(__double_as_longlong(x) & 0xFFFFFFFFll); }
static inline double __hiloint2double(int hi, int lo) {
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
static inline int __builtin_amdgcn_frexp_exp(double x) { int e; (void)std::frexp(x, &e); return e; }
static inline double __builtin_amdgcn_frexp_mant(double x) { int e; return std::frexp(x, &e); }
static inline double __builtin_amdgcn_ldexp(double x, int e) { return std::ldexp(x, e); }
static inline double __builtin_amdgcn_fract(double x) { return x - std::floor(x); }
static inline double __builtin_amdgcn_trig_preop(double, int) { return 0.0; }
static inline unsigned __lane_id() { return 0; }
// wave intrinsics with ONE-lane semantics: only code that does not depend on other lanes may be exercised on the host
static inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_xor(T v, int) { return v; }
static inline int __builtin_amdgcn_mov_dpp(int x, int, int, int, bool) { return x; }
static inline void __syncthreads() {}
static struct { unsigned x; } threadIdx;
extern "C" double g_seed_rel_error;  // +-2^-23 (or 0)
static inline double __builtin_amdgcn_rcp(double d) { return (1.0 / d) * (1.0 + g_seed_rel_error); }
static inline double __builtin_amdgcn_rsq(double d) { return (1.0 / std::sqrt(d)) * (1.0 + g_seed_rel_error); }
