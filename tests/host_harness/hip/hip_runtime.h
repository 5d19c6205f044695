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
extern "C" double g_seed_rel_error;  // +-2^-23 (or 0)
static inline double __builtin_amdgcn_rcp(double d) { return (1.0 / d) * (1.0 + g_seed_rel_error); }
static inline double __builtin_amdgcn_rsq(double d) { return (1.0 / std::sqrt(d)) * (1.0 + g_seed_rel_error); }
