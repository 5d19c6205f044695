// Host build of the device's lane-form triplet selection (tests/test_k0_host.py): outfit_amd/csrc/device/triplets.hpp
// compiled with g++ against the stand-in hip_runtime.h — the very code of K0's one-lane-per-trajectory form, incl. the
// exact row skip of round 4, so that it can be held against the oracle's literal state machine without a GPU.
// (The wave form needs wave intrinsics and is covered on the device: tests/test_gpu_long_arcs.py.)
#include <cstdint>
#include <vector>
#include "hip/hip_runtime.h"
#include "device/triplets.hpp"
extern "C" int k0_lane_form(const double* mjd, unsigned long long n_obs, double dt_min, double dt_max, double dtw,
                            unsigned long long max_keep, unsigned k_cap, unsigned long long* out) {
  outfit::TripletParams P{dt_min, dt_max, 1.0 / dtw, max_keep, k_cap};
  std::vector<double> w(k_cap + 1);
  std::vector<unsigned long long> id(k_cap + 1);
  outfit::HeapView h{w.data(), id.data(), 1};
  const unsigned n = outfit::select_triplets(mjd, n_obs, P, h);
  for (unsigned s = 0; s < n; ++s) out[s] = h.I(s);
  return (int)n;
}
