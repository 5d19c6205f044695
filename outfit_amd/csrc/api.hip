// api.hip — host side of the C ABI declared in include/outfit_hip.h.
//
// Mirrors what the reference does per call of FitIOD::fit_full_iod after
// prepare_iod (src/initial_orbit_determination/obs_dataset_api.rs:145-207):
// every trajectory is an independent task (process_traj :277-296); here the
// tasks of one device are batched into four kernels per chunk of trajectories.
// There is no CPU fallback: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <cstddef>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/outfit_hip.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

// process-wide DEFAULT math mode (include/outfit_hip.h); -1 = not yet initialised from the environment. A call or a
// session that names its mode (outfit_hip_call_options_t) never reads it.
std::atomic<int> g_math_mode{-1};
int current_math_mode() {
  int m = g_math_mode.load();
  if (m < 0) {
    const char* e = std::getenv("OUTFIT_MATH_MODE");
    m = (e && (std::string(e) == "cr" || std::string(e) == "0")) ? OUTFIT_MATH_CORRECTLY_ROUNDED : OUTFIT_MATH_GLIBC;
    int expected = -1;
    if (!g_math_mode.compare_exchange_strong(expected, m)) m = expected;
  }
  return m;
}
// what one call's options say: the math mode (else the process-wide default at this moment) and, since ABI v7, how many
// rows / doubles the caller's arrays hold (0 = not stated: the library then trusts traj_offset, as the reference trusts
// its own owned data, obs_dataset_api.rs:210-226)
struct CallOpts {
  int mode = 0;
  uint64_t n_obs_capacity = 0, noise_z_capacity = 0;
};
int resolve_options(const outfit_hip_call_options_t* o, CallOpts* out) {
  *out = CallOpts{};
  if (!o) { out->mode = current_math_mode(); return OUTFIT_OK; }
  // struct_size first: only the bytes it covers may be read. A struct of an OLDER (shorter) ABI is accepted as long as
  // it reaches math_mode; fields it does not have take their defaults. reserved[] must be zero so that a later version
  // can give it a meaning.
  if (o->struct_size < offsetof(outfit_hip_call_options_t, math_mode) + sizeof(o->math_mode)) {
    g_last_error = "call options: struct_size does not cover math_mode (set it to sizeof(outfit_hip_call_options_t))";
    return OUTFIT_ERR_INVALID_PARAM;
  }
  auto covers = [o](size_t off) { return o->struct_size >= off + sizeof(uint64_t); };
  if (covers(offsetof(outfit_hip_call_options_t, n_obs_capacity))) out->n_obs_capacity = o->n_obs_capacity;
  if (covers(offsetof(outfit_hip_call_options_t, noise_z_capacity))) out->noise_z_capacity = o->noise_z_capacity;
  if (covers(offsetof(outfit_hip_call_options_t, reserved)) && o->reserved[0] != 0) {
    g_last_error = "call options: reserved words must be zero";
    return OUTFIT_ERR_INVALID_PARAM;
  }
  if (o->math_mode == OUTFIT_MATH_DEFAULT) { out->mode = current_math_mode(); return OUTFIT_OK; }
  if (o->math_mode != OUTFIT_MATH_CORRECTLY_ROUNDED && o->math_mode != OUTFIT_MATH_GLIBC) {
    g_last_error = "call options: math_mode must be OUTFIT_MATH_DEFAULT, _CORRECTLY_ROUNDED or _GLIBC";
    return OUTFIT_ERR_INVALID_PARAM;
  }
  out->mode = o->math_mode;
  return OUTFIT_OK;
}

int fail_nothrow(int code, const char* a, const char* b) noexcept {
  try {
    g_last_error = a;
    if (b) g_last_error += b;
  } catch (...) {
    g_last_error.clear();  // noexcept
  }
  return code;
}
// what a catch block turns the exception in flight into (a shard thread stores it instead of returning it)
int code_of_current_exception(std::string* msg) noexcept {
  int code = OUTFIT_ERR_INTERNAL;
  const char *a = "internal error: unknown exception", *b = nullptr;
  try { throw; }
  catch (const std::bad_alloc&) { code = OUTFIT_ERR_NO_MEMORY; a = "host allocation failed"; }
  catch (const std::exception& e) { a = "internal error: "; b = e.what(); }
  catch (...) {}
  try { *msg = a; if (b) *msg += b; } catch (...) { msg->clear(); }
  return code;
}
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// Nothing may unwind through the C ABI (the caller is Rust, C or ctypes): every entry point that can allocate or start a
// thread is a function-try-block ending in this handler.
// The handlers themselves must not throw: fail_nothrow() falls back to an empty message when even that allocation fails.
#define OUTFIT_ABI_CATCH                                                                               \
  catch (const std::bad_alloc&) { return fail_nothrow(OUTFIT_ERR_NO_MEMORY, "host allocation failed", nullptr); } \
  catch (const std::exception& e) { return fail_nothrow(OUTFIT_ERR_INTERNAL, "internal error: ", e.what()); }     \
  catch (...) { return fail_nothrow(OUTFIT_ERR_INTERNAL, "internal error: unknown exception", nullptr); }

}  // namespace
namespace outfit {
int set_last_error(int code, const std::string& msg) { return fail(code, msg); }  // for ephem_file.hip
}
namespace {

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return fail(OUTFIT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

// IODParamsBuilder::build (mod.rs:544-624); partial_cmp semantics: NaN invalid, +inf allowed
const char* validate_params(const outfit_iod_params_t* p) {
  auto gt0 = [](double x) { return x > 0.0; };
  auto ge0 = [](double x) { return x >= 0.0; };
  if (!ge0(p->noise_scale)) return "noise_scale must be non-negative";
  if (!ge0(p->dt_min) || !ge0(p->dt_max_triplet) || !ge0(p->dtmax)) return "time parameters must be non-negative";
  if (!ge0(p->max_ecc)) return "max_ecc must be >= 0";
  if (!ge0(p->root_imag_eps)) return "root_imag_eps must be >= 0";
  if (!gt0(p->max_perihelion_au)) return "max_perihelion_au must be > 0";
  if (!gt0(p->min_rho2_au)) return "min_rho2_au must be > 0";
  if (!gt0(p->aberth_eps)) return "aberth_eps must be > 0";
  if (!gt0(p->kepler_eps)) return "kepler_eps must be > 0";
  if (!gt0(p->newton_eps)) return "newton_eps must be > 0";
  if (p->newton_max_it == 0) return "newton_max_it must be >= 1";
  if (p->aberth_max_iter == 0) return "aberth_max_iter must be >= 1";
  if (p->max_tested_solutions < 1) return "max_tested_solutions must be >= 1";
  if (!(gt0(p->r2_min_au) && gt0(p->r2_max_au) && p->r2_min_au <= p->r2_max_au))
    return "require 0 < r2_min_au <= r2_max_au";
  return nullptr;
}

unsigned saturate_u32(uint64_t v) {
  return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)v;
}

// Device-memory cache. A blocking call builds and tears down a session: ~25 hipMalloc (GBs of candidate scratch) and
// as many hipFree, each of which synchronises the device — 10+ ms per 100 k-object call. Freed blocks are kept per
// device and handed to the next request of a similar size (never more than 2x + 1 MiB larger than asked), so a host
// that calls repeatedly with batches of one shape allocates once. Owners synchronise their streams before
// releasing (outfit_hip_session::free_all); contents are never assumed to be zero. outfit_hip_trim_cache() returns
// everything to the driver (call it before handing the device to another allocator — torch, RCCL — in the same
// process: parked blocks are invisible to them); OUTFIT_NO_CACHE=1 disables the cache, OUTFIT_CACHE_LIMIT_MB bounds it.
// A session's three streams and 18 events: creating them costs ~10 ms and destroying them ~9 ms (measured), so
// they are parked per device like the memory blocks.
struct StreamSet {
  hipStream_t stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t event[18] = {};
  bool valid = false;
};
struct DeviceCache {
  std::mutex m;
  std::multimap<size_t, void*> blocks[64];
  std::vector<StreamSet> sets[64];
  size_t parked[64] = {};  // bytes held per device; a host that keeps changing batch shapes must not grow this for ever
  size_t limit = [] {
    const char* e = std::getenv("OUTFIT_CACHE_LIMIT_MB");
    return (size_t)(e ? std::max(0ll, std::atoll(e)) : 16384ll) << 20;   // default 16 GiB: the scratch of one 10^6-object session
  }();
  // debugging aid (OUTFIT_CACHE_POISON=1): every block handed out is first filled with 0xFF bytes (NaN doubles, huge
  // indices), recycled or fresh, so that a kernel reading a column before its upload has landed cannot go unnoticed
  bool poison = std::getenv("OUTFIT_CACHE_POISON") != nullptr;
  hipError_t get_streams(StreamSet* out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!disabled && dev >= 0 && dev < 64) {
      std::lock_guard<std::mutex> g(m);
      if (!sets[dev].empty()) { *out = sets[dev].back(); sets[dev].pop_back(); return hipSuccess; }
    }
    StreamSet ss;
    // (stream priorities were measured for the side streams — general kernel beside K2c, K0 / K1 ahead — and change
    // nothing on this runtime: what lets the general kernel in beside K2c is its register footprint, kernels.hip)
    for (auto& st : ss.stream) {
      hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      if (e != hipSuccess) { destroy_streams(ss); return e; }
    }
    for (auto& ev : ss.event) {
      hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) { destroy_streams(ss); return e; }
    }
    ss.valid = true;
    *out = ss;
    return hipSuccess;
  }
  static void destroy_streams(StreamSet& ss) {
    for (auto& st : ss.stream) if (st) (void)hipStreamDestroy(st);
    for (auto& ev : ss.event) if (ev) (void)hipEventDestroy(ev);
    ss = StreamSet();
  }
  void put_streams(StreamSet& ss) {  // the owner has synchronised the streams
    if (!ss.valid) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (disabled || dev < 0 || dev >= 64) { destroy_streams(ss); return; }
    std::lock_guard<std::mutex> g(m);
    sets[dev].push_back(ss);
    ss = StreamSet();
  }
  bool disabled = std::getenv("OUTFIT_NO_CACHE") != nullptr;
  hipError_t get(size_t bytes, void** out, size_t* cap) {
    const hipError_t e = get_block(bytes, out, cap);
    if (e != hipSuccess || !poison) return e;
    hipError_t p = hipMemset(*out, 0xFF, *cap);
    if (p == hipSuccess) p = hipDeviceSynchronize();
    return p;
  }
  hipError_t get_block(size_t bytes, void** out, size_t* cap) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!disabled && dev >= 0 && dev < 64) {
      std::lock_guard<std::mutex> g(m);
      auto it = blocks[dev].lower_bound(bytes);
      if (it != blocks[dev].end() && it->first <= 2 * bytes + (1u << 20)) {
        *out = it->second;
        *cap = it->first;
        parked[dev] -= it->first;
        blocks[dev].erase(it);
        return hipSuccess;
      }
    }
    *cap = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(out, *cap);
    if (e != hipSuccess && !disabled) {  // out of memory with blocks parked in the cache: give them back and retry
      trim();
      e = hipMalloc(out, *cap);
    }
    return e;
  }
  void put(void* p, size_t cap) noexcept {  // called from destructors: a failed map insertion frees the block instead
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (disabled || dev < 0 || dev >= 64) { (void)hipFree(p); return; }
    try {
      std::lock_guard<std::mutex> g(m);
      if (parked[dev] + cap <= limit) {
        blocks[dev].emplace(cap, p);
        parked[dev] += cap;
        return;
      }
    } catch (...) {}
    (void)hipFree(p);  // over the limit: back to the driver
  }
  void trim() {
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::lock_guard<std::mutex> g(m);
    for (int d = 0; d < 64; ++d) {
      if (blocks[d].empty() && sets[d].empty()) continue;
      (void)hipSetDevice(d);
      for (auto& kv : blocks[d]) (void)hipFree(kv.second);
      blocks[d].clear();
      parked[d] = 0;
      for (auto& ss : sets[d]) destroy_streams(ss);
      sets[d].clear();
    }
    (void)hipSetDevice(cur);
  }
};
DeviceCache g_cache;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) { p = nullptr; return hipSuccess; }
    return g_cache.get(count * sizeof(T), reinterpret_cast<void**>(&p), &cap);
  }
  void release() {
    if (p) g_cache.put(p, cap);
    p = nullptr;
    n = cap = 0;
  }
  // a buffer that goes out of scope (an early return, an exception on the way to OUTFIT_ERR_NO_MEMORY) gives its block
  // back to the cache; sessions release theirs in free_all(), where the order against stream synchronisation matters
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
};

enum KernelId { K_TRIPLETS = 0, K_PREPASS, K_ROOTS, K_CORRECT_FAST, K_CORRECT, K_RMS, K_RMS_SLOW, K_SELECT, K_REFINE, K_DIFFCOR, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_triplet_select", "k_observer_prepass",   "k_gauss_roots",
                                           "k_gauss_correct_fast", "k_gauss_correct",
                                           "k_orbit_rms",      "k_orbit_rms_slow",     "k_select_best",
                                           "k_refine_best",    "k_diffcor"};

}  // namespace

namespace {
// OUTFIT_TRACE=1: host-side phase times of the blocking call on stderr (tuning aid)
struct Trace {
  bool on = std::getenv("OUTFIT_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[outfit trace] %-28s +%.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
    t0 = t;
  }
};
thread_local Trace g_trace;
}  // namespace

struct outfit_hip_session {
  int device = 0;
  outfit_iod_params_t params{};
  int math_mode = 0;  // OUTFIT_MATH_* captured at creation
  uint64_t n_traj = 0, n_obs = 0;
  bool has_ephem = false, has_geo = false;
  unsigned n_real = 1, k_cap = 0, cand_per_traj = 0;
  uint64_t chunk = 0;

  DevBuf<unsigned long long> traj_offset, traj_key;
  bool noise_device = false;
  unsigned long long noise_seed = 0, traj_key_base = 0;
  DevBuf<double> mjd, ra, dec, sra, sdec, helio, geo, obs_pos, cos_dec, obs_rec, unit[2], noise, coef;
  DevBuf<double> heap_w, sum, dirS, roots, state, psi0, sel_bound;
  DevBuf<unsigned long long> heap_id, triplets[2], queue, slow_list, slow_count;
  DevBuf<unsigned> sel_list;  // K3 -> K3b: contender candidates per trajectory of the chunk (kernels.h)
  DevBuf<unsigned> n_triplets[2], flags, n_win;
  unsigned n_cu = 256;
  bool triplet_lds = false;  // K0 keeps its heap in LDS (small k_cap, <= 1024 reduced observations)
  // K0, wave form: trajectories whose reduced length reaches wave_min_obs (absolute ids, ascending; 0 = form unused)
  unsigned wave_min_obs = 0;
  size_t wave_lds = 0;
  std::vector<unsigned> wave_list_host;
  uint64_t max_traj_len = 0;   // observations of the longest trajectory (chooses the screen's route, session_run)
  DevBuf<unsigned> wave_list;
  DevBuf<outfit::GaussResultRec> results;
  DevBuf<int> error_flag;
  // differential correction (allocated by the first run_lsq)
  DevBuf<double> dc_res, dc_cov, dc_qr;
  DevBuf<unsigned char> dc_sel;
  DevBuf<outfit::GaussResultRec> dc_initial;
  DevBuf<outfit::LsqResultRec> dc_out;
  DevBuf<outfit::ObsFitRec> dc_fit;
  bool prepass_done = false, lsq_ran = false;
  // fit_single_traj sorts a trajectory's observations by epoch (obs_dataset_api.rs:222-223); when the caller's rows are
  // not already in that order, resident row i holds the caller's row perm[i] (both rebased); empty = identity
  std::vector<uint64_t> perm;
  // Streamed blocking call (outfit_hip_fit_full_iod): only traj_offset and mjd are uploaded when the session is built;
  // the other columns follow in two groups while kernels that do not need them yet are already running
  // (group 1 = ra, dec, obs_helio_equ — and sigma_ra, sigma_dec when noisy realisations are drawn: before K1's
  // direction cosines and K2a; group 2 = sigma_ra, sigma_dec otherwise, obs_geo_ecl, the Chebyshev table: before K1's
  // observer positions and K2c). See session_run.
  struct Deferred { DevBuf<double>* d; const double* src; size_t rows; unsigned width; int group; bool use_perm; };
  std::vector<Deferred> deferred;
  bool streaming = false;
  const outfit::GaussResultRec* iod_out = nullptr;  // where the last IOD run wrote (d_out or `results`)
  outfit::EphemTable table{};
  outfit::GaussParams gp{};
  outfit::TripletParams tp{};

  StreamSet streams;  // owns every stream and event handle below (device cache)
  hipStream_t own_stream = nullptr;
  hipStream_t aux_stream = nullptr;  // general correction kernel, beside K2c
  hipStream_t k0_stream = nullptr;   // triplet selection, ahead of the candidate kernels (and of the previous run's tail)
  hipEvent_t ev_k3[2] = {nullptr, nullptr};  // last consumer of triplet buffer b has finished
  bool k3_recorded[2] = {false, false};
  hipEvent_t ev_unit_free[2] = {nullptr, nullptr};  // the last reader of unit buffer b (K2a, both correction kernels) has finished
  bool unit_free_recorded[2] = {false, false};
  unsigned long long chunk_seq = 0, run_seq = 0;
  hipEvent_t ev_start = nullptr, ev_prepass = nullptr, ev_fast = nullptr, ev_general = nullptr, ev_unit = nullptr;
  hipEvent_t ev_sub[8] = {};
  hipStream_t last_stream = nullptr;  // informational only: never synchronised (it is the caller's handle)
  hipEvent_t ev_done = nullptr;  // recorded after the last kernel of every run / run_lsq, on the stream it went to
  bool done_recorded = false;
  bool profiling = false;
  struct Timed { int id; hipEvent_t a, b; };
  std::vector<Timed> timed;
  float kernel_ms[K_COUNT] = {};
  int kernel_launches[K_COUNT] = {};

  void free_all() {
    // blocks go back to the cache, not to hipFree (which would wait for the device): nothing may still use them
    // (the caller's stream may no longer exist: its last run is awaited through an event of ours)
    if (ev_done && done_recorded) (void)hipEventSynchronize(ev_done);
    for (hipStream_t st : {own_stream, aux_stream, k0_stream})
      if (st) (void)hipStreamSynchronize(st);
    g_trace.mark("  destroy: sync");
    traj_offset.release(); traj_key.release(); mjd.release(); ra.release(); dec.release(); sra.release(); sdec.release();
    helio.release(); geo.release(); obs_pos.release(); cos_dec.release(); obs_rec.release(); unit[0].release(); unit[1].release(); noise.release(); coef.release();
    heap_w.release(); sum.release(); dirS.release(); roots.release(); state.release();
    psi0.release(); slow_list.release(); slow_count.release(); sel_bound.release(); sel_list.release();
    heap_id.release(); triplets[0].release(); triplets[1].release(); queue.release(); wave_list.release();
    n_triplets[0].release(); n_triplets[1].release(); flags.release();
    n_win.release(); results.release(); error_flag.release();
    dc_res.release(); dc_cov.release(); dc_qr.release(); dc_sel.release();
    dc_initial.release(); dc_out.release(); dc_fit.release();
    for (auto& t : timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    timed.clear();
    g_trace.mark("  destroy: buffers");
    g_cache.put_streams(streams);
    own_stream = aux_stream = k0_stream = nullptr;
    ev_done = ev_start = ev_prepass = ev_fast = ev_general = ev_unit = nullptr;
    for (int b = 0; b < 2; ++b) ev_k3[b] = ev_unit_free[b] = nullptr;
    for (auto& e : ev_sub) e = nullptr;
    g_trace.mark("  destroy: streams+events");
  }
};

namespace {

// Wait for the last run / run_lsq. The stream it went to is the CALLER's handle and may no longer exist (a torch
// stream of a finished step): the wait goes through the session's own event, recorded behind the last kernel.
hipError_t wait_last_run(outfit_hip_session* s) {
  if (s->ev_done && s->done_recorded) return hipEventSynchronize(s->ev_done);
  return s->own_stream ? hipStreamSynchronize(s->own_stream) : hipSuccess;
}

int check_batch(const outfit_iod_params_t* params, const outfit_obs_batch_t* b, const outfit_ephem_t* eph,
                const CallOpts& opts = CallOpts{}) {
  if (!b) return fail(OUTFIT_ERR_INVALID_BATCH, "batch is NULL");
  if (b->n_traj > 0 && !b->traj_offset) return fail(OUTFIT_ERR_INVALID_BATCH, "traj_offset is NULL");
  if (b->n_traj == 0) return OUTFIT_OK;
  const uint64_t o0 = b->traj_offset[0];
  for (uint64_t t = 0; t < b->n_traj; ++t) {
    if (b->traj_offset[t + 1] < b->traj_offset[t])
      return fail(OUTFIT_ERR_INVALID_BATCH, "traj_offset must be non-decreasing");
    if (b->traj_offset[t + 1] - b->traj_offset[t] >= (1ull << 21))
      return fail(OUTFIT_ERR_INVALID_BATCH, "a trajectory has 2^21 or more observations");
  }
  const uint64_t n_obs = b->traj_offset[b->n_traj] - o0;
  // the columns are indexed by the offsets themselves (row traj_offset[t] .. traj_offset[t+1]-1): a table that is
  // monotone but runs past the caller's arrays is an out-of-bounds read unless the caller says how long they are
  if (opts.n_obs_capacity && b->traj_offset[b->n_traj] > opts.n_obs_capacity)
    return fail(OUTFIT_ERR_INVALID_BATCH, "traj_offset[n_traj] = " + std::to_string(b->traj_offset[b->n_traj]) +
                " runs past the observation columns (options.n_obs_capacity = " + std::to_string(opts.n_obs_capacity) + " rows)");
  const bool device_geometry = b->obs_body_fixed != nullptr || b->mjd_ut1 != nullptr;
  if (device_geometry && (!b->obs_body_fixed || !b->mjd_ut1 || !eph))
    return fail(OUTFIT_ERR_INVALID_BATCH, "obs_body_fixed, mjd_ut1 and an ephemeris are needed together");
  if (n_obs > 0 && (!b->mjd_tt || !b->ra || !b->dec || !b->sigma_ra || !b->sigma_dec ||
                    (!b->obs_helio_equ && !device_geometry)))
    return fail(OUTFIT_ERR_INVALID_BATCH, "an observation array is NULL");
  if (b->noise_source != OUTFIT_NOISE_HOST && b->noise_source != OUTFIT_NOISE_DEVICE)
    return fail(OUTFIT_ERR_INVALID_BATCH, "noise_source must be OUTFIT_NOISE_HOST or OUTFIT_NOISE_DEVICE");
  if (params->n_noise_realizations > 0 && b->noise_source == OUTFIT_NOISE_HOST && !b->noise_z)
    return fail(OUTFIT_ERR_INVALID_BATCH,
                "noise_z is required when n_noise_realizations > 0 (or set noise_source = OUTFIT_NOISE_DEVICE)");
  const uint64_t cpt = (uint64_t)params->max_triplets * (1 + params->n_noise_realizations);
  if (params->n_noise_realizations >= (1ull << 31) || cpt >= (1ull << 31))
    return fail(OUTFIT_ERR_INVALID_PARAM, "max_triplets * (1 + n_noise_realizations) must be < 2^31");
  if (opts.noise_z_capacity && b->noise_z && params->n_noise_realizations > 0 && b->noise_source == OUTFIT_NOISE_HOST) {
    const uint64_t per_traj = (uint64_t)params->max_triplets * params->n_noise_realizations * 6;   // < 2^34
    if (b->n_traj > opts.noise_z_capacity / per_traj)
      return fail(OUTFIT_ERR_INVALID_BATCH, "noise_z holds " + std::to_string(opts.noise_z_capacity) + " doubles (options."
                  "noise_z_capacity); n_traj x max_triplets x n_noise_realizations x 6 are needed");
  }
  if (eph) {
    if (!eph->coef || eph->n_blocks == 0) return fail(OUTFIT_ERR_INVALID_BATCH, "ephemeris table is empty");
    for (int k = 0; k < 3; ++k)
      if (eph->ncoef[k] < 2 || eph->ncoef[k] > 32 || eph->nsub[k] < 1)
        return fail(OUTFIT_ERR_INVALID_BATCH, "ephemeris ncoef must be in [2,32] and nsub >= 1");
    if (!(eph->block_days > 0.0)) return fail(OUTFIT_ERR_INVALID_BATCH, "ephemeris block_days must be > 0");
  }
  return OUTFIT_OK;
}

#define TRY_RC(expr)              \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != OUTFIT_OK) return rc__; \
  } while (0)

// f64::total_cmp as a signed-integer key (library/core/src/num/f64.rs)
inline int64_t total_order_key(double x) {
  int64_t b;
  std::memcpy(&b, &x, sizeof b);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}

// `obs_vec_refs.sort_by(|a, b| a.mjd_tt().total_cmp(&b.mjd_tt()))` (obs_dataset_api.rs:222-223; sort_by is stable):
// fills perm (rebased row -> caller's rebased row) when some trajectory is not already in that order.
void sort_permutation(const outfit_obs_batch_t* b, std::vector<uint64_t>& perm) {
  perm.clear();
  if (!b->n_traj || !b->mjd_tt) return;
  const uint64_t o0 = b->traj_offset[0];
  // the check reads every epoch once (16 MB for 2 x 10^6 observations: ~1 ms on one core, in front of the first
  // kernel of a blocking call): large batches are checked by up to four threads, a range of trajectories each
  auto range_sorted = [b](uint64_t t0, uint64_t t1) {
    for (uint64_t t = t0; t < t1; ++t)
      for (uint64_t i = b->traj_offset[t] + 1; i < b->traj_offset[t + 1]; ++i)
        if (total_order_key(b->mjd_tt[i - 1]) > total_order_key(b->mjd_tt[i])) return false;
    return true;
  };
  bool sorted = true;
  const uint64_t n_all = b->traj_offset[b->n_traj] - o0;
  const unsigned n_thr = n_all >= (1u << 18) && b->n_traj >= 4 ? 4u : 1u;
  if (n_thr == 1) {
    sorted = range_sorted(0, b->n_traj);
  } else {
    bool ok[4] = {true, true, true, true};
    std::thread th[3];
    unsigned started = 0;  // a thread that cannot be created (resource limits) leaves its range to this one: no exception
    for (unsigned k = 1; k < n_thr; ++k) {  //                                                  may cross the C ABI
      try {
        th[k - 1] = std::thread([&ok, &range_sorted, b, k, n_thr] {
          ok[k] = range_sorted(b->n_traj * k / n_thr, b->n_traj * (k + 1) / n_thr);
        });
        started = k;
      } catch (const std::system_error&) {
        break;
      }
    }
    ok[0] = range_sorted(0, b->n_traj / n_thr);
    for (unsigned k = started + 1; k < n_thr; ++k) ok[k] = range_sorted(b->n_traj * k / n_thr, b->n_traj * (k + 1) / n_thr);
    for (unsigned k = 1; k <= started; ++k) th[k - 1].join();
    sorted = ok[0] && ok[1] && ok[2] && ok[3];
  }
  if (sorted) return;
  const uint64_t n = b->traj_offset[b->n_traj] - o0;
  perm.resize(n);
  for (uint64_t i = 0; i < n; ++i) perm[i] = i;
  for (uint64_t t = 0; t < b->n_traj; ++t)
    std::stable_sort(perm.begin() + (b->traj_offset[t] - o0), perm.begin() + (b->traj_offset[t + 1] - o0),
                     [&](uint64_t x, uint64_t y) { return total_order_key(b->mjd_tt[o0 + x]) < total_order_key(b->mjd_tt[o0 + y]); });
}

template <typename T>
int upload(DevBuf<T>& d, const T* src, size_t n);

// a per-observation column of `width` doubles per row, gathered through the sort permutation when there is one
int upload_rows(DevBuf<double>& d, const double* src, size_t n_rows, unsigned width, const std::vector<uint64_t>& perm) {
  if (perm.empty() || !n_rows) return upload(d, src, n_rows * width);
  std::vector<double> tmp(n_rows * width);
  for (size_t i = 0; i < n_rows; ++i)
    for (unsigned k = 0; k < width; ++k) tmp[i * width + k] = src[perm[i] * width + k];
  return upload(d, tmp.data(), tmp.size());
}

int copy_rows(DevBuf<double>& d, const double* src, size_t n_rows, unsigned width, const std::vector<uint64_t>* perm) {
  if (!n_rows) return OUTFIT_OK;
  if (!perm || perm->empty()) {
    HIP_TRY(hipMemcpy(d.p, src, n_rows * width * sizeof(double), hipMemcpyHostToDevice));
    return OUTFIT_OK;
  }
  std::vector<double> tmp(n_rows * width);
  for (size_t i = 0; i < n_rows; ++i)
    for (unsigned k = 0; k < width; ++k) tmp[i * width + k] = src[(*perm)[i] * width + k];
  HIP_TRY(hipMemcpy(d.p, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
  return OUTFIT_OK;
}
// allocate a resident column and upload it now, or (streamed call, group > 0) when its group is flushed
int put_rows(outfit_hip_session* s, DevBuf<double>& d, const double* src, size_t n_rows, unsigned width, int group,
             bool use_perm = true) {
  HIP_TRY(d.alloc(n_rows * width));
  if (s->streaming && group > 0) {
    s->deferred.push_back({&d, src, n_rows, width, group, use_perm});
    return OUTFIT_OK;
  }
  return copy_rows(d, src, n_rows, width, use_perm ? &s->perm : nullptr);
}

int flush_group(outfit_hip_session* s, int group) {
  for (const auto& u : s->deferred)
    if (u.group == group) TRY_RC(copy_rows(*u.d, u.src, u.rows, u.width, u.use_perm ? &s->perm : nullptr));
  g_trace.mark(group == 1 ? "upload group 1" : "upload group 2");
  return OUTFIT_OK;
}

template <typename T>
int upload(DevBuf<T>& d, const T* src, size_t n) {
  HIP_TRY(d.alloc(n));
  if (n) HIP_TRY(hipMemcpy(d.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return OUTFIT_OK;
}


int session_build(outfit_hip_session* s, const outfit_iod_params_t* params, const outfit_obs_batch_t* b,
                  const outfit_ephem_t* eph) {
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(g_cache.get_streams(&s->streams));
  {
    StreamSet& ss = s->streams;
    s->own_stream = ss.stream[0]; s->aux_stream = ss.stream[1]; s->k0_stream = ss.stream[2];
    int k = 0;
    s->ev_done = ss.event[k++];
    for (hipEvent_t& e : s->ev_k3) e = ss.event[k++];
    for (hipEvent_t& e : s->ev_unit_free) e = ss.event[k++];
    s->ev_start = ss.event[k++]; s->ev_prepass = ss.event[k++]; s->ev_fast = ss.event[k++];
    s->ev_general = ss.event[k++]; s->ev_unit = ss.event[k++];
    for (hipEvent_t& e : s->ev_sub) e = ss.event[k++];  // 1 + 2 + 2 + 5 + 8 = 18
  }
  g_trace.mark("  streams+events");
  s->params = *params;  // (s->math_mode was fixed by session_create)
  s->n_traj = b->n_traj;
  const uint64_t o0 = b->n_traj ? b->traj_offset[0] : 0;
  s->n_obs = b->n_traj ? b->traj_offset[b->n_traj] - o0 : 0;
  s->n_real = (unsigned)(1 + params->n_noise_realizations);
  s->k_cap = params->max_triplets;
  s->cand_per_traj = s->k_cap * s->n_real;
  s->has_ephem = eph != nullptr;
  s->has_geo = b->obs_geo_ecl != nullptr || (b->obs_body_fixed && b->mjd_ut1 && eph);

  // rebased offsets
  std::vector<unsigned long long> off(s->n_traj + 1, 0ull);
  for (uint64_t t = 0; t <= s->n_traj && s->n_traj; ++t) off[t] = b->traj_offset[t] - o0;
  TRY_RC(upload(s->traj_offset, off.data(), off.size()));
  const size_t n = (size_t)s->n_obs;
  g_trace.mark("  offsets");
  sort_permutation(b, s->perm);
  g_trace.mark("  sortedness");
  if (b->obs_body_fixed && b->mjd_ut1 && eph) s->streaming = false;  // the geometry pre-pass below needs everything resident
  TRY_RC(put_rows(s, s->mjd, n ? b->mjd_tt + o0 : nullptr, n, 1, 0));
  TRY_RC(put_rows(s, s->ra, n ? b->ra + o0 : nullptr, n, 1, 1));
  TRY_RC(put_rows(s, s->dec, n ? b->dec + o0 : nullptr, n, 1, 1));
  // the uncertainties are first read by K2c — unless there are noisy realisations: gather_angles (K2a) scales the
  // draws with them (gauss.rs:355-364), so they must be resident before the first K2a slice (group 1)
  const int sigma_group = params->n_noise_realizations > 0 ? 1 : 2;
  TRY_RC(put_rows(s, s->sra, n ? b->sigma_ra + o0 : nullptr, n, 1, sigma_group));
  TRY_RC(put_rows(s, s->sdec, n ? b->sigma_dec + o0 : nullptr, n, 1, sigma_group));
  const bool device_geometry = b->obs_body_fixed && b->mjd_ut1 && eph;
  if (device_geometry) HIP_TRY(s->helio.alloc(3 * n));
  else TRY_RC(put_rows(s, s->helio, n ? b->obs_helio_equ + 3 * o0 : nullptr, n, 3, 1));
  HIP_TRY(s->cos_dec.alloc(n));
  HIP_TRY(s->obs_rec.alloc((size_t)outfit::kObsRecDoubles * n));
  for (int b = 0; b < 2; ++b) HIP_TRY(s->unit[b].alloc(3 * n));  // double-buffered: written ahead on k0_stream
  if (s->has_ephem && s->has_geo) {
    if (device_geometry) HIP_TRY(s->geo.alloc(3 * n));
    else TRY_RC(put_rows(s, s->geo, n ? b->obs_geo_ecl + 3 * o0 : nullptr, n, 3, 2));
    HIP_TRY(s->obs_pos.alloc(3 * n));
    unsigned per_block = 0;
    for (int k = 0; k < 3; ++k) {
      s->table.ncoef[k] = eph->ncoef[k];
      s->table.nsub[k] = eph->nsub[k];
      s->table.off[k] = per_block;
      per_block += 3u * eph->ncoef[k] * eph->nsub[k];
    }
    s->table.per_block = per_block;
    s->table.jd_start = eph->jd_start; s->table.jd_end = eph->jd_end;
    s->table.block_days = eph->block_days; s->table.emrat = eph->emrat;
    s->table.n_blocks = eph->n_blocks;
    TRY_RC(put_rows(s, s->coef, eph->coef, (size_t)eph->n_blocks * per_block, 1, 2, /*use_perm=*/false));
    s->table.coef = s->coef.p;
  }
  s->noise_device = params->n_noise_realizations > 0 && b->noise_source == OUTFIT_NOISE_DEVICE;
  s->noise_seed = b->noise_seed;
  s->traj_key_base = b->traj_key_base;
  if (s->noise_device) {
    if (b->traj_key) TRY_RC(upload(s->traj_key, reinterpret_cast<const unsigned long long*>(b->traj_key), (size_t)s->n_traj));
  } else if (params->n_noise_realizations > 0) {
    TRY_RC(upload(s->noise, b->noise_z,
                  (size_t)s->n_traj * s->k_cap * (size_t)params->n_noise_realizations * 6));
  }
  g_trace.mark("  input columns");
  HIP_TRY(s->results.alloc((size_t)s->n_traj));
  // status -1 until an IOD run fills the records (outfit_hip_session_run_lsq reads them as "no seed")
  if (s->n_traj) HIP_TRY(hipMemset(s->results.p, 0xFF, (size_t)s->n_traj * sizeof(outfit::GaussResultRec)));
  HIP_TRY(s->error_flag.alloc(1));
  HIP_TRY(hipMemset(s->error_flag.p, 0, sizeof(int)));
  if (device_geometry && n) {
    // row n2: what OutfitCache::build computes per observation, once per session (inputs are resident)
    DevBuf<double> ut1, fixed;
    TRY_RC(upload_rows(ut1, b->mjd_ut1 + o0, n, 1, s->perm));
    TRY_RC(upload_rows(fixed, b->obs_body_fixed + 3 * o0, n, 3, s->perm));
    outfit::GeometryArgs ga{};
    ga.n_obs = s->n_obs; ga.mjd_tt = s->mjd.p; ga.mjd_ut1 = ut1.p; ga.body_fixed = fixed.p; ga.T = s->table;
    ga.geo_ecl = s->geo.p; ga.helio_equ = s->helio.p; ga.error_flag = s->error_flag.p;
    hipError_t e = outfit::launch_observer_geometry(ga, s->own_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->own_stream);
    ut1.release();
    fixed.release();
    if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  }

  g_trace.mark("  results+flags");
  // chunking: bound the candidate scratch (~200 B per candidate) and the heap scratch. Sized for a 288 GB device:
  // up to 64 Mi candidates (~13 GB of scratch) and 1 Mi trajectories per chunk — 10^6 objects x 100 triplets are two
  // chunks; every chunk boundary costs a fill/drain of the persistent kernels (measured at 10 chunks: +8 %).
  uint64_t chunk = 1ull << 20;
  if (s->cand_per_traj) chunk = std::min<uint64_t>(chunk, std::max<uint64_t>(1, (64ull << 20) / s->cand_per_traj));
  if (const char* e = std::getenv("OUTFIT_IOD_CHUNK")) {  // testing aid: force the multi-chunk path on small batches
    const long long c = std::atoll(e);
    if (c > 0) chunk = std::min<uint64_t>(chunk, (uint64_t)c);
  }
  chunk = std::max<uint64_t>(1, std::min<uint64_t>(chunk, s->n_traj));
  s->chunk = chunk;
  const size_t n_cand = (size_t)chunk * s->cand_per_traj;
  const size_t heap_stride = (size_t)((chunk + 63) / 64 * 64);
  HIP_TRY(s->heap_w.alloc(heap_stride * s->k_cap));
  HIP_TRY(s->heap_id.alloc(heap_stride * s->k_cap));
  for (int b = 0; b < 2; ++b) {  // double-buffered: K0 of the next chunk / run may run ahead
    HIP_TRY(s->triplets[b].alloc((size_t)chunk * s->k_cap));
    HIP_TRY(s->n_triplets[b].alloc((size_t)chunk));
  }
  HIP_TRY(s->flags.alloc(n_cand));
  HIP_TRY(s->n_win.alloc(n_cand));
  HIP_TRY(s->sum.alloc(n_cand));
  HIP_TRY(s->dirS.alloc(9 * n_cand));
  HIP_TRY(s->roots.alloc((size_t)outfit::kRootSlots * n_cand));
  HIP_TRY(s->state.alloc(7 * n_cand));
  HIP_TRY(s->queue.alloc(16));  // [0] fast kernel, [1] general kernel, [2] k_diffcor; [8..15] diagnostic builds
  HIP_TRY(s->psi0.alloc(2 * n_cand));
  HIP_TRY(s->slow_list.alloc(n_cand));
  HIP_TRY(s->slow_count.alloc(2));
  HIP_TRY(s->sel_bound.alloc(2 * chunk));
  HIP_TRY(s->sel_list.alloc((size_t)chunk * (outfit::kContenderListCap + 1u)));
  g_trace.mark("  scratch alloc");
  int n_cu = 0;
  HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s->device));
  s->n_cu = n_cu > 0 ? (unsigned)n_cu : 256u;

  s->gp.max_ecc = params->max_ecc;
  s->gp.max_perihelion_au = params->max_perihelion_au;
  s->gp.min_rho2_au = params->min_rho2_au;
  s->gp.aberth_eps = params->aberth_eps;
  s->gp.kepler_eps = params->kepler_eps;
  s->gp.r2_min_au = params->r2_min_au;
  s->gp.r2_max_au = params->r2_max_au;
  s->gp.newton_eps = params->newton_eps;
  s->gp.root_imag_eps = params->root_imag_eps;
  s->gp.aberth_max_iter = params->aberth_max_iter;
  s->gp.max_tested_solutions = saturate_u32(params->max_tested_solutions);
  s->gp.newton_max_it = saturate_u32(params->newton_max_it);
  s->gp.math_mode = s->math_mode;
  s->gp.derive();
  s->tp.dt_min = params->dt_min;
  s->tp.dt_max = params->dt_max_triplet;
  s->tp.inv_dtw = 1.0 / params->optimal_interval_time;  // f64::recip (mod.rs:347)
  s->tp.max_keep = params->max_obs_for_triplets;
  s->tp.k_cap = params->max_triplets;
  {
    uint64_t max_len = 0;
    for (uint64_t t = 0; t < s->n_traj; ++t) max_len = std::max<uint64_t>(max_len, off[t + 1] - off[t]);
    s->max_traj_len = max_len;
    const uint64_t reduced = std::min<uint64_t>(max_len, std::max<uint64_t>(params->max_obs_for_triplets, 3));
    s->triplet_lds = reduced <= 1024 && (uint64_t)s->k_cap * 64 * 12 <= 160 * 1024 - 1024 && s->k_cap > 0;
    // Long trajectories go to the wave-per-trajectory form of K0 (device/triplets.hpp): from 48 reduced epochs on
    // (C(48,3) = 17 296 scored triplets) one lane's chain is longer than a whole small batch's other kernels; the
    // bench cadence (20 epochs) stays with the lane form. LDS per wave: epochs + heap (weight, 64-bit id).
    // A SMALL batch takes the wave form whatever the length: with <= 4096 trajectories every wave is resident at once and
    // a trajectory's selection is as long as ITS chain (64 triplets scored per step, heap operations only for the
    // survivors), not as long as one lane's walk over all C(n,3) — what an isolated run or a blocking call waits for.
    unsigned wave_min = s->n_traj <= 4096 ? 3 : 48;
    if (const char* e = std::getenv("OUTFIT_K0_WAVE_MIN")) {  // testing / measurement aid: 3 = every trajectory, 0 = never
      const long long v = std::atoll(e);
      if (v >= 0) wave_min = (unsigned)std::min<long long>(v, 1ll << 30);
    }
    const uint64_t lds = (((reduced + 1) & ~1ull) + 2ull * s->k_cap) * sizeof(double);  // epochs (padded to 16 B), heap slots of 16 B
    if (wave_min && s->k_cap > 0 && reduced >= wave_min && lds <= 60ull * 1024 && s->n_traj < (1ull << 32)) {
      for (uint64_t t = 0; t < s->n_traj; ++t)   // the lane kernels skip a trajectory by the same function
        if (outfit::wave_form_owns(off[t + 1] - off[t], s->tp.max_keep, wave_min))
          s->wave_list_host.push_back((unsigned)t);
      if (!s->wave_list_host.empty()) {
        s->wave_min_obs = wave_min;
        s->wave_lds = (size_t)lds;
        TRY_RC(upload(s->wave_list, s->wave_list_host.data(), s->wave_list_host.size()));
      }
    }
  }
  return OUTFIT_OK;
}

struct Stamp {
  outfit_hip_session* s;
  hipStream_t st;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  Stamp(outfit_hip_session* s_, hipStream_t st_, int id_) : s(s_), st(st_), id(id_) {
    if (s->profiling) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, st);
    }
  }
  ~Stamp() {
    if (s->profiling) {
      (void)hipEventRecord(b, st);
      s->timed.push_back({id, a, b});
    }
  }
};

int session_run(outfit_hip_session* s, hipStream_t st, outfit::GaussResultRec* d_out) {
  HIP_TRY(hipSetDevice(s->device));
  s->last_stream = st;
  s->iod_out = d_out ? d_out : s->results.p;
  if (s->n_traj == 0) return OUTFIT_OK;
  hipStream_t aux = s->aux_stream;

  const bool on_device_earth = s->has_ephem && s->has_geo;
  const double* obs_pos = on_device_earth ? s->obs_pos.p : s->helio.p;
  // K1 in two halves. The direction cosines feed K2a: evaluated on k0_stream ahead of the K0 slices, into the
  // unit buffer of this run (the previous run's K2a may still be reading the other one). Observer positions
  // and cos(dec) are first read by K2c: evaluated on aux beside K2a, once everything enqueued on the caller's
  // stream so far (previous run's K2c and corrector, which read them) has finished.
  const unsigned ub = (unsigned)(s->run_seq++ & 1u);
  outfit::PrepassArgs pa{};
  pa.n_obs = s->n_obs; pa.mjd = s->mjd.p; pa.dec = s->dec.p; pa.geo_ecl = s->geo.p; pa.obs_pos = s->obs_pos.p;
  pa.cos_dec = s->cos_dec.p; pa.unit = s->unit[ub].p; pa.ra = s->ra.p; pa.has_table = on_device_earth;
  pa.obs_rec = s->obs_rec.p; pa.sra = s->sra.p; pa.sdec = s->sdec.p; pa.helio = s->helio.p;
  pa.T = s->table; pa.error_flag = s->error_flag.p; pa.math_mode = s->math_mode;
  // Streamed blocking call: the inputs of these two launches are still on their way; each is enqueued (first chunk)
  // right after its upload group, while K0 / the candidate kernels already run (see `Deferred`).
  const bool streamed = s->streaming && !s->deferred.empty();
  auto launch_unit = [&]() -> int {
    // the buffer's previous readers: the candidate kernels of the run before the previous one
    if (s->unit_free_recorded[ub]) HIP_TRY(hipStreamWaitEvent(s->k0_stream, s->ev_unit_free[ub], 0));
    pa.parts = outfit::kPrepassUnit;
    Stamp stamp(s, s->k0_stream, K_PREPASS);
    HIP_TRY(outfit::launch_observer_prepass(pa, s->k0_stream));
    return OUTFIT_OK;
  };
  auto launch_observer = [&]() -> int {
    pa.parts = outfit::kPrepassObserver;
    Stamp stamp(s, aux, K_PREPASS);
    HIP_TRY(outfit::launch_observer_prepass(pa, aux));
    HIP_TRY(hipEventRecord(s->ev_prepass, aux));
    s->prepass_done = true;
    return OUTFIT_OK;
  };
  HIP_TRY(hipEventRecord(s->ev_start, st));
  HIP_TRY(hipStreamWaitEvent(aux, s->ev_start, 0));
  if (!streamed) {
    TRY_RC(launch_unit());
    TRY_RC(launch_observer());
  }
  // Two latency-bound kernels run on their own streams in the shadow of VALU-bound ones:
  //   * K0 (sequential heap per lane) in kSub slices on k0_stream, each one slice ahead of the K2a slice
  //     that consumes its triplets. K0 reads only resident inputs and writes a double-buffered triplet
  //     table, so when runs are enqueued back to back it also runs under the tail of the previous run;
  //   * the general correction kernel (few items, long serial chains) on aux beside K2c, which first
  //     handles every regular candidate and then the slow list.
  //     k0  :  K0(0) K0(1) K0(2) K0(3)
  //     aux :                                               general
  //     main:        K2a(0) K2a(1) K2a(2) K2a(3) prelim fast         rms(regular) rms(slow) select
  unsigned kSub = 4;
  if (const char* e = std::getenv("OUTFIT_K0_SLICES")) {  // tuning aid
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) kSub = (unsigned)v;
  }
  for (uint64_t t0 = 0; t0 < s->n_traj; t0 += s->chunk) {
    const uint64_t nt = std::min<uint64_t>(s->chunk, s->n_traj - t0);
    const unsigned long long n_cand = (unsigned long long)nt * s->cand_per_traj;
    const unsigned n_sub = nt >= 4096 * kSub ? kSub : 1;

    outfit::CandidateIndex ci{};
    ci.n_cand = n_cand; ci.traj0 = t0; ci.cand_per_traj = s->cand_per_traj; ci.n_real = s->n_real;
    ci.k_cap = s->k_cap; ci.traj_offset = s->traj_offset.p;
    const unsigned tb = (unsigned)(s->chunk_seq++ & 1u);  // triplet buffer of this chunk
    ci.n_triplets = s->n_triplets[tb].p; ci.triplets = s->triplets[tb].p;
    ci.mjd = s->mjd.p; ci.ra = s->ra.p; ci.dec = s->dec.p; ci.sra = s->sra.p; ci.sdec = s->sdec.p;
    ci.helio = s->helio.p; ci.noise_z = s->noise.p; ci.noise_scale = s->params.noise_scale;
    ci.unit = s->unit[ub].p; ci.n_obs_total = s->n_obs;
    ci.noise_device = s->noise_device; ci.noise_seed = s->noise_seed; ci.traj_key_base = s->traj_key_base;
    ci.traj_key = s->traj_key.p;
    outfit::CandidateScratch w{};
    w.flags = s->flags.p; w.S = s->dirS.p; w.roots = s->roots.p; w.state = s->state.p;
    w.sum = s->sum.p; w.n_win = s->n_win.p; w.psi0 = s->psi0.p; w.slow_list = s->slow_list.p;
    w.slow_count = s->slow_count.p;

    // K0 reads only the resident inputs: it may start as soon as the last consumer of its triplet
    // buffer (K3 two chunks / runs ago) has finished — i.e. under the tail of the previous run
    hipStream_t k0 = s->k0_stream;
    if (s->k3_recorded[tb]) HIP_TRY(hipStreamWaitEvent(k0, s->ev_k3[tb], 0));
    for (unsigned i = 0; i < n_sub; ++i) {
      const uint64_t a0 = nt * i / n_sub, a1 = nt * (i + 1) / n_sub;
      outfit::TripletArgs ta{};
      ta.n_traj = a1 - a0; ta.traj0 = t0 + a0; ta.traj_offset = s->traj_offset.p; ta.mjd = s->mjd.p; ta.P = s->tp;
      ta.heap_w = s->heap_w.p; ta.heap_id = s->heap_id.p;
      ta.heap_stride = (unsigned long long)((s->chunk + 63) / 64 * 64);
      ta.n_triplets = s->n_triplets[tb].p + a0; ta.triplets = s->triplets[tb].p + a0 * s->k_cap;
      // the slice's long trajectories: one wave each; the lane kernel skips them (and is not launched for a slice
      // made of long trajectories only)
      ta.wave_min_obs = s->wave_min_obs;
      const auto w0 = std::lower_bound(s->wave_list_host.begin(), s->wave_list_host.end(), (unsigned)(t0 + a0));
      const auto w1 = std::lower_bound(w0, s->wave_list_host.end(), (unsigned)(t0 + a1));
      ta.wave_list = s->wave_list.p ? s->wave_list.p + (w0 - s->wave_list_host.begin()) : nullptr;
      ta.n_wave_list = (unsigned)(w1 - w0);
      {
        Stamp stamp(s, k0, K_TRIPLETS);
        if (ta.n_wave_list < a1 - a0) HIP_TRY(outfit::launch_triplet_select(ta, s->triplet_lds, k0));
        HIP_TRY(outfit::launch_triplet_select_wave(ta, s->wave_lds, k0));
      }
      HIP_TRY(hipEventRecord(s->ev_sub[i], k0));
    }
    if (streamed && t0 == 0) {
      g_trace.mark("K0 enqueued");
      // K0 is running on mjd alone; now ra, dec and the observer vectors go up, then the direction cosines (same
      // stream as K0, behind its slices) — the first K2a launch below waits for them through ev_unit
      TRY_RC(flush_group(s, 1));
      TRY_RC(launch_unit());
      HIP_TRY(hipEventRecord(s->ev_unit, k0));
      HIP_TRY(hipStreamWaitEvent(st, s->ev_unit, 0));
    }
    HIP_TRY(outfit::reset_slow_list(w, st));
    for (unsigned i = 0; i < n_sub; ++i) {
      const uint64_t a0 = nt * i / n_sub, a1 = nt * (i + 1) / n_sub;
      HIP_TRY(hipStreamWaitEvent(st, s->ev_sub[i], 0));
      outfit::RootsArgs ra{ci, w, s->gp, a0 * s->cand_per_traj, (a1 - a0) * s->cand_per_traj};
      Stamp stamp(s, st, K_ROOTS);
      HIP_TRY(outfit::launch_gauss_roots(ra, st));
    }
    const outfit::CorrectArgs ca{ci, w, s->gp, s->queue.p};
    {
      Stamp stamp(s, st, K_CORRECT_FAST);
      HIP_TRY(outfit::launch_gauss_correct_fast(ca, s->n_cu, st));
    }
    HIP_TRY(hipEventRecord(s->ev_fast, st));
    if (streamed && t0 == 0) {
      g_trace.mark("roots..fast enqueued");
      // roots / cold starts / corrections are enqueued; the uncertainties, geocentric vectors and the Chebyshev table
      // go up under them, then the observer positions (first read by K2c)
      TRY_RC(flush_group(s, 2));
      TRY_RC(launch_observer());
      s->deferred.clear();
    }
    static const bool general_serial = std::getenv("OUTFIT_GENERAL_SERIAL") != nullptr;  // measurement aid: no overlap with K2c
    hipStream_t gs = general_serial ? st : aux;
    HIP_TRY(hipStreamWaitEvent(gs, s->ev_fast, 0));
    {
      outfit::CorrectArgs cg = ca;
      cg.queue = s->queue.p + 1;  // its own queue head
      Stamp stamp(s, gs, K_CORRECT);
      HIP_TRY(outfit::launch_gauss_correct(cg, s->n_cu, gs));
    }
    HIP_TRY(hipEventRecord(s->ev_general, gs));
    static const bool exact_all_env = std::getenv("OUTFIT_RMS_EXACT_ALL") != nullptr;  // verification aid (kernels.hip, K3b)
    const bool exact_all = exact_all_env && s->math_mode == OUTFIT_MATH_GLIBC;
    static const bool no_screen_env = std::getenv("OUTFIT_RMS_NO_SCREEN") != nullptr;  // measurement aid: exact platform-function K2c + K3b
    // K2c: 0 exact with the platform functions | 1 exact with the glibc-faithful ones | 2 screen. `refine`: K3b follows
    // and re-evaluates the contenders exactly — glibc-faithful mode: glibc's functions (always, unless K2c already did);
    // correctly-rounded mode: the platform functions, i.e. the bits of its exact K2c, after a screening K2c.
    const bool glibc_mode = s->math_mode == OUTFIT_MATH_GLIBC;
    const int rms_mode = exact_all ? 1 : (no_screen_env || (!glibc_mode && exact_all_env)) ? 0 : 2;
    const bool refine = glibc_mode ? !exact_all : rms_mode == 2;
    // the screen's direct state -> elements route pays on short trajectories (the conversion is a fifth of a 20-observation
    // candidate's evaluation) and costs on long ones (kernels.hip rms_candidate): batches whose LONGEST trajectory has up
    // to 64 observations (one long trajectory's declined candidates would be the slow-list pass's chain).
    // OUTFIT_SCREEN_DIRECT = 0 / 1 forces either (measurement / verification aid).
    static const char* direct_env = std::getenv("OUTFIT_SCREEN_DIRECT");
    const int screen_direct = direct_env ? (std::atoi(direct_env) != 0) : (s->max_traj_len <= 64);
    const outfit::RmsArgs rm{ci, w, obs_pos, s->cos_dec.p, s->obs_rec.p, s->params.extf, s->params.dtmax, outfit::sincos_consts(),
                             screen_direct, rms_mode};
    HIP_TRY(hipStreamWaitEvent(st, s->ev_prepass, 0));
    {
      Stamp stamp(s, st, K_RMS);
      HIP_TRY(outfit::launch_orbit_rms(rm, /*slow_list=*/false, st));
    }
    HIP_TRY(hipStreamWaitEvent(st, s->ev_general, 0));
    {
      Stamp stamp(s, st, K_RMS_SLOW);
      HIP_TRY(outfit::launch_orbit_rms(rm, /*slow_list=*/true, st));
    }
    outfit::SelectArgs sa{};
    sa.n_traj = nt; sa.traj0 = t0; sa.n_cand = n_cand; sa.cand_per_traj = s->cand_per_traj;
    sa.n_real = s->n_real; sa.n_triplets = s->n_triplets[tb].p; sa.W = w;
    sa.out = d_out ? d_out : s->results.p;
    sa.math_mode = s->math_mode;
    sa.bound = refine ? s->sel_bound.p : nullptr;
    sa.list = s->sel_list.p;
    sa.traj_offset = ci.traj_offset; sa.sra = ci.sra; sa.sdec = ci.sdec;
    {
      Stamp stamp(s, st, K_SELECT);
      HIP_TRY(outfit::launch_select_best(sa, st));
    }
    if (refine) {  // K3b: the contenders, exactly
      Stamp stamp(s, st, K_REFINE);
      HIP_TRY(outfit::launch_refine_best(sa, rm, st));
    }
    HIP_TRY(hipEventRecord(s->ev_k3[tb], st));
    s->k3_recorded[tb] = true;
  }
  // unit[ub] is read by K2a and, through load_geometry, by both correction kernels (the general one on aux, joined
  // into `st` before the slow-list RMS pass): only now may the direction-cosine prepass of run N+2 rewrite it
  HIP_TRY(hipEventRecord(s->ev_unit_free[ub], st));
  s->unit_free_recorded[ub] = true;
  HIP_TRY(hipEventRecord(s->ev_done, st));
  s->done_recorded = true;
  return OUTFIT_OK;
}

const char* validate_diffcor(const outfit_diffcor_config_t* c) {
  if (c->propagator != OUTFIT_PROPAGATOR_TWOBODY)
    return "propagator: only PropagatorKind::TwoBody is implemented";
  return nullptr;
}

outfit::DiffCorConfig make_diffcor_config(const outfit_diffcor_config_t* c) {
  outfit::DiffCorConfig d{};
  d.max_newton_iterations = saturate_u32(c->max_newton_iterations);
  // `0..=passes` (diff_cor.rs:316): keep passes + 1 representable
  d.max_outlier_rejection_passes = std::min<unsigned>(saturate_u32(c->max_outlier_rejection_passes), 0xFFFFFFFEu);
  d.max_stagnation_iterations = saturate_u32(c->max_stagnation_iterations);
  d.enable_outlier_rejection = c->enable_outlier_rejection ? 1u : 0u;
  d.convergence_threshold = c->convergence_threshold;
  d.convergence_before_rejection_threshold = c->convergence_before_rejection_threshold;
  d.rms_stagnation_ratio = c->rms_stagnation_ratio;
  d.rms_divergence_ratio = c->rms_divergence_ratio;
  d.chi2_reject = c->chi_squared_rejection_threshold;
  d.chi2_recover = c->chi_squared_recovery_threshold;
  d.eccentricity_limit = c->eccentricity_limit;
  d.min_a = c->min_semi_major_axis; d.max_a = c->max_semi_major_axis;
  d.min_peri = c->min_periapsis_distance; d.max_apo = c->max_apoapsis_distance;
  unsigned nf = 0;
  for (int j = 0; j < 6; ++j) { d.free_el[j] = c->free_elements[j] ? 1 : 0; nf += d.free_el[j]; }
  d.n_free = (unsigned char)nf;
  return d;
}

int session_run_lsq(outfit_hip_session* s, const outfit_diffcor_config_t* cfg,
                    const outfit_gauss_result_t* initial, hipStream_t st) {
  HIP_TRY(hipSetDevice(s->device));
  if (const char* m = validate_diffcor(cfg)) return fail(OUTFIT_ERR_INVALID_PARAM, m);
  s->last_stream = st;
  s->lsq_ran = true;
  if (s->n_traj == 0) return OUTFIT_OK;
  const size_t n_obs = (size_t)s->n_obs, n_traj = (size_t)s->n_traj;
  if (!s->dc_out.p) {
    HIP_TRY(s->dc_res.alloc(6 * n_obs));
    HIP_TRY(s->dc_sel.alloc(n_obs));
    HIP_TRY(s->dc_cov.alloc(72 * n_traj));
    HIP_TRY(s->dc_qr.alloc(42 * n_traj * outfit::kDiffcorMaxGroup));
    HIP_TRY(s->dc_fit.alloc(n_obs));
    HIP_TRY(s->dc_initial.alloc(n_traj));
    HIP_TRY(s->dc_out.alloc(n_traj));
  }
  const bool on_device_earth = s->has_ephem && s->has_geo;
  if (!s->prepass_done) {
    outfit::PrepassArgs pa{};
    pa.n_obs = s->n_obs; pa.mjd = s->mjd.p; pa.dec = s->dec.p; pa.geo_ecl = s->geo.p; pa.obs_pos = s->obs_pos.p;
    pa.cos_dec = s->cos_dec.p; pa.unit = s->unit[0].p; pa.ra = s->ra.p; pa.has_table = on_device_earth;
    pa.obs_rec = s->obs_rec.p; pa.sra = s->sra.p; pa.sdec = s->sdec.p; pa.helio = s->helio.p;
    pa.parts = outfit::kPrepassObserver;  // the corrector reads obs_pos only
    pa.T = s->table; pa.error_flag = s->error_flag.p; pa.math_mode = s->math_mode;
    Stamp stamp(s, st, K_PREPASS);
    HIP_TRY(outfit::launch_observer_prepass(pa, st));
    s->prepass_done = true;
  }
  if (initial) {
    // The caller may free `initial` as soon as this call returns and an earlier run on `st` may still
    // read dc_initial: drain the stream, then copy synchronously.
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(s->dc_initial.p, initial, n_traj * sizeof(outfit_gauss_result_t), hipMemcpyHostToDevice));
  }
  outfit::DiffCorArgs a{};
  a.n_traj = s->n_traj; a.n_obs = s->n_obs; a.traj_offset = s->traj_offset.p;
  a.mjd = s->mjd.p; a.ra = s->ra.p; a.dec = s->dec.p; a.sra = s->sra.p; a.sdec = s->sdec.p;
  a.obs_pos = on_device_earth ? s->obs_pos.p : s->helio.p;
  a.iod = s->iod_out ? s->iod_out : s->results.p;
  a.initial = initial ? s->dc_initial.p : nullptr;
  a.cfg = make_diffcor_config(cfg);
  a.res = s->dc_res.p; a.sel = s->dc_sel.p; a.cov = s->dc_cov.p; a.qr_work = s->dc_qr.p;
  a.out = s->dc_out.p; a.obs_fit = s->dc_fit.p; a.queue = s->queue.p + 2;
  a.math_mode = s->math_mode;
  {
    Stamp stamp(s, st, K_DIFFCOR);
    // lanes per trajectory: enough to split the observation loop, not more than the mean arc length pays for
    // (measured at 20 observations per trajectory, work-queue kernel, correctly-rounded functions:
    //  1 -> 2.87 ms, 2 -> 1.97, 4 -> 1.82, 8 -> 2.14 per 100k objects — one lane per trajectory has the best
    //  throughput and the longest tail: the last 11-iteration trajectory alone then takes 11 x 20 observations)
    const double mean_obs = (double)s->n_obs / (double)s->n_traj;
    unsigned group = mean_obs >= 12.0 ? 4u : (mean_obs >= 6.0 ? 2u : 1u);
    if (const char* e = std::getenv("OUTFIT_DIFFCOR_GROUP")) {  // tuning aid
      const unsigned g = (unsigned)std::atoi(e);
      if (g == 1 || g == 2 || g == 4 || g == 8) group = g;
    }
    HIP_TRY(outfit::launch_diffcor(a, group, s->n_cu, st));
  }
  HIP_TRY(hipEventRecord(s->ev_done, st));
  s->done_recorded = true;
  return OUTFIT_OK;
}

// Contiguous trajectory ranges balanced by observation count, one host thread per device, no exchange
// step (all tasks are independent: obs_dataset_api.rs:277-296). `work(device, sub_batch, t0)` runs one shard.
template <typename Work>
int run_sharded(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch, int n_devices, Work work) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  // Testing aid: OUTFIT_SHARD_DEVICES="0,0" runs that many shards, shard k on the listed device — the per-device
  // host-thread path (one thread, one session, one slice of `out` per shard) on a single-GPU box.
  std::vector<int> dev_map;
  if (const char* e = std::getenv("OUTFIT_SHARD_DEVICES")) {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      if (v >= 0 && v < n_dev) dev_map.push_back((int)v);
      p = (*end == ',') ? end + 1 : end;
    }
  }
  if (!dev_map.empty()) n_devices = (int)dev_map.size();
  else if (n_devices <= 0 || n_devices > n_dev) n_devices = n_dev;
  if (batch->n_traj == 0) return OUTFIT_OK;
  n_devices = (int)std::min<uint64_t>((uint64_t)n_devices, batch->n_traj);
  const uint64_t o0 = batch->traj_offset[0], total = batch->traj_offset[batch->n_traj] - o0;
  std::vector<uint64_t> cut(n_devices + 1, batch->n_traj);
  cut[0] = 0;
  for (int d = 1; d < n_devices; ++d) {
    const uint64_t target = o0 + total * (uint64_t)d / (uint64_t)n_devices;
    const uint64_t* lo = std::lower_bound(batch->traj_offset, batch->traj_offset + batch->n_traj, target);
    cut[d] = std::max<uint64_t>(cut[d - 1], (uint64_t)(lo - batch->traj_offset));
  }
  std::vector<int> rc(n_devices, OUTFIT_OK);
  std::vector<std::string> msg(n_devices);
  auto shard = [&](int d) {
    const uint64_t t0 = cut[d], t1 = cut[d + 1];
    if (t1 <= t0) return;
    outfit_obs_batch_t sub = *batch;
    sub.n_traj = t1 - t0;
    sub.traj_offset = batch->traj_offset + t0;
    if (batch->noise_z)
      sub.noise_z = batch->noise_z + (size_t)t0 * params->max_triplets * (size_t)params->n_noise_realizations * 6;
    if (batch->traj_key) sub.traj_key = batch->traj_key + t0;
    else sub.traj_key_base = batch->traj_key_base + t0;
    // a worker thread is not behind the entry point's function-try-block: an exception that left this function would
    // end in std::terminate, so it becomes the shard's return code here
    try {
      rc[d] = work(dev_map.empty() ? d : dev_map[d], sub, t0);
      if (rc[d] != OUTFIT_OK) msg[d] = g_last_error;  // thread-local: keep the shard's message
    } catch (...) {
      rc[d] = code_of_current_exception(&msg[d]);
    }
  };
  if (n_devices == 1) {
    shard(0);
  } else {
    // threads that cannot be created (std::system_error) or stored: the shard runs on the calling thread instead, and
    // every thread that did start is joined before anything can unwind past the vector
    std::vector<std::thread> th;
    try { th.reserve(n_devices); } catch (...) {}
    for (int d = 0; d < n_devices; ++d) {
      bool started = false;
      if (th.size() < th.capacity()) {
        try {
          th.emplace_back(shard, d);
          started = true;
        } catch (...) {}
      }
      if (!started) shard(d);
    }
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < n_devices; ++d)
    if (rc[d] != OUTFIT_OK) return fail(rc[d], msg[d]);
  return OUTFIT_OK;
}

int collect_times(outfit_hip_session* s) {
  for (int k = 0; k < K_COUNT; ++k) { s->kernel_ms[k] = 0.f; s->kernel_launches[k] = 0; }
  for (auto& t : s->timed) {
    HIP_TRY(hipEventSynchronize(t.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
    s->kernel_ms[t.id] += ms;
    s->kernel_launches[t.id] += 1;
  }
  return OUTFIT_OK;
}

}  // namespace

extern "C" {

void outfit_hip_iod_params_default(outfit_iod_params_t* p) {
  std::memset(p, 0, sizeof *p);
  p->n_noise_realizations = 20; p->noise_scale = 1.0; p->extf = -1.0; p->dtmax = 30.0;
  p->dt_min = 0.03; p->dt_max_triplet = 150.0; p->optimal_interval_time = 20.0;
  p->max_obs_for_triplets = 100; p->max_triplets = 10; p->gap_max = 8.0 / 24.0;
  p->max_ecc = 5.0; p->max_perihelion_au = 1.0e3; p->min_rho2_au = 0.01;
  p->aberth_max_iter = 50; p->aberth_eps = 1.0e-6;
  p->kepler_eps = 1e3 * std::numeric_limits<double>::epsilon();
  p->max_tested_solutions = 3; p->r2_min_au = 0.05; p->r2_max_au = 200.0;
  p->newton_eps = 1.0e-10; p->newton_max_it = 50; p->root_imag_eps = 1.0e-6;
}

int outfit_hip_iod_params_validate(const outfit_iod_params_t* p) try {
  if (!p) return fail(OUTFIT_ERR_INVALID_PARAM, "params is NULL");
  if (const char* m = validate_params(p)) return fail(OUTFIT_ERR_INVALID_PARAM, m);
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

const char* outfit_hip_last_error(void) { return g_last_error.c_str(); }

int outfit_hip_set_math_mode(int mode) try {
  if (mode != OUTFIT_MATH_CORRECTLY_ROUNDED && mode != OUTFIT_MATH_GLIBC)
    return fail(OUTFIT_ERR_INVALID_PARAM, "math mode must be OUTFIT_MATH_CORRECTLY_ROUNDED or OUTFIT_MATH_GLIBC");
  g_math_mode.store(mode);
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH
int outfit_hip_get_math_mode(void) { return current_math_mode(); }
int outfit_hip_abi_version(void) { return OUTFIT_HIP_ABI_VERSION; }

int outfit_hip_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} OUTFIT_ABI_CATCH

}  // extern "C"
namespace {
int session_create(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch, const outfit_ephem_t* ephem,
                   int device, bool streaming, int math_mode, outfit_hip_session_t** out);
}
extern "C" {
void outfit_hip_call_options_default(outfit_hip_call_options_t* o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->struct_size = (uint32_t)sizeof *o;
  o->math_mode = OUTFIT_MATH_DEFAULT;
}
int outfit_hip_session_create_ex(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch,
                                 const outfit_ephem_t* ephem, int device, const outfit_hip_call_options_t* options,
                                 outfit_hip_session_t** out) try {
  CallOpts opts;
  TRY_RC(resolve_options(options, &opts));
  if (opts.n_obs_capacity || opts.noise_z_capacity) {   // (session_create repeats the other checks)
    TRY_RC(outfit_hip_iod_params_validate(params));
    TRY_RC(check_batch(params, batch, ephem, opts));
  }
  return session_create(params, batch, ephem, device, /*streaming=*/false, opts.mode, out);
} OUTFIT_ABI_CATCH
int outfit_hip_session_create(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch,
                              const outfit_ephem_t* ephem, int device, outfit_hip_session_t** out) try {
  return outfit_hip_session_create_ex(params, batch, ephem, device, nullptr, out);
} OUTFIT_ABI_CATCH
}  // extern "C"
namespace {
int session_create(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch, const outfit_ephem_t* ephem,
                   int device, bool streaming, int math_mode, outfit_hip_session_t** out) {
  if (!out) return fail(OUTFIT_ERR_BAD_HANDLE, "out_session is NULL");
  *out = nullptr;
  TRY_RC(outfit_hip_iod_params_validate(params));
  TRY_RC(check_batch(params, batch, ephem));
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(OUTFIT_ERR_NO_DEVICE, "device index out of range");
  // owned by a holder until it is handed out: an exception inside session_build (its vectors, the wave list) gives
  // the cached device blocks, streams and events back before the caller is told OUTFIT_ERR_NO_MEMORY
  struct Release {
    void operator()(outfit_hip_session* p) const {
      p->free_all();
      delete p;
    }
  };
  std::unique_ptr<outfit_hip_session, Release> s(new outfit_hip_session());
  s->device = device;
  s->streaming = streaming;
  s->math_mode = math_mode;
  const int rc = session_build(s.get(), params, batch, ephem);
  if (rc != OUTFIT_OK) return rc;
  *out = s.release();
  return OUTFIT_OK;
}
}  // namespace
extern "C" {

int outfit_hip_session_run(outfit_hip_session_t* s, void* stream, void* d_out) try {
  if (!s) return fail(OUTFIT_ERR_BAD_HANDLE, "session is NULL");
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : s->own_stream;
  return session_run(s, st, reinterpret_cast<outfit::GaussResultRec*>(d_out));
} OUTFIT_ABI_CATCH

int outfit_hip_session_results(outfit_hip_session_t* s, outfit_gauss_result_t* out) try {
  if (!s || !out) return fail(OUTFIT_ERR_BAD_HANDLE, "session or out is NULL");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(wait_last_run(s));
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, s->error_flag.p, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) return fail(OUTFIT_ERR_EPHEM_RANGE, "an observation epoch lies outside the ephemeris table");
  if (s->n_traj)
    HIP_TRY(hipMemcpy(out, s->results.p, (size_t)s->n_traj * sizeof(outfit_gauss_result_t),
                      hipMemcpyDeviceToHost));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_trim_cache(void) try {
  g_cache.trim();
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_session_check(outfit_hip_session_t* s) try {
  if (!s) return fail(OUTFIT_ERR_BAD_HANDLE, "session is NULL");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(wait_last_run(s));
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, s->error_flag.p, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) return fail(OUTFIT_ERR_EPHEM_RANGE, "an observation epoch lies outside the ephemeris table");
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

void* outfit_hip_session_device_results(outfit_hip_session_t* s) { return s ? s->results.p : nullptr; }

int outfit_hip_session_candidate_scores(outfit_hip_session_t* s, uint64_t capacity, double* sum, uint32_t* n_win,
                                        uint32_t* flags, uint64_t* n_out) try {
  if (!s || !n_out) return fail(OUTFIT_ERR_BAD_HANDLE, "session or n_out is NULL");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(wait_last_run(s));
  const uint64_t last_chunk = s->n_traj ? (s->n_traj - 1) % s->chunk + 1 : 0;
  const uint64_t n = last_chunk * s->cand_per_traj;
  *n_out = n;
  if (n > capacity) return fail(OUTFIT_ERR_INVALID_BATCH, "candidate_scores: capacity is smaller than the last chunk");
  if (n && sum) HIP_TRY(hipMemcpy(sum, s->sum.p, n * sizeof(double), hipMemcpyDeviceToHost));
  if (n && n_win) HIP_TRY(hipMemcpy(n_win, s->n_win.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (n && flags) HIP_TRY(hipMemcpy(flags, s->flags.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_session_set_profiling(outfit_hip_session_t* s, int enabled) try {
  if (!s) return fail(OUTFIT_ERR_BAD_HANDLE, "session is NULL");
  s->profiling = enabled != 0;
  if (s->profiling) {
    for (auto& t : s->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    s->timed.clear();
  }
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_session_kernel_times(outfit_hip_session_t* s, int max_n, const char** names, float* ms,
                                    int* launches) try {
  if (!s) return fail(OUTFIT_ERR_BAD_HANDLE, "session is NULL");
  if (collect_times(s) != OUTFIT_OK) return OUTFIT_ERR_HIP;
  int n = 0;
  for (int k = 0; k < K_COUNT && n < max_n; ++k) {
    if (s->kernel_launches[k] == 0) continue;
    if (names) names[n] = kKernelNames[k];
    if (ms) ms[n] = s->kernel_ms[k];
    if (launches) launches[n] = s->kernel_launches[k];
    ++n;
  }
  return n;
} OUTFIT_ABI_CATCH

void outfit_hip_session_destroy(outfit_hip_session_t* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  s->free_all();
  delete s;
}

int outfit_hip_fit_full_iod(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch,
                            const outfit_ephem_t* ephem, int n_devices, outfit_gauss_result_t* out) try {
  return outfit_hip_fit_full_iod_ex(params, batch, ephem, n_devices, nullptr, out);
} OUTFIT_ABI_CATCH

int outfit_hip_fit_full_iod_ex(const outfit_iod_params_t* params, const outfit_obs_batch_t* batch,
                               const outfit_ephem_t* ephem, int n_devices, const outfit_hip_call_options_t* options,
                               outfit_gauss_result_t* out) try {
  TRY_RC(outfit_hip_iod_params_validate(params));
  CallOpts opts;
  TRY_RC(resolve_options(options, &opts));  // once per call: every shard runs in the same mode
  TRY_RC(check_batch(params, batch, ephem, opts));
  if (!out && batch->n_traj) return fail(OUTFIT_ERR_INVALID_BATCH, "out is NULL");
  const int mode = opts.mode;
  static const bool streamed_upload = std::getenv("OUTFIT_NO_STREAMED_UPLOAD") == nullptr;
  return run_sharded(params, batch, n_devices, [&](int d, const outfit_obs_batch_t& sub, uint64_t t0) {
    outfit_hip_session_t* s = nullptr;
    g_trace.mark("(start)");
    int r = session_create(params, &sub, ephem, d, /*streaming=*/streamed_upload, mode, &s);
    g_trace.mark("session_create");
    if (r == OUTFIT_OK) r = outfit_hip_session_run(s, nullptr, nullptr);
    g_trace.mark("session_run (enqueue)");
    if (r == OUTFIT_OK) r = outfit_hip_session_results(s, out + t0);
    g_trace.mark("session_results");
    const std::string keep = g_last_error;
    outfit_hip_session_destroy(s);
    g_trace.mark("session_destroy");
    g_last_error = keep;
    return r;
  });
} OUTFIT_ABI_CATCH

void outfit_hip_diffcor_config_default(outfit_diffcor_config_t* c) {
  std::memset(c, 0, sizeof *c);
  c->max_newton_iterations = 30; c->max_outlier_rejection_passes = 10;
  c->convergence_threshold = 1e-4; c->convergence_before_rejection_threshold = 2.0;
  c->rms_stagnation_ratio = 0.98; c->rms_divergence_ratio = 1.5; c->max_stagnation_iterations = 3;
  c->enable_outlier_rejection = 1; c->propagator = OUTFIT_PROPAGATOR_TWOBODY;
  c->chi_squared_rejection_threshold = 25.0; c->chi_squared_recovery_threshold = 9.0;
  c->eccentricity_limit = 1.2; c->min_semi_major_axis = 1e-6; c->max_semi_major_axis = 1e4;
  c->min_periapsis_distance = 1e-6; c->max_apoapsis_distance = 1e4;
  for (int j = 0; j < 6; ++j) c->free_elements[j] = 1;
}

int outfit_hip_diffcor_config_validate(const outfit_diffcor_config_t* c) try {
  if (!c) return fail(OUTFIT_ERR_INVALID_PARAM, "config is NULL");
  if (const char* m = validate_diffcor(c)) return fail(OUTFIT_ERR_INVALID_PARAM, m);
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_session_run_lsq(outfit_hip_session_t* s, const outfit_diffcor_config_t* config,
                               const outfit_gauss_result_t* initial_orbits, void* stream) try {
  if (!s) return fail(OUTFIT_ERR_BAD_HANDLE, "session is NULL");
  if (!config) return fail(OUTFIT_ERR_INVALID_PARAM, "config is NULL");
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : s->own_stream;
  return session_run_lsq(s, config, initial_orbits, st);
} OUTFIT_ABI_CATCH

int outfit_hip_session_lsq_results(outfit_hip_session_t* s, outfit_lsq_result_t* out, outfit_obs_fit_t* obs_fit) try {
  if (!s || !out) return fail(OUTFIT_ERR_BAD_HANDLE, "session or out is NULL");
  if (!s->lsq_ran) return fail(OUTFIT_ERR_BAD_HANDLE, "outfit_hip_session_run_lsq has not been called");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(wait_last_run(s));
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, s->error_flag.p, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) return fail(OUTFIT_ERR_EPHEM_RANGE, "an observation epoch lies outside the ephemeris table");
  if (s->n_traj)
    HIP_TRY(hipMemcpy(out, s->dc_out.p, (size_t)s->n_traj * sizeof(outfit_lsq_result_t), hipMemcpyDeviceToHost));
  if (obs_fit && s->n_obs) {
    if (s->perm.empty()) {
      HIP_TRY(hipMemcpy(obs_fit, s->dc_fit.p, (size_t)s->n_obs * sizeof(outfit_obs_fit_t), hipMemcpyDeviceToHost));
    } else {  // back to the caller's row order
      std::vector<outfit_obs_fit_t> tmp((size_t)s->n_obs);
      HIP_TRY(hipMemcpy(tmp.data(), s->dc_fit.p, tmp.size() * sizeof(outfit_obs_fit_t), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < tmp.size(); ++i) obs_fit[s->perm[i]] = tmp[i];
    }
  }
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_fit_lsq(const outfit_iod_params_t* params, const outfit_diffcor_config_t* config,
                       const outfit_obs_batch_t* batch, const outfit_ephem_t* ephem,
                       const outfit_gauss_result_t* initial, int n_devices, outfit_lsq_result_t* out,
                       outfit_obs_fit_t* obs_fit) try {
  return outfit_hip_fit_lsq_ex(params, config, batch, ephem, initial, n_devices, nullptr, out, obs_fit);
} OUTFIT_ABI_CATCH

int outfit_hip_fit_lsq_ex(const outfit_iod_params_t* params, const outfit_diffcor_config_t* config,
                          const outfit_obs_batch_t* batch, const outfit_ephem_t* ephem,
                          const outfit_gauss_result_t* initial, int n_devices, const outfit_hip_call_options_t* options,
                          outfit_lsq_result_t* out, outfit_obs_fit_t* obs_fit) try {
  TRY_RC(outfit_hip_iod_params_validate(params));
  TRY_RC(outfit_hip_diffcor_config_validate(config));
  CallOpts opts;
  TRY_RC(resolve_options(options, &opts));
  TRY_RC(check_batch(params, batch, ephem, opts));
  if (!out && batch->n_traj) return fail(OUTFIT_ERR_INVALID_BATCH, "out is NULL");
  const int mode = opts.mode;
  return run_sharded(params, batch, n_devices, [&](int d, const outfit_obs_batch_t& sub, uint64_t t0) {
    // the IOD runs only when some trajectory has no usable seed (obs_dataset_api.rs:118-120, mod.rs:86)
    bool need_iod = initial == nullptr;
    for (uint64_t t = t0; t < t0 + sub.n_traj && !need_iod; ++t) need_iod = initial[t].status != OUTFIT_TRAJ_OK;
    outfit_hip_session_t* s = nullptr;
    int r = session_create(params, &sub, ephem, d, /*streaming=*/false, mode, &s);
    if (r == OUTFIT_OK && need_iod) r = outfit_hip_session_run(s, nullptr, nullptr);
    if (r == OUTFIT_OK) r = outfit_hip_session_run_lsq(s, config, initial ? initial + t0 : nullptr, nullptr);
    if (r == OUTFIT_OK)
      r = outfit_hip_session_lsq_results(s, out + t0, obs_fit ? obs_fit + batch->traj_offset[t0] : nullptr);
    const std::string keep = g_last_error;
    outfit_hip_session_destroy(s);
    g_last_error = keep;
    return r;
  });
} OUTFIT_ABI_CATCH

int outfit_hip_observer_positions(const outfit_ephem_t* eph, uint64_t n, const double* mjd_tt, const double* mjd_ut1,
                                  const double* body_fixed, double* geo_ecl, double* helio_equ) try {
  if (n == 0) return OUTFIT_OK;
  if (!eph || !mjd_tt || !mjd_ut1 || !body_fixed || !geo_ecl || !helio_equ)
    return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  if (!eph->coef || eph->n_blocks == 0 || !(eph->block_days > 0.0))
    return fail(OUTFIT_ERR_INVALID_BATCH, "ephemeris table is empty");
  for (int k = 0; k < 3; ++k)
    if (eph->ncoef[k] < 2 || eph->ncoef[k] > 32 || eph->nsub[k] < 1)
      return fail(OUTFIT_ERR_INVALID_BATCH, "ephemeris ncoef must be in [2,32] and nsub >= 1");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<double> tt, ut1, fx, geo, hel, coef;
  DevBuf<int> flag;
  outfit::GeometryArgs ga{};
  unsigned per_block = 0;
  for (int k = 0; k < 3; ++k) {
    ga.T.ncoef[k] = eph->ncoef[k]; ga.T.nsub[k] = eph->nsub[k]; ga.T.off[k] = per_block;
    per_block += 3u * eph->ncoef[k] * eph->nsub[k];
  }
  ga.T.per_block = per_block; ga.T.jd_start = eph->jd_start; ga.T.jd_end = eph->jd_end;
  ga.T.block_days = eph->block_days; ga.T.emrat = eph->emrat; ga.T.n_blocks = eph->n_blocks;
  auto cleanup = [&]() { tt.release(); ut1.release(); fx.release(); geo.release(); hel.release(); coef.release(); flag.release(); };
  int rc = upload(tt, mjd_tt, (size_t)n);
  if (rc == OUTFIT_OK) rc = upload(ut1, mjd_ut1, (size_t)n);
  if (rc == OUTFIT_OK) rc = upload(fx, body_fixed, (size_t)(3 * n));
  if (rc == OUTFIT_OK) rc = upload(coef, eph->coef, (size_t)eph->n_blocks * per_block);
  hipError_t e = hipSuccess;
  if (rc == OUTFIT_OK) {
    e = geo.alloc((size_t)(3 * n));
    if (e == hipSuccess) e = hel.alloc((size_t)(3 * n));
    if (e == hipSuccess) e = flag.alloc(1);
    if (e == hipSuccess) e = hipMemset(flag.p, 0, sizeof(int));
    ga.n_obs = n; ga.mjd_tt = tt.p; ga.mjd_ut1 = ut1.p; ga.body_fixed = fx.p; ga.T.coef = coef.p;
    ga.geo_ecl = geo.p; ga.helio_equ = hel.p; ga.error_flag = flag.p;
    if (e == hipSuccess) e = outfit::launch_observer_geometry(ga, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    int f = 0;
    if (e == hipSuccess) e = hipMemcpy(&f, flag.p, sizeof f, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(geo_ecl, geo.p, (size_t)(3 * n) * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(helio_equ, hel.p, (size_t)(3 * n) * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && f) { cleanup(); return fail(OUTFIT_ERR_EPHEM_RANGE, "an observation epoch lies outside the ephemeris table"); }
  }
  cleanup();
  if (rc != OUTFIT_OK) return rc;
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_device_normals(uint64_t noise_seed, const uint64_t* traj_key, uint64_t traj_key_base, uint64_t n_traj,
                              uint32_t max_triplets, uint64_t n_noise, double* out) try {
  const uint64_t n = n_traj * max_triplets * n_noise * 6;
  if (n == 0) return OUTFIT_OK;
  if (!out) return fail(OUTFIT_ERR_INVALID_BATCH, "out is NULL");
  if (n_noise >= (1ull << 30)) return fail(OUTFIT_ERR_INVALID_PARAM, "n_noise too large");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<double> d;
  DevBuf<unsigned long long> k;
  HIP_TRY(d.alloc((size_t)n));
  hipError_t e = hipSuccess;
  if (traj_key) {
    e = k.alloc((size_t)n_traj);
    if (e == hipSuccess) e = hipMemcpy(k.p, traj_key, (size_t)n_traj * sizeof(uint64_t), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess)
    e = outfit::launch_device_normals(noise_seed, k.p, traj_key_base, n_traj, max_triplets, (unsigned)n_noise, d.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  d.release();
  k.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_selftest_aberth(uint64_t n_polys, uint64_t seed, uint64_t* counts) try {
  if (!counts) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<unsigned long long> d;
  HIP_TRY(d.alloc(2));
  hipError_t e = hipMemset(d.p, 0, 2 * sizeof(unsigned long long));
  const unsigned long long threads = 1ull << 16, per = (n_polys + threads - 1) / threads;
  if (e == hipSuccess) e = outfit::launch_selftest_aberth(threads, per, seed, d.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost);
  d.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  counts[0] = h[0];
  counts[1] = h[1];
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_selftest_crmath(uint64_t n_args, uint64_t seed, uint64_t* counts) try {
  if (!counts) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<unsigned long long> d;
  HIP_TRY(d.alloc(4));
  hipError_t e = hipMemset(d.p, 0, 4 * sizeof(unsigned long long));
  const unsigned long long threads = 1ull << 18, per = (n_args + threads - 1) / threads;
  if (e == hipSuccess) e = outfit::launch_selftest_crmath(threads, per, seed, d.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long h[4] = {0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost);
  d.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  for (int i = 0; i < 4; ++i) counts[i] = h[i];
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_crmath_eval(int fn, uint64_t n, const double* x, double* out) try {
  if ((fn & 15) > 12 || (fn & ~31)) return fail(OUTFIT_ERR_INVALID_PARAM, "fn: 0 sin, 1 cos, 2 acos, 3 sinh, 4 cosh, 5 log, 6 atan2 (+16: full evaluation), 7 / 8 sin / cos of the residual loop, 9..12 glibc-faithful sin / cos / sinh / cosh");
  if (n && (!x || !out)) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (n == 0) return OUTFIT_OK;
  DevBuf<double> dx, dy;
  TRY_RC(upload(dx, x, (size_t)n * ((fn & 15) == 6 ? 2 : 1)));
  HIP_TRY(dy.alloc((size_t)n));
  hipError_t e = outfit::launch_crmath_eval(fn, n, dx.p, dy.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  dx.release();
  dy.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_glibc_eval(int fn, uint64_t n, const double* x, const double* y, double* out) try {
  if (fn < 0 || fn > 6) return fail(OUTFIT_ERR_INVALID_PARAM, "fn: 0 sin1, 1 cos1, 2 tan, 3 atan2(x, y), 4 hypot(x, y), 5 acos, 6 log");
  const bool two = fn == 3 || fn == 4;
  if (n && (!x || !out || (two && !y))) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (n == 0) return OUTFIT_OK;
  DevBuf<double> dx, dy, dz;
  TRY_RC(upload(dx, x, (size_t)n));
  if (two) TRY_RC(upload(dy, y, (size_t)n));
  HIP_TRY(dz.alloc((size_t)n));
  hipError_t e = outfit::launch_glibc_eval(fn, n, dx.p, dy.p, dz.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dz.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  dx.release();
  dy.release();
  dz.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_calibration_stream(uint64_t n_doubles, int repeats) try {
  if (n_doubles == 0 || repeats <= 0) return OUTFIT_OK;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<double> a, b;
  HIP_TRY(a.alloc((size_t)n_doubles));
  hipError_t e = b.alloc((size_t)n_doubles);
  if (e == hipSuccess) e = hipMemset(a.p, 0, (size_t)n_doubles * sizeof(double));
  for (int r = 0; r < repeats && e == hipSuccess; ++r) e = outfit::launch_calibration_stream(n_doubles, a.p, b.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  a.release();
  b.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_prelim_kepuni(uint64_t n, const double* problems, double* out) try {
  if (n == 0) return OUTFIT_OK;
  if (!problems || !out) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<double> dx, dy;
  TRY_RC(upload(dx, problems, (size_t)n * 6));
  HIP_TRY(dy.alloc((size_t)n));
  hipError_t e = outfit::launch_prelim_kepuni(n, dx.p, current_math_mode(), dy.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  dx.release();
  dy.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_selftest_abi_guard(int kind) try {
  if (kind == 0) throw std::bad_alloc();
  if (kind == 1) throw std::runtime_error("selftest");
  if (kind == 2) throw 42;
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

int outfit_hip_selftest_division(uint64_t n_pairs, uint64_t seed, uint64_t* mismatches) try {
  if (!mismatches) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<unsigned long long> d;
  HIP_TRY(d.alloc(1));
  hipError_t e = hipMemset(d.p, 0, sizeof(unsigned long long));
  const unsigned long long threads = 1ull << 16, per = (n_pairs + threads - 1) / threads;
  if (e == hipSuccess) e = outfit::launch_selftest_division(threads, per, seed, d.p, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpy(&h, d.p, sizeof h, hipMemcpyDeviceToHost);
  d.release();
  if (e != hipSuccess) return fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  *mismatches = h;
  return OUTFIT_OK;
} OUTFIT_ABI_CATCH

void outfit_hip_solver_type_default(outfit_solver_type_t* s) {
  std::memset(s, 0, sizeof *s);
  s->kind = OUTFIT_SOLVER_NEWTON_RAPHSON;
  s->parabolic_solving_method = OUTFIT_PARABOLIC_CARDANO;
  s->convergency = 100.0 * std::numeric_limits<double>::epsilon();
  s->max_iter_prelim_kepuni = 20;
}

int outfit_hip_propagate_universal(uint64_t n, const double* r0v0, const double* t0, const double* t1,
                                   double convergency, double* out, int32_t* status) try {
  outfit_solver_type_t st;
  outfit_hip_solver_type_default(&st);
  st.convergency = convergency;
  return outfit_hip_propagate_universal_ex(n, r0v0, t0, t1, &st, out, status);
} OUTFIT_ABI_CATCH

int outfit_hip_propagate_universal_ex(uint64_t n, const double* r0v0, const double* t0, const double* t1,
                                      const outfit_solver_type_t* solver, double* out, int32_t* status) try {
  if (!solver) return fail(OUTFIT_ERR_INVALID_PARAM, "solver is NULL");
  if (solver->kind < 0 || solver->kind > OUTFIT_SOLVER_AUTO)
    return fail(OUTFIT_ERR_INVALID_PARAM, "solver kind must be OUTFIT_SOLVER_NEWTON_RAPHSON, _BRENT_DECKER or _AUTO");
  if (solver->parabolic_solving_method != OUTFIT_PARABOLIC_CARDANO && solver->parabolic_solving_method != OUTFIT_PARABOLIC_NEWTON)
    return fail(OUTFIT_ERR_INVALID_PARAM, "parabolic_solving_method must be OUTFIT_PARABOLIC_CARDANO or _NEWTON");
  outfit::SolverSpec sv;
  sv.kind = solver->kind;
  sv.parabolic_newton = solver->parabolic_solving_method == OUTFIT_PARABOLIC_NEWTON;
  sv.max_iter_prelim = solver->max_iter_prelim_kepuni > 0x7fffffffull ? 0x7fffffff : (int)solver->max_iter_prelim_kepuni;
  sv.has_guess = solver->has_psi_guess != 0;
  sv.guess = solver->psi_guess;
  sv.convergency = solver->convergency;
  if (n == 0) return OUTFIT_OK;
  if (!r0v0 || !t0 || !t1 || !out || !status) return fail(OUTFIT_ERR_INVALID_BATCH, "NULL argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(OUTFIT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  DevBuf<double> d_rv, d_t0, d_t1, d_out;
  DevBuf<int> d_st;
  auto cleanup = [&]() { d_rv.release(); d_t0.release(); d_t1.release(); d_out.release(); d_st.release(); };
  int rc = OUTFIT_OK;
  do {
    if ((rc = upload(d_rv, r0v0, (size_t)6 * n)) != OUTFIT_OK) break;
    if ((rc = upload(d_t0, t0, (size_t)n)) != OUTFIT_OK) break;
    if ((rc = upload(d_t1, t1, (size_t)n)) != OUTFIT_OK) break;
    if (d_out.alloc((size_t)11 * n) != hipSuccess || d_st.alloc((size_t)n) != hipSuccess) {
      rc = fail(OUTFIT_ERR_HIP, "hipMalloc failed");
      break;
    }
    hipError_t e = outfit::launch_propagate_universal(n, d_rv.p, d_t0.p, d_t1.p, sv, current_math_mode(), d_out.p, d_st.p,
                                                      nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out.p, (size_t)11 * n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(status, d_st.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(OUTFIT_ERR_HIP, hipGetErrorString(e));
  } while (false);
  cleanup();
  return rc;
} OUTFIT_ABI_CATCH

}  // extern "C"
