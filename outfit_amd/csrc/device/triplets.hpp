// triplets.hpp — device-side triplet enumeration and best-K selection, one lane
// per trajectory.
//
// Replaces generate_triplets (reference
// src/initial_orbit_determination/triplet_generation/mod.rs:328-408) with its
// lazy index stream (index_generator.rs:133-271), the down-sampling rule
// (index_generator.rs:66-75) and the weight (mod.rs:229-274).
//
// The reference keeps the K best triplets in a std BinaryHeap (max-heap on the
// weight) and sorts the survivors by weight. Which of several EQUAL weights
// survives depends on the heap's sift order, so the heap is emulated operation
// for operation; the heap lives in a per-lane slice of a global scratch buffer
// ([slot][lane] interleaved so that a wave touches 64 consecutive words).
#pragma once
#include <hip/hip_runtime.h>

namespace outfit {

#ifdef OUTFIT_K0_STATS  // diagnostic build (tools/perf_k0_stats.py): what one trajectory's wave-form selection does
// 0 trajectories, 1 middle groups (64 rows tested at once), 2 rows enumerated, 3 last-chunks scored (ballots),
// 4 chunks with a survivor, 5 survivors absorbed, 6 pushes while filling, 7 replacements (pop + push),
// 8 survivors dropped (top fell since the ballot), 9 sift-down levels, 10 sift-up levels, 11 cycles
__device__ unsigned long long g_k0_stat[16];
#define K0STAT(slot, n) do { if ((threadIdx.x & 63u) == 0) k0_local[slot] += (n); } while (0)
#else
#define K0STAT(slot, n) do { } while (0)
#endif

struct TripletParams {
  double dt_min, dt_max, inv_dtw;
  unsigned long long max_keep;  // IODParams::max_obs_for_triplets
  unsigned k_cap;               // IODParams::max_triplets
};

__device__ __forceinline__ unsigned long long pack_triplet(unsigned i, unsigned j, unsigned k) {
  return ((unsigned long long)i << 42) | ((unsigned long long)j << 21) | (unsigned long long)k;
}
__device__ __forceinline__ void unpack_triplet(unsigned long long p, unsigned& i, unsigned& j, unsigned& k) {
  i = (unsigned)(p >> 42) & 0x1FFFFFu;
  j = (unsigned)(p >> 21) & 0x1FFFFFu;
  k = (unsigned)p & 0x1FFFFFu;
}

// Per-lane views of the heap: slot s of this lane is at [s * stride]. Two storages:
// global scratch with 64-bit ids (any size), and LDS with 32-bit ids (10 bits per index,
// reduced trajectories of <= 1024 observations) for the common small case.
struct HeapView {
  double* w;
  unsigned long long* id;
  unsigned long long stride;
  __device__ __forceinline__ double W(unsigned s) const { return w[s * stride]; }
  __device__ __forceinline__ unsigned long long I(unsigned s) const { return id[s * stride]; }
  __device__ __forceinline__ void set(unsigned s, double ww, unsigned long long ii) const {
    w[s * stride] = ww;
    id[s * stride] = ii;
  }
  __device__ __forceinline__ void move(unsigned dst, unsigned src) const {
    w[dst * stride] = w[src * stride];
    id[dst * stride] = id[src * stride];
  }
  // the stored form of an id, for slot-to-slot copies that do not need to decode it
  typedef unsigned long long Raw;
  __device__ __forceinline__ Raw raw(unsigned s) const { return id[s * stride]; }
  __device__ __forceinline__ Raw encode(unsigned long long ii) const { return ii; }
  __device__ __forceinline__ void set_raw(unsigned s, double ww, Raw r) const {
    w[s * stride] = ww;
    id[s * stride] = r;
  }
  __device__ __forceinline__ void load(unsigned s, double& ww, Raw& r) const { ww = w[s * stride]; r = id[s * stride]; }
};
struct LdsHeapView {
  double* w;
  unsigned* id;
  unsigned stride;
  __device__ __forceinline__ double W(unsigned s) const { return w[s * stride]; }
  __device__ __forceinline__ unsigned long long I(unsigned s) const {
    const unsigned p = id[s * stride];  // 10-bit fields -> the 21-bit packing
    return pack_triplet(p >> 20, (p >> 10) & 0x3FFu, p & 0x3FFu);
  }
  __device__ __forceinline__ void set(unsigned s, double ww, unsigned long long ii) const {
    unsigned i, j, k;
    unpack_triplet(ii, i, j, k);
    w[s * stride] = ww;
    id[s * stride] = (i << 20) | (j << 10) | k;
  }
  __device__ __forceinline__ void move(unsigned dst, unsigned src) const {
    w[dst * stride] = w[src * stride];
    id[dst * stride] = id[src * stride];
  }
  typedef unsigned Raw;
  __device__ __forceinline__ Raw raw(unsigned s) const { return id[s * stride]; }
  __device__ __forceinline__ Raw encode(unsigned long long ii) const {
    unsigned i, j, k;
    unpack_triplet(ii, i, j, k);
    return (i << 20) | (j << 10) | k;
  }
  __device__ __forceinline__ void set_raw(unsigned s, double ww, Raw r) const {
    w[s * stride] = ww;
    id[s * stride] = r;
  }
  __device__ __forceinline__ void load(unsigned s, double& ww, Raw& r) const { ww = w[s * stride]; r = id[s * stride]; }
};

// WeightedTriplet::cmp (mod.rs:135-146): weight first; indices only when a weight is NaN.
// Weights here are finite (non-finite ones are skipped, mod.rs:382-384), so `a <= b` is wa <= wb.
__device__ __forceinline__ bool wt_le(double wa, double wb) { return wa <= wb; }

// std::collections::BinaryHeap::sift_up(start, pos) with the element held out of the array
// Every level reads the weight AND the stored id of the slots it may copy in one batch of independent LDS loads (one
// round trip of latency per level instead of two or three): K0 is a chain of dependent LDS accesses per lane.
template <class View>
__device__ __forceinline__ unsigned heap_sift_up_raw(const View& h, unsigned start, unsigned pos, double ew,
                                                     typename View::Raw er) {
  while (pos > start) {
    const unsigned parent = (pos - 1) / 2;
    double wp;
    typename View::Raw rp;
    h.load(parent, wp, rp);
#ifdef OUTFIT_K0_STATS
    atomicAdd(&g_k0_stat[10], 1ull);
#endif
    if (wt_le(ew, wp)) break;
    h.set_raw(pos, wp, rp);
    pos = parent;
  }
  h.set_raw(pos, ew, er);
  return pos;
}
template <class View>
__device__ __forceinline__ unsigned heap_sift_up(const View& h, unsigned start, unsigned pos, double ew,
                                        unsigned long long ei) {
  return heap_sift_up_raw(h, start, pos, ew, h.encode(ei));
}

// BinaryHeap::pop(): swap the last element into the root, sift_down_to_bottom(0), sift_up
template <class View>
__device__ __forceinline__ void heap_pop(const View& h, unsigned& len) {
  double lw;
  typename View::Raw lr;
  h.load(len - 1, lw, lr);
  --len;
  if (len == 0) return;
  const unsigned end = len;
  unsigned pos = 0, child = 1;
  while (end >= 2 && child <= end - 2) {
#ifdef OUTFIT_K0_STATS
    atomicAdd(&g_k0_stat[9], 1ull);
#endif
    double wl, wr;
    typename View::Raw rl, rr;
    h.load(child, wl, rl);
    h.load(child + 1, wr, rr);
    const bool right = wt_le(wl, wr);
    h.set_raw(pos, right ? wr : wl, right ? rr : rl);
    pos = child + (right ? 1u : 0u);
    child = 2 * pos + 1;
  }
  if (child == end - 1) { h.move(pos, child); pos = child; }
  heap_sift_up_raw(h, 0, pos, lw, lr);
}

// index_generator.rs:66-75 — reduced index -> original index
__device__ __forceinline__ unsigned long long keep_index(unsigned long long i, unsigned long long n,
                                                         unsigned long long max_keep) {
  if (max_keep >= n) return i;
  if (max_keep <= 3) return i == 0 ? 0 : (i == 1 ? n / 2 : n - 1);
  return i * (n - 1) / (max_keep - 1);
}
__host__ __device__ __forceinline__ unsigned long long reduced_len(unsigned long long n, unsigned long long max_keep) {
  if (n == 0) return 0;
  if (max_keep >= n) return n;
  if (max_keep <= 3) return 3;
  return max_keep;
}
// Which form of K0 owns a trajectory: THE predicate of both sides — the host builds the wave kernel's list with it and
// the lane kernels skip exactly those trajectories (`max_keep` is the raw max_obs_for_triplets: 0, 1 and 2 are legal
// and behave like 3, index_generator.rs:66-75). Fewer than 3 observations have no triplet: always the lane form, which
// writes n_triplets = 0 (a 1- or 2-observation trajectory with max_keep below its length has reduced_len 3).
__host__ __device__ __forceinline__ bool wave_form_owns(unsigned long long n_obs, unsigned long long max_keep,
                                                        unsigned wave_min_obs) {
  return wave_min_obs != 0 && n_obs >= 3 && reduced_len(n_obs, max_keep) >= wave_min_obs;
}

// mod.rs:265-274
__device__ __forceinline__ double s_gap(double dt, double inv_dtw) {
  const double r = dt * inv_dtw;
  return (r <= 1.0) ? 1.0 / r : 1.0 + r;
}

// Enumerate, keep the best K, sort by weight (stable on the heap's vector order).
// Returns the number of triplets; slot s of the heap then holds triplet s.
//
// The reference's TripletIndexGenerator (index_generator.rs:228-271) is a state machine that
// emits, for each anchor with a non-empty window [lo, hi], every middle in (anchor, hi) and every
// last in [max(lo, middle+1), hi] in lexicographic order; the nested loops below produce the
// same sequence (the oracle keeps the literal state machine). The first gap's term of the weight
// depends on (anchor, middle) only and is hoisted out of the innermost loop (pure function).
template <class View>
__device__ __forceinline__ unsigned select_triplets(const double* mjd /* this trajectory */, unsigned long long n_obs64,
                                           const TripletParams& P, const View& h) {
  if (P.k_cap == 0 || n_obs64 < 3) return 0;
  const unsigned n_obs = (unsigned)n_obs64;                       // < 2^21 (checked by the host)
  const unsigned n = (unsigned)reduced_len(n_obs64, P.max_keep);
  const bool identity = P.max_keep >= n_obs64;
  auto ep = [&](unsigned i) -> double {
    return mjd[identity ? i : (unsigned)keep_index(i, n_obs, P.max_keep)];
  };
  unsigned len = 0;
  double root_w = 0.0;  // weight at the top of the heap, kept in a register (one LDS round trip less per candidate)
  for (unsigned anchor = 0; anchor + 2 < n; ++anchor) {
    // LastWindow::compute (index_generator.rs:94-114)
    const double t0 = ep(anchor);
    unsigned lo = anchor + 2;
    while (lo < n && ep(lo) - t0 < P.dt_min) ++lo;
    unsigned hi = lo - 1;
    if (hi < anchor + 1) hi = anchor + 1;
    while (hi + 1 < n && ep(hi + 1) - t0 <= P.dt_max) ++hi;
    if (lo >= n || lo > hi || hi <= anchor + 1) continue;  // LastWindow::is_empty
    for (unsigned middle = anchor + 1; middle < hi; ++middle) {
      const double tm = ep(middle);
      const double g12 = s_gap(tm - t0, P.inv_dtw);
      // Exact row skip. s_gap(dt) >= 1 for every dt >= 0 (1/r with r <= 1, else 1 + r; NaN stays NaN), and epochs
      // are sorted, so every weight of this (anchor, middle) row is >= fl(g12 + 1) or not finite. With the heap full
      // its top only decreases, and a candidate with w >= top is not pushed (`w < heap[0].w`, mod.rs:396-403): if
      // fl(g12 + 1) >= top now, no candidate of the row can change the heap — the row is not enumerated. With the
      // reference's defaults (K = 10 of up to C(100,3) = 161 700 triplets) this removes all but the rows whose
      // first gap is near the optimal interval.
      if (len == P.k_cap && !(g12 + 1.0 < root_w)) continue;
      unsigned last = lo > middle + 1 ? lo : middle + 1;
      double t_last = last <= hi ? ep(last) : 0.0;
      for (; last <= hi; ++last) {
        const double tl = t_last;
        if (last < hi) t_last = ep(last + 1);  // next epoch on its way while this candidate is processed
        const double w = g12 + s_gap(tl - tm, P.inv_dtw);
        if (!isfinite(w)) continue;
        const unsigned long long id = pack_triplet(anchor, middle, last);
        if (len < P.k_cap) {
          heap_sift_up(h, 0, len, w, id);
          ++len;
          root_w = h.W(0);
        } else if (w < root_w) {
          heap_pop(h, len);
          heap_sift_up(h, 0, len, w, id);
          ++len;
          root_w = h.W(0);
        }
      }
    }
  }
  // sort_unstable_by(weight) restated as a stable insertion sort (equal weights keep heap order). The insertion
  // point is found by bisection (upper bound: after every element <= x, which is what the linear scan from the right
  // stops at), then the tail is shifted by copies that do not wait for a comparison — the same result as the linear
  // scan, without its ~K/4 dependent LDS round trips per element.
  for (unsigned i = 1; i < len; ++i) {
    const double xw = h.W(i);
    const typename View::Raw xr = h.raw(i);
    unsigned lo = 0, hi = i;            // first position p in [0, i] with W(p) > xw (or i)
    while (lo < hi) {
      const unsigned mid = (lo + hi) / 2;
      if (h.W(mid) > xw) hi = mid; else lo = mid + 1;
    }
    for (unsigned j = i; j > lo; --j) h.move(j, j - 1);
    h.set_raw(lo, xw, xr);
  }
  return len;
}

// ------------------------------------------------------------------------------------------------------------------
// The same selection by ONE WAVE per trajectory, for long trajectories (the reference's default caps the search at
// max_obs_for_triplets = 100 epochs: up to C(100,3) = 161 700 scored triplets, a chain of that length for the
// one-lane form above whatever the batch size).
//
// Exact, because of what the reference's loop (mod.rs:365-408) can and cannot observe:
//   * a candidate whose weight is not below the heap's top is dropped without touching the heap, and with the heap
//     full the top only decreases — so a candidate with w >= top(now) is dropped whenever it comes;
//   * the candidates that DO touch the heap must do so in the generator's order (index_generator.rs:228-271:
//     anchor, then middle, then last), each seeing the top its predecessors left.
// The wave therefore scores 64 consecutive `last` indices of one (anchor, middle) row at once, ballots
// `finite && (heap not full || w < top)`, and only the survivors go through the literal heap operations, in lane
// order = generator order, with the top re-read after each — lane 0 runs the very sift code of the one-lane form on
// a heap in LDS. Rows and whole groups of rows are skipped by the same argument (every weight of a row is
// >= fl(g12 + 1)): 64 middles of an anchor are tested at once. The final stable sort is a rank computation
// (rank = #{w_j < w_i} + #{j < i : w_j == w_i}), which is what a stable insertion sort produces.
// LDS: the reduced epochs [n] and the heap [k_cap] x (weight, 64-bit id); the launcher checks that it fits.
// Returns the number of triplets (wave-uniform) and writes them, sorted, to `out` (global).
struct WaveHeap {   // heap of one trajectory in LDS; same interface as HeapView
  // Round 5: slot s is ONE 16-byte element (weight, 64-bit id) at a 16-byte aligned address: a slot is read or written
  // by a single DS instruction (ds_read_b128 / ds_write_b128) instead of two — the kernel issues one-lane LDS
  // instructions for most of its time (profiles/round05_k0_multi_dropped.log), and half of those were the ids.
  double2* e;
  __device__ __forceinline__ double W(unsigned s) const { return e[s].x; }
  __device__ __forceinline__ unsigned long long I(unsigned s) const { return (unsigned long long)__double_as_longlong(e[s].y); }
  __device__ __forceinline__ void move(unsigned dst, unsigned src) const { e[dst] = e[src]; }
  typedef unsigned long long Raw;
  __device__ __forceinline__ Raw raw(unsigned s) const { return I(s); }
  __device__ __forceinline__ Raw encode(unsigned long long ii) const { return ii; }
  __device__ __forceinline__ void set_raw(unsigned s, double ww, Raw r) const {
    e[s] = double2{ww, __longlong_as_double((long long)r)};
  }
  __device__ __forceinline__ void load(unsigned s, double& ww, Raw& r) const {
    const double2 v = e[s];
    ww = v.x;
    r = (unsigned long long)__double_as_longlong(v.y);
  }
};

// value of `x` in lane `l` (wave-uniform l): v_readlane_b32 on the vector ALU instead of the ds_bpermute_b32 that
// __shfl issues — the kernel is short of LDS issue slots, not of ALU ones
__device__ __forceinline__ double lane_value(double x, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
#else
  (void)l;
  return x;
#endif
}
__device__ __forceinline__ unsigned long long lane_value(unsigned long long x, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(x >> 32), l) << 32) |
         (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(x & 0xFFFFFFFFull), l);
#else
  (void)l;
  return x;
#endif
}

// BinaryHeap::pop() followed by push(new) on a FULL heap of K <= kWaveReplaceMaxK slots, by the whole wave: every lane
// calls it with the same arguments; returns the new top weight. The array after the call is the array the literal
// sequence (heap_pop + heap_sift_up above, lane 0 walking 5.7 + 3.7 levels with an LDS round trip each at K = 100)
// leaves — tests/test_k0_wave_heap_model.py holds this formulation against the literal one operation by operation on
// streams full of equal weights, tests/test_gpu_long_arcs.py the kernel against the oracle:
//   1. lane p looks at node p: the larger of its children below `end` (ties: the right one, heap_pop's `wl <= wr`) and
//      that child's element — the choice sift_down_to_bottom makes IF it comes through p;
//   2. the path from the root is then a chase through the lanes' choices (v_readlane with a scalar index per level, on
//      the scalar unit), ending in the hole at the bottom;
//   3. sift_up of the former last element from the hole: the path's nodes hold their chosen child's element after the
//      shift, the element climbs past every node it is greater than — one ballot finds the deepest node that holds it
//      back; the nodes above take their child's element in ONE masked store, the element goes below that node;
//   4. push (heap_push_wave): the ancestors of the last slot are a fixed chain; lane i reads the i-th, one ballot finds
//      where the new element stops, one store shifts the others down.
// Three LDS round trips and ~80 instructions (half of them scalar) instead of ~9.4 round trips and ~140 one-lane
// vector instructions.
constexpr unsigned kWaveReplaceMaxK = 128;  // internal nodes of the heap <= 64: one per lane
#ifndef OUTFIT_K0_WAVE_REPLACE
#define OUTFIT_K0_WAVE_REPLACE 1   // 0: lane 0 walks the levels (the literal sequence; measurement aid)
#endif
constexpr bool kWaveReplace = OUTFIT_K0_WAVE_REPLACE != 0;
__device__ __forceinline__ void wave_lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
  // LDS instructions of one wave execute in order; this only keeps the COMPILER from moving a lane's loads across the
  // other lanes' stores (no s_waitcnt, no s_barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// BinaryHeap::push(new) into slot `slot` (= the heap's length before the push) by the whole wave: sift_up(0, slot, new)
// along the slot's ancestors A_i = ((slot + 1) >> i) - 1, A_0 = slot, A_m = 0 — lane i reads the i-th, one ballot finds
// the first that holds the element back (`if (ew <= wp) break`), one store shifts the ones below it down a level.
// Returns the new top weight (wave-uniform). Any slot < 2^31.
__device__ __forceinline__ double heap_push_wave(const WaveHeap& h, unsigned slot, double nw, unsigned long long ni,
                                                 unsigned lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double2 fresh = {nw, __longlong_as_double((long long)ni)};
  const unsigned m = 31u - (unsigned)__clz((int)(slot + 1u));
  const bool anc = lane >= 1u && lane <= m;
  const double2 a = h.e[anc ? ((slot + 1u) >> lane) - 1u : 0u];
  const unsigned long long halt = __ballot(anc && wt_le(nw, a.x));
  const unsigned s = halt ? (unsigned)__ffsll((long long)halt) - 1u : m + 1u;
  if (lane >= 1u && lane < s) h.e[((slot + 1u) >> (lane - 1u)) - 1u] = a;
  if (lane == 0u) h.e[((slot + 1u) >> (s - 1u)) - 1u] = fresh;
  wave_lds_order();
  const double top = h.e[0].x;
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(top)),
                          __builtin_amdgcn_readfirstlane(__double2loint(top)));
#else
  (void)h; (void)slot; (void)ni; (void)lane;
  return nw;
#endif
}
__device__ __forceinline__ double heap_replace_wave(const WaveHeap& h, unsigned K, double nw, unsigned long long ni,
                                                    unsigned lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned end = K - 1u;  // the heap's length after pop's --len; slot `end` holds the element pop moves
  const double2 fresh = {nw, __longlong_as_double((long long)ni)};
  if (end == 0u) {  // K == 1: pop empties the heap, push refills slot 0
    if (lane == 0u) h.e[0] = fresh;
    wave_lds_order();
    return nw;
  }
  const double2 last = h.e[end];
  // 1. this lane's node
  const unsigned c1 = 2u * lane + 1u, c2 = c1 + 1u;
  const bool has2 = c2 < end;
  const double2 L = h.e[c1 < end ? c1 : 0u];
  const double2 R = h.e[has2 ? c2 : 0u];
  const bool right = has2 && wt_le(L.x, R.x);
  const unsigned nxt = right ? c2 : c1;
  const double2 cv = right ? R : L;
  // 2. the path (wave-uniform: scalar registers)
  unsigned pos = 0u;
  unsigned long long path = 0ull;
  while (2u * pos + 2u < end) {  // heap_pop: `child <= end - 2` with child = 2 pos + 1
    path |= 1ull << pos;
    pos = (unsigned)__builtin_amdgcn_readlane((int)nxt, (int)pos);
  }
  if (2u * pos + 1u == end - 1u) {  // a single child at the very end moves up too
    path |= 1ull << pos;
    pos = 2u * pos + 1u;
  }
  // 3. sift_up(0, hole, last): `if (ew <= wp) break` from the bottom — the deepest path node that holds it back
  const bool on_path = (path >> lane) & 1ull;
  const unsigned long long stop = __ballot(on_path && wt_le(last.x, cv.x));
  unsigned dest = 0u;
  unsigned long long shift = 0ull;
  if (stop) {
    const int j = 63 - __clzll((long long)stop);
    dest = (unsigned)__builtin_amdgcn_readlane((int)nxt, j);
    shift = path & ((2ull << j) - 1ull);
  }
  if ((shift >> lane) & 1ull) h.e[lane] = cv;
  if (lane == 0u) h.e[dest] = last;
  wave_lds_order();
  // 4. push
  return heap_push_wave(h, end, nw, ni, lane);
#else
  (void)h; (void)K; (void)ni; (void)lane;
  return nw;
#endif
}

__device__ __forceinline__ unsigned select_triplets_wave(const double* mjd /* this trajectory */, unsigned long long n_obs64,
                                                         const TripletParams& P, double* ep /* LDS [n] */,
                                                         const WaveHeap& h, unsigned long long* out) {
  if (P.k_cap == 0 || n_obs64 < 3) return 0;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned n_obs = (unsigned)n_obs64;
  const unsigned n = (unsigned)reduced_len(n_obs64, P.max_keep);
  for (unsigned i = lane; i < n; i += 64u) ep[i] = mjd[(unsigned)keep_index(i, n_obs, P.max_keep)];
  __syncthreads();
  unsigned len = 0;       // wave-uniform
  double root_w = 0.0;    // wave-uniform copy of the heap's top weight
#ifdef OUTFIT_K0_STATS
  unsigned long long k0_local[12] = {};
  const unsigned long long k0_t0 = clock64();
#endif
  // survivors of one ballot, in lane order; `w` / `id` are the lanes' candidates
  auto absorb = [&](unsigned long long mask, double w, unsigned long long id) {
    K0STAT(4, 1);
    while (mask) {
      const int l = __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      const double wl = lane_value(w, l);
      const unsigned long long il = lane_value(id, l);
      K0STAT(5, 1);
      if (len < P.k_cap) {
        K0STAT(6, 1);
        if (kWaveReplace) {
          root_w = heap_push_wave(h, len, wl, il, lane);
          ++len;
          continue;
        }
        if (lane == 0) heap_sift_up(h, 0, len, wl, il);
        ++len;
      } else if (wl < root_w) {
        K0STAT(7, 1);
        if (kWaveReplace && P.k_cap <= kWaveReplaceMaxK) {  // the whole wave, three LDS round trips (heap_replace_wave)
          root_w = heap_replace_wave(h, P.k_cap, wl, il, lane);
          continue;
        }
        if (lane == 0) {
          unsigned ll = len;
          heap_pop(h, ll);
          heap_sift_up(h, 0, ll, wl, il);
        }
      } else {
        K0STAT(8, 1);
        continue;  // the top has dropped below this candidate since the ballot
      }
      __syncthreads();
      root_w = h.W(0);
    }
  };
  for (unsigned anchor = 0; anchor + 2 < n; ++anchor) {
    const double t0 = ep[anchor];
    // LastWindow::compute (index_generator.rs:94-114): the two `while` scans, 64 indices per step; each stops at
    // the first index that fails its test, exactly as the sequential loops do (no monotonicity is assumed)
    unsigned lo = anchor + 2;
    for (;;) {
      const unsigned i = lo + lane;
      const unsigned long long fail = __ballot(!(i < n && ep[i < n ? i : 0] - t0 < P.dt_min));
      const unsigned adv = fail ? (unsigned)__ffsll((long long)fail) - 1u : 64u;
      lo += adv;
      if (adv < 64u) break;
    }
    unsigned hi = lo - 1;
    if (hi < anchor + 1) hi = anchor + 1;
    for (;;) {
      const unsigned i = hi + 1 + lane;
      const unsigned long long fail = __ballot(!(i < n && ep[i < n ? i : 0] - t0 <= P.dt_max));
      const unsigned adv = fail ? (unsigned)__ffsll((long long)fail) - 1u : 64u;
      hi += adv;
      if (adv < 64u) break;
    }
    if (lo >= n || lo > hi || hi <= anchor + 1) continue;  // LastWindow::is_empty
    for (unsigned mb = anchor + 1; mb < hi; mb += 64u) {
      // 64 middles at once: first-gap term of the weight and the row test
      const unsigned m_mine = mb + lane;
      const bool m_in = m_mine < hi;
      const double tm_mine = ep[m_in ? m_mine : 0];
      const double g12_mine = s_gap(tm_mine - t0, P.inv_dtw);
      unsigned long long rows = __ballot(m_in && (len < P.k_cap || g12_mine + 1.0 < root_w));
      K0STAT(1, 1);
      while (rows) {
        const int rl = __ffsll((long long)rows) - 1;
        rows &= rows - 1ull;
        const double g12 = lane_value(g12_mine, rl);
        if (len == P.k_cap && !(g12 + 1.0 < root_w)) continue;  // the top has dropped since the row ballot
        const unsigned middle = mb + (unsigned)rl;
        const double tm = lane_value(tm_mine, rl);
        const unsigned first = lo > middle + 1 ? lo : middle + 1;
        K0STAT(2, 1);
        for (unsigned lb = first; lb <= hi; lb += 64u) {
          K0STAT(3, 1);
          const unsigned last = lb + lane;
          const bool in = last <= hi;
          const double w = g12 + s_gap(ep[in ? last : 0] - tm, P.inv_dtw);
          const unsigned long long mask = __ballot(in && isfinite(w) && (len < P.k_cap || w < root_w));
          if (mask) absorb(mask, w, pack_triplet(anchor, middle, last));
        }
      }
    }
  }
  __syncthreads();
#ifdef OUTFIT_K0_STATS
  if (lane == 0) {
    atomicAdd(&g_k0_stat[0], 1ull);
    for (int q = 1; q <= 8; ++q) atomicAdd(&g_k0_stat[q], k0_local[q]);
    atomicAdd(&g_k0_stat[11], (unsigned long long)(clock64() - k0_t0));
  }
#endif
  // sort_unstable_by(weight) as the stable order of the heap's vector (see select_triplets): rank of every slot
  for (unsigned i = lane; i < len; i += 64u) {
    const double wi = h.W(i);
    unsigned rank = 0;
    for (unsigned j = 0; j < len; ++j) {
      const double wj = h.W(j);
      rank += (wj < wi || (wj == wi && j < i)) ? 1u : 0u;
    }
    out[rank] = h.I(i);
  }
  return len;
}

}  // namespace outfit
