/*
 * triplets.c — oracle restatement of
 * src/initial_orbit_determination/triplet_generation/{mod,index_generator}.rs.
 * The best-K selection goes through Rust's std::collections::BinaryHeap; its
 * sift-up / sift-down-to-bottom are restated so that ties in weight keep the
 * same survivors (quirk Q10). The final `sort_unstable_by` (weight only) is
 * restated as a stable insertion sort: the relative order of EQUAL weights is
 * not pinned (it only permutes candidates of equal weight).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* index_generator.rs:66-75 */
size_t orc_downsample_uniform_with_edges(size_t n, size_t max_keep, size_t *keep) {
  if (n == 0) return 0;
  if (max_keep >= n) { for (size_t i = 0; i < n; ++i) keep[i] = i; return n; }
  if (max_keep <= 3) { keep[0] = 0; keep[1] = n / 2; keep[2] = n - 1; return 3; }
  for (size_t i = 0; i < max_keep; ++i) keep[i] = i * (n - 1) / (max_keep - 1);
  return max_keep;
}

typedef struct { double w; uint32_t i, j, k; } wtrip;

/* triplet_generation/mod.rs:135-146 — Ord: weight.partial_cmp, NaN falls back to indices */
static int wt_cmp(const wtrip *a, const wtrip *b) {
  if (a->w < b->w) return -1;
  if (a->w > b->w) return 1;
  if (a->w == b->w) return 0;
  if (a->i != b->i) return a->i < b->i ? -1 : 1;
  if (a->j != b->j) return a->j < b->j ? -1 : 1;
  if (a->k != b->k) return a->k < b->k ? -1 : 1;
  return 0;
}
static int wt_le(const wtrip *a, const wtrip *b) { return wt_cmp(a, b) <= 0; }

/* std BinaryHeap::sift_up(start, pos) */
static size_t heap_sift_up(wtrip *d, size_t start, size_t pos) {
  wtrip elt = d[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (wt_le(&elt, &d[parent])) break;
    d[pos] = d[parent];
    pos = parent;
  }
  d[pos] = elt;
  return pos;
}
/* std BinaryHeap::sift_down_to_bottom(0) */
static void heap_sift_down_to_bottom(wtrip *d, size_t end) {
  size_t start = 0, pos = 0;
  wtrip elt = d[pos];
  size_t child = 2 * pos + 1;
  size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
  while (child <= lim && end >= 2) {
    if (wt_le(&d[child], &d[child + 1])) child += 1;
    d[pos] = d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) { d[pos] = d[child]; pos = child; }
  d[pos] = elt;
  heap_sift_up(d, start, pos);
}
static void heap_push(wtrip *d, size_t *len, wtrip x) {
  d[*len] = x;
  heap_sift_up(d, 0, *len);
  ++*len;
}
static void heap_pop(wtrip *d, size_t *len) {
  wtrip last = d[*len - 1];
  --*len;
  if (*len > 0) { d[0] = last; heap_sift_down_to_bottom(d, *len); }
}

/* triplet_generation/mod.rs:265-274 */
static double s_gap(double dt, double inv_dtw) {
  double r = dt * inv_dtw;
  return (r <= 1.0) ? 1.0 / r : 1.0 + r;
}

/* index_generator.rs:94-114 */
typedef struct { size_t lo, hi; } lastwin;
static lastwin win_compute(size_t anchor, const double *ep, size_t n, double dt_min, double dt_max) {
  double t0 = ep[anchor];
  size_t lo = anchor + 2;
  while (lo < n && ep[lo] - t0 < dt_min) ++lo;
  size_t hi = (lo > 0 ? lo - 1 : 0);
  if (hi < anchor + 1) hi = anchor + 1;
  while (hi + 1 < n && ep[hi + 1] - t0 <= dt_max) ++hi;
  return (lastwin){lo, hi};
}
static int win_empty(lastwin w, size_t anchor, size_t n) {
  return w.lo >= n || w.lo > w.hi || w.hi <= anchor + 1;
}

int orc_tie_order = 0; /* test aid, see below */

/* triplet_generation/mod.rs:328-408 + index_generator.rs:133-271.
 * NOTE quirk Q1: the reduced indices are returned as-is (no keep[] remap). */
size_t orc_generate_triplet_indices(const double *mjd, size_t n_obs, const outfit_iod_params_t *ip,
                                    uint32_t *idx, double *wout) {
  if (ip->max_triplets == 0 || n_obs < 3) return 0;
  size_t *keep = (size_t *)malloc(sizeof(size_t) * (n_obs + 3));
  size_t n = orc_downsample_uniform_with_edges(n_obs, (size_t)ip->max_obs_for_triplets, keep);
  double *ep = (double *)malloc(sizeof(double) * n);
  for (size_t i = 0; i < n; ++i) ep[i] = mjd[keep[i]];
  free(keep);

  size_t kcap = ip->max_triplets;
  double inv_dtw = 1.0 / ip->optimal_interval_time;
  wtrip *heap = (wtrip *)malloc(sizeof(wtrip) * (kcap + 1));
  size_t hlen = 0;

  /* TripletIndexGenerator::new */
  lastwin win = (n >= 3) ? win_compute(0, ep, n, ip->dt_min, ip->dt_max_triplet) : (lastwin){n, 0};
  size_t anchor = 0, middle = 1, last = win.lo > 2 ? win.lo : 2;
  for (;;) {
    /* Iterator::next */
    if (anchor + 2 >= n) break;
    int advance = 0;
    if (win_empty(win, anchor, n)) advance = 1;
    else if (middle >= win.hi) advance = 1;
    if (advance) {
      anchor += 1;
      if (anchor + 2 >= n) break;
      win = win_compute(anchor, ep, n, ip->dt_min, ip->dt_max_triplet);
      middle = anchor + 1;
      last = win.lo > middle + 1 ? win.lo : middle + 1;
      continue;
    }
    if (last <= middle) last = win.lo > middle + 1 ? win.lo : middle + 1;
    if (last > win.hi) {
      middle += 1;
      last = win.lo > middle + 1 ? win.lo : middle + 1;
      continue;
    }
    size_t f = anchor, m = middle, l = last;
    last += 1;
    /* collect_best_k_triplets body */
    double dt12 = ep[m] - ep[f], dt23 = ep[l] - ep[m];
    double w = s_gap(dt12, inv_dtw) + s_gap(dt23, inv_dtw);
    if (!isfinite(w)) continue;
    wtrip x = {w, (uint32_t)f, (uint32_t)m, (uint32_t)l};
    if (hlen < kcap) heap_push(heap, &hlen, x);
    else if (hlen > 0 && w < heap[0].w) { heap_pop(heap, &hlen); heap_push(heap, &hlen, x); }
  }
  /* into_vec + sort by weight (stable insertion sort on the heap's vector order) */
  for (size_t i = 1; i < hlen; ++i) {
    wtrip x = heap[i];
    size_t j = i;
    while (j > 0 && heap[j - 1].w > x.w) { heap[j] = heap[j - 1]; --j; }
    heap[j] = x;
  }
  /* Rust's sort_unstable_by (ipnsort for K > 20) leaves the order of EQUAL weights unspecified and is not
   * restated; orc_tie_order = 1 reverses every run of equal weights, so that tests can show what that order does
   * and does not decide (tests/test_oracle_pipeline.py::test_order_of_equal_weights...). */
  if (orc_tie_order)
    for (size_t a = 0; a < hlen;) {
      size_t e = a + 1;
      while (e < hlen && heap[e].w == heap[a].w) ++e;
      for (size_t lo = a, hi = e; lo + 1 < hi; ++lo, --hi) { wtrip t = heap[lo]; heap[lo] = heap[hi - 1]; heap[hi - 1] = t; }
      a = e;
    }
  for (size_t i = 0; i < hlen; ++i) {
    idx[3 * i] = heap[i].i; idx[3 * i + 1] = heap[i].j; idx[3 * i + 2] = heap[i].k;
    if (wout) wout[i] = heap[i].w;
  }
  free(ep);
  free(heap);
  return hlen;
}
