"""The wave form of K0 replaces the top of its bounded heap (BinaryHeap::pop + push, mod.rs:396-403) with a handful of
wave-wide steps instead of lane 0 walking the levels one LDS round trip at a time (device/triplets.hpp
`heap_replace_wave`). This is the model of that formulation, held against the literal sequence of
std::collections::BinaryHeap (pop = swap_remove(0) + sift_down_to_bottom(0) + sift_up; push = sift_up from the end) on
streams with many equal weights: the ARRAY must be the same after every operation — the array order is what decides
which of several equal weights leaves next and how equal weights are ordered in the end.

No GPU, no oracle: plain Python on both sides; the device code is held against the oracle by tests/test_gpu_long_arcs.py.
"""
import random


# ---- the literal heap (the same statements as device/triplets.hpp heap_pop / heap_sift_up_raw) ----------------------
def sift_up(h, start, pos, elem):
    while pos > start:
        parent = (pos - 1) // 2
        if elem[0] <= h[parent][0]:
            break
        h[pos] = h[parent]
        pos = parent
    h[pos] = elem


def literal_replace(h, new):
    """heap full: pop() then push(new); h is the Vec of the BinaryHeap (max-heap on weight)."""
    last = h.pop()
    end = len(h)
    if end > 0:
        pos, child = 0, 1
        while end >= 2 and child <= end - 2:
            right = h[child][0] <= h[child + 1][0]
            h[pos] = h[child + 1] if right else h[child]
            pos = child + (1 if right else 0)
            child = 2 * pos + 1
        if child == end - 1:
            h[pos] = h[child]
            pos = child
        sift_up(h, 0, pos, last)
    h.append(None)
    sift_up(h, 0, len(h) - 1, new)


# ---- the wave formulation: every "lane" p works on node p; scalar steps are plain Python ------------------------------
def wave_replace(h, new):
    K = len(h)
    end = K - 1
    if end == 0:
        h[0] = new
        return
    last = h[end]
    lanes = range(64)
    # 1. every lane: the larger child of its node (ties: the right one), its value
    nxt, cv = [0] * 64, [None] * 64
    for p in lanes:
        c1, c2 = 2 * p + 1, 2 * p + 2
        has1, has2 = c1 < end, c2 < end
        L = h[c1] if has1 else None
        R = h[c2] if has2 else None
        right = has2 and L[0] <= R[0]
        nxt[p] = c2 if right else c1
        cv[p] = R if right else L
    # 2. scalar chase of the path from the root
    pos, mask = 0, 0
    while 2 * pos + 2 < end:
        mask |= 1 << pos
        pos = nxt[pos]
    if 2 * pos + 1 == end - 1:
        mask |= 1 << pos
        pos = 2 * pos + 1
    hole = pos
    # 3. sift_up of `last` from the hole: the deepest path node that holds it back
    stop = 0
    for p in lanes:
        if (mask >> p) & 1 and not (last[0] > cv[p][0]):
            stop |= 1 << p
    if stop == 0:
        dest, shift = 0, 0
    else:
        j = stop.bit_length() - 1
        dest = nxt[j]
        shift = mask & ((2 << j) - 1)
    assert dest == hole or (mask >> dest) & 1
    for p in lanes:                        # one masked store
        if (shift >> p) & 1:
            h[p] = cv[p]
    h[dest] = last
    # 4. push
    wave_push(h, end, new)


def wave_push(h, slot, new):
    """heap_push_wave: sift_up of `new` from `slot` along its chain of ancestors A_i = ((slot + 1) >> i) - 1."""
    m = (slot + 1).bit_length() - 1
    A = [((slot + 1) >> i) - 1 for i in range(m + 1)]
    a = [h[A[i]] if 1 <= i <= m else None for i in range(m + 1)]   # lane i reads the i-th ancestor
    s = m + 1
    for i in range(1, m + 1):              # (a ballot: the first lane whose ancestor holds the element back)
        if new[0] <= a[i][0]:
            s = i
            break
    for i in range(1, s):                  # one masked store (lane i), lane 0 stores `new`
        h[A[i - 1]] = a[i]
    h[A[s - 1]] = new


def run(K, n_ops, values, seed):
    rng = random.Random(seed)
    lit, wav = [], []
    for i in range(K):                     # fill by pushes
        e = (rng.choice(values), -1 - i)
        lit.append(None)
        sift_up(lit, 0, i, e)
        wav.append(None)
        wave_push(wav, i, e)
        assert wav == lit, (K, seed, "fill", i)
    done = 0
    for t in range(n_ops):
        w = rng.choice(values)
        if not (w < lit[0][0]):
            continue
        new = (w, t)
        literal_replace(lit, new)
        wave_replace(wav, new)
        assert wav == lit, (K, seed, t)
        done += 1
    return done


def test_wave_replace_is_the_literal_pop_push_on_the_array():
    total = 0
    for K in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128):
        for seed in range(6):
            # few distinct values: ties everywhere, at the top, along the path and among the ancestors of the last slot
            vals = [float(v) for v in range(1, 4 + seed * 3)]
            total += run(K, 400, vals, seed)
            # descending stream: every candidate replaces the top
            rng = random.Random(100 + seed)
            desc = sorted((rng.random() for _ in range(600)), reverse=True)
            lit = []
            for i in range(K):
                lit.append(None)
                sift_up(lit, 0, i, (desc[i], i))
            wav = list(lit)
            for t, w in enumerate(desc[K:]):
                if w < lit[0][0]:
                    literal_replace(lit, (w, K + t))
                    wave_replace(wav, (w, K + t))
                    assert wav == lit
                    total += 1
    assert total > 20000
