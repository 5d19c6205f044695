"""Where a small step's wall time goes: kernel time vs gaps between kernels (measurement aid).
  run  (GPU box, under the profiler):  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/small_batch_timeline.py run 1000
  read (anywhere):                     python3 tools/small_batch_timeline.py read DIR
`run` does 5 warm-up steps, then 20 timed steps back to back and one fence; `read` takes the kernel trace, keeps the
library's kernels of the last 20 steps and reports, per step on average: the span from the first kernel's start to the last
kernel's end, the union of the intervals in which at least one kernel runs (busy), and span - busy (the device idles between
dependent kernels: launch latency, not arithmetic)."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n):
    sys.path[:0] = [ROOT]
    import outfit_amd
    from outfit_amd import synth
    ed = synth.make_ephemeris()
    ds = synth.make_dataset(n, ephem=ed)
    s = outfit_amd.IODSession(outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(ed),
                              synth.bench_params(outfit_amd.IODParams()))
    for _ in range(5):
        s.run()
    s.results()
    t = time.perf_counter()
    for _ in range(20):
        s.run()
    s.results()
    print("wall per step %.3f ms (20 steps back to back, %d objects)" % (1e3 * (time.perf_counter() - t) / 20, n), flush=True)


def read(d, steps=20):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "outfit::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("outfit::", "").replace("void ", "")))
    rows.sort()
    names = [r[2] for r in rows]
    first = "k_triplet_select"
    starts = [i for i, nm in enumerate(names) if nm.startswith(first)]
    # a step = from one K0 launch group to the next; keep the last `steps` complete ones
    per_step = len(rows) // max(1, len([1 for i in starts if i == 0 or not names[i - 1].startswith(first)]))
    groups = [i for i in starts if i == 0 or not names[i - 1].startswith(first)]
    groups = groups[-steps:]
    end_idx = groups[1:] + [len(rows)]
    span = busy = ksum = 0.0
    by_kernel = {}
    for a, b in zip(groups, end_idx):
        ks = rows[a:b]
        span += max(k[1] for k in ks) - ks[0][0]
        cur_s, cur_e, u = ks[0][0], ks[0][1], 0
        for s0, e0, _ in ks[1:]:
            if s0 > cur_e:
                u += cur_e - cur_s
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        u += cur_e - cur_s
        busy += u
        for s0, e0, nm in ks:
            ksum += e0 - s0
            by_kernel.setdefault(nm, [0, 0.0])
            by_kernel[nm][0] += 1
            by_kernel[nm][1] += e0 - s0
    n = len(groups)
    total = (rows[-1][1] - rows[groups[0]][0]) / n
    print("steps %d, kernels per step %.1f" % (n, per_step))
    print("per step: first start -> next step's first start %.1f us; span %.1f us; busy (union) %.1f us; idle inside the span %.1f us; sum of kernels %.1f us"
          % (total / 1e3, span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3, ksum / n / 1e3))
    for nm, (c, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %5.1f launches/step %8.1f us/step" % (nm[:60], c / n, t / n / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
    else:
        read(sys.argv[2])
