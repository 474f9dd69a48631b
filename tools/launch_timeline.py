#!/usr/bin/env python3
"""Where a batch's time goes BY BOUNCE: reads a rocprofv3 --kernel-trace CSV of one bench.py run and prints, per bounce iteration of the longest
complete batch, every kernel's launch duration, the gaps between launches, and the share of the batch spent in launches shorter than a
threshold (the "tail": queues too small to fill 256 CUs).

usage: tools/launch_timeline.py <dir with *_kernel_trace.csv> [short_us=200]
       (tools/run_gpu.sh timeline <tag> <bench args...> runs the trace and this on the GPU box)
"""
import csv, glob, re, sys, collections

def short(name):
    n = name
    for k in ("k_trace_primary", "k_trace_closest", "k_trace_shadow", "k_shadow_resolve", "k_shade_records", "k_shade", "k_finish_sample", "k_flush_survivors", "k_generate"):
        if k in n:
            if k == "k_shade":
                m = re.search(r"k_shade<([^>]*)>", n)  # <COUNT, SIMPLE, FIRST>
                return "k_shade_first" if m and m.group(1).replace(" ", "").endswith("true") else "k_shade"
            return k
    return None

def main():
    d = sys.argv[1]; short_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    # batches: a k_trace_primary launch opens one, the next k_finish_sample closes it
    batches, cur = [], None
    for s, e, k in rows:
        if k == "k_trace_primary" and (cur is None or cur["closed"]):
            cur = dict(rows=[], closed=False); batches.append(cur)
        if cur is None:
            continue
        cur["rows"].append((s, e, k))
        if k == "k_finish_sample":
            cur["closed"] = True
    batches = [b for b in batches if b["closed"]]
    if not batches:
        print("no complete batch in the trace"); return
    b = max(batches, key=lambda x: x["rows"][-1][1] - x["rows"][0][0])["rows"]  # the longest batch: a timed one (the trace also holds the 1-frame counting pass)
    t0, t1 = b[0][0], b[-1][1]
    wall = (t1 - t0) / 1e3
    busy = sum(e - s for s, e, _ in b) / 1e3
    print(f"batches in trace {len(batches)}; longest batch: {len(b)} launches, wall {wall / 1e3:.2f} ms, sum of launch durations {busy / 1e3:.2f} ms (gaps {100 * (1 - busy / wall):.1f} %)")
    # iterations: each k_trace_closest opens a bounce
    it, iters = [], []
    for s, e, k in b:
        if k == "k_trace_closest" and it:
            iters.append(it); it = []
        it.append((s, e, k))
    iters.append(it)
    print("bounce  " + "  ".join(f"{k:>16s}" for k in ("k_trace_closest", "k_shade", "k_trace_shadow", "k_shadow_resolve")) + "    total us   cumulative %")
    cum = 0.0
    for i, it in enumerate(iters):
        acc = collections.Counter()
        for s, e, k in it:
            acc[k] += (e - s) / 1e3
        tot = sum(acc.values()); cum += tot
        if i == 0:
            print("  (bounce 0 row includes " + ", ".join(f"{k} {v:.0f}" for k, v in acc.items() if k not in ("k_trace_closest", "k_shade", "k_trace_shadow", "k_shadow_resolve")) + ")")
        print(f"{i:5d}   " + "  ".join(f"{acc.get(k, 0.0):16.1f}" for k in ("k_trace_closest", "k_shade", "k_trace_shadow", "k_shadow_resolve")) + f"  {tot:10.1f}   {100 * cum / busy:6.1f}")
    tail = sum(e - s for s, e, _ in b if (e - s) / 1e3 < short_us) / 1e3
    n_tail = sum(1 for s, e, _ in b if (e - s) / 1e3 < short_us)
    gaps = [(b[i + 1][0] - b[i][1]) / 1e3 for i in range(len(b) - 1)]
    print(f"launches shorter than {short_us:.0f} us: {n_tail} of {len(b)}, {tail / 1e3:.3f} ms = {100 * tail / wall:.2f} % of the batch; gaps between launches: median {sorted(gaps)[len(gaps) // 2]:.1f} us, "
          f"sum {sum(g for g in gaps if g > 0) / 1e3:.3f} ms")

if __name__ == "__main__":
    main()
