#!/usr/bin/env python3
"""How much the kernels of several batches in flight overlap, from a rocprofv3 --kernel-trace CSV.

    python tools/inflight_overlap.py gpurun_out/pmc_cfg3/inflight

Looks at the part of the trace in which more than one hardware queue carries decode kernels (the warm-up and timed graph
replays of bench.py --in-flight N; the one-batch-at-a-time measurement that follows uses a single queue) and prints,
as markdown: the time during which 0 / 1 / 2 / 3+ decode kernels were running, the mean duration of each kernel there
against its duration when it runs alone, and a sample of the timeline.
"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--alone", default="", help="name=ms,... durations of the kernels running alone (for the comparison column)")
    a = ap.parse_args()
    f = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]   # the newest run
    ev = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("pjd_k"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
    ev.sort()
    count = defaultdict(int)
    for e in ev:
        count[e[3]] += 1
    main_q = max(count, key=count.get)
    other = [e for e in ev if e[3] != main_q]
    t0, t1 = min(e[0] for e in other), max(e[1] for e in other)
    sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    pts = sorted([(e[0], 1) for e in sel] + [(e[1], -1) for e in sel])
    conc, last, level = defaultdict(int), sel[0][0], 0
    for t, d in pts:
        conc[min(level, 3)] += t - last
        last, level = t, level + d
    span = sel[-1][1] - sel[0][0]
    alone = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.alone.split(",") if "=" in kv)
    print(f"window: {len(sel)} kernel launches on {len(set(e[3] for e in sel))} hardware queues, span {span / 1e6:.2f} ms\n")
    print("| decode kernels running at once | time | share |\n|---|---|---|")
    for k in range(4):
        print(f"| {k}{'+' if k == 3 else ''} | {conc[k] / 1e6:.2f} ms | {conc[k] / span:.1%} |")
    dur = defaultdict(list)
    for s, e, n, _ in sel:
        dur[n].append((e - s) / 1e6)
    print("\n| kernel | launches | mean ms in this window | ms alone (--in-flight 1) |\n|---|---|---|---|")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{n}` | {len(v)} | {sum(v) / len(v):.3f} | {alone.get(n, float('nan')):.3f} |")
    n_huff = sum(1 for e in sel if e[2] == "pjd_k_huff_lanes")
    print(f"\nbatches completed in the window: {n_huff} -> {span / 1e6 / max(n_huff, 1):.3f} ms per batch; sum of kernel durations "
          f"{sum(e[1] - e[0] for e in sel) / 1e6:.2f} ms = {sum(e[1] - e[0] for e in sel) / span:.2f} kernels running on average\n")
    print("timeline sample (ms from window start; queue; kernel):\n\n```")
    for s, e, n, st in sel[:36]:
        print(f"{(s - sel[0][0]) / 1e6:8.3f} -> {(e - sel[0][0]) / 1e6:8.3f}  q{st:>3}  {n}")
    print("```")


if __name__ == "__main__":
    main()
