#!/usr/bin/env python3
"""Offline report of a PJD_DEBUG_DUMP wave timeline (gpurun_out/r4_dbg.bin): phases per wave, who is running when, the slowest waves."""
import sys
import numpy as np
d = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r4_dbg.bin', np.uint32).reshape(-1, 32)
t0 = d[:, 0].astype(np.int64); t0 -= t0.min()
ph = d[:, 1:6].astype(np.int64)
start = t0 / 100.0
endA = (t0 + ph[:, 0]) / 100.0; endR = (t0 + ph[:, :2].sum(1)) / 100.0; endC = (t0 + ph[:, :4].sum(1)) / 100.0; end = (t0 + ph.sum(1)) / 100.0
for name, a in (("A", ph[:, 0]), ("R", ph[:, 1]), ("stitch", ph[:, 2]), ("C", ph[:, 3]), ("W", ph[:, 4])):
    a = a / 100.0
    print(name, "mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (a.mean(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max()))
print("end: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (end.mean(), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
for t in range(0, int(end.max()) + 200, 200):
    print(t, "running", int(((start <= t) & (end > t)).sum()), "in rounds", int(((endA <= t) & (endR > t)).sum()), "in write", int(((endC <= t) & (end > t)).sum()), "stitch/wait", int(((endR <= t) & (endC > t)).sum()))
r = d[:, 8:32]
act = (r >> 24) & 0x7f; walked = (r >> 31) & 1; tm = (r & 0xffffff) / 100.0
nr = (r != 0).sum(1)
print("rounds per wave mean %.2f" % nr.mean())
for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
    m = (r != 0) & (walked == 0) & (act >= lo) & (act <= hi)
    if m.any():
        print("rounds with %d-%d active: n=%d mean %.0f us" % (lo, hi, m.sum(), tm[m].mean()))
m = (r != 0) & (walked == 1)
print("walks: n=%d lanes mean %.1f time mean %.0f us; per lane %.0f" % (m.sum(), act[m].mean(), tm[m].mean(), (tm[m] / np.maximum(act[m], 1)).mean()))
for i in np.argsort(-end)[:12]:
    print("wave", i, "img", d[i, 6], "lanes", d[i, 7] & 0xff, "A %.0f R %.0f S %.0f C %.0f W %.0f end %.0f" % (*(ph[i] / 100.0), end[i]), [(int(act[i, k]), int(walked[i, k]), int(tm[i, k])) for k in range(nr[i])])
