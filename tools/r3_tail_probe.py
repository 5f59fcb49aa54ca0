#!/usr/bin/env python3
"""Which pictures of the default batch make the entropy decoder's tail?  Decodes the cfg3 batch once alone with the wave timeline
on (PJD_DEBUG_STATS + PJD_DEBUG_DUMP) and relates every picture's re-sync rounds to its density (bytes of stream per MCU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["PJD_DEBUG_STATS"] = "1"
os.environ["PJD_DEBUG_DUMP"] = "/tmp/pjd_dbg.bin"
import pjd_amd
import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
jp = synth.cfg3_imagenet_like(n, seed=3, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True)
sc = [pjd_amd.Scanned(j) for j in jp]
ctx = pjd_amd.Context(0)
b = ctx.batch([s.desc for s in sc], pjd_amd.OUT_BMP)
b.upload(); b.decode(); b.sync(); b.decode(); b.sync()
b.info()
d = np.fromfile("/tmp/pjd_dbg.bin", np.uint32).reshape(-1, 32)
img = d[:, 6]
rounds = (d[:, 8:32] != 0).sum(1)
t_rounds = d[:, 2] / 100.0
end = (d[:, 0] - d[:, 0].min() + d[:, 1:6].sum(1)) / 100.0
rows = []
for i, s in enumerate(sc):
    w = img == i
    if not w.any():
        continue
    de = s.desc
    mcus = ((de.width + 15) // 16) * ((de.height + 15) // 16)
    rows.append((i, int(de.ecs_len) / mcus, int(w.sum()), int(rounds[w].max()), float(t_rounds[w].max()), float(end[w].max())))
rows.sort(key=lambda r: -r[5])
print("picture  bytes/MCU  waves  max rounds  rounds us  end us")
for r in rows[:25]:
    print("%6d  %8.1f  %5d  %9d  %9.0f  %7.0f" % r)
a = np.array([(r[1], r[3], r[5]) for r in rows])
for lo, hi in ((0, 100), (100, 150), (150, 200), (200, 250), (250, 300), (300, 1000)):
    m = (a[:, 0] >= lo) & (a[:, 0] < hi)
    if m.any():
        print(f"bytes/MCU {lo:4d}-{hi:4d}: {int(m.sum()):4d} pictures, max rounds mean {a[m, 1].mean():.1f} max {a[m, 1].max():.0f}, end us mean {a[m, 2].mean():.0f} max {a[m, 2].max():.0f}")
b.destroy(); ctx.close()
