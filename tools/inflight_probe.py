"""Probe: K decodes of the cfg3 batch with 1, 2, 3 batches in flight (one context = one HIP stream each)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pjd_amd
import synth

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
jpegs = synth.cfg3_imagenet_like(n_img, seed=3)
scanned = [pjd_amd.Scanned(j) for j in jpegs]
descs = [s.desc for s in scanned]
K = 40
for nfl in (1, 2, 3, 4):
    ctxs = [pjd_amd.Context(0) for _ in range(nfl)]
    bs = [c.batch(descs, pjd_amd.OUT_RGB8) for c in ctxs]
    for b in bs:
        b.upload(); b.capture(); b.decode(); b.sync()
    t0 = time.perf_counter()
    for i in range(K):
        b = bs[i % nfl]
        if i >= nfl:
            b.sync()                 # the decode issued nfl steps ago on this batch
        b.decode()
    for b in bs:
        b.sync()
    dt = time.perf_counter() - t0
    pix = bs[0].info()["pixels"]
    print(f"in flight {nfl}: {dt / K * 1e3:.3f} ms/step  {pix * K / dt / 1e9:.1f} GPix/s", flush=True)
    for b in bs:
        b.destroy()
    for c in ctxs:
        c.close()
