#!/usr/bin/env python3
"""GPU box: 48 large seeded pictures (700-2600 px, all sampling modes, q85-100, plain and optimised tables) in one batch and
in a batch of ten, BMP output, against the oracle: long multi-wave pictures, nothing may fall back."""
import sys, os
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, oracle_lib, pjd_amd, synth
from concurrent.futures import ThreadPoolExecutor
port = oracle_lib.Port(); ctx = pjd_amd.Context(0); rng = np.random.default_rng(77)
jp=[]
for k in range(48):
    w, h = int(rng.integers(700, 2600)), int(rng.integers(700, 2200))
    sub = int(rng.choice([synth.SUB_444, synth.SUB_422, synth.SUB_420, synth.SUB_440, synth.SUB_GREY]))
    jp.append(synth.make(w, h, 4400+k, int(rng.choice([85, 97, 100])), sub, 0, float(rng.choice([1.0, synth.DENSE_DETAIL])), bool(k&1)))
sc=[pjd_amd.Scanned(j) for j in jp]
bad=0
for group in (list(range(48)), list(range(0,48,5))):
    with ctx.batch([sc[i].desc for i in group], pjd_amd.OUT_BMP) as b:
        b.upload(); b.decode(); outs, st = b.download(); info = b.info()
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(lambda t: outs[t[0]].tobytes() == port.decode(jp[t[1]])["bmp"], list(enumerate(group))))
    bad += res.count(False)
    print("group", len(group), "wrong", res.count(False), "status", sum(1 for s in st if s), "fallback", info["n_fallback"], info["flag_waves"], "lanes", info["n_subsequences"], "S", info["sub_bytes"], flush=True)
print("big fuzz bad", bad)
