#!/usr/bin/env python3
"""GPU box: how full the lane regions get (pjd_batch_info.lane_fill_x1024) on several kinds of pictures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pjd_amd, synth
ctx = pjd_amd.Context(0)
sets = {
  "flat q5 fitted": [synth.make(444, 460, 91, 5, synth.SUB_422, 0, 1.0, True), synth.make(649, 513, 92, 5, synth.SUB_444, 0, 1.0, True), synth.make(300, 200, 94, 5, synth.SUB_GREY, 0, 1.0, True)],
  "flat q5 annex-K": [synth.make(444, 460, 91, 5, synth.SUB_420, 0, 1.0, False), synth.make(649, 513, 92, 5, synth.SUB_444, 0, 1.0, False)],
  "q30 fitted": [synth.make(800, 600, 6, 30, synth.SUB_420, 0, 1.0, True)],
  "q30 annex-K": [synth.make(800, 600, 6, 30, synth.SUB_420, 0, 1.0, False)],
  "cfg3 dense x64": synth.cfg3_imagenet_like(64, seed=3, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True),
  "cfg3lite x64": synth.cfg3_imagenet_like(64, seed=3),
  "q100 dense": [synth.make(700, 500, 8, 100, synth.SUB_444, 0, synth.DENSE_DETAIL, True)],
}
for name, jpegs in sets.items():
    sc = [pjd_amd.Scanned(j) for j in jpegs]
    with ctx.batch([s.desc for s in sc]) as b:
        b.upload(); b.decode(); b.sync(); i = b.info()
    print(f"{name}: fill {i['lane_fill_x1024'] / 1024:.3f}, mu {[round(pjd_amd.plan_step_bits(s.desc), 2) for s in sc[:3]]}, symbols/step {i['n_entries'] / max(1, i['n_steps']):.2f}, overflow {i['flag_waves'][5]}, S {i['sub_bytes']}", flush=True)
