#!/usr/bin/env python3
"""GPU box: PCIe-inclusive rate of the pipelined batcher against the number of slots per device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pjd_amd, synth
jpegs = synth.cfg3_imagenet_like(1024, seed=3, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True)
for slots in (3, 4, 5, 6, 3):
    pjd_amd.pipe_run(jpegs=jpegs * 8, batch_images=1024, scan_threads=8, slots=slots, sink=None, device=0)
    ps = pjd_amd.pipe_run(jpegs=jpegs * 32, batch_images=1024, scan_threads=8, slots=slots, sink=None, device=0)
    pjd_amd.pipe_release()
    print(f"slots {slots}: {ps['pixels'] / ps['wall_s'] / 1e9:.2f} GPix/s, {ps['out_bytes'] / ps['wall_s'] / 1e9:.1f} GB/s D2H, wall {ps['wall_s'] * 1e3:.0f} ms; worker ms", {k[:-2]: round(ps[k] * 1e3) for k in ("scan_s", "create_s", "upload_s", "exec_s", "download_s")}, flush=True)
