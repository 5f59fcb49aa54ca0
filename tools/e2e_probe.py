"""Probe of the pipelined batcher: PCIe-inclusive throughput for several thread/slot settings."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pjd_amd
import synth

jpegs = synth.cfg3_imagenet_like(1024, seed=3)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pjd_amd.pipe_run(jpegs=jpegs * 8, batch_images=1024, slots=4, sink=None)
for scan_threads, slots, batch in [(6, 3, 1024), (12, 3, 1024), (12, 2, 1024), (12, 4, 1024), (12, 3, 512), (12, 3, 2048), (14, 1, 1024)]:
    ps = pjd_amd.pipe_run(jpegs=jpegs * nb, batch_images=batch, scan_threads=scan_threads, slots=slots, sink=None)
    print(json.dumps({"scan_threads": scan_threads, "slots": slots, "batch": batch, "MPix/s": round(ps["pixels"] / ps["wall_s"] / 1e6, 1),
                      "wall_ms": round(ps["wall_s"] * 1e3, 1),
                      "per_batch_ms": {k[:-2]: round(ps[k] * 1e3 / ps["n_batches"], 2) for k in ("scan_s", "create_s", "upload_s", "exec_s", "download_s")}}), flush=True)
