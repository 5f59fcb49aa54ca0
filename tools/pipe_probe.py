#!/usr/bin/env python3
"""PCIe-inclusive rate of the pipelined batcher (libpjdpipe) on the bench's default picture set, by number of slots.

    PJD_PIPE_TRACE=1 python tools/pipe_probe.py [--slots 3,4,6] [--batches 16] [--workload cfg3]

With PJD_PIPE_TRACE=1 the library prints one line per batch (stage times), which is what profiles/r02_pcie.md quotes.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", default="3,4,6")
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--scan-threads", type=int, default=8)
    args = ap.parse_args()
    import bench
    import pjd_amd
    jpegs, _ = bench.make_workload(args.workload, args.images, 3, 16384, 0)
    for slots in [int(x) for x in args.slots.split(",")]:
        pjd_amd.pipe_run(jpegs=jpegs * (2 * slots), batch_images=len(jpegs), scan_threads=args.scan_threads, slots=slots, sink=None)
        print(f"--- slots {slots}", file=sys.stderr, flush=True)
        ps = pjd_amd.pipe_run(jpegs=jpegs * args.batches, batch_images=len(jpegs), scan_threads=args.scan_threads, slots=slots, sink=None)
        pjd_amd.pipe_release()
        print(json.dumps({"slots": slots, "MPix_per_s": round(ps["pixels"] / ps["wall_s"] / 1e6, 1), "d2h_GBps": round(ps["out_bytes"] / ps["wall_s"] / 1e9, 2),
                          "wall_ms": round(ps["wall_s"] * 1e3, 1), "exact_kernel_images": ps["n_exact_images"],
                          "worker_ms_per_batch": {k[:-2]: round(ps[k] * 1e3 / ps["n_batches"], 2) for k in ("scan_s", "create_s", "upload_s", "exec_s", "download_s")}}), flush=True)


if __name__ == "__main__":
    main()
