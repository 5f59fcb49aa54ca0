#!/usr/bin/env python3
"""bench.py -- JPEG -> RGB throughput of the MI355X decode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg2rst|cfg5] [--images M]

One process per GPU (the driver launches N ranks with torch.distributed.run); every rank decodes its
OWN batch (weak scaling, no data-path collective: images are independent).  A step = one pass of the
whole hot path (Huffman entropy decode -> dequantise -> IDCT -> upsample -> YCbCr->RGB) over one
batch whose bitstreams, tables and work lists are already resident in HBM; pictures stay in HBM.
By default two identical batches are resident and steps alternate between them, each batch on its own HIP stream
(`--in-flight 2`), so step i is issued while step i-1 still runs -- a serving loop; every step's results are
drained and checked (`pjd_batch_sync`) before its batch is decoded again.  `one_batch_in_flight` reports the
same K steps strictly serialised, and the per-kernel durations / `roofline` come from serialised launches too.
Rank 0 prints ONE JSON line.  Beside the contract fields it carries `roofline` (dominant kernel: algorithmic
bytes / HIP-event duration against the HBM peak, PMC traffic from profiles/), `cpu_baseline` (oracle/_ref = the
reference's own host code on one core, same JPEGs) and, at N=1, `pcie_inclusive`: the same files through the
pipelined batcher (libpjdpipe) from JPEG bytes in host memory to BMP bytes in page-locked host memory -- never `value`.

Workloads (synthetic and seeded -- there is no dataset on the box; tools/synth.py):
    cfg3     M (default 1024) ImageNet-like 4:2:0 JPEGs of mixed sizes per GPU, hipGraph replay  [default]
    cfg2     one 3840x2160 4:2:0 q85 JPEG without restart markers
    cfg2rst  the same picture with one restart interval per MCU row
    cfg5     one 8192x8192 4:4:4 JPEG, one restart interval per MCU row (size via --tile)
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def make_workload(name, n_images, seed, tile):
    import synth
    if name == "cfg3":
        return synth.cfg3_imagenet_like(n_images, seed=seed), f"{n_images} ImageNet-like 4:2:0 JPEGs (mixed sizes, q75-95, seed {seed}) per GPU, hipGraph replay"
    if name == "cfg2":
        return [synth.cfg2_single_4k(seed=seed)], "one 3840x2160 4:2:0 q85 JPEG, no restart markers"
    if name == "cfg2rst":
        return [synth.cfg2_single_4k(seed=seed, restart_rows=True)], "one 3840x2160 4:2:0 q85 JPEG, restart interval = one MCU row"
    if name == "cfg5":
        return [synth.cfg5_tile(tile, seed=seed)], f"one {tile}x{tile} 4:4:4 q85 JPEG, restart interval = one MCU row"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(jpegs, budget_s=12.0):
    """Time the reference's CPU path on a bounded sample of the same JPEGs, one core.

    Preferred: oracle/_ref -- the reference's own read_JPEG + decode_Huffman_data + write_BMP compiled
    in place, with oracle/dpu_stages.c standing in for the UPMEM device stage (kind "reference").
    Otherwise the plain-C port oracle/liboracle.so (kind "port")."""
    import oracle_lib
    tmp = tempfile.mkdtemp(prefix="pjd_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        use_ref = oracle_lib.Ref.available()
        ref = oracle_lib.Ref() if use_ref else None
        port = None if use_ref else oracle_lib.Port()
        import pjd_amd
        pix, n, t_used = 0, 0, 0.0
        for k, data in enumerate(jpegs):
            s = pjd_amd.Scanned(data)
            w, h = int(s.desc.width), int(s.desc.height)
            jp, bp = os.path.join(tmp, f"{k}.jpg"), os.path.join(tmp, f"{k}.bmp")
            if use_ref:
                with open(jp, "wb") as f:
                    f.write(data)
                t0 = time.perf_counter()
                rc = ref.L.ref_decode_file(jp.encode(), bp.encode())
                t_used += time.perf_counter() - t0
                assert rc == 0
                os.remove(jp)
                os.remove(bp)
            else:
                t0 = time.perf_counter()
                port.decode(data)
                t_used += time.perf_counter() - t0
            pix += w * h
            n += 1
            if t_used > budget_s:
                break
        out = {"value": round(pix / t_used / 1e6, 3), "unit": "MPix/s", "cores": 1,
               "kind": "reference" if use_ref else "port",
               "sample": f"first {n} JPEGs of the workload ({pix / 1e6:.1f} MPix, {t_used:.1f} s), file -> BMP file, "
                         + ("oracle/_ref: reference scanner+Huffman+BMP writer, restated DPU stages" if use_ref else "oracle/liboracle.so")}
        if use_ref:
            # the same code on all host cores of this box's share: one image per task, threads (ctypes drops the GIL and
            # the reference's functions keep no static state) -- the reference itself is single-producer/single-consumer
            from concurrent.futures import ThreadPoolExecutor
            threads = max(1, min(16, os.cpu_count() or 1))
            sample = jpegs[:n]
            for k, data in enumerate(sample):
                with open(os.path.join(tmp, f"{k}.jpg"), "wb") as f:
                    f.write(data)

            def one(k):
                return ref.L.ref_decode_file(os.path.join(tmp, f"{k}.jpg").encode(), os.path.join(tmp, f"{k}.bmp").encode())

            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                rcs = list(ex.map(one, range(len(sample))))
            t_par = time.perf_counter() - t0
            assert not any(rcs)
            out["all_cores"] = {"value": round(pix / t_par / 1e6, 2), "unit": "MPix/s", "cores": threads,
                                "sample": f"the same {n} JPEGs, one image per task on {threads} threads ({t_par:.2f} s)"}
        return out
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--tile", type=int, default=8192)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="resident batches decoded round-robin, each on its own HIP stream: step i is issued while step i-1 is "
                         "still running, as a serving loop would (1 = strictly one step after the other)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check a few pictures against the oracle after the run")
    ap.add_argument("--e2e-batches", type=int, default=16,
                    help="also run the pipelined batcher (JPEG bytes in host memory -> BMP bytes in pinned host memory, "
                         "PCIe both ways) over this many batches of the workload; 0 = skip.  Reported as pcie_inclusive, never as value")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            raise SystemExit(2)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pjd_amd
    t_gen = time.perf_counter()
    jpegs, label = make_workload(args.workload, args.images, 3 + rank, args.tile)
    t_gen = time.perf_counter() - t_gen
    t_scan = time.perf_counter()
    scanned = [pjd_amd.Scanned(j) for j in jpegs]
    t_scan = time.perf_counter() - t_scan
    assert all(s.valid for s in scanned)
    if args.workload == "cfg2rst":
        # 4:2:0 + DRI: the reference's own restart rule garbles such files; decode per ITU-T.81
        for s in scanned:
            s.desc.flags = pjd_amd.F_STANDARD_RESTART

    # One context = one HIP stream.  `--in-flight` identical batches are resident; step i decodes batch i % in_flight,
    # so consecutive steps overlap (the slow tail of one step's entropy decode runs beside the next step's bulk).
    nfl = max(1, args.in_flight)
    ctxs = [pjd_amd.Context(local_rank) for _ in range(nfl)]      # raises if the HIP library / a gfx950 device is missing
    batches = [c.batch([s.desc for s in scanned], pjd_amd.OUT_RGB8) for c in ctxs]
    ctx, batch = ctxs[0], batches[0]
    t_up = time.perf_counter()
    batch.upload()
    t_up = time.perf_counter() - t_up
    for b in batches[1:]:
        b.upload()
    info = batch.info()
    if not args.no_graph:
        for b in batches:
            b.capture()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n, group):
        """n steps round-robin over `group`; every step's decode is drained and its status words read (sync)."""
        g = len(group)
        for i in range(n):
            b = group[i % g]
            if i >= g:
                b.sync()              # the step issued g steps ago on this batch: drain, read statuses, redo flagged images
            b.decode()
        for b in group[:min(n, g)]:
            b.sync()

    run_steps(max(args.warmup, nfl), batches)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, batches)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the same K steps strictly one after the other (reported beside `value`, never instead of it)
    dt_serial = None
    if nfl > 1:
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps, batches[:1])
        barrier()
        dt_serial = time.perf_counter() - t0
    info = batch.info()

    # per-kernel durations, HIP events on the library's own stream (ungraphed launches of the same work)
    ktimes, ktotal, reps = {}, 0.0, 5
    if not args.no_graph:
        pass
    for _ in range(reps):
        kt, tot = batch.decode_timed()
        batch.sync()
        for k, v in kt.items():
            ktimes[k] = ktimes.get(k, 0.0) + v / reps
        ktotal += tot / reps

    verify = None
    if args.verify and rank == 0:
        import numpy as np
        import oracle_lib
        port = oracle_lib.Port()
        outs, st = batch.download()
        idx = list(range(min(4, len(jpegs))))
        verify = all(np.array_equal(outs[i], port.decode(jpegs[i])["rgb"]) for i in idx)

    if rank == 0:
        pixels = info["pixels"]
        value = world * pixels * args.steps / dt / 1e6
        dom = max(ktimes, key=ktimes.get)
        alg_bytes = info["ecs_bytes"] + info["out_bytes"]            # ECS read once + RGB8 written once
        achieved = alg_bytes / (ktimes[dom] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel per launch, from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE, both KiB); only valid for the workload it was collected on
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_final_onepass_cfg3_traffic.json")))
            if args.workload == "cfg3" and args.images == 1024:
                for k, v in tj["per_kernel"].items():
                    if dom in k:
                        traffic = int((2 * (v.get("fetch_kib_raw") or 0) + (v.get("write_kib") or 0)) * 1024)
        except Exception:
            traffic = None
        line = {
            "metric": "MPixels/sec JPEG->RGB (bit-exact BMP)", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/int32 (integer IDCT), u8 out", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {label}", "images_per_gpu": info["n_images"],
                       "pixels_per_gpu": pixels, "ecs_bytes_per_gpu": info["ecs_bytes"],
                       "huffman_lanes": info["n_subsequences"], "exact_kernel_images": info["n_sequential"] + info["n_fallback"],
                       "hip_graph": not args.no_graph, "batches_in_flight": nfl,
                       "sync": {k: info[k] for k in ("n_huff_workgroups", "sync_rounds", "sync_lane_passes", "fix_rounds", "fix_lane_passes")}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(ktimes[dom], 4)},
            "kernels_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "kernel_pipeline_ms": round(ktotal, 4),
            "host_ms": {"generate": round(t_gen * 1e3, 1), "scan": round(t_scan * 1e3, 1), "upload": round(t_up * 1e3, 1)},
        }
        if dt_serial is not None:
            line["one_batch_in_flight"] = {"value": round(world * pixels * args.steps / dt_serial / 1e6, 2), "unit": "MPix/s",
                                           "ms_per_step": round(dt_serial / args.steps * 1e3, 4),
                                           "note": "rank 0's clock, steps strictly one after the other on one stream"}
        if verify is not None:
            line["verified_against_oracle"] = bool(verify)
        if world == 1 and args.e2e_batches > 0:
            # PCIe-inclusive rate: host scan + H2D + kernels + D2H, all overlapped by libpjdpipe (3 GPU slots)
            pipe_jpegs = jpegs * args.e2e_batches
            # warm-up: every slot allocates its HBM pool and page-locks its output buffer once
            pjd_amd.pipe_run(jpegs=jpegs * 6, batch_images=len(jpegs), scan_threads=6, slots=3, sink=None, device=local_rank)
            ps = pjd_amd.pipe_run(jpegs=pipe_jpegs, batch_images=len(jpegs), scan_threads=6, slots=3, sink=None, device=local_rank)
            pjd_amd.pipe_release()
            line["pcie_inclusive"] = {
                "value": round(ps["pixels"] / ps["wall_s"] / 1e6, 2), "unit": "MPix/s", "out_format": "bmp",
                "inputs": ps["n_inputs"], "batches": ps["n_batches"], "wall_ms": round(ps["wall_s"] * 1e3, 2),
                "d2h_GBps": round(ps["out_bytes"] / ps["wall_s"] / 1e9, 2),
                "worker_ms": {k[:-2]: round(ps[k] * 1e3, 1) for k in ("scan_s", "create_s", "upload_s", "exec_s", "download_s")},
                "note": "JPEG bytes in host memory -> BMP bytes in page-locked host memory; pictures are not consumed further"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(jpegs)
        print(json.dumps(line))
    for b in batches:
        b.destroy()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
