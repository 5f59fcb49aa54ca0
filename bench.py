#!/usr/bin/env python3
"""bench.py -- JPEG -> RGB throughput of the MI355X decode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg2rst|cfg5] [--images M]

One process per GPU (the driver launches N ranks with torch.distributed.run); every rank decodes its
OWN batch (weak scaling, no data-path collective: images are independent).  A step = one pass of the
whole hot path (Huffman entropy decode -> dequantise -> IDCT -> upsample -> YCbCr->RGB) over one
batch whose bitstreams, tables and work lists are already resident in HBM; pictures stay in HBM.
By default four identical batches are resident and steps rotate over them, each batch on its own HIP stream
(`--in-flight 4`: four streams = the HIP runtime's four hardware queues; with GPU_MAX_HW_QUEUES=8 in the environment 6 or 8 in flight measure
within 1.5 % of it at the end of round 3 -- 109.7 / 110.0 / 111.3 GPix/s for 4 / 6 / 8 -- and 4 has the shortest fill and drain when few steps are timed),
so step i is issued while step i-1 still runs -- a serving loop; every step's results are
drained and checked (`pjd_batch_sync`) before its batch is decoded again.  `one_batch_in_flight` reports the
same K steps strictly serialised, and the per-kernel durations / `roofline` come from serialised launches too.
Rank 0 prints ONE JSON line.  Beside the contract fields it carries `roofline` (dominant kernel: algorithmic
bytes / HIP-event duration against the HBM peak, PMC traffic from profiles/), `cpu_baseline` (oracle/_ref = the
reference's own host code on one core, same JPEGs) and, at N=1, `pcie_inclusive`: the same files through the
pipelined batcher (libpjdpipe) from JPEG bytes in host memory to BMP bytes in page-locked host memory -- never `value`.

Workloads (synthetic and seeded -- there is no dataset on the box; tools/synth.py):
    cfg3     M (default 1024) ImageNet-like 4:2:0 JPEGs of mixed sizes per GPU, hipGraph replay  [default].  The set has the
             density of the one real ImageNet file at hand (~0.58 B/px against its 0.58), every picture carries its own
             optimised Huffman tables (four distinct ones, like that file), and picture 0 is that file re-encoded 4:2:0.
             `variants` in the output line repeats the measurement on the lighter round-1 set (Annex-K tables, 0.32 B/px).
    cfg3lite the round-1 set itself as the main workload
    cfg2     one 3840x2160 4:2:0 q85 JPEG without restart markers
    cfg2rst  the same picture with one restart interval per MCU row
    cfg5     one 8192x8192 4:4:4 JPEG, one restart interval per MCU row (size via --tile); every rank decodes its own
    cfg5split  BASELINE config 5 as stated: ONE picture (--tile, default 8192; 16384 is the config's size) scanned on rank 0,
             descriptor broadcast + entropy-coded bytes scattered over the process group (nccl = RCCL over xGMI), every
             rank decodes only its restart-segment range; value = the picture's pixels / slowest rank (strong scaling).
             With --gpus 1 it is the plain cfg5 measurement.
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


def bundled_rgb(device):
    """The bundled ImageNet sample as pixels, decoded by THIS library (the product, not the oracle)."""
    import numpy as np
    import pjd_amd
    data = open(os.path.join(ROOT, "tests", "golden", "ilsvrc_val_00000001.jpg"), "rb").read()
    s = pjd_amd.Scanned(data)
    ctx = pjd_amd.Context(device)
    outs, st = ctx.decode([s.desc], pjd_amd.OUT_RGB8)
    ctx.close()
    assert st == [0]
    return np.asarray(outs[0]).reshape(int(s.desc.height), int(s.desc.width), 3)


def make_workload(name, n_images, seed, tile, device=0):
    import synth
    if name == "cfg3":
        jp = synth.cfg3_imagenet_like(n_images, seed=seed, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True,
                                      extra=[bundled_rgb(device)])
        return jp, (f"{n_images} ImageNet-like 4:2:0 JPEGs (mixed sizes, q88-97, per-picture optimised Huffman tables, seed {seed}; "
                    "picture 0 = the bundled ImageNet sample re-encoded) per GPU, hipGraph replay")
    if name == "cfg3lite":
        return synth.cfg3_imagenet_like(n_images, seed=seed), f"{n_images} ImageNet-like 4:2:0 JPEGs (mixed sizes, q75-95, Annex-K tables, seed {seed}) per GPU, hipGraph replay"
    if name == "cfg2":
        return [synth.cfg2_single_4k(seed=seed)], "one 3840x2160 4:2:0 q85 JPEG, no restart markers"
    if name == "cfg2rst":
        return [synth.cfg2_single_4k(seed=seed, restart_rows=True)], "one 3840x2160 4:2:0 q85 JPEG, restart interval = one MCU row"
    if name in ("cfg5", "cfg5split"):
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
            threads = max(1, os.cpu_count() or 1)
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
            out["host_nproc"] = os.cpu_count()
            out["all_cores"] = {"value": round(pix / t_par / 1e6, 2), "unit": "MPix/s", "cores": threads,
                                "sample": f"the same {n} JPEGs, one image per task on {threads} threads ({t_par:.2f} s)"}
        return out
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def cli_end_to_end(jpegs, device):
    """The drop-in path file to file: `bin/decoder --pipeline` over the SAME JPEGs as files in a tmpfs directory, BMP files written
    next to them -- the reference's own figure of merit is its "End-to-end execution time" over files (src/decoder_host.cpp:379-394).
    A child process (it pays process start, HIP initialisation and pool warm-up like any CLI run), timed by its own "Profiles:" block."""
    import hashlib
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "bin", "decoder")
    if not os.path.exists(exe):
        return {"error": "bin/decoder is not built"}
    tmp = tempfile.mkdtemp(prefix="pjd_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        names = []
        for k, data in enumerate(jpegs):
            names.append(os.path.join(tmp, f"{k:05d}.jpg"))
            with open(names[-1], "wb") as f:
                f.write(data)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "--pipeline", "--device", str(device)] + names, capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        rows, pixels, pictures = {}, None, None
        for line in p.stdout.splitlines():
            line = line.strip()
            if line.startswith("End-to-end execution time:"):
                rows["end_to_end_s"] = float(line.split(":")[1].strip().rstrip("s"))
            elif line.startswith("- ") and "time:" in line:
                k, v = line[2:].split(":")
                rows[k.strip().replace(" ", "_").replace("-", "_")] = float(v.strip().rstrip("s"))
            elif line.startswith("- Total"):
                w = line.split()
                pictures, pixels = int(w[4]), float(w[6]) * 1e6          # "- Total <calls> calls, <pictures> pictures, <MPixels> MPixels"
        n_bmp = sum(1 for k in range(len(jpegs)) if os.path.exists(os.path.join(tmp, f"{k:05d}.bmp")))
        out = {"returncode": p.returncode, "files": len(jpegs), "bmp_files_written": n_bmp, "process_wall_s": round(wall, 3), "profiles": rows}
        if "end_to_end_s" in rows and pixels:
            out.update(value=round(pixels / rows["end_to_end_s"] / 1e6, 2), unit="MPix/s", pictures=pictures,
                       note="JPEG files on tmpfs -> BMP files on tmpfs through bin/decoder --pipeline (scan | H2D | kernels | D2H | write overlapped); "
                            "value = pixels / the CLI's own End-to-end execution time, process start and HIP initialisation excluded (process_wall_s has them)")
            # one picture checked by content: the BMP file equals what this library decodes in memory (the oracle pins that in tests/)
            k = 0
            import pjd_amd
            sc = pjd_amd.Scanned(jpegs[k])
            ctx = pjd_amd.Context(device)
            outs, _ = ctx.decode([sc.desc], pjd_amd.OUT_BMP)
            ctx.close()
            bmp = open(os.path.join(tmp, f"{k:05d}.bmp"), "rb").read() if n_bmp else b""
            out["first_bmp_identical_to_library_decode"] = hashlib.sha256(bmp).hexdigest() == hashlib.sha256(bytes(outs[0])).hexdigest()
        else:
            out["error"] = (p.stdout[-300:] + p.stderr[-300:])
        return out
    except Exception as e:          # never lose the bench line over this extra
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cxx_split_child(tile, seed, devices):
    """pjd_split_decode (the C++ path: one host thread per device, ONE ncclBroadcast of the descriptor) over `devices` in a child
    process with a time limit, compared with the one-device decode.  Evidence for BASELINE config 5 at N > 1; never `value`."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import sys, json, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, pjd_amd, synth
        data = synth.cfg5_tile(%d, seed=%d)
        sc = pjd_amd.Scanned(data)
        devs = %r
        got, status, stats = pjd_amd.split_decode(sc.desc, devs, pjd_amd.OUT_BMP)
        got2, status2, stats2 = pjd_amd.split_decode(sc.desc, devs, pjd_amd.OUT_BMP)      # communicators cached: the steady state
        ctx = pjd_amd.Context(devs[0])
        whole, st = ctx.decode([sc.desc], pjd_amd.OUT_BMP)
        ctx.close()
        same = hashlib.sha256(np.asarray(got2).tobytes()).hexdigest() == hashlib.sha256(np.asarray(whole[0]).tobytes()).hexdigest()
        keep = ("wall_s", "broadcast_s", "upload_s", "exec_s", "download_s", "blob_bytes", "n_segments", "n_ranks", "n_exact", "rccl_used", "redone_whole")
        print("RESULT " + json.dumps({"status": int(status2), "equals_one_device_decode": bool(same), "first_call": {k: stats[k] for k in keep},
                                      "second_call": {k: stats2[k] for k in keep}, "pixels": int(sc.desc.width) * int(sc.desc.height)}))
    """) % (os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"), os.path.join(ROOT, "tools"), tile, seed, list(devices))
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420)
        for line in p.stdout.splitlines():
            if line.startswith("RESULT "):
                return json.loads(line[7:])
        return {"error": (p.stdout[-200:] + p.stderr[-300:]), "returncode": p.returncode}
    except Exception as e:
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default 300: a timed region of about 0.6 s on the default workload)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--tile", type=int, default=8192)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="resident batches decoded round-robin, each on its own HIP stream: step i is issued while step i-1 is "
                         "still running, as a serving loop would (1 = strictly one step after the other)")
    ap.add_argument("--plan-mode", default="auto", choices=["auto", "latency", "throughput"],
                    help="pjd_set_plan_mode of the contexts: auto = throughput for the batches kept in flight, latency for the batch that is "
                         "decoded alone (one_batch_in_flight, kernels_ms, roofline)")
    ap.add_argument("--force-exact", action="store_true",
                    help="decode everything with the exact one-lane kernel (PJD_F_FORCE_SEQUENTIAL): the bound of the fallback path, not a product mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra measurement on the lighter cfg3lite set")
    ap.add_argument("--out-format", default="bmp", choices=["bmp", "rgb8"],
                    help="what the device writes per picture: the BMP file image the reference's writer would emit (default), or tight RGB8")
    ap.add_argument("--verify", action="store_true", help="check a few pictures against the oracle after the run")
    ap.add_argument("--no-cli", action="store_true", help="skip cli_end_to_end (bin/decoder --pipeline over the workload as files on tmpfs)")
    ap.add_argument("--e2e-batches", type=int, default=32,
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
    # rehearsal on a one-GPU box (never set by the driver): PJD_BENCH_DEVICE puts every rank on that device and
    # PJD_BENCH_BACKEND=gloo replaces RCCL, so that the multi-rank control flow (barriers, reductions, totals) can be run
    if "PJD_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PJD_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PJD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    # how many ranks the collective backend really joins: one all-reduce of a 1 per rank (RCCL over xGMI with the nccl backend)
    comm = {"backend": None, "ranks": 1}
    if world > 1:
        t = torch.ones(1, dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        comm = {"backend": dist.get_backend(), "ranks": int(t.item())}

    import pjd_amd
    out_fmt = pjd_amd.OUT_BMP if args.out_format == "bmp" else pjd_amd.OUT_RGB8
    nfl = max(1, args.in_flight)

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

    def measure(workload, steps, warmup, full):
        """Generate `workload`, make `nfl` resident batches, time `steps` steps.  full: also the serialised run and the
        per-kernel HIP-event timings."""
        split = workload == "cfg5split" and world > 1
        t_gen = time.perf_counter()
        if split and rank != 0:
            jpegs, label = [], ""
        else:
            jpegs, label = make_workload(workload, args.images, 3 + (0 if split else rank), args.tile, local_rank)
        t_gen = time.perf_counter() - t_gen
        t_scan = time.perf_counter()
        scanned = [pjd_amd.Scanned(j) for j in jpegs]
        t_scan = time.perf_counter() - t_scan
        assert all(s.valid for s in scanned)
        t_dist, keep = 0.0, None
        if split:
            # rank 0 scanned the file: one broadcast of the descriptor, every rank receives its slice of the bitstream
            from pjd_amd import parallel
            t_dist = time.perf_counter()
            d, keep, blob_bytes = parallel.distribute_image(scanned[0] if rank == 0 else None, src=0, device=torch.device("cuda", local_rank))
            t_dist = time.perf_counter() - t_dist
            descs = [d] if d is not None else []
            label = label or f"one {args.tile}x{args.tile} 4:4:4 q85 JPEG, restart interval = one MCU row"
            label += f"; split by restart segment over {world} ranks (descriptor blob {blob_bytes} B broadcast)"
            split_info = {"blob_bytes": int(blob_bytes), "rank0_segments": int(d.shard_n_segs) if d is not None else 0,
                          "rank0_slice_bytes": int(len(keep[0])) if keep is not None else 0, "collective_backend": dist.get_backend()}
        else:
            descs = [s.desc for s in scanned]
        if args.force_exact:
            for s in scanned:
                s.desc.flags = int(s.desc.flags) | pjd_amd.F_FORCE_SEQUENTIAL
        if workload == "cfg2rst":
            # 4:2:0 + DRI: the reference's own restart rule garbles such files; decode per ITU-T.81
            for s in scanned:
                s.desc.flags = int(s.desc.flags) | pjd_amd.F_STANDARD_RESTART
        # One context = one HIP stream.  `--in-flight` identical batches are resident; step i decodes batch i % in_flight,
        # so consecutive steps overlap (the slow tail of one step's entropy decode runs beside the next step's bulk).
        # Batches kept in flight are planned for pictures per second, a batch decoded alone for its own latency (include/pjd.h,
        # pjd_set_plan_mode: how much stream an entropy-decoder lane takes); --plan-mode overrides both.
        mode_of = {"latency": pjd_amd.PLAN_LATENCY, "throughput": pjd_amd.PLAN_THROUGHPUT}
        main_mode = mode_of.get(args.plan_mode, pjd_amd.PLAN_THROUGHPUT if nfl > 1 else pjd_amd.PLAN_LATENCY)
        ctxs = [pjd_amd.Context(local_rank, plan_mode=main_mode) for _ in range(nfl)]      # raises if the HIP library / a gfx950 device is missing
        batches = [c.batch(descs, out_fmt) for c in ctxs]
        batch = batches[0]
        t_up = time.perf_counter()
        batch.upload()
        t_up = time.perf_counter() - t_up
        for b in batches[1:]:
            b.upload()
        if not args.no_graph:
            for b in batches:
                b.capture()
        run_steps(max(warmup, nfl), batches)
        barrier()
        t0 = time.perf_counter()
        run_steps(steps, batches)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if os.environ.get("PJD_DEBUG_STATS") and rank == 0:
            print("[bench] wave timeline of the last decode issued with batches in flight:", file=sys.stderr, flush=True)
            batches[-1].info()            # the library prints the timeline of this batch's last decode (stderr)
        r = {"jpegs": jpegs, "label": label, "dt": dt, "steps": steps, "split": split, "host_ms": {"generate": round(t_gen * 1e3, 1),
             "scan": round(t_scan * 1e3, 1), "upload": round(t_up * 1e3, 1)}}
        if split:
            r["host_ms"]["distribute"] = round(t_dist * 1e3, 1)
            r["split_info"] = split_info
        # the same K steps strictly one after the other (reported beside `value`, never instead of it)
        r["info"] = batch.info()
        alone = batch                 # the batch that is decoded ALONE: planned for latency
        if nfl > 1:
            alone_mode = mode_of.get(args.plan_mode, pjd_amd.PLAN_LATENCY)
            if alone_mode != main_mode:
                ctxs.append(pjd_amd.Context(local_rank, plan_mode=alone_mode))
                alone = ctxs[-1].batch(descs, out_fmt)
                batches.append(alone)
                alone.upload()
                if not args.no_graph:
                    alone.capture()
                run_steps(2, [alone])
            barrier()
            t0 = time.perf_counter()
            run_steps(steps, [alone])
            barrier()
            r["dt_serial"] = time.perf_counter() - t0
            r["info_alone"] = alone.info()
        # whole-job totals: ranks decode different seeded batches, so sum what each one really processed per step
        tot = [float(r["info"]["pixels"]), float(r["info"]["ecs_bytes"]), float(r["info"]["n_entries"])]
        if world > 1 and not split:
            tt = torch.tensor(tot, dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            tot = [float(x) for x in tt.tolist()]
        r["tot"] = tot
        if full:
            # per-kernel durations, HIP events on the library's own stream (ungraphed launches of the same work)
            ktimes, ktotal, reps = {}, 0.0, 5
            for _ in range(reps):
                kt, tot = alone.decode_timed()
                alone.sync()
                for k, v in kt.items():
                    ktimes[k] = ktimes.get(k, 0.0) + v / reps
                ktotal += tot / reps
            r["ktimes"], r["ktotal"] = ktimes, ktotal
            if args.verify and rank == 0:
                import numpy as np
                import oracle_lib
                port = oracle_lib.Port()
                outs, st = batch.download()
                idx = list(range(min(4, len(jpegs))))
                if out_fmt == pjd_amd.OUT_BMP:
                    r["verify"] = all(outs[i].tobytes() == port.decode(jpegs[i])["bmp"] for i in idx)
                else:
                    r["verify"] = all(np.array_equal(outs[i], port.decode(jpegs[i])["rgb"]) for i in idx)
        for b in batches:
            b.destroy()
        for c in ctxs:
            c.close()
        return r

    def rates(r):
        """Throughput figures of one measurement (whole job: all ranks)."""
        info, dt, k = r["info"], r["dt"], r["steps"]
        mult = 1 if r.get("split") else world        # a split picture is counted once (every rank's plan names the whole picture)
        pix, ecs, ent = r["tot"]                     # summed over ranks (a split picture: rank 0's plan names the whole picture)
        if r.get("split"):
            ecs, ent = world * ecs, world * ent      # every rank holds and decodes its own slice (approximately equal shares)
        o = {"value": round(pix * k / dt / 1e6, 2), "unit": "MPix/s", "ms_per_step": round(dt / k * 1e3, 4),
             "ecs_GBps": round(ecs * k / dt / 1e9, 2),
             "huffman_symbols_per_s": round(ent * k / dt, 0),
             "bytes_per_pixel": round(info["ecs_bytes"] / info["pixels"], 3), "table_sets": info["n_table_sets"],
             "huffman_lanes": info["n_subsequences"], "sub_bytes": info["sub_bytes"],
             "exact_kernel_images": info["n_sequential"] + info["n_fallback"]}
        if "dt_serial" in r:
            ia = r["info_alone"]
            o["one_batch_in_flight"] = {"value": round(mult * info["pixels"] * k / r["dt_serial"] / 1e6, 2), "unit": "MPix/s",
                                        "ms_per_step": round(r["dt_serial"] / k * 1e3, 4),
                                        "plan_mode": "throughput" if ia["plan_mode"] else "latency",
                                        "huffman_lanes": ia["n_subsequences"], "sub_bytes": ia["sub_bytes"], "huffman_waves": ia["n_huff_waves"],
                                        "lane_passes_per_lane": round(2.0 + ia["sync_lane_passes"] / max(1, ia["n_subsequences"]), 3)}
        return o

    R = measure(args.workload, args.steps, args.warmup, True)
    variants = {}
    if args.workload == "cfg3" and not args.no_variants:
        # the lighter round-1 set, same run (Annex-K tables shared by every picture, 0.32 B/px)
        V = measure("cfg3lite", max(5, args.steps // 2), args.warmup, False)
        variants["cfg3lite"] = dict(rates(V), workload=V["label"])
        if world > 1:
            # BASELINE config 5 beside the batches: ONE picture split by restart segment over the ranks -- the descriptor broadcast and
            # the slices travel through the process group (RCCL).  Strong scaling; never `value` of this line.
            S = measure("cfg5split", max(5, args.steps // 4), args.warmup, False)
            variants["cfg5split"] = dict(rates(S), workload=S["label"], scaling="strong", split=S.get("split_info"), host_ms=S["host_ms"])

    if rank == 0:
        info, ktimes = R["info"], R["ktimes"]
        main_rates = rates(R)
        pixels = info["pixels"]
        dom = max(ktimes, key=ktimes.get)
        alg_bytes = info["ecs_bytes"] + info["out_bytes"]            # ECS read once + picture written once
        achieved = alg_bytes / (ktimes[dom] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel per launch and the VALU issue fraction, from the committed rocprofv3 PMC passes
        # (profiles/<tag>_counters.json, written by tools/make_profile.py); only valid for the workload they were collected on
        traffic, valu_issue_frac, prof_src = None, None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r04_cfg3_counters.json")))
            if args.workload == tj.get("workload_key") and args.images == 1024:
                prof_src = tj.get("source")
                for k, v in tj["per_kernel"].items():
                    if dom in k:
                        traffic = v.get("traffic_bytes")
                        valu_issue_frac = v.get("valu_issue_frac")
        except Exception:
            pass
        line = {
            "metric": "MPixels/sec JPEG->RGB (bit-exact BMP)", "value": main_rates["value"], "unit": "MPix/s",
            "value_mode": (f"{nfl} batches in flight (each resident batch on its own HIP stream; every step drained and checked before its batch is "
                           "decoded again; planned with PJD_PLAN_THROUGHPUT); one_batch_in_flight = the same steps strictly serialised on ONE "
                           "batch planned with PJD_PLAN_LATENCY (the library's default); --plan-mode forces one plan for both") if nfl > 1 else "one batch at a time",
            "n_gpus": world, "rccl_ranks": comm["ranks"], "collective_backend": comm["backend"], "steps": args.steps, "warmup": max(args.warmup, nfl),
            "ms_per_step": main_rates["ms_per_step"], "higher_is_better": True, "scaling": "strong" if R.get("split") else "weak",
            "vs_baseline": None, "dtype": "int16/int32 (integer IDCT), u8 out", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {R['label']}", "out_format": args.out_format,
                       "images_per_gpu": info["n_images"], "pixels_per_gpu": pixels, "ecs_bytes_per_gpu": info["ecs_bytes"],
                       "bytes_per_pixel": main_rates["bytes_per_pixel"], "table_sets": info["n_table_sets"],
                       "huffman_lanes": info["n_subsequences"], "sub_bytes": info["sub_bytes"],
                       "device_bytes_per_batch": int(info["device_bytes"]), "lane_stream_capacity_bytes": int(info["coef_bytes"]),
                       "lane_stream_bytes_used": int(info["n_steps"] * 4 * 8 // 7), "fullest_lane_region": round(info["lane_fill_x1024"] / 1024.0, 3),
                       "exact_kernel_images": main_rates["exact_kernel_images"],
                       "hip_graph": not args.no_graph, "batches_in_flight": nfl, "host_nproc": os.cpu_count(),
                       "plan_mode": "throughput" if info["plan_mode"] else "latency",
                       "sync": dict({k: info[k] for k in ("n_huff_workgroups", "n_huff_waves", "sync_rounds", "sync_lane_passes", "fix_rounds", "fix_lane_passes", "walks", "walk_lanes")},
                                    lane_passes_per_lane=round(2.0 + info["sync_lane_passes"] / max(1, info["n_subsequences"]), 3),
                                    symbols=int(info["n_entries"]), write_steps=int(info["n_steps"]),
                                    symbols_per_step=round(info["n_entries"] / max(1, info["n_steps"]), 3),
                                    note="lane_passes_per_lane = speculative pass + re-sync passes + write pass, per lane (work); a wave's "
                                         "re-sync round lasts as long as its slowest lane; walks = rounds a wave finished cooperatively "
                                         "(few lanes left), walk_lanes = lanes re-decoded that way (counted in sync_lane_passes too)")},
            "ecs_GBps": main_rates["ecs_GBps"], "huffman_symbols_per_s": main_rates["huffman_symbols_per_s"],
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_issue_frac": valu_issue_frac,
                         "counters_from": prof_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(ktimes[dom], 4),
                         "plan_mode": "throughput" if R.get("info_alone", info)["plan_mode"] else "latency"},
            "kernels_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "kernel_pipeline_ms": round(R["ktotal"], 4),
            "host_ms": R["host_ms"],
        }
        if "one_batch_in_flight" in main_rates:
            serial_ms = main_rates["one_batch_in_flight"]["ms_per_step"]
            line["one_batch_in_flight"] = dict(main_rates["one_batch_in_flight"],
                                               note="rank 0's clock, steps strictly one after the other on one stream: BASELINE config 3 as written "
                                                    "(one batch, one graph); `roofline` (serialised launches of the dominant kernel) belongs to this figure",
                                               roofline={"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                                                         "whole_step_achieved": round(alg_bytes / (serial_ms * 1e-3) / 1e9, 2),
                                                         "whole_step_frac": round(alg_bytes / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
        if variants:
            line["variants"] = variants
        if R.get("split_info"):
            line["split"] = R["split_info"]
        if world > 1 and (args.workload == "cfg5split" or (args.workload == "cfg3" and not args.no_variants)):
            # the C++ route to the same thing (pjd_split_decode: host threads + ONE ncclBroadcast), over the node's devices, in a child
            rehearsal = "PJD_BENCH_DEVICE" in os.environ          # one-GPU rehearsal: every "rank" is that device (RCCL refuses duplicates: host copies)
            if rehearsal:
                os.environ["PJD_PIPE_ALLOW_DUP_DEVICES"] = "1"
            line["cxx_split_decode"] = cxx_split_child(min(args.tile, 8192), 3, [local_rank] * world if rehearsal else range(world))
        if R.get("verify") is not None:
            line["verified_against_oracle"] = bool(R["verify"])
        jpegs = R["jpegs"]
        if world == 1 and args.e2e_batches > 0:
            # PCIe-inclusive rate: host scan + H2D + kernels + D2H, all overlapped by libpjdpipe (3 GPU slots)
            pipe_jpegs = jpegs * args.e2e_batches
            # warm-up: every slot allocates its HBM pool and page-locks its output buffer once
            pjd_amd.pipe_run(jpegs=jpegs * 8, batch_images=len(jpegs), scan_threads=8, slots=3, sink=None, device=local_rank)
            ps = pjd_amd.pipe_run(jpegs=pipe_jpegs, batch_images=len(jpegs), scan_threads=8, slots=3, sink=None, device=local_rank)
            pjd_amd.pipe_release()
            line["pcie_inclusive"] = {
                "value": round(ps["pixels"] / ps["wall_s"] / 1e6, 2), "unit": "MPix/s", "out_format": "bmp",
                "inputs": ps["n_inputs"], "batches": ps["n_batches"], "wall_ms": round(ps["wall_s"] * 1e3, 2),
                "d2h_GBps": round(ps["out_bytes"] / ps["wall_s"] / 1e9, 2), "exact_kernel_images": ps["n_exact_images"],
                "worker_ms": {k[:-2]: round(ps[k] * 1e3, 1) for k in ("scan_s", "create_s", "upload_s", "exec_s", "download_s")},
                "note": "JPEG bytes in host memory -> BMP bytes in page-locked host memory; pictures are not consumed further"}
        if world == 1 and not args.no_cli and args.workload in ("cfg3", "cfg3lite"):
            line["cli_end_to_end"] = cli_end_to_end(jpegs, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(jpegs)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
