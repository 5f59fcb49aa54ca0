#!/usr/bin/env python3
"""GPU box: random pictures with restart intervals, decoded in shards (config-5 mechanics) and through the pipelined batcher.

    python tools/fuzz_shards.py [--n 200] [--seed 1]

Part 1: every picture (4:4:4 or grey, random DRI) is split over a random number of ranks by restart segment; each shard is
decoded as its own batch from its slice of the bitstream; the union of the MCUs the shards own must equal the oracle's picture.
Part 2: all pictures (plus subsampled ones without DRI) go through libpjdpipe with random batch sizes / slots; BMP bytes must
equal the oracle's.
"""
import argparse
import hashlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import oracle_lib
    import pjd_amd
    import synth
    from pjd_amd import parallel
    port = oracle_lib.Port()
    ctx = pjd_amd.Context(0)
    rng = np.random.default_rng(a.seed)
    bad = 0
    jpegs = []
    for k in range(a.n):
        w, h = int(rng.integers(16, 500)), int(rng.integers(16, 500))
        sub = int(rng.choice([synth.SUB_444, synth.SUB_444, synth.SUB_GREY]))
        mcux = (w + 7) // 8
        ri = int(rng.choice([1, 3, mcux, 2 * mcux, 5 * mcux + 1]))
        jp = synth.make(w, h, 777_000 * a.seed + k, int(rng.choice([50, 85, 97])), sub, ri, float(rng.choice([1.0, synth.DENSE_DETAIL])), bool(k & 1))
        jpegs.append(jp)
        s = pjd_amd.Scanned(jp)
        segs, ecs = s.seg_offsets(), s.ecs()
        want = port.decode(jp)["rgb"]
        got = np.zeros_like(want)
        d0 = s.desc
        n_mcu = mcux * ((h + 7) // 8)
        world = int(rng.integers(1, 9))
        for r in range(world):
            f, c = parallel.segment_range(len(segs), r, world)
            if c == 0:
                continue
            lo = int(segs[f]); hi = int(segs[f + c]) if f + c < len(segs) else len(ecs)
            d, keep = parallel.shard_descriptor(d0, segs, ecs[lo:hi], lo, r, world)
            outs, st = ctx.decode([d], pjd_amd.OUT_RGB8)
            if st != [0]:
                bad += 1; print(f"SHARD STATUS picture {k} rank {r}/{world}: {st}", flush=True)
            for m in range(f * ri, min((f + c) * ri, n_mcu)):
                y0, x0 = (m // mcux) * 8, (m % mcux) * 8
                got[y0:y0 + 8, x0:x0 + 8] = outs[0][y0:y0 + 8, x0:x0 + 8]
        if not np.array_equal(got, want):
            bad += 1; print(f"SHARD MISMATCH picture {k}: {w}x{h} RI {ri} world {world} segments {len(segs)}", flush=True)
        if k % 50 == 49:
            print(f"shards: {k + 1} pictures, bad {bad}", flush=True)
    # part 2: the batcher
    extra = [synth.make(int(rng.integers(8, 400)), int(rng.integers(8, 400)), 999_000 + k, 90, int(rng.choice([synth.SUB_420, synth.SUB_422, synth.SUB_440])), 0,
                        1.0, bool(k & 1)) for k in range(a.n // 2)]
    allj = jpegs + extra
    want_sha = [hashlib.sha256(port.decode(j)["bmp"]).hexdigest() for j in allj]
    for trial in range(3):
        got, lock = {}, threading.Lock()

        def sink(index, name, log, status, data):
            with lock:
                got[index] = (status, None if data is None else hashlib.sha256(data.tobytes()).hexdigest())
        bs, slots = int(rng.integers(1, 64)), int(rng.integers(1, 5))
        st = pjd_amd.pipe_run(jpegs=allj, batch_images=bs, scan_threads=4, slots=slots, sink_threads=3, sink=sink)
        wrong = [i for i in range(len(allj)) if got.get(i, (None, None)) != (0, want_sha[i])]
        print(f"pipeline trial {trial}: batch {bs}, slots {slots}: {st['n_batches']} batches, {len(wrong)} wrong, exact kernel {st['n_exact_images']}", flush=True)
        bad += len(wrong)
    pjd_amd.pipe_release()
    ctx.close()
    print(f"fuzz shards+pipeline: {a.n} sharded pictures, {len(allj)} pipelined x3, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
