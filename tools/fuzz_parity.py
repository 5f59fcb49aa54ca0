#!/usr/bin/env python3
"""GPU box: seeded random pictures through the C ABI against the oracle, many more than the test suite runs.

    python tools/fuzz_parity.py [--n 3000] [--seed 1] [--batch 250]

Every sampling mode, qualities 5..100, sizes 1..700 px per side, restart intervals, default and optimised Huffman tables,
plain and detailed pictures; BMP and RGB8 output alternate per batch.  Prints the first mismatches and a summary line.
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


REASONS = ["symbol", "segment", "nosync", "stitch", "timeout", "overflow", "verify", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--corrupt", type=int, default=0, help="overwrite this many random bytes in the second half of every file: error paths "
                    "(scanner accept/reject set, Huffman error classes, partial pictures) against the oracle")
    a = ap.parse_args()
    import oracle_lib
    import pjd_amd
    import synth
    port = oracle_lib.Port()
    ctx = pjd_amd.Context(0)
    rng = np.random.default_rng(a.seed)
    bad = fb = seq = 0
    for b0 in range(0, a.n, a.batch):
        jpegs, flags, plain = [], [], []
        for k in range(b0, min(a.n, b0 + a.batch)):
            big = rng.random() < 0.15
            w, h = (int(rng.integers(200, 701)), int(rng.integers(200, 701))) if big else (int(rng.integers(1, 201)), int(rng.integers(1, 201)))
            sub = int(rng.choice([synth.SUB_444, synth.SUB_422, synth.SUB_420, synth.SUB_440, synth.SUB_GREY]))
            q = int(rng.choice([5, 25, 50, 75, 90, 97, 100]))
            hs = 2 if sub in (synth.SUB_422, synth.SUB_420) else 1
            mcux = (w + 8 * hs - 1) // (8 * hs)
            ri = int(rng.choice([0, 0, 0, 1, 2, 7, mcux, 3 * mcux]))
            detail = float(rng.choice([1.0, 1.0, synth.DENSE_DETAIL]))
            opt = bool(rng.random() < 0.5)
            jp = synth.make(w, h, 10_000 * a.seed + k, q, sub, ri, detail, opt)
            if a.corrupt:
                ba = bytearray(jp)
                for _ in range(a.corrupt):
                    ba[int(rng.integers(len(ba) // 2, len(ba) - 2))] = int(rng.integers(0, 256))
                jp = bytes(ba)
            jpegs.append(jp)
            std = (not a.corrupt) and ri != 0 and sub in (synth.SUB_422, synth.SUB_420, synth.SUB_440) and k % 2 == 0
            flags.append(pjd_amd.F_STANDARD_RESTART if std else 0)
            plain.append(synth.make(w, h, 10_000 * a.seed + k, q, sub, 0, detail, opt) if std else None)
        scanned = [pjd_amd.Scanned(j) for j in jpegs]
        if a.corrupt:                                          # the scanners must agree on what is a JPEG; the rest goes on
            keep = []
            for i, (j, s) in enumerate(zip(jpegs, scanned)):
                ov = bool(port.parse(j)["info"]["valid"])
                if ov != bool(s.valid):
                    bad += 1
                    print(f"SCANNER MISMATCH picture {b0 + i}: oracle valid {ov}, scanner valid {s.valid}", flush=True)
                elif ov:
                    keep.append(i)
            rejected = len(jpegs) - len(keep)
            jpegs = [jpegs[i] for i in keep]; scanned = [scanned[i] for i in keep]; flags = [flags[i] for i in keep]; plain = [plain[i] for i in keep]
            print(f"    corrupted: {rejected} rejected by both scanners, {len(keep)} decoded", flush=True)
        for s, f in zip(scanned, flags):
            s.desc.flags = f
        fmt = pjd_amd.OUT_BMP if (b0 // a.batch) % 2 else pjd_amd.OUT_RGB8
        with ctx.batch([s.desc for s in scanned], fmt) as b:
            b.upload(); b.decode()
            outs, st = b.download()
            info = b.info()
        fb += info["n_fallback"]; seq += info["n_sequential"]

        def check(i):
            want = port.decode(plain[i] if plain[i] is not None else jpegs[i])
            ok = st[i] == want["huff_rc"] and (outs[i].tobytes() == want["bmp"] if fmt == pjd_amd.OUT_BMP else np.array_equal(outs[i], want["rgb"]))
            return ok
        with ThreadPoolExecutor(8) as ex:
            res = list(ex.map(check, range(len(jpegs))))
        for i, ok in enumerate(res):
            if not ok:
                bad += 1
                if bad <= 10:
                    d = scanned[i].desc
                    print(f"MISMATCH picture {b0 + i}: {d.width}x{d.height} comps {d.num_components} samp {d.h_samp}x{d.v_samp} RI {d.restart_interval} flags {flags[i]} len {len(jpegs[i])}", flush=True)
        why = ",".join(f"{REASONS[r]}:{v}" for r, v in enumerate(info["flag_waves"]) if v)
        print(f"batch at {b0}: {len(jpegs)} pictures, bad so far {bad}, exact kernel {info['n_sequential']} routed + {info['n_fallback']} flagged"
              f"{' (' + why + ')' if why else ''}, lanes {info['n_subsequences']}, S {info['sub_bytes']}", flush=True)
        if info["n_fallback"]:
            with ctx.batch([s.desc for s in scanned], fmt) as b2:          # which pictures: decode them one by one
                pass
            for i, s in enumerate(scanned):
                with ctx.batch([s.desc], fmt) as b1:
                    b1.upload(); b1.decode(); b1.download()
                    i1 = b1.info()
                if i1["n_fallback"]:
                    d = s.desc
                    w1 = ",".join(f"{REASONS[r]}:{v}" for r, v in enumerate(i1["flag_waves"]) if v)
                    print(f"    alone, picture {b0 + i} also falls back ({w1}): {d.width}x{d.height} samp {d.h_samp}x{d.v_samp} RI {d.restart_interval} ecs {d.ecs_len} S {i1['sub_bytes']}", flush=True)
    ctx.close()
    print(f"fuzz: {a.n} pictures, {bad} mismatches, {seq} routed to the exact kernel, {fb} flagged by the parallel decoder")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
