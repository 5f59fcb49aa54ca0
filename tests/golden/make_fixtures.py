#!/usr/bin/env python3
"""Generate tests/golden/*.jpg and tests/golden/manifest.json.

Run HERE (the build container, where /root/reference exists and oracle/_ref can
be built).  JPEG bytes come from Pillow (libjpeg-turbo) on seeded synthetic
pictures plus a few byte-level edits; the EXPECTED values in the manifest are
produced by oracle/_ref, i.e. by the reference's own read_JPEG /
decode_Huffman_data / write_BMP compiled in place, with oracle/dpu_stages.c for
the device stage (pinned by the SURVEY section 0.4 hash, fixture
`ilsvrc_val_00000001`).  The JPEG bytes are committed because encoder output
varies between libjpeg builds.

    python tests/golden/make_fixtures.py
"""
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

REFERENCE_SAMPLE = "/root/reference/ILSVRC2012_val_00000001.JPEG"


def picture(w, h, seed, kind="smooth"):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    if kind == "smooth":
        r = 127 + 100 * np.sin(xx / 9.0 + seed) * np.cos(yy / 13.0)
        g = 127 + 90 * np.cos(xx / 17.0) + 30 * np.sin(yy / 5.0 + seed)
        b = (xx * 255 / max(w - 1, 1) + yy * 255 / max(h - 1, 1)) / 2
        img = np.stack([r, g, b], -1) + rng.normal(0, 6, (h, w, 3))
    elif kind == "noise":          # saturated noise: stresses clamps and long codes
        img = rng.integers(0, 2, (h, w, 3)) * 255.0
    elif kind == "flat":
        img = np.full((h, w, 3), 200.0) + rng.normal(0, 0.4, (h, w, 3))
    else:
        raise ValueError(kind)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")


def enc(img, **kw):
    bio = io.BytesIO()
    img.save(bio, "JPEG", **kw)
    return bio.getvalue()


def find_marker(data, m):
    i = data.find(bytes([0xFF, m]))
    assert i >= 0
    return i


def sos_end(data):
    i = find_marker(data, 0xDA)
    return i + 2 + ((data[i + 2] << 8) | data[i + 3])


def rewrite_dqt(d, value_of, precision16=True):
    """Rewrite every DQT segment of `d`: entry k (zigzag order) of table `tid` becomes value_of(tid, k, old), written as a
    16-bit (precision 1) or 8-bit table."""
    out = bytearray(d[:2])
    p = 2
    while True:
        assert d[p] == 0xFF
        m = d[p + 1]
        ln = (d[p + 2] << 8) | d[p + 3]
        if m == 0xDB:
            body = d[p + 4:p + 2 + ln]
            new = bytearray()
            q = 0
            while q < len(body):
                assert body[q] >> 4 == 0
                tid = body[q] & 15
                new.append((0x10 if precision16 else 0x00) | tid)
                for k, v in enumerate(body[q + 1:q + 65]):
                    nv = int(value_of(tid, k, v))
                    new += bytes([nv >> 8, nv & 255]) if precision16 else bytes([nv & 255])
                q += 65
            out += bytes([0xFF, 0xDB, (len(new) + 2) >> 8, (len(new) + 2) & 255]) + new
        else:
            out += d[p:p + 2 + ln]
        p += 2 + ln
        if m == 0xDA:
            break
    out += d[p:]
    return bytes(out)


def relabel_h1v2(d):
    """A 4:2:2 (h2v1) stream with swapped dimensions is a valid 4:4:0 (h1v2) stream (Pillow cannot emit that mode)."""
    d = bytearray(d)
    i = find_marker(d, 0xC0)
    hh, ww = (d[i + 5] << 8) | d[i + 6], (d[i + 7] << 8) | d[i + 8]
    d[i + 5:i + 9] = bytes([ww >> 8, ww & 255, hh >> 8, hh & 255])
    assert d[i + 11] == 0x21
    d[i + 11] = 0x12
    return bytes(d)


def build_set():
    S = {}
    # --- the parity envelope ---------------------------------
    k = 0
    for (w, h) in [(64, 48), (61, 45), (72, 40), (128, 96), (17, 9), (8, 8), (1, 1), (200, 150)]:
        for sub, tag in [(0, "444"), (1, "422"), (2, "420")]:
            q = [85, 30, 100][k % 3]
            opt = bool(k % 2)
            k += 1
            S[f"env_{w}x{h}_{tag}_q{q}{'_opt' if opt else ''}"] = enc(picture(w, h, k), quality=q, subsampling=sub, optimize=opt)
    for (w, h) in [(64, 48), (61, 45), (33, 70)]:
        S[f"gray_{w}x{h}"] = enc(picture(w, h, 90 + w).convert("L"), quality=80)
    S["noise_96x80_444_q100"] = enc(picture(96, 80, 7, "noise"), quality=100, subsampling=0)
    S["noise_96x80_420_q95"] = enc(picture(96, 80, 8, "noise"), quality=95, subsampling=2)
    S["noise_80x96_422_q50_opt"] = enc(picture(80, 96, 9, "noise"), quality=50, subsampling=1, optimize=True)
    S["flat_120x88_420_q90"] = enc(picture(120, 88, 10, "flat"), quality=90, subsampling=2)
    S["big_640x480_420_q85"] = enc(picture(640, 480, 11), quality=85, subsampling=2)
    S["big_500x375_444_q92_opt"] = enc(picture(500, 375, 12), quality=92, subsampling=0, optimize=True)
    S["wide_1200x64_420_q75"] = enc(picture(1200, 64, 13), quality=75, subsampling=2)
    S["tall_40x900_422_q75"] = enc(picture(40, 900, 14), quality=75, subsampling=1)
    # restart intervals where the reference handles them (luma 1x1)
    S["rst4_128x96_444"] = enc(picture(128, 96, 20), quality=85, subsampling=0, restart_marker_blocks=4)
    S["rst1_61x45_444"] = enc(picture(61, 45, 21), quality=70, subsampling=0, restart_marker_blocks=1)
    S["rstrow_200x150_444_opt"] = enc(picture(200, 150, 22), quality=90, subsampling=0, restart_marker_rows=1, optimize=True)
    S["rstrow_gray_100x60"] = enc(picture(100, 60, 23).convert("L"), quality=85, restart_marker_rows=1)
    S["rst7_gray_61x45"] = enc(picture(61, 45, 24).convert("L"), quality=60, restart_marker_blocks=7)
    # 4:4:0 (h1v2): Pillow cannot emit it; re-label a 4:2:2 stream with swapped
    # dimensions -- same data units per MCU, same MCU count, a valid h1v2 stream.
    for (w, h, seed) in [(64, 48, 30), (61, 45, 31), (72, 104, 32)]:
        d = bytearray(enc(picture(w, h, seed), quality=85, subsampling=1))
        i = find_marker(d, 0xC0)
        hh, ww = (d[i + 5] << 8) | d[i + 6], (d[i + 7] << 8) | d[i + 8]
        d[i + 5:i + 9] = bytes([ww >> 8, ww & 255, hh >> 8, hh & 255])
        assert d[i + 11] == 0x21
        d[i + 11] = 0x12
        S[f"h1v2_{h}x{w}"] = bytes(d)
    # 2-component frame (accepted by the reference): drop Cr from a 4:4:4 header is not
    # a valid stream; skip.  Zero-based component ids: patch ids 1,2,3 -> 0,1,2.
    d = bytearray(enc(picture(48, 32, 40), quality=85, subsampling=2))
    i = find_marker(d, 0xC0)
    for c in range(3):
        d[i + 10 + 3 * c] -= 1
    j = find_marker(d, 0xDA)
    for c in range(3):
        d[j + 5 + 2 * c] -= 1
    S["zero_based_ids_48x32_420"] = bytes(d)
    # 16-bit quantisation table: rewrite the DQT segments as precision-1 tables
    d = enc(picture(64, 48, 41), quality=40, subsampling=0)
    out = bytearray()
    p = 2
    out += d[:2]
    while True:
        assert d[p] == 0xFF
        m = d[p + 1]
        ln = (d[p + 2] << 8) | d[p + 3]
        if m == 0xDB:
            body = d[p + 4:p + 2 + ln]
            new = bytearray()
            q = 0
            while q < len(body):
                tid = body[q] & 15
                new.append(0x10 | tid)
                for v in body[q + 1:q + 65]:
                    new += bytes([0, v])
                q += 65
            out += bytes([0xFF, 0xDB, (len(new) + 2) >> 8, (len(new) + 2) & 255]) + new
        else:
            out += d[p:p + 2 + ln]
        p += 2 + ln
        if m == 0xDA:
            break
    out += d[p:]
    S["dqt16_64x48_444"] = bytes(out)

    # --- int16 wrap through the WHOLE path (round 3): saturated noise at q100 (coefficients up to +-1000) with the quantisation
    #     tables replaced by huge entries, so that (int16)(coef * Q) wraps in dequantisation (reference src/decoder_dpu.c:169-172),
    #     both IDCT passes store wrapped int16 (:218-320) and |Cb|, |Cr| leave the range where the colour products fit 32 bits
    #     (:376-382).  All four sampling modes + grey; 16-bit tables with 65535 / 40000 / mixed entries and 8-bit tables of 255.
    wrap_q = {"q65535": (True, lambda t, k, v: 65535), "q40000": (True, lambda t, k, v: 40000 if (k + t) % 3 else 65535 - 7 * k),
              "q255": (False, lambda t, k, v: 255)}
    wk = 0
    for tag, sub in [("444", 0), ("422", 1), ("420", 2), ("440", 1), ("gray", None)]:
        for qn, (p16, fn) in wrap_q.items():
            wk += 1
            w, h = (88, 56) if tag != "440" else (56, 88)
            img = picture(w, h, 300 + wk, "noise")
            if sub is None:
                d = enc(img.convert("L"), quality=100)
            else:
                d = enc(img, quality=100, subsampling=sub)
            if tag == "440":
                d = relabel_h1v2(d)
            S[f"wrap_{tag}_{qn}"] = rewrite_dqt(d, fn, p16)

    # --- negative / divergent cases ---------------------------------------------
    S["neg_progressive_64x48"] = enc(picture(64, 48, 50), quality=85, progressive=True)
    S["div_rst_420_64x48"] = enc(picture(64, 48, 51), quality=85, subsampling=2, restart_marker_blocks=2)
    S["div_rst_422_61x45"] = enc(picture(61, 45, 52), quality=85, subsampling=1, restart_marker_rows=1)
    good = enc(picture(96, 64, 53), quality=85, subsampling=2)
    e = sos_end(good)
    S["neg_truncated_noeoi"] = good[: e + (len(good) - e) // 2]
    S["err_truncated_eoi_420"] = good[: e + (len(good) - e) // 2] + b"\xff\xd9"
    good4 = enc(picture(96, 64, 54), quality=90, subsampling=0)
    e4 = sos_end(good4)
    S["err_truncated_eoi_444"] = good4[: e4 + (len(good4) - e4) // 3] + b"\xff\xd9"
    rng = np.random.default_rng(55)
    for n in range(4):
        d = bytearray(good if n % 2 else good4)
        e0 = sos_end(bytes(d))
        for _ in range(3):
            pos = int(rng.integers(e0 + 8, len(d) - 4))
            v = int(rng.integers(0, 255))
            if v == 0xFF or d[pos] == 0xFF or d[pos - 1] == 0xFF:
                continue
            d[pos] = v
        S[f"err_corrupt_{n}"] = bytes(d)
    S["neg_not_jpeg"] = b"BM" + bytes(64)
    S["neg_empty"] = b""
    cmyk = io.BytesIO()
    Image.new("CMYK", (16, 16), (10, 20, 30, 40)).save(cmyk, "JPEG")
    S["neg_cmyk"] = cmyk.getvalue()
    d = bytearray(enc(picture(32, 32, 56), quality=85, subsampling=2))
    i = find_marker(d, 0xC0)
    d[i + 11] = 0x41          # luma sampling 4x1: unsupported
    S["neg_sampling_41"] = bytes(d)
    d = bytearray(enc(picture(32, 32, 57), quality=85, subsampling=0))
    i = find_marker(d, 0xC0)
    d[i + 4] = 12             # 12-bit precision
    S["neg_precision12"] = bytes(d)
    # trailing garbage after a complete scan is ignored by the reference
    S["tail_garbage_64x48_420"] = enc(picture(64, 48, 58), quality=85, subsampling=2)[:-2] + b"\xff\xd9" + bytes(range(40))
    # fill bytes (FF FF) before a marker inside the scan, and a COM segment
    d = enc(picture(64, 48, 59), quality=85, subsampling=0, restart_marker_blocks=6)
    e0 = sos_end(d)
    k = d.find(b"\xff\xd0", e0)
    S["fill_ff_before_rst_444"] = d[:k] + b"\xff\xff" + d[k:]
    # --- odd Huffman tables (the scan then decodes to whatever the reference makes of it) ---------
    base = enc(picture(96, 64, 61), quality=90, subsampling=0)
    ac_syms = [s for s in range(256) if (s & 15) <= 10 and (s & 15 or s in (0x00, 0xF0))]      # the 162 baseline AC symbols
    # 3 short codes + 159 codes of 11 bits in BOTH AC tables: 80 ten-bit prefixes with long codes per table,
    # more second-level tables than the parallel decoder keeps in LDS -> exact kernel
    counts = [1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 159, 0, 0, 0, 0, 0]
    S["huff_longtail_96x64_444"] = replace_dht(base, {(1, 0): (counts, ac_syms), (1, 1): (counts, ac_syms[:3] + ac_syms[:2:-1])})
    # over-subscribed table: three codes of one bit (reference generate_codes just keeps counting)
    counts = [3, 1, 2, 4] + [0] * 12
    S["huff_oversub_96x64_444"] = replace_dht(base, {(1, 0): (counts, ac_syms[:10])})
    return S


def replace_dht(data, new):
    """Rewrite the DHT segments: `new` maps (table class, table id) -> (16 counts, symbols); others are kept."""
    out, i = bytearray(data[:2]), 2
    while i < len(data):
        assert data[i] == 0xFF
        m = data[i + 1]
        if m == 0xDA:
            out += data[i:]
            break
        n = (data[i + 2] << 8) | data[i + 3]
        seg = data[i + 4:i + 2 + n]
        if m == 0xC4:
            j, body = 0, bytearray()
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                cnt = list(seg[j + 1:j + 17])
                syms = list(seg[j + 17:j + 17 + sum(cnt)])
                j += 17 + sum(cnt)
                if (tc, th) in new:
                    cnt, syms = new[(tc, th)]
                    assert len(cnt) == 16 and sum(cnt) == len(syms)
                body += bytes([(tc << 4) | th]) + bytes(cnt) + bytes(syms)
            out += bytes([0xFF, 0xC4, (len(body) + 2) >> 8, (len(body) + 2) & 255]) + body
        else:
            out += data[i:i + 2 + n]
        i += 2 + n
    return bytes(out)


def main():
    oracle_lib.build_oracle()
    assert oracle_lib.Ref.available(), "oracle/_ref must be buildable where fixtures are generated"
    ref = oracle_lib.Ref()
    S = build_set()
    with open(REFERENCE_SAMPLE, "rb") as f:
        S["ilsvrc_val_00000001"] = f.read()     # the reference's bundled sample (data, 109,527 B)

    # files already committed are kept byte for byte (encoder output varies between libjpeg builds); only new names are written
    for fn in os.listdir(HERE):
        if fn.endswith(".jpg") and fn[:-4] not in S:
            os.remove(os.path.join(HERE, fn))
    for name in sorted(S):
        fp = os.path.join(HERE, name + ".jpg")
        if os.path.exists(fp):
            S[name] = open(fp, "rb").read()
    manifest = {}
    tmp = tempfile.mkdtemp()
    try:
        for name in sorted(S):
            data = S[name]
            with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
                f.write(data)
            jp = os.path.join(tmp, name + ".jpg")
            bp = os.path.join(tmp, name + ".bmp")
            shutil.copy(os.path.join(HERE, name + ".jpg"), jp)
            rc, out = ref.run_cli(jp, bp)
            ent = {"bytes": len(data), "rc": rc, "stdout": out.replace(jp, "{path}")}
            if rc == 0:
                bmp = open(bp, "rb").read()
                ent["bmp_len"] = len(bmp)
                ent["bmp_sha256"] = hashlib.sha256(bmp).hexdigest()
                r = ref.parse_and_huffman(jp)
                ent["huff_ok"] = r["huff_ok"]
                ent["coef_sha256"] = hashlib.sha256(r["coef"].tobytes()).hexdigest()
                ent["ecs_sha256"] = hashlib.sha256(r["ecs"].tobytes()).hexdigest()
                i = r["info"]
                ent["dims"] = [i["width"], i["height"], i["ncomp"], i["hsamp"], i["vsamp"], i["restart_interval"]]
            manifest[name] = ent
    finally:
        shutil.rmtree(tmp)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    ok = sum(1 for e in manifest.values() if e["rc"] == 0)
    print(f"{len(manifest)} fixtures, {ok} decoded, total {sum(len(v) for v in S.values())} bytes")


if __name__ == "__main__":
    main()
