"""Seeded synthetic JPEG workloads (bench / test tooling; wraps tools/jpeg_synth.c).

    cfg2_single_4k()      one 3840x2160 4:2:0 q85 picture                  (BASELINE config 2)
    cfg3_imagenet_like(n) n mixed-size 4:2:0 pictures, ImageNet-like sizes (BASELINE config 3/4)
    cfg5_tile(size)       one size x size 4:4:4 picture, one restart interval per MCU row (config 5)
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libjpegsynth.so")
SRC = os.path.join(HERE, "jpeg_synth.c")
_lib = None

SUB_444, SUB_422, SUB_420, SUB_440, SUB_GREY = 0, 1, 2, 3, 4


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", SO, SRC, "-lm"], check=True)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        L.synth_picture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64]
        L.synth_encode.restype = C.c_long
        L.synth_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.synth_make.restype = C.c_long
        L.synth_make.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.synth_make2.restype = C.c_long
        L.synth_make2.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.synth_encode2.restype = C.c_long
        L.synth_encode2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.synth_init()
        _lib = L
    return _lib


def picture(w, h, seed):
    rgb = np.zeros((h, w, 3), np.uint8)
    lib().synth_picture(rgb.ctypes.data, w, h, seed)
    return rgb


def encode(rgb, quality=85, subsampling=SUB_420, restart_interval=0, optimize=False):
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb)
    cap = w * h * 3 + 4096
    out = np.zeros(cap, np.uint8)
    n = lib().synth_encode2(rgb.ctypes.data, w, h, quality, subsampling, restart_interval, int(optimize), out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()


def make(w, h, seed, quality=85, subsampling=SUB_420, restart_interval=0, detail=1.0, optimize=False):
    """detail 1.0 = the round-1 pictures (~0.32 B/px at q75-95 4:2:0); DENSE_DETAIL gives the density of the bundled
    ImageNet sample's class (~0.58 B/px).  optimize: Huffman tables fitted to the picture (4 distinct tables)."""
    cap = w * h * 3 + 4096
    out = np.zeros(cap, np.uint8)
    n = lib().synth_make2(w, h, seed, quality, subsampling, restart_interval, int(round(detail * 100)), int(optimize), out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()


def imagenet_like_specs(n, seed=3):
    """(w, h, seed, quality) tuples: widths ~500, heights ~375, clipped to [64, 1024], ~10 % portrait."""
    rng = np.random.default_rng(seed)
    specs = []
    for k in range(n):
        w = int(np.clip(rng.normal(500, 90), 64, 1024))
        h = int(np.clip(rng.normal(375, 70), 64, 1024))
        if rng.random() < 0.10:
            w, h = h, w
        q = int(rng.choice([75, 85, 90, 95]))
        specs.append((w, h, seed * 1000003 + k, q))
    return specs


DENSE_DETAIL = 2.2      # with QUALITY_SHIFT: mean density of the cfg3 set ~0.58 B/px (tests/test_synth.py checks >= 0.55)


QUALITY_SHIFT = {75: 88, 85: 92, 90: 95, 95: 97}   # the denser set: higher qualities, as ImageNet originals have


def cfg3_imagenet_like(n=1024, seed=3, threads=None, detail=1.0, optimize=False, extra=(), quality_shift=False):
    """n mixed-size 4:2:0 JPEGs.  `extra`: RGB arrays re-encoded (4:2:0, q90) in place of the first pictures of the set --
    the bench puts the bundled ImageNet sample there.  detail=DENSE_DETAIL with quality_shift gives ~0.58 B/px."""
    lib()
    specs = imagenet_like_specs(n, seed)
    if quality_shift:
        specs = [(w, h, sd, QUALITY_SHIFT[q]) for (w, h, sd, q) in specs]
    threads = threads or min(16, os.cpu_count() or 4)
    with ThreadPoolExecutor(threads) as ex:      # ctypes drops the GIL inside the C call
        out = list(ex.map(lambda s: make(s[0], s[1], s[2], s[3], SUB_420, 0, detail, optimize), specs))
    for k, rgb in enumerate(extra):
        out[k] = encode(rgb, 90, SUB_420, 0, optimize)
    return out


def cfg2_single_4k(seed=2, restart_rows=False):
    ri = (3840 // 16) if restart_rows else 0
    return make(3840, 2160, seed, 85, SUB_420, ri)


def cfg5_tile(size=16384, seed=5):
    return make(size, size, seed, 85, SUB_444, size // 8)
