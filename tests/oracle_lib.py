"""ctypes bindings to the CHECKERS under oracle/ (test infrastructure only).

* ``Port``  -> oracle/liboracle.so   plain-C restatement (always available once built)
* ``Ref``   -> oracle/_ref/libpjdref.so   the reference's own scanner / Huffman
  decoder / BMP writer compiled in place (present wherever it was built; the
  built .so travels to the GPU box, the reference sources do not).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PORT_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libpjdref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "ref_decode")


class Info(C.Structure):
    _fields_ = [
        ("valid", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32),
        ("hsamp", C.c_int32), ("vsamp", C.c_int32),
        ("mcu_w", C.c_int32), ("mcu_h", C.c_int32), ("mcu_w_real", C.c_int32), ("mcu_h_real", C.c_int32),
        ("restart_interval", C.c_int32), ("frame_type", C.c_int32), ("n_dpus", C.c_int32),
        ("ecs_len", C.c_int64),
        ("comp_h", C.c_uint8 * 3), ("comp_v", C.c_uint8 * 3), ("comp_qt", C.c_uint8 * 3),
        ("comp_dc", C.c_uint8 * 3), ("comp_ac", C.c_uint8 * 3),
        ("qt_set", C.c_uint8 * 4), ("dc_set", C.c_uint8 * 4), ("ac_set", C.c_uint8 * 4),
        ("qt", (C.c_uint32 * 64) * 4),
        ("dc_offsets", (C.c_uint8 * 17) * 4), ("dc_symbols", (C.c_uint8 * 162) * 4),
        ("ac_offsets", (C.c_uint8 * 17) * 4), ("ac_symbols", (C.c_uint8 * 162) * 4),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = np.ctypeslib.as_array(v).copy().tolist() if hasattr(v, "_length_") else int(v)
        return d


def build_oracle():
    """(Re)build oracle/liboracle.so, and oracle/_ref when the reference tree is present."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "port", "_ref"], check=True)


def _u8(buf):
    return (C.c_uint8 * len(buf)).from_buffer_copy(buf)


class Port:
    """oracle/liboracle.so"""

    def __init__(self):
        if not os.path.exists(PORT_SO):
            build_oracle()
        L = C.CDLL(PORT_SO)
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_int64]
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.orc_get_ecs.restype = C.c_int64
        L.orc_get_ecs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_get_metadata.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_huffman.restype = C.c_int
        L.orc_huffman.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_dpu_exec.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_dpu_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_bmp.restype = C.c_int64
        L.orc_bmp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rgb_from_mcus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_standard_zigzag.argtypes = [C.c_int]
        self.L = L

    def standard_zigzag(self, on: bool):
        """NOT the reference's behaviour: zigzag entry 48 -> 58 (tests of PJD_F_STANDARD_ZIGZAG only; switch it off again)."""
        self.L.orc_set_standard_zigzag(1 if on else 0)

    def parse(self, data: bytes, name: str = "x.jpg"):
        """-> dict(info, ecs, metadata, log, handle-free). Scanner only."""
        log = C.create_string_buffer(8192)
        buf = _u8(data)
        h = self.L.orc_open(buf, len(data), name.encode(), log, len(log))
        try:
            info = Info()
            self.L.orc_get_info(h, C.byref(info))
            out = {"info": info.as_dict(), "log": log.value.decode()}
            if info.valid:
                ecs = np.zeros(max(int(info.ecs_len), 1), np.uint8)
                self.L.orc_get_ecs(h, ecs.ctypes.data, ecs.size)
                out["ecs"] = ecs[: int(info.ecs_len)].copy()
                meta = np.zeros(276, np.uint32)
                self.L.orc_get_metadata(h, meta.ctypes.data)
                out["metadata"] = meta
            return out
        finally:
            self.L.orc_close(h)

    def decode(self, data: bytes, name: str = "x.jpg"):
        """Whole path. -> dict(valid, log, huff_rc, coef (n_dpus x 19200 int16, after Huffman),
        mcus (after the DPU stages), metadata, bmp (bytes), rgb (H x W x 3 uint8))."""
        log = C.create_string_buffer(8192)
        buf = _u8(data)
        h = self.L.orc_open(buf, len(data), name.encode(), log, len(log))
        try:
            info = Info()
            self.L.orc_get_info(h, C.byref(info))
            out = {"valid": bool(info.valid), "info": info.as_dict()}
            if not info.valid:
                out["log"] = log.value.decode() + f"{name}: Error - Invalid JPEG\n"
                return out
            meta = np.zeros(276, np.uint32)
            self.L.orc_get_metadata(h, meta.ctypes.data)
            mcus = np.zeros((info.n_dpus, 19200), np.int16)
            out["huff_rc"] = int(self.L.orc_huffman(h, mcus.ctypes.data))
            out["coef"] = mcus.copy()
            for d in range(info.n_dpus):
                self.L.orc_dpu_exec(meta.ctypes.data, mcus[d].ctypes.data)
            out["mcus"] = mcus
            out["metadata"] = meta
            n = self.L.orc_bmp(meta.ctypes.data, mcus.ctypes.data, None)
            bmp = np.zeros(n, np.uint8)
            self.L.orc_bmp(meta.ctypes.data, mcus.ctypes.data, bmp.ctypes.data)
            out["bmp"] = bmp.tobytes()
            rgb = np.zeros((info.height, info.width, 3), np.uint8)
            self.L.orc_rgb_from_mcus(meta.ctypes.data, mcus.ctypes.data, rgb.ctypes.data)
            out["rgb"] = rgb
            out["log"] = log.value.decode()
            return out
        finally:
            self.L.orc_close(h)

    def dpu_exec(self, meta: np.ndarray, mcus: np.ndarray):
        """In place: one 19200-int16 DPU payload through dequant/IDCT/colour."""
        assert meta.dtype == np.uint32 and meta.size == 276
        assert mcus.dtype == np.int16 and mcus.size == 19200 and mcus.flags.c_contiguous
        self.L.orc_dpu_exec(meta.ctypes.data, mcus.ctypes.data)

    def dpu_stage(self, meta, mcus, stage):
        self.L.orc_dpu_stage(meta.ctypes.data, mcus.ctypes.data, stage)


class Ref:
    """oracle/_ref/libpjdref.so -- the reference's own code (file-path based)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO) and os.path.exists(REF_BIN)

    def __init__(self):
        L = C.CDLL(REF_SO)
        L.ref_open.restype = C.c_void_p
        L.ref_open.argtypes = [C.c_char_p]
        L.ref_close.argtypes = [C.c_void_p]
        L.ref_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.ref_get_ecs.restype = C.c_int64
        L.ref_get_ecs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_get_metadata.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_huffman.restype = C.c_int
        L.ref_huffman.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_write_bmp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        L.ref_decode_file.restype = C.c_int
        L.ref_decode_file.argtypes = [C.c_char_p, C.c_char_p]
        self.L = L

    def parse_and_huffman(self, path: str):
        """-> dict(info, ecs, metadata, coef, huff_ok) using read_JPEG + decode_Huffman_data."""
        h = self.L.ref_open(path.encode())
        if not h:
            return None
        try:
            info = Info()
            self.L.ref_get_info(h, C.byref(info))
            out = {"info": info.as_dict()}
            if info.valid:
                ecs = np.zeros(max(int(info.ecs_len), 1), np.uint8)
                self.L.ref_get_ecs(h, ecs.ctypes.data, ecs.size)
                out["ecs"] = ecs[: int(info.ecs_len)].copy()
                meta = np.zeros(276, np.uint32)
                self.L.ref_get_metadata(h, meta.ctypes.data)
                out["metadata"] = meta
                coef = np.zeros((info.n_dpus, 19200), np.int16)
                out["huff_ok"] = int(self.L.ref_huffman(h, coef.ctypes.data, info.n_dpus))
                out["coef"] = coef
            return out
        finally:
            self.L.ref_close(h)

    @staticmethod
    def run_cli(in_path: str, out_path: str):
        """Run oracle/_ref/ref_decode as a child; -> (returncode, stdout text)."""
        p = subprocess.run([REF_BIN, in_path, out_path], capture_output=True, text=True, timeout=600)
        return p.returncode, p.stdout
