"""pjd_amd -- thin ctypes binding of the C ABI in include/pjd.h and include/pjd_host.h.

Python is plumbing here (tests, bench harness); the product is lib/libpjd.so (HIP kernels for
gfx950 behind the C ABI) and lib/libpjdhost.so (scanner + BMP helpers).  There is no CPU decode
path in this package: without a gfx950 device `Context()` raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_DIR = os.path.join(_PKG, "lib")
LIBPJD = os.path.join(LIB_DIR, "libpjd.so")
LIBHOST = os.path.join(LIB_DIR, "libpjdhost.so")
LIBPIPE = os.path.join(LIB_DIR, "libpjdpipe.so")

OUT_RGB8, OUT_BMP = 0, 1
PLAN_LATENCY, PLAN_THROUGHPUT = 0, 1      # pjd_set_plan_mode
F_STANDARD_RESTART, F_FORCE_SEQUENTIAL, F_STANDARD_ZIGZAG, F_PROGRESSIVE = 1, 2, 4, 8
SCAN_PROGRESSIVE = 1
MAX_KERNELS = 16
ABI_VERSION = 5          # PJD_VERSION of include/pjd.h these ctypes structs mirror


class HuffTable(C.Structure):
    _fields_ = [("offsets", C.c_uint8 * 17), ("symbols", C.c_uint8 * 162), ("set", C.c_uint8)]


class ScanDesc(C.Structure):
    _fields_ = [("n_comp", C.c_uint8), ("comp", C.c_uint8 * 3), ("ss", C.c_uint8), ("se", C.c_uint8), ("ah", C.c_uint8), ("al", C.c_uint8),
                ("restart_interval", C.c_uint32), ("table", HuffTable * 3), ("ecs", C.c_void_p), ("ecs_len", C.c_uint64)]


class ImageDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("num_components", C.c_uint8), ("h_samp", C.c_uint8), ("v_samp", C.c_uint8),
        ("comp_h", C.c_uint8 * 3), ("comp_v", C.c_uint8 * 3),
        ("comp_qt", C.c_uint8 * 3), ("comp_dc", C.c_uint8 * 3), ("comp_ac", C.c_uint8 * 3),
        ("qt_set", C.c_uint8 * 4),
        ("qt", (C.c_uint32 * 64) * 4),
        ("dc", HuffTable * 4), ("ac", HuffTable * 4),
        ("restart_interval", C.c_uint32),
        ("ecs", C.c_void_p), ("ecs_len", C.c_uint64),
        ("seg_offsets", C.c_void_p), ("n_segments", C.c_uint32),
        ("flags", C.c_uint32),
        ("shard_first_seg", C.c_uint32), ("shard_n_segs", C.c_uint32),
        ("qt_slot48", C.c_uint32 * 4),
        ("scans", C.POINTER(ScanDesc)), ("n_scans", C.c_uint32), ("reserved_", C.c_uint32),
    ]


class Timings(C.Structure):
    _fields_ = [("n", C.c_int32), ("ms", C.c_float * MAX_KERNELS),
                ("name", (C.c_char * 32) * MAX_KERNELS), ("total_ms", C.c_float)]

    def as_dict(self):
        return {self.name[i].value.decode(): float(self.ms[i]) for i in range(self.n)}


class BatchInfo(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("pixels", C.c_uint64), ("ecs_bytes", C.c_uint64),
                ("out_bytes", C.c_uint64), ("coef_bytes", C.c_uint64), ("n_data_units", C.c_uint64),
                ("n_subsequences", C.c_uint64), ("device_bytes", C.c_uint64),
                ("n_sequential", C.c_int32), ("n_fallback", C.c_int32),
                ("n_huff_workgroups", C.c_uint64), ("sync_rounds", C.c_uint64), ("sync_lane_passes", C.c_uint64),
                ("fix_rounds", C.c_uint64), ("fix_lane_passes", C.c_uint64),
                ("sub_bytes", C.c_uint32), ("n_table_sets", C.c_uint32), ("n_huff_waves", C.c_uint64),
                ("n_entries", C.c_uint64), ("exact_fallback_ms", C.c_float), ("n_entropy_errors", C.c_uint32),
                ("flag_waves", C.c_uint64 * 8), ("huff_lds_bytes", C.c_uint32), ("plan_mode", C.c_uint32), ("walks", C.c_uint64), ("walk_lanes", C.c_uint64), ("n_steps", C.c_uint64),
                ("lane_fill_x1024", C.c_uint32), ("reserved2_", C.c_uint32)]


SPLIT_MAX_DEVICES = 16


class SplitStats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("broadcast_s", C.c_double), ("upload_s", C.c_double), ("exec_s", C.c_double), ("download_s", C.c_double),
                ("blob_bytes", C.c_uint64), ("ecs_bytes", C.c_uint64 * SPLIT_MAX_DEVICES),
                ("n_segments", C.c_uint32), ("n_ranks", C.c_uint32), ("n_exact", C.c_uint32),
                ("rccl_used", C.c_int32), ("redone_whole", C.c_int32)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k == "ecs_bytes" else getattr(self, k)) for k, _ in self._fields_}


class PjdError(RuntimeError):
    pass


_host = None
_dev = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(LIBHOST):
            raise PjdError(f"{LIBHOST} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIBHOST)
        L.pjd_scan_memory.restype = C.c_int
        L.pjd_scan_memory.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_void_p)]
        L.pjd_scan_file.restype = C.c_int
        L.pjd_scan_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.pjd_scan_memory_ex.restype = C.c_int
        L.pjd_scan_memory_ex.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.pjd_scan_file_ex.restype = C.c_int
        L.pjd_scan_file_ex.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.pjd_scanned_desc.restype = C.POINTER(ImageDesc)
        L.pjd_scanned_desc.argtypes = [C.c_void_p]
        L.pjd_scanned_log.restype = C.c_char_p
        L.pjd_scanned_log.argtypes = [C.c_void_p]
        L.pjd_scanned_valid.restype = C.c_int
        L.pjd_scanned_valid.argtypes = [C.c_void_p]
        L.pjd_scanned_free.argtypes = [C.c_void_p]
        L.pjd_scanned_metadata.argtypes = [C.c_void_p, C.c_void_p]
        L.pjd_rgb_to_bmp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pjd_write_file.restype = C.c_int
        L.pjd_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        _host = L
    return _host


def dev_lib():
    """Load libpjd.so.  Fails loudly when the HIP extension has not been built."""
    global _dev
    if _dev is None:
        if not os.path.exists(LIBPJD):
            raise PjdError(f"{LIBPJD} missing: the HIP extension is not built (no CPU fallback exists)")
        L = C.CDLL(LIBPJD)
        vp, i32 = C.c_void_p, C.c_int
        L.pjd_version.restype = i32
        if L.pjd_version() != ABI_VERSION:
            raise PjdError(f"{LIBPJD} has ABI version {L.pjd_version()}, these bindings mirror version {ABI_VERSION}: rebuild")
        L.pjd_open.restype = i32
        L.pjd_open.argtypes = [i32, C.POINTER(vp)]
        L.pjd_close.argtypes = [vp]
        L.pjd_set_plan_mode.restype = i32
        L.pjd_set_plan_mode.argtypes = [vp, i32]
        L.pjd_last_error.restype = C.c_char_p
        L.pjd_last_error.argtypes = [vp]
        L.pjd_status_string.restype = C.c_char_p
        L.pjd_status_string.argtypes = [i32]
        L.pjd_stream.restype = vp
        L.pjd_stream.argtypes = [vp]
        L.pjd_batch_create.restype = i32
        L.pjd_batch_create.argtypes = [vp, C.POINTER(ImageDesc), i32, i32, C.POINTER(vp)]
        for fn in ("pjd_batch_upload", "pjd_batch_decode", "pjd_batch_capture", "pjd_batch_sync"):
            getattr(L, fn).restype = i32
            getattr(L, fn).argtypes = [vp]
        L.pjd_batch_decode_timed.restype = i32
        L.pjd_batch_decode_timed.argtypes = [vp, C.POINTER(Timings)]
        L.pjd_batch_download.restype = i32
        L.pjd_batch_download.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32)]
        L.pjd_batch_get_info.restype = i32
        L.pjd_batch_get_info.argtypes = [vp, C.POINTER(BatchInfo)]
        L.pjd_batch_download_packed.restype = i32
        L.pjd_batch_download_packed.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_int32)]
        L.pjd_batch_packed_size.restype = C.c_uint64
        L.pjd_batch_packed_size.argtypes = [vp]
        L.pjd_batch_output_offset.restype = C.c_uint64
        L.pjd_batch_output_offset.argtypes = [vp, i32]
        L.pjd_host_alloc.restype = vp
        L.pjd_host_alloc.argtypes = [C.c_uint64]
        L.pjd_host_free.argtypes = [vp]
        L.pjd_batch_output_size.restype = C.c_uint64
        L.pjd_batch_output_size.argtypes = [vp, i32]
        L.pjd_batch_device_output.restype = vp
        L.pjd_batch_device_output.argtypes = [vp, i32]
        L.pjd_batch_device_status.restype = vp
        L.pjd_batch_device_status.argtypes = [vp]
        L.pjd_batch_destroy.argtypes = [vp]
        L.pjd_decode_batch.restype = i32
        L.pjd_decode_batch.argtypes = [vp, C.POINTER(ImageDesc), i32, i32, C.POINTER(vp), C.POINTER(C.c_int32)]
        L.pjd_exec_dpu_payload.restype = i32
        L.pjd_exec_dpu_payload.argtypes = [vp, vp, vp, i32]
        L.pjd_output_size.restype = C.c_uint64
        L.pjd_output_size.argtypes = [C.c_uint32, C.c_uint32, i32]
        L.pjd_coefficients_size.restype = C.c_uint64
        L.pjd_coefficients_size.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8]
        L.pjd_batch_download_coefficients.restype = i32
        L.pjd_batch_download_coefficients.argtypes = [vp, i32, vp, C.c_uint64]
        L.pjd_split_decode.restype = i32
        L.pjd_split_decode.argtypes = [C.POINTER(ImageDesc), C.POINTER(C.c_int32), i32, i32, vp, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(SplitStats)]
        L.pjd_split_plan.restype = i32
        L.pjd_split_plan.argtypes = [C.POINTER(ImageDesc), i32, i32, C.POINTER(ImageDesc), vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pjd_split_release.restype = None
        L.pjd_plan_info.restype = i32
        L.pjd_plan_info.argtypes = [C.POINTER(ImageDesc), i32, i32, C.POINTER(BatchInfo)]
        _dev = L
    return _dev


class Scanned:
    """A parsed JPEG (host side).  Mirrors the reference's `Header` for the decode path."""

    def __init__(self, data: bytes = None, name: str = "x.jpg", path: str = None, options: int = 0):
        """options: SCAN_PROGRESSIVE parses a progressive file scan by scan (not reference behaviour: the reference rejects it)."""
        L = host_lib()
        h = C.c_void_p()
        if path is not None:
            rc = L.pjd_scan_file_ex(path.encode(), options, C.byref(h))
            if rc == 2:
                raise FileNotFoundError(path)
        else:
            self._data = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
            rc = L.pjd_scan_memory_ex(self._data.ctypes.data, len(data), name.encode(), options, C.byref(h))
        self._h = h
        self.rc = rc
        self.valid = bool(L.pjd_scanned_valid(h))
        self.log = L.pjd_scanned_log(h).decode()
        self.desc = L.pjd_scanned_desc(h).contents

    def metadata(self):
        m = np.zeros(276, np.uint32)
        host_lib().pjd_scanned_metadata(self._h, m.ctypes.data)
        return m

    def ecs(self):
        n = int(self.desc.ecs_len)
        if n == 0:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(self.desc.ecs, C.POINTER(C.c_uint8)), (n,)).copy()

    def seg_offsets(self):
        n = int(self.desc.n_segments)
        return np.ctypeslib.as_array(C.cast(self.desc.seg_offsets, C.POINTER(C.c_uint64)), (n,)).copy()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                host_lib().pjd_scanned_free(self._h)
                self._h = None
        except Exception:
            pass


def rgb_to_bmp(rgb: np.ndarray) -> bytes:
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.zeros(26 + h * (3 * w + w % 4), np.uint8)
    host_lib().pjd_rgb_to_bmp(rgb.ctypes.data, w, h, out.ctypes.data)
    return out.tobytes()


class Context:
    """One GPU (replaces DpuSet::allocate + load of the reference)."""

    def __init__(self, device: int = 0, plan_mode: int = None):
        L = dev_lib()
        h = C.c_void_p()
        rc = L.pjd_open(device, C.byref(h))
        if rc != 0:
            raise PjdError(f"pjd_open({device}) failed with {rc}: no usable gfx950 device")
        self._h = h
        self.L = L
        if plan_mode is not None:
            self.set_plan_mode(plan_mode)

    def set_plan_mode(self, mode: int):
        """PLAN_LATENCY (a batch decoded alone finishes sooner) or PLAN_THROUGHPUT (batches kept in flight: more pictures per second);
        applies to batches created afterwards (include/pjd.h, pjd_set_plan_mode)."""
        self._check(self.L.pjd_set_plan_mode(self._h, int(mode)), "pjd_set_plan_mode")

    def close(self):
        if self._h:
            self.L.pjd_close(self._h)
            self._h = None

    def _check(self, rc, what):
        if rc != 0:
            raise PjdError(f"{what} failed ({rc}): {self.L.pjd_last_error(self._h).decode()}")

    @property
    def stream(self):
        return self.L.pjd_stream(self._h)

    def batch(self, descs, out_format=OUT_RGB8):
        return Batch(self, descs, out_format)

    def decode(self, descs, out_format=OUT_RGB8):
        """One-shot decode -> (list of np.uint8 arrays, list of status ints)."""
        with self.batch(descs, out_format) as b:
            b.upload()
            b.decode()
            return b.download()

    def exec_dpu_payload(self, metadata: np.ndarray, mcus: np.ndarray):
        """The literal DPU contract: metadata (n,276) uint32, mcus (n,19200) int16 in/out."""
        metadata = np.ascontiguousarray(metadata, np.uint32).reshape(-1, 276)
        assert mcus.dtype == np.int16 and mcus.flags.c_contiguous
        n = metadata.shape[0]
        assert mcus.size == n * 19200
        self._check(self.L.pjd_exec_dpu_payload(self._h, metadata.ctypes.data, mcus.ctypes.data, n), "pjd_exec_dpu_payload")
        return mcus


class Batch:
    def __init__(self, ctx: Context, descs, out_format):
        self.ctx, self.L = ctx, ctx.L
        self.n = len(descs)
        self.out_format = out_format
        arr = (ImageDesc * max(self.n, 1))()
        for i, d in enumerate(descs):
            C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(ImageDesc))
        self._descs = arr
        self._keep = descs
        h = C.c_void_p()
        ctx._check(self.L.pjd_batch_create(ctx._h, arr, self.n, out_format, C.byref(h)), "pjd_batch_create")
        self._h = h

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.destroy()

    def destroy(self):
        if self._h:
            self.L.pjd_batch_destroy(self._h)
            self._h = None

    def upload(self):
        self.ctx._check(self.L.pjd_batch_upload(self._h), "pjd_batch_upload")

    def decode(self):
        self.ctx._check(self.L.pjd_batch_decode(self._h), "pjd_batch_decode")

    def decode_timed(self):
        t = Timings()
        self.ctx._check(self.L.pjd_batch_decode_timed(self._h, C.byref(t)), "pjd_batch_decode_timed")
        return t.as_dict(), float(t.total_ms)

    def capture(self):
        self.ctx._check(self.L.pjd_batch_capture(self._h), "pjd_batch_capture")

    def sync(self):
        self.ctx._check(self.L.pjd_batch_sync(self._h), "pjd_batch_sync")

    def info(self):
        bi = BatchInfo()
        self.ctx._check(self.L.pjd_batch_get_info(self._h, C.byref(bi)), "pjd_batch_get_info")
        return {k: (list(getattr(bi, k)) if k == "flag_waves" else (float(getattr(bi, k)) if k == "exact_fallback_ms" else int(getattr(bi, k)))) for k, _ in bi._fields_}

    def output_size(self, i):
        return int(self.L.pjd_batch_output_size(self._h, i))

    def device_output(self, i):
        return self.L.pjd_batch_device_output(self._h, i)

    def download(self):
        outs = [np.zeros(self.output_size(i), np.uint8) for i in range(self.n)]
        ptrs = (C.c_void_p * max(self.n, 1))(*[o.ctypes.data for o in outs])
        st = (C.c_int32 * max(self.n, 1))()
        self.ctx._check(self.L.pjd_batch_download(self._h, ptrs, st), "pjd_batch_download")
        if self.out_format == OUT_RGB8:
            outs = [o.reshape(int(self._descs[i].height), int(self._descs[i].width), 3) for i, o in enumerate(outs)]
        return outs, [int(st[i]) for i in range(self.n)]


    def coefficients(self, i):
        """Stage-level parity: image i's coefficients after entropy decoding, in the reference's MCU_buffer layout
        (n_dpus x 19200 int16, reference src/jpeg_scanner.cpp:733-741)."""
        d = self._descs[i]
        n = int(self.L.pjd_coefficients_size(d.width, d.height, d.h_samp, d.v_samp))
        out = np.zeros(n, np.int16)
        self.ctx._check(self.L.pjd_batch_download_coefficients(self._h, i, out.ctypes.data, n), "pjd_batch_download_coefficients")
        return out.reshape(-1, 19200)

    def download_packed(self):
        """All pictures in one D2H copy into page-locked memory; returns (list of arrays, statuses)."""
        size = int(self.L.pjd_batch_packed_size(self._h))
        host = self.L.pjd_host_alloc(size)
        if not host:
            raise PjdError("pjd_host_alloc failed")
        try:
            st = (C.c_int32 * max(self.n, 1))()
            self.ctx._check(self.L.pjd_batch_download_packed(self._h, host, size, st), "pjd_batch_download_packed")
            whole = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), shape=(size,))
            outs = []
            for i in range(self.n):
                off, n = int(self.L.pjd_batch_output_offset(self._h, i)), self.output_size(i)
                outs.append(whole[off:off + n].copy())
        finally:
            self.L.pjd_host_free(host)
        return outs, [int(st[i]) for i in range(self.n)]


# ---- pipelined batcher (include/pjd_pipeline.h) ---------------------------------------------
SINK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint8), C.c_uint64)


class PipeOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("out_format", C.c_int32), ("batch_images", C.c_int32),
                ("scan_threads", C.c_int32), ("slots", C.c_int32), ("sink_threads", C.c_int32),
                ("sink", SINK_FN), ("sink_user", C.c_void_p),
                ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_int32), ("scan_options", C.c_uint32)]


PIPE_MAX_DEVICES = 16


class PipeStats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("scan_s", C.c_double), ("create_s", C.c_double), ("upload_s", C.c_double),
                ("exec_s", C.c_double), ("download_s", C.c_double), ("sink_s", C.c_double),
                ("n_inputs", C.c_uint64), ("n_decoded", C.c_uint64), ("n_rejected", C.c_uint64),
                ("n_batches", C.c_uint64), ("n_batch_failures", C.c_uint64),
                ("pixels", C.c_uint64), ("in_bytes", C.c_uint64), ("ecs_bytes", C.c_uint64), ("out_bytes", C.c_uint64),
                ("n_devices", C.c_uint64), ("n_stolen", C.c_uint64),
                ("device_batches", C.c_uint64 * PIPE_MAX_DEVICES), ("device_in_bytes", C.c_uint64 * PIPE_MAX_DEVICES),
                ("n_exact_images", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k.startswith("device_") else getattr(self, k)) for k, _ in self._fields_}


_pipe = None


def pipe_lib():
    global _pipe
    if _pipe is None:
        if not os.path.exists(LIBPIPE):
            raise PjdError(f"{LIBPIPE} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        dev_lib(), host_lib()
        L = C.CDLL(LIBPIPE)
        L.pjd_pipe_run_files.restype = C.c_int
        L.pjd_pipe_run_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(PipeOpts), C.POINTER(PipeStats)]
        L.pjd_pipe_run_memory.restype = C.c_int
        L.pjd_pipe_run_memory.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p), C.c_int,
                                          C.POINTER(PipeOpts), C.POINTER(PipeStats)]
        L.pjd_pipe_release.restype = None
        L.pjd_pipe_assign.restype = C.c_int
        L.pjd_pipe_assign.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_int32)]
        _pipe = L
    return _pipe


def pipe_release():
    if _pipe is not None:
        _pipe.pjd_pipe_release()


def pipe_assign(costs, n_devices):
    """The batcher's dealing rule (pjd_pipe_assign): device index per item, longest first onto the least loaded."""
    L = pipe_lib()
    n = len(costs)
    c = (C.c_uint64 * max(n, 1))(*[int(x) for x in costs])
    out = (C.c_int32 * max(n, 1))()
    rc = L.pjd_pipe_assign(c, n, n_devices, out)
    if rc != 0:
        raise PjdError(f"pjd_pipe_assign failed ({rc})")
    return [int(out[k]) for k in range(n)]


def pipe_run(jpegs=None, names=None, paths=None, out_format=OUT_BMP, batch_images=1024, scan_threads=0, slots=0,
             sink_threads=0, sink=None, device=0, devices=None, scan_options=0):
    """Run the pipelined batcher over in-memory JPEGs (`jpegs`: list of bytes) or files (`paths`).
    `devices`: HIP ordinals to spread the batches over (default: `device` alone).

    `sink(index, name, log, status, data)` is called from worker threads with `data` a numpy copy of the
    picture (or None).  Returns the statistics as a dict."""
    L = pipe_lib()
    o = PipeOpts()
    o.device, o.out_format, o.batch_images = device, out_format, batch_images
    o.scan_threads, o.slots, o.sink_threads = scan_threads, slots, sink_threads
    o.scan_options = scan_options
    if devices is not None:
        dv = (C.c_int32 * max(len(devices), 1))(*[int(d) for d in devices])
        o.devices, o.n_devices = C.cast(dv, C.POINTER(C.c_int32)), len(devices)

    def _tramp(user, index, name, log, status, data, length):
        pic = np.ctypeslib.as_array(data, shape=(length,)).copy() if data and length else None
        sink(index, name.decode(), log.decode(), status, pic)

    cb = SINK_FN(_tramp) if sink else SINK_FN()
    o.sink = cb
    st = PipeStats()
    if paths is not None:
        arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
        rc = L.pjd_pipe_run_files(arr, len(paths), C.byref(o), C.byref(st))
    else:
        n = len(jpegs)
        keep = [np.frombuffer(j, np.uint8) for j in jpegs]
        ptrs = (C.c_void_p * max(n, 1))(*[k.ctypes.data for k in keep])
        lens = (C.c_uint64 * max(n, 1))(*[len(j) for j in jpegs])
        nm = (C.c_char_p * max(n, 1))(*[(names[i] if names else f"mem{i}.jpg").encode() for i in range(n)])
        rc = L.pjd_pipe_run_memory(ptrs, lens, nm, n, C.byref(o), C.byref(st))
    if rc != 0:
        raise PjdError(f"pipeline failed ({rc})")
    return st.as_dict()


def split_decode(desc, devices, out_format=OUT_RGB8):
    """pjd_split_decode: ONE picture over several devices (restart-segment ranges, RCCL broadcast of the descriptor).
    -> (picture as np.uint8 array, status, stats dict)."""
    L = dev_lib()
    n = int(L.pjd_output_size(desc.width, desc.height, out_format))
    out = np.zeros(n, np.uint8)
    dv = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    st, status = SplitStats(), C.c_int32(0)
    rc = L.pjd_split_decode(C.byref(desc), dv, len(devices), out_format, out.ctypes.data, n, C.byref(status), C.byref(st))
    if rc != 0:
        raise PjdError(f"pjd_split_decode failed ({rc})")
    if out_format == OUT_RGB8:
        out = out.reshape(int(desc.height), int(desc.width), 3)
    return out, int(status.value), st.as_dict()


def split_plan(desc, world, rank):
    """pjd_split_plan (host only) -> None if the rank has no segment, else dict(first_seg, n_segs, byte_lo, byte_hi, first_mcu, last_mcu,
    seg_offsets of the shard descriptor, ecs_len of the shard)."""
    L = dev_lib()
    shard = ImageDesc()
    scratch = np.zeros(max(int(desc.n_segments), 1), np.uint64)
    lo, hi, m0, m1 = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
    rc = L.pjd_split_plan(C.byref(desc), world, rank, C.byref(shard), scratch.ctypes.data, C.byref(lo), C.byref(hi), C.byref(m0), C.byref(m1))
    if rc == 1:
        return None
    if rc != 0:
        raise PjdError(f"pjd_split_plan failed ({rc})")
    return {"first_seg": int(shard.shard_first_seg), "n_segs": int(shard.shard_n_segs), "byte_lo": lo.value, "byte_hi": hi.value,
            "first_mcu": m0.value, "last_mcu": m1.value, "seg_offsets": scratch.copy(), "ecs_len": int(shard.ecs_len),
            "ecs_delta": (int(shard.ecs or 0) - int(desc.ecs or 0))}


def plan_step_bits(desc):
    """Host-only, debug: fewest bits of stream per write-pass step the picture's tables can be made to sustain (a float; pjd.h)."""
    L = dev_lib()
    L.pjd_plan_step_bits.restype = C.c_int32
    L.pjd_plan_step_bits.argtypes = [C.POINTER(ImageDesc), C.POINTER(C.c_uint32)]
    v = C.c_uint32()
    rc = L.pjd_plan_step_bits(C.byref(desc), C.byref(v))
    if rc != 0:
        raise PjdError(f"pjd_plan_step_bits failed ({rc})")
    return v.value / 256.0


def plan_info(descs, out_format=OUT_RGB8):
    """Host-only: what a batch of these images would occupy (no GPU needed)."""
    L = dev_lib()
    arr = (ImageDesc * max(len(descs), 1))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(ImageDesc))
    bi = BatchInfo()
    rc = L.pjd_plan_info(arr, len(descs), out_format, C.byref(bi))
    if rc != 0:
        raise PjdError(f"pjd_plan_info failed ({rc})")
    return {k: (list(getattr(bi, k)) if k == "flag_waves" else (float(getattr(bi, k)) if k == "exact_fallback_ms" else int(getattr(bi, k)))) for k, _ in bi._fields_}
