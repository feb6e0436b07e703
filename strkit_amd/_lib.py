"""ctypes binding of libstrkit_amd.so — the thin layer above the C ABI (include/strkit_amd.h).

Fails loudly: if the HIP library cannot be loaded, or no gfx950 device can be initialised, every
compute entry point raises.  There is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _build

STRK_SG_ALL = 15
STRK_TIE_FIRST, STRK_TIE_LAST = 0, 1
STRK_NARROW_NONE, STRK_NARROW_DECREMENT, STRK_NARROW_HALVE, STRK_NARROW_AFTER_SEED = 0, 1, 2, 3
STRK_I16_CELL_MAY_SATURATE, STRK_I16_SCORE_SATURATES = 1, 2
STRK_E_EMPTY = -61
STRK_E_INVALID, STRK_E_NOMEM, STRK_E_DEVICE, STRK_E_NODEV = -22, -12, -5, -19

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class StrkParams(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("local_search_range", C.c_int32), ("step_size", C.c_int32),
                ("tie_rule", C.c_int32), ("end_flags", C.c_int32), ("feedback", C.c_int32), ("window", C.c_int32),
                ("no_dedupe", C.c_int32), ("no_band", C.c_int32), ("narrowing", C.c_int32)]


class StrkBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_loci", C.c_int32), ("seqs", C.c_void_p), ("seq_off", C.c_void_p),
                ("nfl", C.c_void_p), ("ntr", C.c_void_p), ("nfr", C.c_void_p), ("est_cn", C.c_void_p),
                ("read_off", C.c_void_p), ("motifs", C.c_void_p), ("motif_off", C.c_void_p)]


class StrkStats(C.Structure):
    _fields_ = [("dp_cells", C.c_int64), ("n_fallback", C.c_int32), ("n_miss_reads", C.c_int32),
                ("n_miss_rounds", C.c_int32), ("kernel_ms", C.c_float), ("dp_kernel_ms", C.c_float),
                ("n_dp_launches", C.c_int32), ("n_dedup_reads", C.c_int32), ("n_band_reads", C.c_int32),
                ("n_band_fallback", C.c_int32), ("band_kernel_ms", C.c_float), ("window_used", C.c_int32),
                ("band_bytes", C.c_int64), ("exact_bytes", C.c_int64), ("band_wide_kernel_ms", C.c_float),
                ("long_kernel_ms", C.c_float), ("generic_kernel_ms", C.c_float), ("head_ms", C.c_float), ("replay_ms", C.c_float),
                ("n_long_reads", C.c_int32), ("wide_bytes", C.c_int64), ("long_bytes", C.c_int64),
                ("band_cells", C.c_int64), ("wide_cells", C.c_int64), ("exact_cells", C.c_int64), ("long_cells", C.c_int64),
                ("window_bucket", C.c_int32 * 5), ("n_sub_batches", C.c_int32)]

    def as_dict(self) -> dict:
        return {k: (list(getattr(self, k)) if k == "window_bucket" else getattr(self, k)) for k, _ in self._fields_}


# Every symbol include/strkit_amd.h declares (tests check the .so exports exactly these).
EXPORTS = ("strk_init", "strk_destroy", "strk_last_error", "strk_version", "strk_adaptive_reset", "strk_device_mem", "strk_host_register", "strk_host_unregister",
           "strk_host_is_pinned", "strk_repeat_count", "strk_count_loci",
           "strk_count_loci_device", "strk_submit_loci_device", "strk_finish", "strk_score_table",
           "strk_score_ref_table", "strk_ref_repeat_count", "strk_ref_repeat_count_batch", "strk_realign", "strk_realign_i16_flags", "strk_bam_scan",
           "strk_extract_reads", "strk_bgzf_inflate", "strk_bgzf_inflate_range", "strk_bam_names", "strk_bam_scan_piece",
           "strk_dbam_open", "strk_dbam_close", "strk_dbam_release_cache", "strk_dbam_inflate", "strk_dbam_inflate_file", "strk_dbam_inflate_file_range", "strk_dbam_file_ms", "strk_dbam_download", "strk_dbam_data", "strk_bgzf_inflate_sw",
           "strk_dbam_download_seqs", "strk_dbam_kernel_ms", "strk_dbam_voffsets", "strk_dbam_scan", "strk_dbam_extract", "strk_dbam_names", "strk_count_loci_dseqs", "strk_read_coords_both")

_lib = None
_lib_lock = threading.Lock()


class StrkError(RuntimeError):
    """Raised for any non-zero return of the C ABI (the worker's catch-all in the reference,
    strkit/call/call_sample.py:159-166, turns it into 'locus skipped + logged')."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"strkit_amd error {code}: {msg}")
        self.code = code


def load(build: bool = True):
    """dlopen the library (building it in-tree first if it is missing/stale and hipcc is present)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        # Callers that keep several batched calls in flight (one context + stream each) need a
        # hardware queue per stream; the HIP runtime's default is 4 for the whole process.  Read at HIP initialisation.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        path = os.environ.get("STRKIT_AMD_LIB", _build.LIB_PATH)   # a privately built copy (profiling aids in tools/)
        build = build and path == _build.LIB_PATH
        if build and _build.stale():
            try:
                _build.build()
            except Exception as e:  # noqa: BLE001
                if not os.path.exists(path):
                    raise RuntimeError(f"libstrkit_amd.so is not built and could not be built: {e}") from e
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m strkit_amd._build` (strkit_amd has no CPU fallback)")
        L = C.CDLL(path)
        L.strk_init.restype = C.c_int
        L.strk_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.strk_destroy.restype = None
        L.strk_destroy.argtypes = [C.c_void_p]
        L.strk_last_error.restype = C.c_char_p
        L.strk_version.restype = C.c_char_p
        L.strk_adaptive_reset.restype = None
        L.strk_adaptive_reset.argtypes = []
        L.strk_repeat_count.restype = C.c_int
        L.strk_repeat_count.argtypes = ([C.c_void_p, C.c_int32] + [C.c_char_p, C.c_int32] * 4 + [C.c_int32] * 3
                                        + [_i32p] * 3)
        L.strk_count_loci.restype = C.c_int
        L.strk_count_loci.argtypes = [C.c_void_p, C.POINTER(StrkBatch), C.POINTER(StrkParams)] + [C.c_void_p] * 4 + [
            C.POINTER(StrkStats)]
        L.strk_count_loci_device.restype = C.c_int
        L.strk_count_loci_device.argtypes = ([C.c_void_p, C.POINTER(StrkBatch), C.POINTER(StrkParams)]
                                             + [C.c_void_p] * 4 + [C.c_void_p, C.POINTER(StrkStats)])
        L.strk_submit_loci_device.restype = C.c_int
        L.strk_submit_loci_device.argtypes = ([C.c_void_p, C.POINTER(StrkBatch), C.POINTER(StrkParams)]
                                              + [C.c_void_p] * 4 + [C.c_void_p])
        L.strk_finish.restype = C.c_int
        L.strk_finish.argtypes = [C.c_void_p, C.POINTER(StrkStats)]
        L.strk_score_table.restype = C.c_int
        L.strk_score_table.argtypes = [C.c_void_p, C.POINTER(StrkBatch), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int32, C.c_void_p, C.POINTER(StrkStats)]
        L.strk_score_ref_table.restype = C.c_int
        L.strk_score_ref_table.argtypes = [C.c_void_p, C.POINTER(StrkBatch), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(StrkStats)]
        L.strk_ref_repeat_count.restype = C.c_int
        L.strk_ref_repeat_count.argtypes = ([C.c_void_p, C.c_int32] + [C.c_char_p, C.c_int32] * 4 + [C.c_int32] * 6
                                            + [_i32p])
        L.strk_ref_repeat_count_batch.restype = C.c_int
        L.strk_ref_repeat_count_batch.argtypes = ([C.c_void_p, C.c_int32] + [C.c_void_p] * 9 + [C.c_int32] + [C.c_void_p] * 3
                                                  + [C.c_int32, C.c_void_p])
        L.strk_bgzf_inflate.restype = C.c_int64
        L.strk_bgzf_inflate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32]
        L.strk_bgzf_inflate_range.restype = C.c_int64
        L.strk_bgzf_inflate_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int32]
        L.strk_dbam_open.restype = C.c_int
        L.strk_dbam_open.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.strk_dbam_release_cache.restype = None
        L.strk_dbam_release_cache.argtypes = []
        L.strk_dbam_close.restype = None
        L.strk_dbam_close.argtypes = [C.c_void_p]
        L.strk_dbam_inflate.restype = C.c_int64
        L.strk_dbam_inflate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.strk_dbam_file_ms.restype = None
        L.strk_dbam_file_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.strk_dbam_inflate_file_range.restype = C.c_int64
        L.strk_dbam_inflate_file_range.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        L.strk_dbam_inflate_file.restype = C.c_int64
        L.strk_dbam_inflate_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]
        L.strk_dbam_download.restype = C.c_int
        L.strk_dbam_download.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.strk_dbam_data.restype = C.c_void_p
        L.strk_dbam_data.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.strk_dbam_download_seqs.restype = C.c_int
        L.strk_dbam_download_seqs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.strk_dbam_kernel_ms.restype = C.c_double
        L.strk_dbam_kernel_ms.argtypes = [C.c_void_p]
        L.strk_dbam_voffsets.restype = C.c_int
        L.strk_dbam_voffsets.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.strk_dbam_scan.restype = C.c_int64
        L.strk_dbam_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 9
        L.strk_dbam_extract.restype = C.c_int
        L.strk_dbam_extract.argtypes = ([C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p] * 6
                                        + [C.POINTER(C.c_void_p)])
        L.strk_dbam_names.restype = C.c_int
        L.strk_dbam_names.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.strk_count_loci_dseqs.restype = C.c_int
        L.strk_count_loci_dseqs.argtypes = ([C.c_void_p, C.POINTER(StrkBatch), C.c_void_p, C.POINTER(StrkParams)] + [C.c_void_p] * 4
                                            + [C.POINTER(StrkStats)])
        L.strk_read_coords_both.restype = C.c_int
        L.strk_read_coords_both.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.strk_bgzf_inflate_sw.restype = C.c_int64
        L.strk_bgzf_inflate_sw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.strk_bam_scan_piece.restype = C.c_int64
        L.strk_bam_scan_piece.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 8 + [C.POINTER(C.c_int64)]
        L.strk_bam_names.restype = C.c_int64
        L.strk_bam_names.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.strk_bam_scan.restype = C.c_int64
        L.strk_bam_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 8
        L.strk_extract_reads.restype = C.c_int
        L.strk_extract_reads.argtypes = ([C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 3
                                         + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p])
        L.strk_realign.restype = C.c_int
        L.strk_realign.argtypes = ([C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p] * 5
                                   + [C.POINTER(StrkStats)])
        L.strk_device_mem.restype = C.c_int
        L.strk_device_mem.argtypes = [C.c_int, _i64p, _i64p]
        L.strk_realign_i16_flags.restype = C.c_int
        L.strk_realign_i16_flags.argtypes = [C.c_int32] + [C.c_void_p] * 4
        L.strk_host_register.restype = C.c_int
        L.strk_host_register.argtypes = [C.c_void_p, C.c_int64]
        L.strk_host_unregister.restype = C.c_int
        L.strk_host_unregister.argtypes = [C.c_void_p]
        L.strk_host_is_pinned.restype = C.c_int
        L.strk_host_is_pinned.argtypes = [C.c_void_p, C.c_int64]
        _lib = L
        return L


def device_mem(device: int = 0) -> tuple[int, int]:
    """(free, total) bytes of a device's memory."""
    f, t = C.c_int64(), C.c_int64()
    check(load().strk_device_mem(int(device), C.byref(f), C.byref(t)))
    return f.value, t.value


def host_register(arr) -> None:
    """Page-locks a C-contiguous numpy array in place (strk_host_register): strk_count_loci then reads it by DMA where it
    lies.  The array must stay alive, and must not be resized, until host_unregister."""
    if arr.nbytes:
        check(load().strk_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes))


def host_unregister(arr) -> None:
    if arr.nbytes:
        check(load().strk_host_unregister(C.c_void_p(arr.ctypes.data)))


def host_is_pinned(arr) -> bool:
    return bool(arr.nbytes) and bool(load().strk_host_is_pinned(C.c_void_p(arr.ctypes.data), arr.nbytes))


def check(rc: int) -> None:
    if rc != 0:
        raise StrkError(rc, load().strk_last_error().decode("utf-8", "replace"))


class Context:
    """One HIP device context (strk_init / strk_destroy).  Created lazily per process and device:
    the reference's workers are forked (strkit/call/call_sample.py:345-356), so nothing touches the
    GPU at import time."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        check(self._lib.strk_init(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._pid = os.getpid()

    def close(self) -> None:
        if getattr(self, "handle", None) and self._pid == os.getpid():
            self._lib.strk_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_ctxs: dict[tuple[int, int], Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("STRKIT_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (os.getpid(), device)
    ctx = _ctxs.get(key)
    if ctx is None or ctx.handle is None:
        ctx = _ctxs[key] = Context(device)
    return ctx
