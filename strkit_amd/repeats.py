"""Drop-in for strkit/call/repeats.py on the GPU backend.

``get_repeat_count`` keeps the reference's signature, memoisation and return contract
(strkit/call/repeats.py:47-70): ``((best size, best score), n_explored, best size - start_count)``.
Where the reference calls ``strkit_rust_ext.get_repeat_count`` (repeats.py:58-68) this module calls
``strk_repeat_count`` of libstrkit_amd.so through ctypes.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache

from . import _lib
from .repeat_count_params import RepeatCountParams

__all__ = ["get_repeat_count", "get_ref_repeat_count"]


def _get_repeat_count(start_count: int, tr_seq: str, flank_left_seq: str, flank_right_seq: str, motif: str,
                      max_iters: int, local_search_range: int, step_size: int,
                      use_shortcuts: bool = False) -> tuple[tuple[int, int], int, int]:
    """Same signature and return value as strkit_rust_ext.get_repeat_count (call site repeats.py:58-68)."""
    ctx = _lib.default_context()
    cn, sc, n = C.c_int32(), C.c_int32(), C.c_int32()
    tr, fl, fr, mo = (s.encode("ascii") for s in (tr_seq, flank_left_seq, flank_right_seq, motif))
    rc = _lib.load().strk_repeat_count(ctx.handle, start_count, tr, len(tr), fl, len(fl), fr, len(fr), mo, len(mo),
                                       max_iters, local_search_range, step_size, C.byref(cn), C.byref(sc), C.byref(n))
    if rc == _lib.STRK_E_EMPTY:
        raise ValueError("max() arg is an empty sequence")
    _lib.check(rc)
    return (cn.value, sc.value), n.value, cn.value - start_count


@lru_cache(maxsize=512)  # repeats.py:47
def get_repeat_count(
    start_count: int,
    tr_seq: str,
    flank_left_seq: str,
    flank_right_seq: str,
    motif: str,
    rc_params: RepeatCountParams,
) -> tuple[tuple[int, int], int, int]:
    # returns: (best size, best score), n_explored, best size - start count
    if rc_params.method == "repalign":
        return _get_repeat_count(
            start_count,
            tr_seq,
            flank_left_seq,
            flank_right_seq,
            motif,
            rc_params.max_iters,
            rc_params.initial_local_search_range,
            rc_params.initial_step_size,
            use_shortcuts=False,
        )
    # repeats.py:69-70 routes "comp" to strkit_rust_ext.get_repeat_count_compostr, an experimental
    # k-mer heuristic outside the DP hot path this backend replaces.
    raise NotImplementedError("rc_method 'comp' is not provided by the GPU backend; use 'repalign'")


def get_ref_repeat_count(
    start_count: int,
    tr_seq: str,
    flank_left_seq: str,
    flank_right_seq: str,
    motif: str,
    ref_size: int,
    vcf_anchor_size: int,
    rc_params: RepeatCountParams,
    respect_coords: bool = False,
) -> tuple[tuple[int, int], int, int, tuple[int, int], tuple[str, str, str]]:
    """Reference-side count with boundary extension — same signature and return value as
    strkit/call/repeats.py:73-192: (final_res, l_offset, r_offset, (n_offset_scores, n_iters),
    (flank_left_seq, tr_seq, flank_right_seq)) with the flank/tract split adjusted by the offsets.
    The parasail profile alignments of score_ref_boundaries (repeats.py:23-43) run on the GPU."""
    if rc_params.method != "repalign":
        raise NotImplementedError("rc_method 'comp' is not provided by the GPU backend; use 'repalign'")
    ctx = _lib.default_context()
    tr, fl, fr, mo = (s.encode("ascii") for s in (tr_seq, flank_left_seq, flank_right_seq, motif))
    out = (C.c_int32 * 9)()
    rc = _lib.load().strk_ref_repeat_count(ctx.handle, start_count, tr, len(tr), fl, len(fl), fr, len(fr), mo, len(mo),
                                           ref_size, vcf_anchor_size, rc_params.max_iters,
                                           rc_params.initial_local_search_range, rc_params.initial_step_size,
                                           int(respect_coords), out)
    if rc == _lib.STRK_E_EMPTY:
        raise ValueError("max() arg is an empty sequence")
    _lib.check(rc)
    db = flank_left_seq + tr_seq + flank_right_seq
    nfl, ntr = out[6], out[7]
    # the reference upper-cases only what it hands to the final count (repeats.py:183); the returned
    # tract keeps the caller's case, as there
    return ((out[0], out[1]), out[2], out[3], (out[4], out[5]), (db[:nfl], db[nfl:nfl + ntr], db[nfl + ntr:]))
