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

__all__ = ["get_ref_repeat_counts", "get_ref_repeat_counts_packed", "get_repeat_count", "get_ref_repeat_count"]


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
    context: "_lib.Context | None" = None,
) -> tuple[tuple[int, int], int, int, tuple[int, int], tuple[str, str, str]]:
    """Reference-side count with boundary extension — same signature and return value as
    strkit/call/repeats.py:73-192: (final_res, l_offset, r_offset, (n_offset_scores, n_iters),
    (flank_left_seq, tr_seq, flank_right_seq)) with the flank/tract split adjusted by the offsets.
    The parasail profile alignments of score_ref_boundaries (repeats.py:23-43) run on the GPU."""
    if rc_params.method != "repalign":
        raise NotImplementedError("rc_method 'comp' is not provided by the GPU backend; use 'repalign'")
    ctx = context or _lib.default_context()   # `context`: an extension for callers that count loci from several threads
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


def get_ref_repeat_counts(
    loci: "list[tuple[int, str, str, str, str, int, RepeatCountParams]]",
    vcf_anchor_size: int,
    respect_coords: bool = False,
    context: "_lib.Context | None" = None,
) -> list:
    """``get_ref_repeat_count`` for a block of loci in one library call (``strk_ref_repeat_count_batch``): every round
    of boundary scoring is a single device launch for all loci.  ``loci`` holds
    ``(start_count, tr_seq, flank_left_seq, flank_right_seq, motif, ref_size, rc_params)`` per locus; the result list
    has the reference's return tuple (repeats.py:190-192) per locus."""
    import numpy as np
    n = len(loci)
    if n == 0:
        return []
    if any(rc.method != "repalign" for *_, rc in loci):
        raise NotImplementedError("rc_method 'comp' is not provided by the GPU backend; use 'repalign'")
    ctx = context or _lib.default_context()
    dbs = [(fl + tr + fr).encode("ascii") for _, tr, fl, fr, *_ in loci]
    mos = [x[4].encode("ascii") for x in loci]
    seq_off = np.zeros(n + 1, np.int64); np.cumsum([len(d) for d in dbs], out=seq_off[1:])
    motif_off = np.zeros(n + 1, np.int32); np.cumsum([len(m) for m in mos], out=motif_off[1:])
    seqs = np.frombuffer(b"".join(dbs) + b"\0", np.uint8)
    motifs = np.frombuffer(b"".join(mos) + b"\0", np.uint8)
    i32 = lambda v: np.ascontiguousarray(v, np.int32)  # noqa: E731
    start = i32([x[0] for x in loci]); ntr = i32([len(x[1]) for x in loci]); nfl = i32([len(x[2]) for x in loci])
    nfr = i32([len(x[3]) for x in loci]); ref_size = i32([x[5] for x in loci])
    mi = i32([x[6].max_iters for x in loci]); lsr = i32([x[6].initial_local_search_range for x in loci])
    st = i32([x[6].initial_step_size for x in loci])
    out = np.zeros(n * 9, np.int32)
    rc = _lib.load().strk_ref_repeat_count_batch(
        ctx.handle, n, start.ctypes.data, seqs.ctypes.data, seq_off.ctypes.data, nfl.ctypes.data, ntr.ctypes.data,
        nfr.ctypes.data, motifs.ctypes.data, motif_off.ctypes.data, ref_size.ctypes.data, int(vcf_anchor_size),
        mi.ctypes.data, lsr.ctypes.data, st.ctypes.data, int(respect_coords), out.ctypes.data)
    if rc == _lib.STRK_E_EMPTY:
        raise ValueError("max() arg is an empty sequence")
    _lib.check(rc)
    res = []
    for i, (_, tr, fl, fr, *_rest) in enumerate(loci):
        o = out[9 * i:9 * i + 9]
        db = fl + tr + fr
        a, b = int(o[6]), int(o[7])
        res.append(((int(o[0]), int(o[1])), int(o[2]), int(o[3]), (int(o[4]), int(o[5])), (db[:a], db[a:a + b], db[a + b:])))
    return res


def get_ref_repeat_counts_packed(start, seqs, seq_off, nfl, ntr, nfr, motifs, motif_off, ref_size, max_iters, lsr, step,
                                 vcf_anchor_size: int, respect_coords: bool = False, context: "_lib.Context | None" = None):
    """`get_ref_repeat_counts` on packed arrays (no Python object per locus): locus i owns ``seqs[seq_off[i]:seq_off[i+1]]``
    laid out fl|tr|fr and ``motifs[motif_off[i]:motif_off[i+1]]``; everything else is one value per locus.  Returns the
    int32 array ``out9[n, 9]`` of strk_ref_repeat_count_batch: cn, score, l_offset, r_offset, n_offset_scores,
    n_iters_final, new fl / tr / fr lengths."""
    import numpy as np
    n = int(len(start))
    out = np.zeros((n, 9), np.int32)
    if n == 0:
        return out
    ctx = context or _lib.default_context()
    i32 = lambda v: np.ascontiguousarray(v, np.int32)  # noqa: E731
    start, nfl, ntr, nfr, ref_size, max_iters, lsr, step = (i32(v) for v in (start, nfl, ntr, nfr, ref_size, max_iters, lsr, step))
    seqs = np.ascontiguousarray(seqs, np.uint8)
    motifs = np.ascontiguousarray(motifs, np.uint8)
    seq_off = np.ascontiguousarray(seq_off, np.int64)
    motif_off = i32(motif_off)
    rc = _lib.load().strk_ref_repeat_count_batch(
        ctx.handle, n, start.ctypes.data, seqs.ctypes.data, seq_off.ctypes.data, nfl.ctypes.data, ntr.ctypes.data,
        nfr.ctypes.data, motifs.ctypes.data, motif_off.ctypes.data, ref_size.ctypes.data, int(vcf_anchor_size),
        max_iters.ctypes.data, lsr.ctypes.data, step.ctypes.data, int(respect_coords), out.ctypes.data)
    if rc == _lib.STRK_E_EMPTY:
        raise ValueError("max() arg is an empty sequence")
    _lib.check(rc)
    return out
