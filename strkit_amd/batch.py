"""Batched per-locus repeat counting on the GPU (host side of strk_count_loci / strk_score_table).

This is the data-parallel form of the reference's per-read loop (strkit/call/call_locus.py:1082-1161):
all reads of a shard of loci go to the device in one CSR-packed batch; the in-order start-count
feedback inside a locus (call_locus.py:1129-1136,1161) is honoured on the device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .repeat_count_params import RepeatCountParams, default_read_rc_params
from .synth import LocusBatch

__all__ = ["count_loci", "score_table", "score_ref_table", "make_params", "batch_struct", "calc_adj_score", "filter_reads"]


def make_params(rc_params: RepeatCountParams | None = None, feedback: bool = True, window: int = 0,
                tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL,
                dedupe: bool = True, band: bool = True, narrowing: int = _lib.STRK_NARROW_NONE) -> _lib.StrkParams:
    rc = rc_params or default_read_rc_params()
    if rc.method != "repalign":
        raise NotImplementedError("only rc_method='repalign' runs on the GPU backend")
    return _lib.StrkParams(max_iters=rc.max_iters, local_search_range=rc.initial_local_search_range,
                           step_size=rc.initial_step_size, tie_rule=tie_rule, end_flags=end_flags,
                           feedback=int(feedback), window=window, no_dedupe=int(not dedupe), no_band=int(not band), narrowing=int(narrowing))


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def batch_struct(b: LocusBatch):
    """(StrkBatch over host memory, keep-alive list of the contiguous arrays it points into)."""
    keep = dict(
        seqs=np.ascontiguousarray(b.seqs, np.uint8) if b.seqs.size else np.zeros(1, np.uint8),
        seq_off=np.ascontiguousarray(b.seq_off, np.int64), nfl=np.ascontiguousarray(b.nfl, np.int32),
        ntr=np.ascontiguousarray(b.ntr, np.int32), nfr=np.ascontiguousarray(b.nfr, np.int32),
        est_cn=np.ascontiguousarray(b.est_cn, np.int32), read_off=np.ascontiguousarray(b.read_off, np.int32),
        motifs=np.ascontiguousarray(b.motifs, np.uint8) if b.motifs.size else np.zeros(1, np.uint8),
        motif_off=np.ascontiguousarray(b.motif_off, np.int32))
    s = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: _ptr(v) for k, v in keep.items()})
    return s, keep


def pin_batch(b: LocusBatch) -> LocusBatch:
    """Page-locks the bases of a batch in place (include/strkit_amd.h: strk_host_register), so that `count_loci` uploads them
    by DMA from where they lie instead of staging them through the library's pinned blocks.  For batches whose arrays are
    reused — registering costs a few milliseconds per 100 MB.  Undo with `unpin_batch` before the array is dropped."""
    b.seqs = np.ascontiguousarray(b.seqs, np.uint8)
    _lib.host_register(b.seqs)
    return b


def unpin_batch(b: LocusBatch) -> None:
    if _lib.host_is_pinned(b.seqs):
        _lib.host_unregister(b.seqs)


def count_loci(b: LocusBatch, rc_params: RepeatCountParams | None = None, feedback: bool = True, window: int = 0,
               tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL, ctx: _lib.Context | None = None,
               with_stats: bool = False, dedupe: bool = True, band: bool = True, narrowing: int = _lib.STRK_NARROW_NONE):
    """Per-read (cn, score, n_iters, start) for every read of the batch, as int32 arrays."""
    ctx = ctx or _lib.default_context()
    p = make_params(rc_params, feedback, window, tie_rule, end_flags, dedupe, band, narrowing)
    s, keep = batch_struct(b)
    n = max(b.n_reads, 1)
    out = {k: np.zeros(n, np.int32) for k in ("cn", "score", "n_iters", "start")}
    st = _lib.StrkStats()
    d_seqs = getattr(b, "d_seqs", None)      # the bases are in device memory already (frontend.native.DeviceBam)
    if d_seqs:
        rc = _lib.load().strk_count_loci_dseqs(ctx.handle, C.byref(s), C.c_void_p(d_seqs), C.byref(p), _ptr(out["cn"]), _ptr(out["score"]),
                                               _ptr(out["n_iters"]), _ptr(out["start"]), C.byref(st))
    else:
        rc = _lib.load().strk_count_loci(ctx.handle, C.byref(s), C.byref(p), _ptr(out["cn"]), _ptr(out["score"]),
                                         _ptr(out["n_iters"]), _ptr(out["start"]), C.byref(st))
    del keep
    if rc == _lib.STRK_E_EMPTY:
        raise ValueError("max() arg is an empty sequence")  # what the reference's max() raises
    _lib.check(rc)
    out = {k: v[:b.n_reads] for k, v in out.items()}
    return (out, st.as_dict()) if with_stats else out


def score_table(b: LocusBatch, lo, n, end_flags: int = _lib.STRK_SG_ALL, force_generic: bool = False,
                ctx: _lib.Context | None = None, with_stats: bool = False):
    """scores[r] = int32 array of the semi-global scores of candidate sizes lo[r] .. lo[r]+n[r]-1."""
    ctx = ctx or _lib.default_context()
    lo = np.ascontiguousarray(lo, np.int32)
    n = np.ascontiguousarray(n, np.int32)
    off = np.zeros(b.n_reads + 1, np.int64)
    np.cumsum(n, out=off[1:])
    flat = np.zeros(max(int(off[-1]), 1), np.int32)
    s, keep = batch_struct(b)
    st = _lib.StrkStats()
    _lib.check(_lib.load().strk_score_table(ctx.handle, C.byref(s), _ptr(lo), _ptr(n), _ptr(off), end_flags,
                                            int(force_generic), _ptr(flat), C.byref(st)))
    del keep
    res = [flat[off[r]:off[r + 1]] for r in range(b.n_reads)]
    return (res, st.as_dict()) if with_stats else res


def score_ref_table(b: LocusBatch, lo, n, force_generic: bool = False, ctx: _lib.Context | None = None):
    """Reference-side scoring (score_ref_boundaries, strkit/call/repeats.py:23-43): per read r two int32
    arrays (score, end_query) of the candidate fl + motif*i, i = lo[r] .. lo[r]+n[r]-1, against the
    window fl+tr+fr with a free end.  For the reversed alignment pass the reversed window."""
    ctx = ctx or _lib.default_context()
    lo = np.ascontiguousarray(lo, np.int32)
    n = np.ascontiguousarray(n, np.int32)
    off = np.zeros(b.n_reads + 1, np.int64)
    np.cumsum(n, out=off[1:])
    sc = np.zeros(max(int(off[-1]), 1), np.int32)
    eq = np.zeros(max(int(off[-1]), 1), np.int32)
    s, keep = batch_struct(b)
    st = _lib.StrkStats()
    _lib.check(_lib.load().strk_score_ref_table(ctx.handle, C.byref(s), _ptr(lo), _ptr(n), _ptr(off), int(force_generic),
                                                _ptr(sc), _ptr(eq), C.byref(st)))
    del keep
    return [(sc[off[r]:off[r + 1]], eq[off[r]:off[r + 1]]) for r in range(b.n_reads)]


# ---- what the caller does with the four integers (strkit/call/call_locus.py:1172,1222-1252,1279-1283) ----------
MIN_READ_ALIGN_SCORE = 0.1          # params.min_read_align_score default (strkit/call/params.py)
EXTREMELY_LOW_READ_ADJ_SCORE = 0.1  # call_locus.py:75
MAX_TERRIBLE_READS = 3              # params.max_terrible_reads default (call_locus.py:1236)


def calc_adj_score(score, nfl, ntr, nfr):
    """``STRkitAlignedSegmentSequenceDataForLocus.calc_adj_score`` (call_locus.py:1172): the alignment score per base
    of the scored window (flanks + tract), ``None`` (NaN here) when the read holds no tract.  The Rust original is
    not in the tree; a perfect read scores 2.0 as in docs/output_formats.md:102."""
    score = np.asarray(score, np.float64)
    total = np.asarray(nfl, np.int64) + np.asarray(ntr, np.int64) + np.asarray(nfr, np.int64)
    adj = np.full(score.shape, np.nan)
    ok = np.asarray(ntr) > 0
    adj[ok] = score[ok] / total[ok]
    return adj


def filter_reads(b: LocusBatch, res: dict, min_read_align_score: float = MIN_READ_ALIGN_SCORE,
                 max_terrible_reads: int = MAX_TERRIBLE_READS) -> dict:
    """Per-read filter of the caller's loop, vectorised over a batch (call_locus.py:1222-1252):

    * a read whose adjusted score is below ``min_read_align_score`` is skipped (``keep[r] = False``);
    * such a read is "extremely poor" when its adjusted score is also below 0.1 (call_locus.py:75,1234; the
      wall-clock half of that test never fires at GPU speeds); once a locus has seen MORE than
      ``max_terrible_reads`` of them, in read order, it is not called (``locus_ok[l] = False``) and the reads
      from that point on are not kept — reads before it were already recorded by the reference's loop.

    Returns ``{"sc": adjusted scores (NaN = None), "keep": bool per read, "locus_ok": bool per locus}`` — ``sc`` and
    ``res["cn"]`` of the kept reads are the ``read_dict`` entries of call_locus.py:1279-1283."""
    adj = calc_adj_score(res["score"], b.nfl, b.ntr, b.nfr)
    low = ~np.isnan(adj) & (adj < min_read_align_score)
    terrible = low & (adj < EXTREMELY_LOW_READ_ADJ_SCORE)
    keep = ~low
    locus_ok = np.ones(b.n_loci, bool)
    # running count of terrible reads inside each locus, in read order
    csum = np.cumsum(terrible)
    base = np.concatenate(([0], csum))[np.asarray(b.read_off[:-1], np.int64)]
    read_locus = np.repeat(np.arange(b.n_loci), np.diff(b.read_off))
    running = csum - base[read_locus]
    voided = running > max_terrible_reads            # true from the read that tipped the locus over
    keep &= ~voided
    locus_ok[np.unique(read_locus[voided])] = False
    return {"sc": adj, "keep": keep, "locus_ok": locus_ok}
