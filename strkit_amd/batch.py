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

__all__ = ["count_loci", "score_table", "score_ref_table", "make_params", "batch_struct"]


def make_params(rc_params: RepeatCountParams | None = None, feedback: bool = True, window: int = 0,
                tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL,
                dedupe: bool = True, band: bool = True) -> _lib.StrkParams:
    rc = rc_params or default_read_rc_params()
    if rc.method != "repalign":
        raise NotImplementedError("only rc_method='repalign' runs on the GPU backend")
    return _lib.StrkParams(max_iters=rc.max_iters, local_search_range=rc.initial_local_search_range,
                           step_size=rc.initial_step_size, tie_rule=tie_rule, end_flags=end_flags,
                           feedback=int(feedback), window=window, no_dedupe=int(not dedupe), no_band=int(not band), reserved=0)


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


def count_loci(b: LocusBatch, rc_params: RepeatCountParams | None = None, feedback: bool = True, window: int = 0,
               tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL, ctx: _lib.Context | None = None,
               with_stats: bool = False, dedupe: bool = True, band: bool = True):
    """Per-read (cn, score, n_iters, start) for every read of the batch, as int32 arrays."""
    ctx = ctx or _lib.default_context()
    p = make_params(rc_params, feedback, window, tie_rule, end_flags, dedupe, band)
    s, keep = batch_struct(b)
    n = max(b.n_reads, 1)
    out = {k: np.zeros(n, np.int32) for k in ("cn", "score", "n_iters", "start")}
    st = _lib.StrkStats()
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
