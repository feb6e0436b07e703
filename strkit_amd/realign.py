"""Read realignment — host mirror of strkit/call/realign.py over the HIP backend.

Same entry points as the reference module: ``realign_read`` (realign.py:34-72) and ``perform_realign``
(realign.py:75-154), plus ``realign_reads`` — the batched form a worker uses to realign every
soft-clipped read of a block of loci in one device call.  The parasail call
``sg_dx_trace_scan_16(ref_seq, query_seq, 7, 0, dna_matrix)`` (realign.py:56-63) and the CIGAR behind
``pr.cigar.seq`` come from ``strk_realign`` (include/strkit_amd.h); ``get_aligned_pair_matches`` mirrors
the ``strkit_rust_ext`` helper of the same name called at realign.py:71.

There is no CPU path: without the HIP library / a gfx950 device every function here raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .segment import calculate_seq_with_wildcards

__all__ = ["realign_read", "perform_realign", "realign_reads", "realign_pairs", "get_aligned_pair_matches",
           "AlignedCoords", "cigar_to_string", "i16_saturation_flags"]

match_score: int = 2                     # strkit/call/align_matrix.py:15
min_realign_score_ratio: float = 0.95    # realign.py:28
realign_indel_open_penalty: int = 7      # realign.py:29
max_ref_len_for_same_proc: int = 2000    # realign.py:30 (kept for interface parity; the device path never spawns)
max_read_len_for_same_proc: int = 25000  # realign.py:31

CIGAR_OPS = "MIDNSHP=X"
_OP_I, _OP_D, _OP_EQ, _OP_X, _OP_M = 1, 2, 7, 8, 0


class AlignedCoords:
    """Matched (query, ref) coordinate pairs — the STRkitAlignedCoords the reference passes on to
    get_read_coords_from_matched_pairs (strkit/call/call_locus.py:875-877)."""

    __slots__ = ("query_coords", "ref_coords")

    def __init__(self, query_coords: np.ndarray, ref_coords: np.ndarray):
        self.query_coords = query_coords
        self.ref_coords = ref_coords

    def __len__(self) -> int:
        return int(self.query_coords.shape[0])

    def pair_at_idx(self, idx: int) -> tuple[int, int]:
        return int(self.query_coords[idx]), int(self.ref_coords[idx])

    def __repr__(self) -> str:
        return f"AlignedCoords(n={len(self)})"


def cigar_to_string(cigar: np.ndarray) -> str:
    """pr.cigar.decode of the reference's debug line (realign.py:69)."""
    return "".join(f"{int(x) >> 4}{CIGAR_OPS[int(x) & 15]}" for x in cigar)


def get_aligned_pair_matches(cigar: np.ndarray, query_start: int, ref_start: int, swap: bool = False) -> AlignedCoords:
    """Coordinates of the aligned (M, =, X) columns of a CIGAR whose first op sits at (query_start, ref_start).

    The reference calls it with the reference window as parasail's "query" and swap=True (realign.py:71),
    so that the result's query coordinates are read positions and its ref coordinates are genome positions."""
    cigar = np.asarray(cigar, dtype=np.uint32)
    ops = (cigar & 15).astype(np.int64)
    lens = (cigar >> 4).astype(np.int64)
    adv_q = np.isin(ops, (_OP_M, _OP_I, _OP_EQ, _OP_X, 4)) * lens
    adv_r = np.isin(ops, (_OP_M, _OP_D, _OP_EQ, _OP_X, 3)) * lens
    q0 = query_start + np.concatenate(([0], np.cumsum(adv_q)[:-1]))
    r0 = ref_start + np.concatenate(([0], np.cumsum(adv_r)[:-1]))
    al = np.isin(ops, (_OP_M, _OP_EQ, _OP_X))
    if not al.any():
        e = np.zeros(0, dtype=np.int64)
        return AlignedCoords(e, e.copy())
    ln = lens[al]
    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    qc = np.repeat(q0[al], ln) + within
    rc = np.repeat(r0[al], ln) + within
    return AlignedCoords(rc, qc) if swap else AlignedCoords(qc, rc)


def _pack(seqs: Sequence[str | bytes]) -> tuple[np.ndarray, np.ndarray]:
    bs = [s.encode("ascii") if isinstance(s, str) else bytes(s) for s in seqs]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off


def realign_pairs(ref_seqs: Sequence[str | bytes], query_seqs: Sequence[str | bytes],
                  open_penalty: int = realign_indel_open_penalty, extend_penalty: int = 0, gap_pref: int = 0,
                  with_stats: bool = False, context: _lib.Context | None = None):
    """Batched parasail ``sg_dx_trace`` (realign.py:56-63): per pair ``(score, end_ref, cigar)`` where cigar is the
    BAM-encoded uint32 array of ``pr.cigar.seq`` and end_ref the read position of the last aligned base."""
    if len(ref_seqs) != len(query_seqs):
        raise ValueError("ref_seqs and query_seqs differ in length")
    n = len(ref_seqs)
    if n == 0:
        return ([], {}) if with_stats else []
    ctx = context or _lib.default_context()
    s1, o1 = _pack(ref_seqs)
    s2, o2 = _pack(query_seqs)
    cig_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(2 * np.diff(o1) + 4, out=cig_off[1:])
    cig = np.zeros(int(cig_off[-1]), dtype=np.uint32)
    score = np.zeros(n, dtype=np.int32)
    end_ref = np.zeros(n, dtype=np.int32)
    ncig = np.zeros(n, dtype=np.int32)
    st = _lib.StrkStats()
    _lib.check(ctx._lib.strk_realign(ctx.handle, n, s1.ctypes.data, o1.ctypes.data, s2.ctypes.data, o2.ctypes.data,
                                     int(open_penalty), int(extend_penalty), int(gap_pref), score.ctypes.data,
                                     end_ref.ctypes.data, ncig.ctypes.data, cig.ctypes.data, cig_off.ctypes.data,
                                     C.byref(st)))
    out = [(int(score[p]), int(end_ref[p]), cig[cig_off[p]:cig_off[p] + ncig[p]].copy()) for p in range(n)]
    return (out, st.as_dict()) if with_stats else out


def i16_saturation_flags(ref_seqs: Sequence[str | bytes], query_seqs: Sequence[str | bytes], scores: Sequence[int]) -> np.ndarray:
    """What the reference's fixed 16-bit parasail kernel (``sg_dx_trace_scan_16``, realign.py:56) would have done with these
    pairs — ``strk_realign`` computes in 32 bits and never saturates: per pair 0, ``STRK_I16_CELL_MAY_SATURATE`` (an
    intermediate cell can reach the 16-bit limit; the score itself fits) or that ``| STRK_I16_SCORE_SATURATES`` (the score
    does not fit: the reference would compare a saturated result with its threshold)."""
    n = len(scores)
    o1 = np.zeros(n + 1, np.int64)
    o2 = np.zeros(n + 1, np.int64)
    np.cumsum([len(x) for x in ref_seqs], out=o1[1:])
    np.cumsum([len(x) for x in query_seqs], out=o2[1:])
    sc = np.ascontiguousarray(scores, np.int32)
    out = np.zeros(max(n, 1), np.int32)
    _lib.check(_lib.load().strk_realign_i16_flags(n, o1.ctypes.data, o2.ctypes.data, sc.ctypes.data, out.ctypes.data))
    return out[:n]


def _gate(flank_size: int) -> float:
    return min_realign_score_ratio * (flank_size * 2 * match_score - realign_indel_open_penalty)   # realign.py:65


def realign_reads(ref_seqs: Sequence[str], query_seqs: Sequence[str], left_flank_coords: Sequence[int], flank_size: int,
                  context: _lib.Context | None = None, drop_i16_saturated: bool = False) -> list[AlignedCoords | None]:
    """``realign_read`` for many (reference window, wildcarded read) pairs in one device call.

    The reference compares whatever score ``sg_dx_trace_scan_16`` returns with its threshold and goes on (realign.py:56-72): it
    has no saturation check, and a saturated 16-bit score is still far above the threshold.  The default therefore KEEPS a pair
    whose score does not fit 16 bits (a window of more than 16 kb) with the exact 32-bit alignment computed here;
    ``drop_i16_saturated=True`` leaves such pairs un-realigned instead (what a caller does who does not want results the
    reference could only have produced from saturated cells; ``i16_saturation_flags`` tells which pairs those are)."""
    res = realign_pairs(ref_seqs, query_seqs, context=context)
    th = _gate(flank_size)
    sat = i16_saturation_flags(ref_seqs, query_seqs, [sc for sc, _, _ in res]) if (res and drop_i16_saturated) else [0] * len(res)
    return [None if (sc < th or (f & _lib.STRK_I16_SCORE_SATURATES)) else get_aligned_pair_matches(cg, int(lfc), 0, swap=True)
            for (sc, _, cg), lfc, f in zip(res, left_flank_coords, sat)]


def realign_read(ref_seq: str, query_seq: str, left_flank_coord: int, flank_size: int, q=None, read_log_str: str = "",
                 log_level: int = 0) -> AlignedCoords | None:
    """strkit/call/realign.py:34-72, same arguments.  ``q`` (a multiprocessing queue) is honoured for interface
    parity; the device path needs no helper process or timeout."""
    res = realign_reads([ref_seq], [query_seq], [left_flank_coord], flank_size)[0]
    if q:
        q.put(res)
        q.close()
    return res


def perform_realign(locus_with_ref_data, segment, params, logger_=None) -> AlignedCoords | None:
    """strkit/call/realign.py:75-154.  Takes the same objects by duck typing: ``locus_with_ref_data.ref_total_seq``,
    ``.locus_def.left_flank_coord``; ``segment.query_sequence`` / ``.query_qualities``; ``params.flank_size``."""
    qs_wc = calculate_seq_with_wildcards(segment.query_sequence, segment.query_qualities, 3)   # realign.py:86
    return realign_read(locus_with_ref_data.ref_total_seq, qs_wc, locus_with_ref_data.locus_def.left_flank_coord,
                        params.flank_size, None, getattr(segment, "name", ""), getattr(params, "log_level", 0))
