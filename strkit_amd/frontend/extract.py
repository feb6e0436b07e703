"""From an aligned read to the (left flank, tract, right flank) triple the repeat counter scores.

Restates, from their call sites, three pieces of `strkit_rust_ext` that are not in the reference tree:
`segment.aligned_coords` / `get_read_coords_from_matched_pairs` (call_locus.py:875-877,929-931),
`find_pair_by_ref_pos` (its only in-tree statement are the vectors in tests/test_caller_utils.py:1-11), and
`segment.get_sequence_data_for_locus(locus, data, min_avg_phred, base_wildcard_threshold)` (call_locus.py:1101-1115).

Choices the call sites leave open (all in one place so they can be flipped):
* a boundary coordinate that falls inside a deletion maps to the next aligned base (lower bound, as the test vectors
  of find_pair_by_ref_pos show: 1007 -> index 6, not found);
* bases inserted exactly at a tract boundary belong to the TRACT: the left flank ends one past the last read base
  aligned left of the tract, the right flank starts at the first read base aligned at or right of its end;
* the mean base quality that can raise LowMeanBaseQual is that of the tract bases only (comment at
  call_locus.py:1099).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..segment import BASE_WILDCARD_THRESHOLD, calculate_seq_with_wildcards

__all__ = ["LocusReadCoords", "LowMeanBaseQual", "find_pair_by_ref_pos", "get_aligned_pairs", "CigarIndex",
           "get_read_coords_from_matched_pairs", "get_read_coords_from_cigar", "get_sequence_data_for_locus", "SequenceDataForLocus",
           "MIN_AVG_PHRED", "FLANK_LEEWAY"]

MIN_AVG_PHRED = 13   # params.min_avg_phred default (strkit/call/params.py)
FLANK_LEEWAY = 10    # "+10" of the comment at call_locus.py:1092-1096

_CONSUMES_QUERY = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1], bool)
_CONSUMES_REF = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1], bool)
_ALIGNED = np.array([1, 0, 0, 0, 0, 0, 0, 1, 1], bool)


class LowMeanBaseQual(Exception):
    def __init__(self, mean_base_qual: float):
        super().__init__(f"low mean base quality: {mean_base_qual:.2f}")
        self.mean_base_qual = mean_base_qual


def get_aligned_pairs(segment) -> tuple[np.ndarray, np.ndarray]:
    """(query_coords, ref_coords) of the aligned (M, =, X) columns — pysam get_aligned_pairs(matches_only=True)."""
    ops = (segment.cigar & 15).astype(np.int64)
    lens = (segment.cigar >> 4).astype(np.int64)
    q0 = np.concatenate(([0], np.cumsum(lens * _CONSUMES_QUERY[ops])[:-1]))
    r0 = segment.start + np.concatenate(([0], np.cumsum(lens * _CONSUMES_REF[ops])[:-1]))
    al = _ALIGNED[ops]
    ln = lens[al]
    if ln.size == 0:
        e = np.zeros(0, np.int64)
        return e, e.copy()
    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    return np.repeat(q0[al], ln) + within, np.repeat(r0[al], ln) + within


class CigarIndex:
    """The aligned (M, =, X) runs of one segment: pair lookups without expanding every aligned pair.

    Pair index k counts aligned columns from the start of the read's alignment; `first_pair_at_or_after(c)` is
    find_pair_by_ref_pos on the expanded pair list."""

    __slots__ = ("q0", "r0", "ln", "k0", "n_pairs")

    def __init__(self, segment):
        ops = (segment.cigar & 15).astype(np.int64)
        lens = (segment.cigar >> 4).astype(np.int64)
        q0 = np.concatenate(([0], np.cumsum(lens * _CONSUMES_QUERY[ops])[:-1]))
        r0 = segment.start + np.concatenate(([0], np.cumsum(lens * _CONSUMES_REF[ops])[:-1]))
        al = _ALIGNED[ops]
        self.q0, self.r0, self.ln = q0[al], r0[al], lens[al]
        self.k0 = np.concatenate(([0], np.cumsum(self.ln)[:-1])) if self.ln.size else np.zeros(0, np.int64)
        self.n_pairs = int(self.ln.sum())

    def first_ref(self) -> int:
        return int(self.r0[0])

    def last_ref(self) -> int:
        return int(self.r0[-1] + self.ln[-1] - 1)

    def first_pair_at_or_after(self, c: int) -> int:
        """Index of the first aligned pair whose reference coordinate is >= c (n_pairs if none)."""
        i = int(np.searchsorted(self.r0 + self.ln, c, side="right"))   # first run that ends past c
        if i >= self.ln.size:
            return self.n_pairs
        return int(self.k0[i] + max(0, c - int(self.r0[i])))

    def query_at(self, k: int) -> int:
        """Read position of aligned pair k."""
        i = int(np.searchsorted(self.k0, k, side="right")) - 1
        return int(self.q0[i] + (k - int(self.k0[i])))


def find_pair_by_ref_pos(ref_coords, target: int) -> tuple[int, bool]:
    """(index of the first pair whose reference coordinate is >= target, whether it equals target)."""
    idx = int(np.searchsorted(ref_coords, target, side="left"))
    return idx, bool(idx < len(ref_coords) and ref_coords[idx] == target)


@dataclass
class LocusReadCoords:
    left_flank_start: int = -1
    left_flank_end: int = -1
    right_flank_start: int = -1
    right_flank_end: int = -1
    full_left_flank: bool = False
    full_right_flank: bool = False

    def is_incomplete(self) -> bool:
        return min(self.left_flank_start, self.left_flank_end, self.right_flank_start, self.right_flank_end) < 0


def get_read_coords_from_matched_pairs(left_flank_coord: int, left_coord: int, right_coord: int, right_flank_coord: int,
                                       query_coords: np.ndarray, ref_coords: np.ndarray,
                                       allow_only_one_full_flank: bool = False) -> LocusReadCoords:
    """Read positions of the four locus boundaries.  The read must have aligned bases at or left of the left flank
    start and at or right of the right flank end, else the coordinates are incomplete (the early-out of
    call_locus.py:907-922) unless one partial flank is allowed."""
    out = LocusReadCoords()
    n = len(ref_coords)
    if n == 0:
        return out
    out.full_left_flank = bool(ref_coords[0] <= left_flank_coord)
    # call_locus.py:907-909: skipped when right_flank_coord >= segment.end, i.e. the last aligned base (end - 1) must
    # lie at or right of right_flank_coord
    out.full_right_flank = bool(ref_coords[-1] >= right_flank_coord)
    if not (out.full_left_flank and out.full_right_flank):
        if not allow_only_one_full_flank or not (out.full_left_flank or out.full_right_flank):
            return out
    i_lfs, _ = find_pair_by_ref_pos(ref_coords, left_flank_coord)
    i_l, _ = find_pair_by_ref_pos(ref_coords, left_coord)          # first pair at or right of the tract start
    i_r, _ = find_pair_by_ref_pos(ref_coords, right_coord)         # first pair at or right of the tract end
    i_rfe, _ = find_pair_by_ref_pos(ref_coords, right_flank_coord)
    if i_l == 0 or i_r >= n:
        return out                                                 # nothing aligned on one side of the tract
    out.left_flank_start = int(query_coords[min(i_lfs, n - 1)])
    out.left_flank_end = int(query_coords[i_l - 1]) + 1            # boundary insertions go to the tract
    out.right_flank_start = int(query_coords[i_r])
    out.right_flank_end = int(query_coords[i_rfe]) if i_rfe < n else int(query_coords[-1]) + 1
    if out.left_flank_end > out.right_flank_start:                 # tract deleted entirely and boundaries crossed
        out.right_flank_start = out.left_flank_end
    return out


@dataclass
class SequenceDataForLocus:
    flank_left_seq_wc: str
    tr_seq_wc: str
    flank_right_seq_wc: str
    tr_seq: str
    tr_len_with_flank: int

    def get_est_copy_num(self, motif_size: int) -> int:
        return round(len(self.tr_seq_wc) / motif_size)      # same expression as call_locus.py:796 on the read tract


def get_read_coords_from_cigar(left_flank_coord: int, left_coord: int, right_coord: int, right_flank_coord: int,
                               segment, allow_only_one_full_flank: bool = False) -> LocusReadCoords:
    """get_read_coords_from_matched_pairs on the segment's own alignment, computed from the CIGAR runs (same result,
    tests/test_frontend.py checks them against each other)."""
    out = LocusReadCoords()
    ix = CigarIndex(segment)
    n = ix.n_pairs
    if n == 0:
        return out
    out.full_left_flank = ix.first_ref() <= left_flank_coord
    out.full_right_flank = ix.last_ref() >= right_flank_coord      # call_locus.py:907-909, as above
    if not (out.full_left_flank and out.full_right_flank):
        if not allow_only_one_full_flank or not (out.full_left_flank or out.full_right_flank):
            return out
    i_lfs = ix.first_pair_at_or_after(left_flank_coord)
    i_l = ix.first_pair_at_or_after(left_coord)
    i_r = ix.first_pair_at_or_after(right_coord)
    i_rfe = ix.first_pair_at_or_after(right_flank_coord)
    if i_l == 0 or i_r >= n:
        return out
    out.left_flank_start = ix.query_at(min(i_lfs, n - 1))
    out.left_flank_end = ix.query_at(i_l - 1) + 1
    out.right_flank_start = ix.query_at(i_r)
    out.right_flank_end = ix.query_at(i_rfe) if i_rfe < n else ix.query_at(n - 1) + 1
    if out.left_flank_end > out.right_flank_start:
        out.right_flank_start = out.left_flank_end
    return out


def get_sequence_data_for_locus(segment, coords: LocusReadCoords, flank_size: int, min_avg_phred: int = MIN_AVG_PHRED,
                                base_wildcard_threshold: int = BASE_WILDCARD_THRESHOLD) -> SequenceDataForLocus:
    qs, quals = segment.query_sequence, segment.query_qualities
    a, b, c, d = coords.left_flank_start, coords.left_flank_end, coords.right_flank_start, coords.right_flank_end
    if quals is not None and c > b:
        mean_q = float(np.mean(quals[b:c]))
        if mean_q < min_avg_phred:
            raise LowMeanBaseQual(mean_q)
    keep = flank_size + FLANK_LEEWAY
    a2, d2 = max(a, b - keep), min(d, c + keep)     # only the part that is kept needs wildcards
    wc = calculate_seq_with_wildcards(qs[a2:d2], None if quals is None else quals[a2:d2], base_wildcard_threshold)
    fl, tr, fr = wc[:b - a2], wc[b - a2:c - a2], wc[c - a2:]
    return SequenceDataForLocus(fl, tr, fr, qs[b:c], len(fl) + (c - b) + len(fr))
