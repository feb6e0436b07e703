"""Per-read helpers of the caller protocol that feed the repeat counter.

In the reference these are methods of Rust objects that are not vendored
(``STRkitAlignedSegmentSequenceDataForLocus``, ``calculate_seq_with_wildcards``); only their call
sites are in the tree, so the two rules below are the documented reading of those call sites:

* ``calculate_seq_with_wildcards(seq, quals, base_wildcard_threshold)`` — bases whose PHRED quality is
  at or below the threshold (3: strkit/call/call_locus.py:79, passed at :1101-1106 and
  strkit/call/realign.py:86) become the wildcard ``X``, which scores 0 against any base
  (strkit/call/align_matrix.py:29,38-39; docs/caller_catalog.md:48-53).  Whether the Rust code
  compares with ``<=`` or ``<`` cannot be read from the tree; ``<=`` is assumed.
* ``get_est_copy_num()`` — the integer start estimate of a read, ``round(len(tr) / len(motif))``
  (same expression as the reference-side estimate at call_locus.py:796).
"""
from __future__ import annotations

import numpy as np

__all__ = ["calculate_seq_with_wildcards", "get_est_copy_num", "BASE_WILDCARD_THRESHOLD"]

BASE_WILDCARD_THRESHOLD = 3  # call_locus.py:79


def calculate_seq_with_wildcards(seq: str, quals, threshold: int = BASE_WILDCARD_THRESHOLD) -> str:
    if quals is None:
        return seq
    q = np.asarray(quals)
    if q.shape[0] != len(seq):
        raise ValueError("sequence and quality lengths differ")
    b = np.frombuffer(seq.encode("ascii"), np.uint8).copy()
    b[q <= threshold] = ord("X")
    return b.tobytes().decode("ascii")


def get_est_copy_num(tr_seq: str, motif: str) -> int:
    return round(len(tr_seq) / len(motif))
