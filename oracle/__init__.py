"""CPU oracle for the per-read repeat-count path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported CPU baseline.  ``strkit_amd`` (the
product) never imports it.  PARITY UNPINNED: see the header of ``strk_oracle.c``.

Thin ctypes wrapper over ``libstrk_oracle.so`` (built by ``oracle/Makefile``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstrk_oracle.so")

SG_ALL = 15
S1_BEG_FREE, S1_END_FREE, S2_BEG_FREE, S2_END_FREE = 1, 2, 4, 8
TIE_FIRST, TIE_LAST = 0, 1

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "strk_oracle.c")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "strk_simd.c")))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libstrk_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.strk_o_matrix.restype = C.POINTER(C.c_int8)
        L.strk_o_encode.restype = C.c_int
        L.strk_o_sg_align.restype = C.c_int32
        L.strk_o_sg_align.argtypes = [_u8p, C.c_int32, _u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]
        L.strk_o_candidate_score.restype = C.c_int32
        L.strk_o_candidate_score.argtypes = [_u8p, C.c_int32] * 4 + [C.c_int32, C.c_int32]
        L.strk_o_repeat_count.restype = C.c_int
        L.strk_o_repeat_count.argtypes = ([C.c_int32] + [_u8p, C.c_int32] * 4 + [C.c_int32] * 5 + [_i32p, _i32p, _i32p, _i64p])
        L.strk_o_count_locus.restype = C.c_int
        L.strk_o_count_locus.argtypes = ([C.c_int32, _u8p, _i64p, _i32p, _i32p, _i32p, _i32p, _u8p] + [C.c_int32] * 8
                                         + [_i32p, _i32p, _i32p, _i32p, _i64p])
        L.strk_o_score_ref_boundaries.restype = None
        L.strk_o_score_ref_boundaries.argtypes = [_u8p, C.c_int32] * 4 + [C.c_int32, C.c_int32, _i32p]
        L.strk_o_ref_repeat_count.restype = C.c_int
        L.strk_o_realign.argtypes = [_u8p, C.c_int32, _u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p,
                                     C.POINTER(C.c_uint32), C.c_int32]
        L.strk_o_realign.restype = C.c_int32
        L.strk_o_ref_repeat_count.argtypes = ([C.c_int32] + [_u8p, C.c_int32] * 4 + [C.c_int32] * 8 + [_i32p])
        L.strk_o_set_simd.restype = C.c_int
        L.strk_o_set_simd.argtypes = [C.c_int]
        L.strk_o_simd_available.restype = C.c_int
        L.strk_o_init()
        _lib = L
    return _lib


def _b(s) -> tuple:
    if isinstance(s, str):
        s = s.encode("ascii")
    a = np.frombuffer(bytes(s), dtype=np.uint8) if len(s) else np.zeros(0, np.uint8)
    buf = (C.c_uint8 * max(len(a), 1)).from_buffer_copy(a.tobytes() if len(a) else b"\0")
    return buf, len(a)


def matrix() -> np.ndarray:
    p = lib().strk_o_matrix()
    return np.ctypeslib.as_array(p, shape=(17, 17)).copy()


def encode(ch: str) -> int:
    return lib().strk_o_encode(ord(ch))


def sg_align(s1, s2, open_: int = 5, ext: int = 5, flags: int = SG_ALL):
    """(score, end1, end2); s1 is parasail's query/profile side, s2 the database side."""
    b1, n1 = _b(s1)
    b2, n2 = _b(s2)
    e1, e2 = C.c_int32(), C.c_int32()
    sc = lib().strk_o_sg_align(b1, n1, b2, n2, open_, ext, flags, C.byref(e1), C.byref(e2))
    return sc, e1.value, e2.value


def candidate_score(tr, fl, fr, motif, i: int, flags: int = SG_ALL) -> int:
    bt, nt = _b(tr); bl, nl = _b(fl); br, nr = _b(fr); bm, nm = _b(motif)
    return lib().strk_o_candidate_score(bt, nt, bl, nl, br, nr, bm, nm, i, flags)


def repeat_count(start_count: int, tr, fl, fr, motif, max_iters: int = 50, lsr: int = 3, step: int = 1,
                 tie_rule: int = TIE_FIRST, flags: int = SG_ALL, with_cells: bool = False, narrowing: int = 0):
    """Reference contract (repeats.py:55-56): ((cn, score), n_explored, cn - start_count).
    `narrowing`: how local_search_range changes inside the search (strk_oracle.c: 0 fixed, the default; 1 / 2 / 3)."""
    tie_rule = (tie_rule & 1) | ((narrowing & 3) << 8)
    bt, nt = _b(tr); bl, nl = _b(fl); br, nr = _b(fr); bm, nm = _b(motif)
    cn, sc, n, cells = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    rc = lib().strk_o_repeat_count(start_count, bt, nt, bl, nl, br, nr, bm, nm, max_iters, lsr, step, tie_rule,
                                   flags, C.byref(cn), C.byref(sc), C.byref(n), C.byref(cells))
    if rc:
        raise ValueError("max() arg is an empty sequence (nothing scored)")
    res = ((cn.value, sc.value), n.value, cn.value - start_count)
    return (res, cells.value) if with_cells else res


def count_locus(seqs: np.ndarray, off: np.ndarray, nfl: np.ndarray, ntr: np.ndarray, nfr: np.ndarray,
                est_cn: np.ndarray, motif, max_iters: int = 50, lsr: int = 3, step: int = 1,
                tie_rule: int = TIE_FIRST, flags: int = SG_ALL, feedback: bool = True, memo: bool = False, narrowing: int = 0):
    """One locus, reads in order with the caller's start-count feedback (call_locus.py:1125-1161).

    Returns dict of int32 arrays cn, score, n_iters, start and the DP cell count."""
    n = len(nfl)
    tie_rule = (tie_rule & 1) | ((narrowing & 3) << 8)
    seqs = np.ascontiguousarray(seqs, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    nfl = np.ascontiguousarray(nfl, np.int32); ntr = np.ascontiguousarray(ntr, np.int32)
    nfr = np.ascontiguousarray(nfr, np.int32); est_cn = np.ascontiguousarray(est_cn, np.int32)
    bm, nm = _b(motif)
    out = {k: np.zeros(n, np.int32) for k in ("cn", "score", "n_iters", "start")}
    cells = C.c_int64()
    p = lambda a, t: a.ctypes.data_as(t)
    if seqs.size == 0:
        seqs = np.zeros(1, np.uint8)
    rc = lib().strk_o_count_locus(n, p(seqs, _u8p), p(off, _i64p), p(nfl, _i32p), p(ntr, _i32p), p(nfr, _i32p),
                                  p(est_cn, _i32p), bm, nm, max_iters, lsr, step, tie_rule, flags, int(feedback), int(memo),
                                  p(out["cn"], _i32p), p(out["score"], _i32p), p(out["n_iters"], _i32p),
                                  p(out["start"], _i32p), C.byref(cells))
    if rc:
        raise ValueError("nothing scored")
    out["cells"] = cells.value
    return out


def score_ref_boundaries(db, fl, fr, motif, i: int, ref_size: int):
    """((fwd score, r_adj), (rev score, l_adj)) — repeats.py:23-43."""
    bd, nd = _b(db); bl, nl = _b(fl); br, nr = _b(fr); bm, nm = _b(motif)
    o = (C.c_int32 * 4)()
    lib().strk_o_score_ref_boundaries(bd, nd, bl, nl, br, nr, bm, nm, i, ref_size, o)
    return (o[0], o[1]), (o[2], o[3])


def ref_repeat_count(start_count: int, tr: str, fl: str, fr: str, motif: str, ref_size: int, vcf_anchor_size: int,
                     max_iters: int, lsr: int, step: int, respect_coords: bool = False,
                     tie_rule: int = TIE_FIRST, flags: int = SG_ALL):
    """Reference contract (repeats.py:190-192)."""
    bt, nt = _b(tr); bl, nl = _b(fl); br, nr = _b(fr); bm, nm = _b(motif)
    o = (C.c_int32 * 9)()
    rc = lib().strk_o_ref_repeat_count(start_count, bt, nt, bl, nl, br, nr, bm, nm, ref_size, vcf_anchor_size,
                                       max_iters, lsr, step, int(respect_coords), tie_rule, flags, o)
    if rc:
        raise ValueError("nothing scored")
    db = fl + tr + fr
    nfl2, ntr2, nfr2 = o[6], o[7], o[8]
    return ((o[0], o[1]), o[2], o[3], (o[4], o[5]),
            (db[:nfl2], db[nfl2:nfl2 + ntr2], db[nfl2 + ntr2:nfl2 + ntr2 + nfr2]))


def realign(s1, s2, open_: int = 7, ext: int = 0, gap_pref: int = 0):
    """parasail sg_dx_trace (realign.py:56-63): (score, end position in s2, CIGAR runs as BAM uint32)."""
    b1, n1 = _b(s1); b2, n2 = _b(s2)
    cap = 2 * n1 + 4
    cig = (C.c_uint32 * cap)()
    sc, e2 = C.c_int32(), C.c_int32()
    n = lib().strk_o_realign(b1, n1, b2, n2, open_, ext, gap_pref, C.byref(sc), C.byref(e2), cig, cap)
    if n < 0:
        raise ValueError("realign: empty input" if n == -2 else "realign: CIGAR capacity")
    return sc.value, e2.value, np.frombuffer(cig, dtype=np.uint32, count=n).copy()


def set_simd(on: bool) -> bool:
    """Candidate scores from the AVX2 inter-sequence pass (strk_simd.c) where it applies: same results, lower cost — the
    "simd" CPU baseline of bench.py.  Returns whether it is in effect (False on a CPU without AVX2)."""
    return bool(lib().strk_o_set_simd(int(on)))
