"""Pure-Python restatement (small cases only) — TEST INFRASTRUCTURE ONLY.

A second, independent statement of the same algorithm as ``strk_oracle.c``, kept as close as
possible to the shape of the reference's Python so that native ``dict`` insertion order,
``list.pop()`` and ``max()`` first-maximum semantics are the real ones, not re-implementations:

* ``dna_matrix``            follows strkit/call/align_matrix.py:15-44 + strkit/iupac.py:9-21
* ``sg_align``              parasail semi-global Gotoh semantics (third-party, not in tree)
* ``get_repeat_count``      contract strkit/call/repeats.py:47-70, search shape repeats.py:100-151
* ``score_ref_boundaries``  strkit/call/repeats.py:23-43
* ``get_ref_repeat_count``  strkit/call/repeats.py:73-192
* ``count_locus``           strkit/call/call_locus.py:1125-1161

PARITY UNPINNED (see strk_oracle.c).  O(n*m) Python loops: keep inputs to a few hundred bases.
"""
from __future__ import annotations

from operator import itemgetter

match_score = 2
mismatch_penalty = 7
indel_penalty = 5

IUPAC = {
    "R": ("A", "G"), "Y": ("C", "T"), "S": ("C", "G"), "W": ("A", "T"), "K": ("G", "T"), "M": ("A", "C"),
    "B": ("C", "G", "T"), "D": ("A", "C", "T"),  # sic: the reference's D equals H (iupac.py:17)
    "H": ("A", "C", "T"), "V": ("A", "C", "G"), "N": ("A", "C", "G", "T"),
}
dna_bases_str = "ACGT" + "".join(IUPAC.keys()) + "X"
dna_bases = {b: i for i, b in enumerate(dna_bases_str)}
dna_codes = {**IUPAC, "X": ("A", "C", "G", "T")}
STAR = len(dna_bases_str)  # parasail's implicit '*' index


def _create_dna_matrix():
    n = len(dna_bases_str)
    m = [[(match_score if i == j else -mismatch_penalty) if (i < n and j < n) else 0 for j in range(n + 1)]
         for i in range(n + 1)]
    for code, code_matches in dna_codes.items():
        for cm in code_matches:
            m[dna_bases[code]][dna_bases[cm]] = 2 if code != "X" else 0
            m[dna_bases[cm]][dna_bases[code]] = 2 if code != "X" else 0
    return m


dna_matrix = _create_dna_matrix()


def _enc(s: str) -> list[int]:
    return [dna_bases.get(ch.upper(), STAR) for ch in s]


def sg_align(s1: str, s2: str, open_: int, ext: int, s1_beg: bool, s1_end: bool, s2_beg: bool, s2_end: bool):
    """Returns (score, end1, end2); s1 = parasail query/profile side (rows), s2 = database side."""
    a, b = _enc(s1), _enc(s2)
    n1, n2 = len(a), len(b)
    NEG = -(10 ** 9)
    H = [0] + [(0 if s2_beg else -(open_ + (j - 1) * ext)) for j in range(1, n2 + 1)]
    F = [NEG] * (n2 + 1)
    lastcol = [None] * (n1 + 1)
    for i in range(1, n1 + 1):
        w = dna_matrix[a[i - 1]]
        diag = H[0]
        H[0] = 0 if s1_beg else -(open_ + (i - 1) * ext)
        E = NEG
        for j in range(1, n2 + 1):
            E = max(E - ext, H[j - 1] - open_)
            F[j] = max(F[j] - ext, H[j] - open_)
            h = max(diag + w[b[j - 1]], E, F[j])
            diag = H[j]
            H[j] = h
        lastcol[i] = H[n2]
    best, e1, e2 = H[n2], n1, n2
    if s1_end:
        for i in range(1, n1 + 1):  # smallest i among equal scores
            if lastcol[i] > best or (lastcol[i] == best and i < e1 and e2 == n2):
                best, e1, e2 = lastcol[i], i, n2
    if s2_end:
        for j in range(1, n2 + 1):
            if H[j] > best:
                best, e1, e2 = H[j], n1, j
    return best, e1 - 1, e2 - 1


def _window(size: int, direction: int, lsr: int, step: int) -> range:
    """Candidate sizes visited for one stack entry (repeats.py:114-119)."""
    widen = step > lsr
    lo = max(size - (lsr if (direction < 1 or widen) else 0), 0)
    hi = size + (lsr if (direction > -1 or widen) else 0)
    return range(lo, hi + 1)


def _hill_climb(start: int, step: int, lsr: int, max_iters: int, evaluate, rank, tie_last: bool = False):
    """Shared search skeleton (shape of repeats.py:100-151).

    ``evaluate(i)`` -> list of entries scored for size i (one for reads, fwd+rev for the ref side);
    ``rank(entry)`` -> sort key.  Returns ({size: entries} in insertion order, n_scored).
    The stack is a Python list popped from the END, so (start, 0) is visited first.
    """
    seen: dict[int, list] = {}
    n_scored = 0
    stack = [(start - step, -1), (start + step, 1), (start, 0)]
    while stack and n_scored < max_iters:
        size, direction = stack.pop()
        if size < 0:
            continue
        columns: list[list[tuple]] = []  # columns[c] = [(i, entry)] for entry kind c, ascending i
        for i in _window(size, direction, lsr, step):
            if i not in seen:
                seen[i] = evaluate(i)
                n_scored += 1
            for c, e in enumerate(seen[i]):
                while len(columns) <= c:
                    columns.append([])
                columns[c].append((i, e))
        flat = [x for col in columns for x in col]  # all fwd entries, then all rev entries (repeats.py:135)
        if tie_last:
            flat = flat[::-1]
        top_i = max(flat, key=lambda x: rank(x[1]))[0]
        for sign in (1, -1):  # at most one of the two can fire (repeats.py:136-151)
            nxt = top_i + sign * step
            if (top_i - size) * sign > 0 and nxt not in seen and nxt >= 0:
                stack.append((nxt, sign))
    return seen, n_scored


def get_repeat_count(start_count: int, tr_seq: str, flank_left_seq: str, flank_right_seq: str, motif: str,
                     max_iters: int = 50, local_search_range: int = 3, step_size: int = 1, tie_last: bool = False):
    """((best size, best score), n_explored, best size - start_count) — repeats.py:55-56."""
    db_seq = flank_left_seq + tr_seq + flank_right_seq
    g = indel_penalty

    def evaluate(i: int):
        cand = flank_left_seq + motif * i + flank_right_seq
        return [sg_align(db_seq, cand, g, g, True, True, True, True)[0]]

    seen, n = _hill_climb(start_count, step_size, local_search_range, max_iters, evaluate, lambda e: e, tie_last)
    items = [(i, v[0]) for i, v in seen.items()]  # dict order = insertion order
    if tie_last:
        items = items[::-1]
    best = max(items, key=itemgetter(1))
    return best, n, best[0] - start_count


def score_ref_boundaries(db_seq: str, tr_candidate: str, fl: str, fr: str, ref_size: int):
    """((fwd score, r_adj), (rev score, l_adj)) — repeats.py:23-43: two query-end-free alignments."""
    g = indel_penalty
    s_f, endq_f, _ = sg_align(db_seq, fl + tr_candidate, g, g, False, True, False, False)
    s_r, endq_r, _ = sg_align(db_seq[::-1], (tr_candidate + fr)[::-1], g, g, False, True, False, False)
    return (s_f, endq_f + 1 - len(fl) - ref_size), (s_r, endq_r + 1 - len(fr) - ref_size)


def get_ref_repeat_count(start_count: int, tr_seq: str, flank_left_seq: str, flank_right_seq: str, motif: str,
                         ref_size: int, vcf_anchor_size: int, max_iters: int, local_search_range: int,
                         step_size: int, respect_coords: bool = False):
    """(final_res, l_offset, r_offset, (n_offset_scores, n_iters), (fl, tr, fr)) — repeats.py:73-192."""
    l_off = r_off = 0
    n_scored = 0
    fl, tr, fr = flank_left_seq, tr_seq, flank_right_seq
    if not respect_coords:
        db_seq = fl + tr + fr
        seen, n_scored = _hill_climb(
            start_count, step_size, local_search_range, max_iters,
            lambda i: list(score_ref_boundaries(db_seq, motif * i, fl, fr, ref_size)),
            lambda e: e)  # entries are (score, adj) tuples, compared lexicographically (repeats.py:135)
        best_fwd = max(seen.items(), key=lambda kv: kv[1][0][0])  # by score only, first max (repeats.py:154)
        best_rev = max(seen.items(), key=lambda kv: kv[1][1][0])
        l_off, r_off = best_rev[1][1][1], best_fwd[1][0][1]
        if l_off >= len(fl) - vcf_anchor_size:  # repeats.py:164-169
            l_off = 0
        if r_off >= len(fr):
            r_off = 0
        if l_off > 0:  # repeats.py:171-176: move flank bases into the tract
            tr, fl = fl[-l_off:] + tr, fl[:-l_off]
        if r_off > 0:
            tr, fr = tr + fr[:r_off], fr[r_off:]
    new_start = round((start_count * len(motif) + max(0, l_off) + max(0, r_off)) / len(motif))  # repeats.py:182
    final, n_final, _ = get_repeat_count(new_start, tr.upper(), fl, fr, motif, max_iters, local_search_range, step_size)
    return final, l_off, r_off, (n_scored, n_final), (fl, tr, fr)


def count_locus(reads, motif: str, est_cns, max_iters: int = 50, local_search_range: int = 3, step_size: int = 1,
                feedback: bool = True, tie_last: bool = False):
    """Caller protocol call_locus.py:1125-1161.  reads: [(fl, tr, fr)].  -> [(cn, score, n_iters, start)]."""
    out = []
    frac = 0.0  # read_offset_frac_from_starting_guess
    for (fl, tr, fr), est in zip(reads, est_cns):
        start = est
        if feedback:
            shift = round(frac * start)  # Python round(): half-to-even on the float64 product
            if shift < -start:
                frac = 0.0
            else:
                start += shift
        (cn, score), n_iters, delta = get_repeat_count(start, tr, fl, fr, motif, max_iters, local_search_range,
                                                       step_size, tie_last)
        frac += delta / max(cn, 1)
        out.append((cn, score, n_iters, start))
    return out
