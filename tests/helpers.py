"""Shared generators for the parity tests (seeded; no reference code involved)."""
from __future__ import annotations

import numpy as np

import oracle
from strkit_amd.synth import LocusBatch

ALPHA_ACGT = "ACGT"
ALPHA_WC = "ACGTXN"
ALPHA_IUPAC = "ACGTRYSWKMBDHVNX"


def rand_seq(rng, n, alpha=ALPHA_ACGT):
    return "".join(alpha[i] for i in rng.integers(len(alpha), size=n))


def noisy_tract(rng, motif, cn, n_edits, alpha):
    tr = list(motif * cn)
    for _ in range(n_edits):
        x = rng.random()
        if tr and x < 0.4:
            tr[rng.integers(len(tr))] = alpha[rng.integers(len(alpha))]
        elif tr and x < 0.7:
            del tr[rng.integers(len(tr))]
        else:
            tr.insert(int(rng.integers(len(tr) + 1)), alpha[rng.integers(len(alpha))])
    return "".join(tr)


def random_locus(rng, n_reads, motif_len=(1, 6), cn=(0, 40), flank=(1, 80), alpha=ALPHA_ACGT, motif_alpha=None,
                 edits=(0, 4)):
    m = int(rng.integers(motif_len[0], motif_len[1] + 1))
    motif = rand_seq(rng, m, motif_alpha or alpha)
    fl0 = rand_seq(rng, int(rng.integers(flank[0], flank[1] + 1)), alpha)
    fr0 = rand_seq(rng, int(rng.integers(flank[0], flank[1] + 1)), alpha)
    base_cn = int(rng.integers(cn[0], cn[1] + 1))
    reads = []
    for _ in range(n_reads):
        c = max(0, base_cn + int(rng.integers(-2, 3)))
        tr = noisy_tract(rng, motif, c, int(rng.integers(edits[0], edits[1] + 1)), alpha)
        fl = noisy_tract(rng, fl0, 1, int(rng.integers(0, 2)), alpha) or fl0
        fr = noisy_tract(rng, fr0, 1, int(rng.integers(0, 2)), alpha) or fr0
        reads.append((fl, tr, fr))
    return motif, reads


def oracle_table(b: LocusBatch, lo, n, flags=15):
    out = []
    for l in range(b.n_loci):
        motif = b.motif(l)
        for r in range(int(b.read_off[l]), int(b.read_off[l + 1])):
            fl, tr, fr = b.read(r)
            out.append(np.array([oracle.candidate_score(tr, fl, fr, motif, int(lo[r]) + k, flags)
                                 for k in range(int(n[r]))], np.int32))
    return out


def oracle_count(b: LocusBatch, max_iters=50, lsr=3, step=1, tie_rule=0, flags=15, feedback=True, narrowing=0):
    res = {k: np.zeros(b.n_reads, np.int32) for k in ("cn", "score", "n_iters", "start")}
    for l in range(b.n_loci):
        r0, r1 = int(b.read_off[l]), int(b.read_off[l + 1])
        if r1 == r0:
            continue
        s0 = int(b.seq_off[r0])
        o = oracle.count_locus(b.seqs[s0:int(b.seq_off[r1])], b.seq_off[r0:r1 + 1] - s0, b.nfl[r0:r1], b.ntr[r0:r1],
                               b.nfr[r0:r1], b.est_cn[r0:r1], b.motif(l), max_iters, lsr, step, tie_rule, flags,
                               feedback, narrowing=narrowing)
        for k in res:
            res[k][r0:r1] = o[k]
    return res


def py_search(start, step, lsr, max_iters, tie_last, narrow, table):
    """The read-side search as a plain Python loop over a {size: score} table (control flow of repeats.py:100-151 as
    strk_search.h states it), with the schedules of local_search_range the library offers (strk_search.h: LsrSchedule).
    Written from the description, not from the C++: (cn, score, n_explored), or "miss" when a size outside the table is needed,
    "empty" when nothing was scored."""
    to_explore = [(start - step, -1, True), (start + step, 1, True), (start, 0, True)]
    seen = {}
    floor1 = min(lsr, 1)
    cur = lsr
    n = 0
    while to_explore and n < max_iters:
        size, direction, seed = to_explore.pop()
        if size < 0:
            continue
        use = (lsr if seed else floor1) if narrow == 3 else cur
        if narrow == 1:
            cur = max(floor1, cur - 1)
        elif narrow == 2:
            cur = max(floor1, cur // 2)
        both = step > use
        w_lo = max(0, size - (use if (direction < 1 or both) else 0))
        w_hi = size + (use if (direction > -1 or both) else 0)
        szs = []
        for i in range(w_lo, w_hi + 1):
            if i not in table:
                return "miss"
            if i not in seen:
                seen[i] = table[i]
                n += 1
            szs.append((i, table[i]))
        mv = szs[0]
        for x in szs[1:]:
            if x[1] > mv[1] or (tie_last and x[1] == mv[1]):
                mv = x
        if mv[0] > size and (mv[0] + step) not in seen and mv[0] + step >= 0:
            to_explore.append((mv[0] + step, 1, False))
        if mv[0] < size and (mv[0] - step) not in seen and mv[0] - step >= 0:
            to_explore.append((mv[0] - step, -1, False))
    if not seen:
        return "empty"
    best = None
    for i, s in seen.items():                                  # insertion order: first (or last) maximum
        if best is None or s > best[1] or (tie_last and s == best[1]):
            best = (i, s)
    return best[0], best[1], n


# ---- realignment (strkit/call/realign.py) ---------------------------------------------------
def mutate(rng, seq, sub=0.01, indel=0.01, alpha=ALPHA_ACGT):
    out = []
    for ch in seq:
        x = rng.random()
        if x < sub:
            out.append(alpha[rng.integers(len(alpha))])
        elif x < sub + indel / 2:
            continue
        elif x < sub + indel:
            out.append(ch)
            out.append(alpha[rng.integers(len(alpha))])
        else:
            out.append(ch)
    return "".join(out)


def realign_pair(rng, n_ref, n_read, ins=0, dele=0, sub=0.01, indel=0.01, alpha=ALPHA_ACGT, wc=0.0):
    """A reference window of n_ref bases and a read of about n_read bases that contains a mutated copy of it
    (optionally with one large insertion / deletion in the middle, the soft-clip case of call_locus.py:860-865)."""
    ref = rand_seq(rng, n_ref, alpha)
    mid = n_ref // 2
    body = ref[:mid] + rand_seq(rng, ins, alpha) + ref[mid + dele:]
    body = mutate(rng, body, sub, indel, alpha)
    left = int(rng.integers(0, max(1, n_read - len(body)) + 1)) if n_read > len(body) else 0
    right = max(0, n_read - len(body) - left)
    read = rand_seq(rng, left, alpha) + body + rand_seq(rng, right, alpha)
    if wc > 0:
        read = "".join("X" if rng.random() < wc else ch for ch in read)
    return ref, read or "A"


def cigar_tuples(cig):
    return [(int(x) >> 4, "MIDNSHP=X"[int(x) & 15]) for x in cig]


def rescore_cigar(ref, read, cig, open_=7, ext=0):
    """Score of the alignment a CIGAR describes (s1 = ref window, s2 = read; leading D run is free)."""
    M = oracle.matrix()
    i = j = 0
    score = 0
    first = True
    for ln, op in cigar_tuples(cig):
        if op in "=X":
            for _ in range(ln):
                score += int(M[oracle.encode(ref[i]), oracle.encode(read[j])])
                i += 1
                j += 1
        elif op == "I":
            score -= open_ + (ln - 1) * ext
            i += ln
        elif op == "D":
            if not (first and i == 0):
                score -= open_ + (ln - 1) * ext
            j += ln
        first = False
    return score, i, j
