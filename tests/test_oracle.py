"""CPU: the oracle (C restatement) against known answers, the pure-Python restatement and the golden
fixtures.  PARITY UNPINNED — see oracle/strk_oracle.c: no reference test or golden vector exists for
this path, so these known answers are derived from the recurrence itself."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import ALPHA_IUPAC, ALPHA_WC, oracle_count, oracle_table, rand_seq, random_locus
from oracle import py_restatement as P
from strkit_amd.synth import LocusBatch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_scoring_matrix_rules():
    """strkit/call/align_matrix.py:15-44 + strkit/iupac.py:9-21."""
    m = oracle.matrix()
    idx = {c: i for i, c in enumerate("ACGTRYSWKMBDHVNX")}
    for c, i in idx.items():
        assert m[i, i] == 2
    assert m[idx["A"], idx["C"]] == -7
    for code, members in {"R": "AG", "Y": "CT", "N": "ACGT", "H": "ACT", "B": "CGT"}.items():
        for b in "ACGT":
            exp = 2 if b in members else -7
            assert m[idx[code], idx[b]] == exp and m[idx[b], idx[code]] == exp
    for b in "ACGT":  # X neither rewards nor penalises a base (docs/caller_catalog.md:48-53)
        assert m[idx["X"], idx[b]] == 0 and m[idx[b], idx["X"]] == 0
    assert m[idx["X"], idx["N"]] == -7 and m[idx["R"], idx["N"]] == -7  # code-vs-code stays a mismatch
    # the reference's IUPAC table gives D the members of H (iupac.py:17; tests/test_iupac.py:8 pins {A,C,T} -> H)
    assert np.array_equal(m[idx["D"], :4], m[idx["H"], :4]) and m[idx["D"], idx["G"]] == -7
    assert (m[16, :] == 0).all() and (m[:, 16] == 0).all()  # parasail's implicit '*'
    assert np.array_equal(m, np.array(P.dna_matrix, np.int8))
    with open(os.path.join(GOLD, "dna_matrix.json")) as f:
        assert json.load(f)["matrix"] == m.tolist()
    assert oracle.encode("a") == oracle.encode("A") == 0 and oracle.encode("?") == 16


def test_alignment_known_answers():
    s = "ACGTTGCAGGCTAAGCTTAGC"
    assert oracle.sg_align(s, s)[0] == 2 * len(s)
    sub = s[:10] + ("A" if s[10] != "A" else "C") + s[11:]
    assert oracle.sg_align(s, sub, flags=0)[0] == 2 * len(s) - 9          # one substitution: -2 -7
    assert oracle.sg_align(s, s[:10] + s[11:], flags=0)[0] == 2 * (len(s) - 1) - 5   # one deleted base
    assert oracle.sg_align(s, s[:10] + "X" + s[11:], flags=0)[0] == 2 * (len(s) - 1)  # X column scores 0
    assert oracle.sg_align("AAGAG", "AARRG", flags=0)[0] == 10              # IUPAC motif vs a member read
    assert oracle.sg_align("G", "D", flags=0)[0] == -7                      # the D quirk
    # free end gaps: a short sequence inside a long one costs nothing at the ends
    assert oracle.sg_align("TTTT" + s + "CCCC", s, flags=oracle.S1_BEG_FREE | oracle.S1_END_FREE)[0] == 2 * len(s)
    assert oracle.sg_align("TTTT" + s + "CCCC", s, flags=0)[0] == 2 * len(s) - 5 * 8
    assert oracle.sg_align("", "ACGT") == (0, -1, -1)


def test_pure_tract_converges_in_nine_alignments():
    """(i) fl + motif*n + fr from start = n: cn = n, perfect score, 7 + 1 + 1 sizes explored."""
    fl, fr = "ACGTTGCATTGACCGTA", "GGATCCATTCGAATGCA"
    for motif, n in (("CAG", 12), ("AT", 7), ("GGCCC", 20), ("A", 30)):
        tr = motif * n
        (cn, score), n_explored, delta = oracle.repeat_count(n, tr, fl, fr, motif)
        assert (cn, score, delta) == (n, 2 * len(fl + tr + fr), 0)
        assert n_explored == 9
        assert P.get_repeat_count(n, tr, fl, fr, motif) == ((cn, score), n_explored, delta)


def test_search_chases_a_distant_start_and_respects_limits():
    fl, fr, motif = "ACGTTGCATTGACCGTA", "GGATCCATTCGAATGCA", "CAG"
    tr = motif * 25
    for start in (15, 35, 0, 3):
        (cn, score), n, delta = oracle.repeat_count(start, tr, fl, fr, motif)
        assert cn == 25 and score == 2 * len(fl + tr + fr) and delta == 25 - start
    (cn, _), n, _ = oracle.repeat_count(10, tr, fl, fr, motif, max_iters=5)  # (v) cut-off
    assert n >= 5 and cn < 25
    with pytest.raises(ValueError):  # every seed negative -> nothing scored (Python max() would raise)
        oracle.repeat_count(-5, tr, fl, fr, motif)
    assert oracle.repeat_count(0, "", fl, fr, motif)[0][0] == 0


def test_partial_copy_rounds_up_under_alignment_scoring():
    """Motif AC, 31-base tract ACAC...A (the shape of the docs example, docs/output_formats.md:92-104,
    which was produced with rc_method "comp" and reports read cn 15): under the repalign scoring a
    16th copy costs one gap but wins one more match than 15 copies, so the aligner's answer is 16."""
    tr = ("AC" * 16)[:31]
    fl, fr = "GATTACAGATTACAGGT", "TTGGCCATAGCTAGGCT"
    (cn, score), _, _ = oracle.repeat_count(round(31 / 2), tr, fl, fr, "AC")
    assert cn == 16 and score == 2 * (len(fl) + 31 + len(fr)) - 5
    assert oracle.candidate_score(tr, fl, fr, "AC", 15) == 2 * (len(fl) + 30 + len(fr)) - 5


@pytest.mark.parametrize("seed", range(4))
def test_docs_anchor_ref_cn_16(seed):
    """The only result the reference's tree holds for this path (docs/output_formats.md:92-104): motif AC, 31-base
    soft-masked tract `acac...a`, `ref_cn: 16`, `start_adj == start`, `end_adj == end`.  ref_cn always comes from
    get_ref_repeat_count with repalign (call_locus.py:799-810), whatever rc_method the reads used."""
    from strkit_amd.repeat_count_params import get_reference_rc_params
    rng = np.random.default_rng(seed)
    fl = "".join("acgt"[i] for i in rng.integers(4, size=69)) + "t"
    fr = "g" + "".join("acgt"[i] for i in rng.integers(4, size=69))
    tr = ("ac" * 16)[:31]
    est = round(31 / 2)
    assert est == 16
    rc = get_reference_rc_params("repalign", est, 250)       # call_locus.py:71,808
    (cn, _), l_off, r_off, _, (_, tr2, _) = oracle.ref_repeat_count(est, tr, fl, fr, "AC", 31, 5, rc.max_iters,
                                                                    rc.initial_local_search_range, rc.initial_step_size)
    assert cn == 16 and max(0, l_off) == 0 and max(0, r_off) == 0 and tr2 == tr


def test_tie_rules_differ_only_on_ties():
    rng = np.random.default_rng(5)
    seen_diff = False
    for _ in range(300):
        motif, reads = random_locus(rng, 1, motif_len=(1, 3), cn=(0, 6), flank=(1, 6), edits=(0, 3))
        fl, tr, fr = reads[0]
        a = oracle.repeat_count(2, tr, fl, fr, motif, tie_rule=oracle.TIE_FIRST)
        b = oracle.repeat_count(2, tr, fl, fr, motif, tie_rule=oracle.TIE_LAST)
        assert a[0][1] == b[0][1] or a[1] != b[1]  # same best score unless the explored sets diverged
        seen_diff |= a != b
        assert P.get_repeat_count(2, tr, fl, fr, motif) == a
        assert P.get_repeat_count(2, tr, fl, fr, motif, tie_last=True) == b
    assert seen_diff


@pytest.mark.parametrize("seed", [0, 1])
def test_c_oracle_matches_python_restatement(seed):
    rng = np.random.default_rng(seed)
    for _ in range(150):
        a, b = rand_seq(rng, int(rng.integers(0, 25)), ALPHA_IUPAC), rand_seq(rng, int(rng.integers(0, 25)), ALPHA_IUPAC)
        flags = int(rng.integers(0, 16))
        got = oracle.sg_align(a, b, 5, 5, flags)
        if not a or not b:
            assert got[0] == 0
            continue
        exp = P.sg_align(a, b, 5, 5, bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8))
        assert got[0] == exp[0], (a, b, flags)
    for _ in range(40):
        motif, reads = random_locus(rng, 4, cn=(0, 12), flank=(3, 20), alpha=ALPHA_WC, motif_alpha=ALPHA_IUPAC)
        b = LocusBatch.from_reads([(motif, reads)])
        b.est_cn = np.maximum(0, b.est_cn + rng.integers(-5, 6, size=b.n_reads)).astype(np.int32)
        exp = P.count_locus(reads, motif, [int(x) for x in b.est_cn])
        got = oracle_count(b)
        assert [tuple(int(got[k][i]) for k in ("cn", "score", "n_iters", "start")) for i in range(b.n_reads)] == exp


def test_reference_side_matches_python_restatement():
    """get_ref_repeat_count / score_ref_boundaries (strkit/call/repeats.py:23-43,73-192)."""
    rng = np.random.default_rng(9)
    for _ in range(60):
        motif = rand_seq(rng, int(rng.integers(1, 6)))
        cn = int(rng.integers(2, 12))
        ext_l, ext_r = int(rng.integers(0, 3)), int(rng.integers(0, 3))  # repeat copies hidden in the flanks
        fl = rand_seq(rng, 25) + motif * ext_l
        fr = motif * ext_r + rand_seq(rng, 25)
        tr = motif * cn
        ref_size = len(tr)
        args = (cn, tr, fl, fr, motif, ref_size, 5, 50, 3, 1)
        assert oracle.ref_repeat_count(*args) == P.get_ref_repeat_count(*args)
        db = fl + tr + fr
        assert oracle.score_ref_boundaries(db, fl, fr, motif, cn, ref_size) == \
            P.score_ref_boundaries(db, motif * cn, fl, fr, ref_size)
        assert oracle.ref_repeat_count(*args, respect_coords=True) == P.get_ref_repeat_count(*args, respect_coords=True)


def _golden_batch(loci_json):
    b = LocusBatch.from_reads([(l["motif"], [tuple(r) for r in l["reads"]]) for l in loci_json],
                              [l["est_cn"] for l in loci_json])
    return b


def test_golden_count_cases():
    with open(os.path.join(GOLD, "count_cases.json")) as f:
        cases = json.load(f)
    assert set(cases) >= {"cfg1_hifi", "cfg3_ont", "cfg5_long", "adversarial_first_max", "adversarial_last_max"}
    for name, case in cases.items():
        b = _golden_batch(case["loci"])
        got = oracle_count(b, **case.get("params", {}))
        for k, v in case["expected"].items():
            assert got[k].tolist() == v, (name, k)


def test_golden_pin_cases():
    """tests/golden/pin_cases.json (make_pin_cases.py): the inputs the oracle's open assumptions hinge on — lower-case windows,
    empty tracts, start 0, IUPAC motifs against wildcards, constructed ties (both tie rules), reference-side boundary ties."""
    with open(os.path.join(GOLD, "pin_cases.json")) as f:
        cases = json.load(f)
    assert set(cases) == {"case", "empty", "start0", "iupac", "ties", "ref_ties"}
    for name, case in cases.items():
        if name == "ref_ties":
            continue
        b = _golden_batch(case["loci"])
        for tag, tie in (("expected_first_max", 0), ("expected_last_max", 1)):
            got = oracle_count(b, tie_rule=tie)
            for k, v in case[tag].items():
                assert got[k].tolist() == v, (name, tag, k)
    # upper- and lower-case windows count alike; the tie cases really depend on the rule
    c = cases["case"]
    b = _golden_batch(c["loci"])
    up = LocusBatch.from_reads([(l["motif"].upper(), [tuple(x.upper() for x in r) for r in l["reads"]]) for l in c["loci"]],
                               [l["est_cn"] for l in c["loci"]])
    assert oracle_count(b)["cn"].tolist() == oracle_count(up)["cn"].tolist() == c["expected_first_max"]["cn"]
    t = cases["ties"]
    assert all(x != y for x, y in zip(t["expected_first_max"]["cn"], t["expected_last_max"]["cn"]))
    ref = cases["ref_ties"]
    k = 0
    for l in ref["loci"]:
        for a, tr, cc in l["reads"]:
            for respect in (False, True):
                res = oracle.ref_repeat_count(round(len(tr) / len(l["motif"])), tr, a, cc, l["motif"], len(tr), 5, 50, 3, 1,
                                              respect_coords=respect)
                assert json.loads(json.dumps(res)) == ref["expected"][k], (l["motif"], respect)
                k += 1
    assert k == len(ref["expected"])


def test_golden_score_tables():
    with open(os.path.join(GOLD, "score_tables.json")) as f:
        g = json.load(f)
    b = _golden_batch(g["loci"])
    for flags, tabs in g["tables"].items():
        got = oracle_table(b, g["lo"], g["n"], int(flags))
        assert [t.tolist() for t in got] == tabs, flags


# ---- realignment restatement (strkit/call/realign.py:56-72) -----------------------------------
def _py_sg_dx_score(s1, s2, open_, ext):
    """Independent three-state Gotoh in pure Python: s1 global, both ends of s2 free."""
    M = oracle.matrix()
    NEG = -10**9
    n2 = len(s2)
    H = [0] * (n2 + 1)
    GI = [NEG] * (n2 + 1)
    for i in range(1, len(s1) + 1):
        a = oracle.encode(s1[i - 1])
        diag, H[0], GD = H[0], -(open_ + (i - 1) * ext), NEG
        for j in range(1, n2 + 1):
            GD = max(GD - ext, H[j - 1] - open_)
            GI[j] = max(GI[j] - ext, H[j] - open_)
            h = max(diag + int(M[a, oracle.encode(s2[j - 1])]), GI[j], GD)
            diag, H[j] = H[j], h
    return max(H[1:])


def test_realign_known_answers():
    from helpers import cigar_tuples
    sc, e2, cg = oracle.realign("ACGTACGTAC", "TTTTACGTACGTACTTT")
    assert (sc, e2, cigar_tuples(cg)) == (20, 13, [(4, "D"), (10, "=")])
    # a 4-base insertion in the window costs one gap of 7 (extend 0): 2*10 - 7
    sc, e2, cg = oracle.realign("ACGTACGGGGGTAC", "TTTTACGTACGTACTTT")
    assert (sc, cigar_tuples(cg)) == (13, [(4, "D"), (6, "="), (4, "I"), (4, "=")])
    # a 24-base expansion in the read costs the same 7
    sc, e2, cg = oracle.realign("ACGTACGTAC", "TTTTACGTAAAAAAAAAAAAAAAAAAAAAAAAACGTACTTT")
    assert (sc, e2, cigar_tuples(cg)) == (13, 37, [(4, "D"), (4, "="), (24, "D"), (6, "=")])
    # X scores 0, a mismatch -7 ('X' op in both cases: the letters differ)
    sc, _, cg = oracle.realign("ACGT", "AXGT")
    assert (sc, cigar_tuples(cg)) == (6, [(1, "="), (1, "X"), (2, "=")])
    # score agrees with the end-gap-flag scorer used by the counting path
    assert oracle.sg_align("ACGTTGCA", "GGACGTGCAGG", 7, 0, oracle.S2_BEG_FREE | oracle.S2_END_FREE)[0] == \
        oracle.realign("ACGTTGCA", "GGACGTGCAGG")[0]


def test_realign_matches_python_restatement_and_its_own_cigar():
    from helpers import ALPHA_WC, rand_seq, realign_pair, rescore_cigar
    rng = np.random.default_rng(31)
    for k in range(40):
        open_, ext = [(7, 0), (7, 1), (5, 5), (2, 1)][k % 4]
        if k % 2:
            r, q = realign_pair(rng, int(rng.integers(1, 40)), int(rng.integers(1, 90)), ins=int(rng.integers(0, 9)),
                                sub=0.05, indel=0.05, alpha=ALPHA_WC)
        else:
            r, q = rand_seq(rng, int(rng.integers(1, 30)), ALPHA_WC), rand_seq(rng, int(rng.integers(1, 60)), ALPHA_WC)
        for pref in (0, 1):
            sc, e2, cg = oracle.realign(r, q, open_, ext, pref)
            assert sc == _py_sg_dx_score(r, q, open_, ext)
            rs, i_end, j_end = rescore_cigar(r, q, cg, open_, ext)
            assert (rs, i_end, j_end - 1) == (sc, len(r), e2)


def test_golden_realign_cases():
    from helpers import cigar_tuples
    with open(os.path.join(GOLD, "realign_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) == 72
    for c in cases:
        sc, e2, cg = oracle.realign(c["ref"], c["read"], c["open"], c["extend"], c["gap_pref"])
        assert (sc, e2, "".join(f"{n}{o}" for n, o in cigar_tuples(cg))) == (c["score"], c["end_ref"], c["cigar"])


def test_simd_scoring_equals_the_scalar_restatement():
    """oracle/strk_simd.c (AVX2, sixteen candidate sizes of a read per pass: bench.py's "simd" CPU baseline) against the
    scalar code: per-candidate scores and whole searches, wildcard and IUPAC alphabets, empty tracts, bad estimates."""
    import ctypes as C
    if not oracle.set_simd(True):
        pytest.skip("this CPU has no AVX2")
    try:
        rng = np.random.default_rng(9)
        L = oracle.lib()
        L.strk_o_simd_scores16.restype = C.c_int64
        for alpha in ("ACGTXN", ALPHA_IUPAC):
            for _ in range(40):
                motif, reads = random_locus(rng, 1, motif_len=(1, 7), cn=(0, 20), flank=(1, 40), alpha=alpha, edits=(0, 5))
                fl, tr, fr = reads[0]
                db = (fl + tr + fr).encode()
                lo = int(rng.integers(0, 12))
                out = (C.c_int32 * 16)()
                cells = L.strk_o_simd_scores16(db, len(db), fl.encode(), len(fl), fr.encode(), len(fr), motif.encode(), len(motif), lo, out)
                assert cells > 0
                oracle.set_simd(False)
                want = [oracle.candidate_score(tr, fl, fr, motif, lo + k) for k in range(16)]
                oracle.set_simd(True)
                assert list(out) == want, (motif, fl, tr, fr, lo)
        for _ in range(60):
            motif, reads = random_locus(rng, 4, motif_len=(1, 6), cn=(0, 30), flank=(1, 70), alpha="ACGTX")
            for fl, tr, fr in reads:
                start = max(0, round(len(tr) / len(motif)) + int(rng.integers(-7, 8)))
                oracle.set_simd(False)
                a = oracle.repeat_count(start, tr, fl, fr, motif)
                oracle.set_simd(True)
                assert oracle.repeat_count(start, tr, fl, fr, motif) == a
    finally:
        oracle.set_simd(False)


def test_narrowing_schedules_of_the_oracle_equal_a_plain_python_search():
    """strk_o_repeat_count's rule word (tie rule + schedule of local_search_range) against the plain Python loop of
    tests/helpers.py::py_search over a table of the oracle's own candidate scores: 4 schedules x 2 tie rules x steps and
    ranges, starts from exact to far off.  (tests/test_host.py holds the product's search_replay against the same loop.)"""
    from helpers import noisy_tract, py_search, rand_seq
    rng = np.random.default_rng(20261010)
    n_diff = 0
    for _ in range(60):
        m = int(rng.integers(1, 7))
        motif = rand_seq(rng, m)
        cn = int(rng.integers(0, 25))
        fl, fr = rand_seq(rng, int(rng.integers(5, 40))), rand_seq(rng, int(rng.integers(5, 40)))
        tr = noisy_tract(rng, motif, cn, int(rng.integers(0, 5)), "ACGT")
        table = {i: oracle.candidate_score(tr, fl, fr, motif, i) for i in range(0, cn + 12 + 50 * 9 + 10)}   # (as far as 50 steps of 3 + 5 can walk)
        for _ in range(12):
            start = max(0, cn + int(rng.integers(-12, 13)))
            step, lsr = int(rng.integers(1, 4)), int(rng.integers(0, 6))
            max_iters, tie = int(rng.choice((3, 10, 50))), int(rng.integers(2))
            res = []
            for narrow in range(4):
                want = py_search(start, step, lsr, max_iters, tie, narrow, table)
                try:
                    got = oracle.repeat_count(start, tr, fl, fr, motif, max_iters, lsr, step, tie, narrowing=narrow)
                    got = (got[0][0], got[0][1], got[1])
                except ValueError:
                    got = "empty"
                assert got == want, (motif, tr, start, step, lsr, max_iters, tie, narrow, got, want)
                res.append(got)
            n_diff += len(set(res)) > 1
    assert n_diff > 20          # the schedules are not one and the same on these inputs
