#!/usr/bin/env python3
"""Regenerates tests/golden/pin_cases.json: the inputs that the assumptions of the oracle hinge on, with
the ORACLE's answers under both tie rules.  They are NOT STRkit outputs — tools/make_reference_vectors.py feeds these same
inputs to an installed STRkit, and tests/test_reference_vectors.py then says which assumption fails.

What is in it (every case: motif, reads (fl, tr, fr), start estimates):
  * case        — lower-case and mixed-case windows (soft-masked reference stretches reach the reads' windows through the
                  reference side; SURVEY.md section 7 "case handling");
  * empty       — an empty tract, with start counts 0, 1 and 5; a tract shorter than one motif copy;
  * start0      — start_count = 0 on real tracts (negative seeds are skipped, repeats.py:108-109);
  * iupac       — IUPAC motifs (AARRG, GCN, YTN ...) against reads with wildcard `X` and `N` bases (align_matrix.py:36-39,
                  the D / H quirk of iupac.py:17-18 included);
  * ties        — windows found by search where two candidate sizes INSIDE the first search window score the same maximum, so
                  that the first-max / last-max rule decides the count (repeats.py:135,154);
  * ref_ties    — reference-side loci whose flanks begin / end with copies of the motif (boundary extension can end at several
                  positions with one score: the `end_query` tie rule of the profile scan, repeats.py:33-41).

    python tests/golden/make_pin_cases.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
from helpers import ALPHA_ACGT, noisy_tract, oracle_count, rand_seq  # noqa: E402
from strkit_amd.synth import LocusBatch  # noqa: E402


def locus(motif, reads, est=None):
    return {"motif": motif, "reads": [list(r) for r in reads],
            "est_cn": [int(e) for e in (est if est is not None else [round(len(r[1]) / len(motif)) for r in reads])]}


def with_expected(loci):
    b = LocusBatch.from_reads([(l["motif"], [tuple(r) for r in l["reads"]]) for l in loci], [l["est_cn"] for l in loci])
    out = {"loci": loci}
    for tag, tie in (("expected_first_max", 0), ("expected_last_max", 1)):
        exp = oracle_count(b, tie_rule=tie)
        out[tag] = {k: v.tolist() for k, v in exp.items()}
    return out


def main() -> None:
    rng = np.random.default_rng(20261005)
    fl, fr = rand_seq(rng, 70), rand_seq(rng, 70)
    cases = {}
    # ---- case handling
    cag = "CAG" * 12
    cases["case"] = with_expected([
        locus("CAG", [(fl, cag.lower(), fr), (fl.lower(), cag, fr.lower()), (fl, "CAGcagCAGcagCAGcagCAGCAGCAGcag", fr),
                      ((fl + cag + fr).lower()[:70], (fl + cag + fr).lower()[70:70 + 36], (fl + cag + fr).lower()[106:])]),
        locus("ac", [(fl, "ac" * 15 + "a", fr), (fl, "AC" * 15 + "A", fr)]),            # docs/output_formats.md:96: a soft-masked tract
        locus("AARRG", [(fl, "aagagaaaggaagag", fr), (fl, "AAGAGAAAGGAAGAG", fr)]),
    ])
    # ---- empty and sub-motif tracts
    cases["empty"] = with_expected([
        locus("CAG", [(fl, "", fr), (fl, "", fr), (fl, "", fr), (fl, "CA", fr), (fl, "C", fr)], est=[0, 1, 5, 0, 1]),
        locus("A", [(fl[:-1] + "C", "", "G" + fr[1:]), (fl[:-1] + "C", "A", "G" + fr[1:])], est=[0, 0]),
        locus("ATTCT", [(fl, "", fr), (fl, "ATT", fr)], est=[2, 1]),
    ])
    # ---- start_count = 0 on real tracts (the search must climb from nothing; window clipped at 0)
    cases["start0"] = with_expected([
        locus("CAG", [(fl, "CAG" * k, fr) for k in (1, 2, 3, 4, 6, 9)], est=[0] * 6),
        locus("AT", [(fl, "AT" * k + "A", fr) for k in (1, 3, 5, 8)], est=[0, 0, 1, 0]),
    ])
    # ---- IUPAC motifs against wildcarded reads
    def wc(s, every, ch="X"):
        return "".join(ch if (i % every) == every - 1 else c for i, c in enumerate(s))
    cases["iupac"] = with_expected([
        locus("AARRG", [(fl, wc("AAGAGAAAGGAAGAGAAAAG", 7), fr), (wc(fl, 11), "AAGAG" * 6, wc(fr, 13)), (fl, wc("AAGGGAAAAGAAGAG", 4, "N"), fr)]),
        locus("GCN", [(fl, wc("GCAGCCGCGGCTGCAGCC", 5), fr), (fl, "GCXGCXGCXGCXGCX", fr), (fl, "GCNGCNGCN", fr)]),
        locus("YTN", [(fl, "CTATTGCTCTTA", fr), (fl, wc("CTATTGCTCTTACTG", 6), fr)]),
        locus("D", [(fl[:-1] + "C", "AGTCGATTG", "C" + fr[1:])]),            # D = (A, C, T) in the reference's table: G is a mismatch
        locus("H", [(fl[:-1] + "G", "ACTCATTAC", "G" + fr[1:])]),
        locus("X", [(fl, "ACGTACGT", fr)], est=[8]),                          # a wildcard MOTIF (valid_motif rejects it upstream; the scorer does not)
    ])
    # ---- ties inside the first search window: random small windows until the oracle's two tie rules disagree
    tie_loci = []
    tries = 0
    while len(tie_loci) < 24 and tries < 200000:
        tries += 1
        m = int(rng.integers(1, 5))
        motif = rand_seq(rng, m, ALPHA_ACGT)
        cn = int(rng.integers(1, 9))
        a, c = rand_seq(rng, int(rng.integers(3, 12))), rand_seq(rng, int(rng.integers(3, 12)))
        tr = noisy_tract(rng, motif, cn, int(rng.integers(0, 3)), ALPHA_ACGT)
        est = max(0, round(len(tr) / m) + int(rng.integers(-1, 2)))
        r0 = oracle.repeat_count(est, tr, a, c, motif, 50, 3, 1, tie_rule=0, flags=15)
        r1 = oracle.repeat_count(est, tr, a, c, motif, 50, 3, 1, tie_rule=1, flags=15)
        if r0[0][0] != r1[0][0]:
            tie_loci.append(locus(motif, [(a, tr, c)], est=[est]))
    cases["ties"] = with_expected(tie_loci)
    # ---- reference-side loci whose flanks run into the tract (get_ref_repeat_count's boundary extension)
    ref_loci = []
    for motif, k_l, k_r, n in (("CAG", 2, 0, 10), ("CAG", 0, 3, 10), ("CAG", 1, 1, 7), ("AT", 3, 2, 12), ("A", 4, 4, 15),
                               ("GGCCCC", 1, 1, 5), ("AAG", 2, 2, 8), ("TTC", 0, 1, 20)):
        a = rand_seq(rng, 50) + motif * k_l
        c = motif * k_r + rand_seq(rng, 50)
        ref_loci.append(locus(motif, [(a, motif * n, c), (a.lower(), (motif * n).lower(), c.lower())]))
    # partial copies at the boundaries
    ref_loci.append(locus("CAG", [(rand_seq(rng, 50) + "AG", "CAG" * 9, "CA" + rand_seq(rng, 50))]))
    ref_loci.append(locus("AT", [(rand_seq(rng, 40) + "T", "AT" * 11, "A" + rand_seq(rng, 40))]))
    ref = {"loci": ref_loci, "expected": []}
    for l in ref_loci:
        for a, tr, c in l["reads"]:
            for respect in (False, True):
                try:
                    res = oracle.ref_repeat_count(round(len(tr) / len(l["motif"])), tr, a, c, l["motif"], len(tr), 5, 50, 3, 1,
                                                  respect_coords=respect)
                    ref["expected"].append(json.loads(json.dumps(res)))
                except ValueError:
                    ref["expected"].append("raises")
    cases["ref_ties"] = ref
    with open(os.path.join(HERE, "pin_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote pin_cases.json:", {k: len(v["loci"]) for k, v in cases.items()}, f"({tries} windows tried for the ties)")


if __name__ == "__main__":
    main()
