#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle.

PARITY UNPINNED: the reference's read-side arithmetic lives in strkit_rust_ext + parasail, neither
vendored nor importable here, and the reference's own tests hold no vectors for this
path.  These fixtures therefore pin the ORACLE (oracle/strk_oracle.c), so that a later change to it,
to the generator or to the HIP path is caught; they are not outputs of STRkit itself.

    python tests/golden/make_golden.py
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
from helpers import ALPHA_IUPAC, ALPHA_WC, oracle_count, oracle_table, random_locus  # noqa: E402
from strkit_amd.synth import LocusBatch, make_config  # noqa: E402


def batch_to_json(b: LocusBatch) -> list:
    return [{"motif": b.motif(l), "reads": [list(b.read(r)) for r in range(int(b.read_off[l]), int(b.read_off[l + 1]))],
             "est_cn": [int(x) for x in b.est_cn[int(b.read_off[l]):int(b.read_off[l + 1])]]} for l in range(b.n_loci)]


def main() -> None:
    rng = np.random.default_rng(20261003)
    # 1. per-read counts on small slices of the BASELINE configs + adversarial loci
    count_cases = {}
    for name, b in (("cfg1_hifi", make_config(1, n_loci=12)), ("cfg3_ont", make_config(3, n_loci=12)),
                    ("cfg5_long", make_config(5, n_loci=2, cn_range=(50, 90), reads_per_locus=4))):
        exp = oracle_count(b)
        count_cases[name] = {"loci": batch_to_json(b), "expected": {k: v.tolist() for k, v in exp.items()}}
    loci = [random_locus(rng, 5, motif_len=(1, 6), cn=(0, 20), flank=(8, 70), alpha=ALPHA_WC, motif_alpha=ALPHA_IUPAC)
            for _ in range(12)]
    b = LocusBatch.from_reads(loci)
    b.est_cn = np.maximum(0, b.est_cn + rng.integers(-6, 7, size=b.n_reads)).astype(np.int32)
    for tag, kw in (("adversarial_first_max", dict(tie_rule=0)), ("adversarial_last_max", dict(tie_rule=1))):
        exp = oracle_count(b, **kw)
        count_cases[tag] = {"loci": batch_to_json(b), "params": kw, "expected": {k: v.tolist() for k, v in exp.items()}}
    with open(os.path.join(HERE, "count_cases.json"), "w") as f:
        json.dump(count_cases, f, separators=(",", ":"))
    # 2. raw score tables (the implementation-independent surface: scores depend only on the recurrence)
    loci = [random_locus(rng, 3, motif_len=(1, 8), cn=(0, 14), flank=(1, 40), alpha=ALPHA_IUPAC) for _ in range(10)]
    loci += [random_locus(rng, 3, motif_len=(2, 6), cn=(3, 30), flank=(60, 70), alpha=ALPHA_WC) for _ in range(10)]
    b = LocusBatch.from_reads(loci)
    lo = np.maximum(0, b.est_cn - 3).astype(np.int32)
    n = np.full(b.n_reads, 7, np.int32)
    tables = {str(flags): [t.tolist() for t in oracle_table(b, lo, n, flags)] for flags in (15, 0, 2, 6, 9)}
    with open(os.path.join(HERE, "score_tables.json"), "w") as f:
        json.dump({"loci": batch_to_json(b), "lo": lo.tolist(), "n": n.tolist(), "tables": tables}, f, separators=(",", ":"))
    # 3. scoring matrix
    with open(os.path.join(HERE, "dna_matrix.json"), "w") as f:
        json.dump({"alphabet": "ACGTRYSWKMBDHVNX*", "matrix": oracle.matrix().tolist()}, f)
    # 4. realignment (strk_o_realign): score, end and CIGAR under both gap preferences and several gap models
    from helpers import cigar_tuples, rand_seq, realign_pair
    rng2 = np.random.default_rng(20261004)
    cases = []
    for k in range(36):
        open_, ext = [(7, 0), (7, 0), (7, 1), (5, 5), (2, 1), (0, 0)][k % 6]
        alpha = [ALPHA_WC, "AC", ALPHA_IUPAC, "ACGT"][k % 4]
        if k % 3:
            r, q = realign_pair(rng2, int(rng2.integers(1, 90)), int(rng2.integers(1, 260)), ins=int(rng2.integers(0, 25)),
                                dele=int(rng2.integers(0, 8)), sub=0.04, indel=0.04, alpha=alpha)
        else:
            r, q = rand_seq(rng2, int(rng2.integers(1, 40)), alpha), rand_seq(rng2, int(rng2.integers(1, 120)), alpha)
        if k % 5 == 0:
            q = q.lower()
        for pref in (0, 1):
            sc, e2, cg = oracle.realign(r, q, open_, ext, pref)
            cases.append({"ref": r, "read": q, "open": open_, "extend": ext, "gap_pref": pref, "score": sc, "end_ref": e2,
                          "cigar": "".join(f"{n}{o}" for n, o in cigar_tuples(cg))})
    with open(os.path.join(HERE, "realign_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote", sorted(x for x in os.listdir(HERE) if x.endswith(".json")))


if __name__ == "__main__":
    main()
