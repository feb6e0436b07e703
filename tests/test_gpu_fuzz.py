"""Randomised differential test: fresh random batches and option combinations against the oracle.
A few seconds by default; STRK_FUZZ_SECONDS=300 python -m pytest tests/test_gpu_fuzz.py -m gpu for a soak run
(the seed is printed on failure)."""
import os
import sys
import time

import numpy as np
import pytest

import oracle
from helpers import ALPHA_ACGT, ALPHA_IUPAC, ALPHA_WC, cigar_tuples, oracle_count, rand_seq, random_locus, realign_pair
from strkit_amd.batch import count_loci
from strkit_amd.realign import realign_pairs
from strkit_amd.repeat_count_params import RepeatCountParams
from strkit_amd.synth import LocusBatch, make_config

pytestmark = pytest.mark.gpu
SECONDS = float(os.environ.get("STRK_FUZZ_SECONDS", "6"))
KEYS = ("cn", "score", "n_iters", "start")


def _one_count_case(rng, ctx):
    kind = int(rng.integers(6))
    if kind == 0:      # slices of the named configurations, different seeds
        b = make_config(int(rng.integers(1, 4)), n_loci=int(rng.integers(1, 60)), seed_shift=int(rng.integers(1 << 20)))
    elif kind == 1:    # long tracts (wide band classes / long-read kernel)
        b = make_config(5, n_loci=int(rng.integers(1, 4)), seed_shift=int(rng.integers(1 << 20)),
                        cn_range=(50, int(rng.integers(60, 700))), reads_per_locus=int(rng.integers(1, 6)))
    else:              # adversarial: odd alphabets, short flanks, bad estimates, empty tracts
        alpha = [ALPHA_ACGT, ALPHA_WC, ALPHA_IUPAC, "AC"][int(rng.integers(4))]
        loci = [random_locus(rng, int(rng.integers(1, 9)), motif_len=(1, int(rng.integers(1, 21))), cn=(0, int(rng.integers(1, 70))),
                             flank=(1, int(rng.integers(1, 100))), alpha=alpha, motif_alpha=[None, ALPHA_IUPAC][int(rng.integers(2))],
                             edits=(0, int(rng.integers(0, 12)))) for _ in range(int(rng.integers(1, 25)))]
        b = LocusBatch.from_reads(loci)
        if rng.random() < 0.5:
            b.est_cn = np.maximum(0, b.est_cn + rng.integers(-12, 13, size=b.n_reads)).astype(np.int32)
    opts = dict(band=bool(rng.integers(2)), dedupe=bool(rng.integers(2)), feedback=bool(rng.integers(2)),
                window=int(rng.choice([0, 0, 0, 1, 2, 5, 11, 15])), tie_rule=int(rng.integers(2)),
                end_flags=int(rng.choice([15, 15, 15, 0, 2, 6, 9, 13])))
    rc = RepeatCountParams("repalign", int(rng.choice([50, 50, 7, 100])), int(rng.choice([3, 3, 1, 5])), int(rng.choice([1, 1, 2])))
    got = count_loci(b, rc, ctx=ctx, **opts)
    exp = oracle_count(b, rc.max_iters, rc.initial_local_search_range, rc.initial_step_size, opts["tie_rule"], opts["end_flags"],
                       opts["feedback"])
    for k in KEYS:
        if not np.array_equal(got[k], exp[k]):
            bad = int(np.nonzero(got[k] != exp[k])[0][0])
            locus = int(np.searchsorted(b.read_off, bad, side="right")) - 1
            raise AssertionError(f"{k} differs: kind {kind} opts {opts!r} rc {rc!r} reads {b.n_reads} first bad read {bad} "
                                 f"(locus {locus}, motif {b.motif(locus)!r}, est {int(b.est_cn[bad])}) got "
                                 f"{[int(got[x][bad]) for x in KEYS]} exp {[int(exp[x][bad]) for x in KEYS]} read {b.read(bad)!r}")
    return b.n_reads


def _one_realign_case(rng):
    n = int(rng.integers(1, 12))
    open_, ext = [(7, 0), (7, 0), (7, 2), (4, 4), (0, 0)][int(rng.integers(5))]
    pref = int(rng.integers(2))
    refs, reads = [], []
    for _ in range(n):
        alpha = [ALPHA_ACGT, ALPHA_WC, ALPHA_IUPAC][int(rng.integers(3))]
        n_ref = int(rng.choice([1, 5, 60, 200, 256, 257, 700, 1500, 2300]))
        r, q = realign_pair(rng, n_ref, int(rng.integers(1, 3 * n_ref + 400)), ins=int(rng.integers(0, 80)),
                            dele=int(rng.integers(0, min(20, max(1, n_ref // 2)))), sub=0.03, indel=0.03, alpha=alpha, wc=0.01)
        refs.append(r)
        reads.append(q)
    got = realign_pairs(refs, reads, open_, ext, pref)
    for p in range(n):
        sc, e2, cg = oracle.realign(refs[p], reads[p], open_, ext, pref)
        assert (got[p][0], got[p][1], cigar_tuples(got[p][2])) == (sc, e2, cigar_tuples(cg)), (p, len(refs[p]), len(reads[p]), open_, ext, pref)
    return n


def _one_ref_case(rng):
    from strkit_amd.repeat_count_params import get_reference_rc_params
    from strkit_amd.repeats import get_ref_repeat_counts
    jobs = []
    for _ in range(int(rng.integers(1, 16))):
        alpha = [ALPHA_ACGT, ALPHA_ACGT, "ACGTN"][int(rng.integers(3))]
        m = int(rng.integers(1, 10))
        motif = rand_seq(rng, m)
        cn = int(rng.integers(0, int(rng.choice([12, 60, 400]))))
        fl, fr = rand_seq(rng, int(rng.integers(6, 90)), alpha), rand_seq(rng, int(rng.integers(6, 90)), alpha)
        tr = motif * cn
        if rng.random() < 0.4:
            fl = fl[:max(1, len(fl) - m)] + motif          # the tract spills into a flank
        if rng.random() < 0.4:
            fr = motif + fr
        if tr and rng.random() < 0.4:
            cut = int(rng.integers(len(tr)))
            tr = tr[:cut] + rand_seq(rng, int(rng.integers(1, 4))) + tr[cut:]
        est = max(0, round(len(tr) / m) + int(rng.integers(-3, 4)))
        jobs.append((est, tr, fl, fr, motif, len(tr), get_reference_rc_params("repalign", est, int(rng.choice([100, 20])))))
    anchor, respect = int(rng.choice([5, 0, 12])), bool(rng.integers(2))
    got = get_ref_repeat_counts(jobs, anchor, respect)
    for (est, tr, fl, fr, motif, ref_size, rc), g in zip(jobs, got):
        exp = oracle.ref_repeat_count(est, tr, fl, fr, motif, ref_size, anchor, rc.max_iters, rc.initial_local_search_range,
                                      rc.initial_step_size, respect)
        assert g == exp, (est, len(tr), len(fl), len(fr), motif, anchor, respect, rc, g[:4], exp[:4])
    return len(jobs)


def test_random_batches_and_options_match_the_oracle(gpu_ctx):
    # the routine run is reproducible; a soak run (STRK_FUZZ_SECONDS set) draws a fresh seed unless one is given
    default_seed = str(int(time.time()) & 0xFFFFFF) if "STRK_FUZZ_SECONDS" in os.environ else "20261004"
    seed = int(os.environ.get("STRK_FUZZ_SEED", default_seed))
    rng = np.random.default_rng(seed)
    t0, cases, reads, pairs, loci = time.time(), 0, 0, 0, 0
    t_note = time.time()
    try:
        while time.time() - t0 < SECONDS or cases < 12:
            if cases % 4 == 3:
                pairs += _one_realign_case(rng)
            elif cases % 8 == 6:
                loci += _one_ref_case(rng)
            else:
                reads += _one_count_case(rng, gpu_ctx)
            cases += 1
            if cases % 40 == 0:
                print(f"[fuzz seed {seed}] {cases} cases so far", flush=True)
            if time.time() - t_note > 60:   # a soak run under pytest's capture must not look hung (the real stderr is not captured)
                t_note = time.time()
                sys.__stderr__.write(f"[fuzz seed {seed}] {cases} cases, {time.time() - t0:.0f} s\n")
                sys.__stderr__.flush()
    except AssertionError as e:
        raise AssertionError(f"fuzz seed {seed}, case {cases}: {e}") from e
    print(f"\n[fuzz seed {seed}] {cases} cases, {reads} reads, {pairs} realignments, {loci} reference-side loci in {time.time() - t0:.1f} s")
