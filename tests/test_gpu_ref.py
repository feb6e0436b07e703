"""GPU parity of the reference-side path (get_ref_repeat_count / score_ref_boundaries) vs the oracle."""
import numpy as np
import pytest

import oracle
from helpers import ALPHA_ACGT, ALPHA_WC, rand_seq, random_locus
from strkit_amd.repeat_count_params import RepeatCountParams, get_reference_rc_params
from strkit_amd.synth import LocusBatch

pytestmark = pytest.mark.gpu


def _ref_batch(loci):
    """Two device reads per locus: the window and its reversal with the flanks swapped."""
    out = []
    for motif, (fl, tr, fr) in loci:
        out.append((motif, [(fl, tr, fr)]))
        out.append((motif[::-1], [(fr[::-1], tr[::-1], fl[::-1])]))
    return LocusBatch.from_reads(out)


@pytest.mark.parametrize("force_generic", [False, True])
def test_score_ref_boundaries(gpu_ctx, force_generic):
    from strkit_amd.batch import score_ref_table
    rng = np.random.default_rng(31)
    loci = []
    for _ in range(40):
        motif, reads = random_locus(rng, 1, motif_len=(1, 7), cn=(1, 30), flank=(10, 70), alpha=ALPHA_WC, edits=(0, 5))
        loci.append((motif, reads[0]))
    b = _ref_batch(loci)
    lo = np.maximum(0, np.repeat([max(0, round(len(t[1]) / len(m)) - 4) for m, t in loci], 2)).astype(np.int32)
    n = np.full(b.n_reads, 9, np.int32)
    got = score_ref_table(b, lo, n, force_generic=force_generic, ctx=gpu_ctx)
    for k, (motif, (fl, tr, fr)) in enumerate(loci):
        db = fl + tr + fr
        ref_size = len(tr)
        for j in range(9):
            i = int(lo[2 * k]) + j
            (fs, fa), (rs, ra) = oracle.score_ref_boundaries(db, fl, fr, motif, i, ref_size)
            g_fs, g_fe = int(got[2 * k][0][j]), int(got[2 * k][1][j])
            g_rs, g_re = int(got[2 * k + 1][0][j]), int(got[2 * k + 1][1][j])
            assert (g_fs, g_fe + 1 - len(fl) - ref_size) == (fs, fa), (k, i, "fwd")
            assert (g_rs, g_re + 1 - len(fr) - ref_size) == (rs, ra), (k, i, "rev")


def test_get_ref_repeat_count_matches_oracle(gpu_ctx):
    from strkit_amd.repeats import get_ref_repeat_count
    rng = np.random.default_rng(32)
    for it in range(60):
        motif = rand_seq(rng, int(rng.integers(1, 7)))
        cn = int(rng.integers(2, 40))
        ext_l, ext_r = int(rng.integers(0, 4)), int(rng.integers(0, 4))  # repeat copies hidden in the flanks
        fl = rand_seq(rng, int(rng.integers(20, 70))) + motif * ext_l
        fr = motif * ext_r + rand_seq(rng, int(rng.integers(20, 70)))
        tr = motif * cn
        if it % 3 == 0:  # soft-masked lower case reference and an imperfect tract
            tr = tr[: len(tr) // 2].lower() + ("A" if tr[len(tr) // 2:len(tr) // 2 + 1] != "A" else "C") + tr[len(tr) // 2 + 1:]
        start = round(len(tr) / len(motif)) + int(rng.integers(-2, 3))
        rc = RepeatCountParams("repalign", 250, 3, 1)
        for respect in (False, True):
            exp = oracle.ref_repeat_count(start, tr, fl, fr, motif, len(tr), 5, rc.max_iters, 3, 1, respect_coords=respect)
            got = get_ref_repeat_count(start, tr, fl, fr, motif, len(tr), 5, rc, respect_coords=respect)
            assert got == exp, (it, respect, motif, fl, tr, fr)


def test_ref_schedule_with_big_steps(gpu_ctx):
    """Large reference copy numbers search in bigger steps (repeat_count_params.py:17-42)."""
    from strkit_amd.repeats import get_ref_repeat_count
    rng = np.random.default_rng(33)
    for cn, est in ((230, 226), (60, 64)):
        motif = "CAG"
        fl, fr = rand_seq(rng, 50, ALPHA_ACGT) + "CAGCAG", "CAG" + rand_seq(rng, 50, ALPHA_ACGT)
        tr = motif * cn
        rc = get_reference_rc_params("repalign", est, 250)
        exp = oracle.ref_repeat_count(est, tr, fl, fr, motif, len(tr), 5, rc.max_iters, rc.initial_local_search_range,
                                      rc.initial_step_size)
        got = get_ref_repeat_count(est, tr, fl, fr, motif, len(tr), 5, rc)
        assert got == exp


def test_batched_reference_side_equals_scalar_and_oracle(gpu_ctx):
    """strk_ref_repeat_count_batch (lock-step rounds over a block of loci) == one strk_ref_repeat_count per locus ==
    the oracle, including loci whose schedule differs (large estimates search in bigger steps)."""
    import numpy as np
    from helpers import rand_seq
    from strkit_amd.repeat_count_params import get_reference_rc_params
    from strkit_amd.repeats import get_ref_repeat_count, get_ref_repeat_counts
    rng = np.random.default_rng(77)
    jobs = []
    for k in range(40):
        m = int(rng.integers(1, 8))
        motif = rand_seq(rng, m)
        cn = int(rng.integers(2, 400 if k % 7 == 0 else 60))
        fl, fr = rand_seq(rng, 70), rand_seq(rng, 70)
        # the tract spills a little into the flanks so that the boundary extension has something to find
        tr = motif * cn
        if k % 3 == 0:
            fl = fl[:70 - m] + motif
        if k % 4 == 0:
            fr = motif + fr[m:]
        if k % 5 == 0:
            tr = tr[:len(tr) // 2] + rand_seq(rng, 2) + tr[len(tr) // 2:]
        est = max(0, round(len(tr) / m) + int(rng.integers(-2, 3)))
        jobs.append((est, tr, fl, fr, motif, len(tr), get_reference_rc_params("repalign", est, 100)))
    assert len({(j[6].max_iters, j[6].initial_local_search_range, j[6].initial_step_size) for j in jobs}) > 1
    for respect in (False, True):
        batch = get_ref_repeat_counts(jobs, 5, respect)
        for job, got in zip(jobs, batch):
            est, tr, fl, fr, motif, ref_size, rc = job
            assert got == get_ref_repeat_count(est, tr, fl, fr, motif, ref_size, 5, rc, respect)
            exp = oracle.ref_repeat_count(est, tr, fl, fr, motif, ref_size, 5, rc.max_iters, rc.initial_local_search_range,
                                          rc.initial_step_size, respect)
            assert got == exp
    assert get_ref_repeat_counts([], 5) == []


def _docs_anchor_locus(seed):
    """The one result the reference's tree holds for this path: docs/output_formats.md:92-104 — motif AC, the 31-base
    soft-masked tract `acac...a`, `ref_cn: 16`, `start_adj == start`, `end_adj == end`.  The read side of that sample
    used rc_method "comp", but the reference side ALWAYS runs repalign (call_locus.py:799-810), so 16 is an answer of
    get_ref_repeat_count.  The flanks are not printed there: any flank that does not continue the repeat will do."""
    rng = np.random.default_rng(seed)
    fl = "".join("acgt"[i] for i in rng.integers(4, size=69)) + "t"      # "ref_start_anchor": "t"
    fr = "g" + "".join("acgt"[i] for i in rng.integers(4, size=69))
    return fl, ("ac" * 16)[:31], fr


@pytest.mark.parametrize("seed", range(4))
def test_docs_anchor_ref_cn_16(gpu_ctx, seed):
    from strkit_amd.repeats import get_ref_repeat_count
    fl, tr, fr = _docs_anchor_locus(seed)
    est = round(len(tr) / 2)
    rc = get_reference_rc_params("repalign", est, 250)
    (cn, score), l_off, r_off, n_is, (fl2, tr2, fr2) = get_ref_repeat_count(est, tr, fl, fr, "AC", 31, 5, rc)
    assert cn == 16 and max(0, l_off) == 0 and max(0, r_off) == 0      # ref_cn 16, start_adj/end_adj unchanged
    assert tr2 == tr                                                     # "ref_seq" keeps the soft-masked case
    assert ((cn, score), l_off, r_off, n_is, (fl2, tr2, fr2)) == oracle.ref_repeat_count(
        est, tr, fl, fr, "AC", 31, 5, rc.max_iters, rc.initial_local_search_range, rc.initial_step_size)


def test_long_boundary_search_runs_250_iterations(gpu_ctx, tmp_path):
    """call_locus.py:71: default_ref_max_iters = 250.  A homopolymer that runs 125 bases past the catalog's right
    coordinate (flank size 160) needs more than 100 offset scores: with a limit of 100 the search stops at +99."""
    from strkit_amd.frontend import Fasta
    from strkit_amd.frontend.call import DEFAULT_REF_MAX_ITERS, get_locus_with_ref_data
    from strkit_amd.frontend.fasta import write_fasta
    from strkit_amd.frontend.loci import Locus
    assert DEFAULT_REF_MAX_ITERS == 250
    rng = np.random.default_rng(3)
    rnd = lambda n: "".join("CGT"[i] for i in rng.integers(3, size=n))  # noqa: E731
    F = 160
    fl, tr, fr = rnd(F), "A" * 20, "A" * 125 + rnd(F - 125)
    chrom = rnd(500) + fl + tr + fr + rnd(500)
    write_fasta(str(tmp_path / "ref.fa"), {"chr1": chrom})
    locus = Locus(1, "locus1", "chr1", 500 + F, 500 + F + 20, "A", F)
    rd = get_locus_with_ref_data(locus, Fasta(str(tmp_path / "ref.fa")), context=gpu_ctx)
    exp = {mi: oracle.ref_repeat_count(20, tr, fl, fr, "A", 20, 5, mi, 3, 1) for mi in (100, 250)}
    assert exp[100][2] == 99 and exp[250][2] == 125 and exp[250][3][0] > 100
    assert (rd["ref_cn"], rd["right_coord_adj"] - locus.right_coord, rd["left_coord_adj"]) == (exp[250][0][0], 125, locus.left_coord)
    assert (rd["ref_left_flank_seq"], rd["ref_seq"], rd["ref_right_flank_seq"]) == exp[250][4]
