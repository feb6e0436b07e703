"""GPU parity of the DP scores: strk_score_table (HIP, through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from helpers import ALPHA_ACGT, ALPHA_IUPAC, ALPHA_WC, oracle_table, rand_seq, random_locus
from strkit_amd.synth import LocusBatch

pytestmark = pytest.mark.gpu


def _check(b, lo, n, flags=15, force_generic=False, ctx=None):
    from strkit_amd.batch import score_table
    got, st = score_table(b, lo, n, flags, force_generic, ctx=ctx, with_stats=True)
    exp = oracle_table(b, lo, n, flags)
    bad = [(r, got[r].tolist(), exp[r].tolist()) for r in range(b.n_reads) if not np.array_equal(got[r], exp[r])]
    assert not bad, f"{len(bad)} of {b.n_reads} reads differ; first: read {bad[0][0]} {b.read(bad[0][0])} " \
                    f"lo={lo[bad[0][0]]} got={bad[0][1]} exp={bad[0][2]}"
    return st


def _windows(rng, b, width=(1, 12)):
    lo = np.maximum(0, b.est_cn + rng.integers(-6, 3, size=b.n_reads)).astype(np.int32)
    n = rng.integers(width[0], width[1] + 1, size=b.n_reads).astype(np.int32)
    return lo, n


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_small_random_all_flags(gpu_ctx, seed):
    rng = np.random.default_rng(seed)
    for flags in (15, 0, 1, 2, 4, 8, 5, 10, 7, 11, 13, 14):
        loci = [random_locus(rng, 4, cn=(0, 12), flank=(1, 20), alpha=ALPHA_WC, edits=(0, 3)) for _ in range(40)]
        b = LocusBatch.from_reads(loci)
        lo, n = _windows(rng, b)
        st = _check(b, lo, n, flags, ctx=gpu_ctx)
        assert st["n_fallback"] == 0


def test_flank70_group16_classes(gpu_ctx):
    rng = np.random.default_rng(10)
    loci = [random_locus(rng, 6, motif_len=(2, 6), cn=(3, 60), flank=(60, 70), alpha=ALPHA_WC) for _ in range(120)]
    b = LocusBatch.from_reads(loci)
    lo, n = _windows(rng, b, (5, 17))
    st = _check(b, lo, n, ctx=gpu_ctx)
    assert st["n_fallback"] == 0


def test_group64_classes(gpu_ctx):
    rng = np.random.default_rng(11)
    loci = [random_locus(rng, 2, motif_len=(2, 20), cn=(20, 80), flank=(70, 70), alpha=ALPHA_WC, edits=(0, 12))
            for _ in range(40)]
    b = LocusBatch.from_reads(loci)
    assert (b.nfl + b.ntr + b.nfr).max() > 448
    lo, n = _windows(rng, b, (3, 9))
    st = _check(b, lo, n, ctx=gpu_ctx)
    assert st["n_fallback"] == 0


def test_iupac_motifs_and_many_symbols(gpu_ctx):
    rng = np.random.default_rng(12)
    loci = [random_locus(rng, 3, cn=(2, 15), flank=(5, 40), alpha=ALPHA_WC, motif_alpha=ALPHA_IUPAC) for _ in range(40)]
    # reads that hold more than 8 distinct symbols go to the generic kernel
    loci += [random_locus(rng, 2, cn=(2, 10), flank=(20, 40), alpha=ALPHA_IUPAC) for _ in range(20)]
    b = LocusBatch.from_reads(loci)
    lo, n = _windows(rng, b, (1, 8))
    st = _check(b, lo, n, ctx=gpu_ctx)
    assert st["n_fallback"] > 0


def test_generic_kernel_and_degenerate_shapes(gpu_ctx):
    rng = np.random.default_rng(13)
    loci = []
    for _ in range(30):
        motif = rand_seq(rng, int(rng.integers(1, 5)))
        reads = []
        for _ in range(3):
            fl = rand_seq(rng, int(rng.integers(0, 4)))
            fr = rand_seq(rng, int(rng.integers(0, 4)))
            tr = motif * int(rng.integers(0, 5))
            reads.append((fl, tr, fr))
        loci.append((motif, reads))
    b = LocusBatch.from_reads(loci)
    lo = rng.integers(0, 3, size=b.n_reads).astype(np.int32)
    n = rng.integers(1, 6, size=b.n_reads).astype(np.int32)
    _check(b, lo, n, ctx=gpu_ctx)
    _check(b, lo, n, 15, force_generic=True, ctx=gpu_ctx)
    for flags in (0, 6, 9):
        _check(b, lo, n, flags, ctx=gpu_ctx)


def test_long_windows_are_chunked(gpu_ctx):
    rng = np.random.default_rng(14)
    loci = [random_locus(rng, 2, cn=(20, 40), flank=(30, 70)) for _ in range(10)]
    b = LocusBatch.from_reads(loci)
    lo = np.zeros(b.n_reads, np.int32)
    n = (b.est_cn + 20).astype(np.int32)
    _check(b, lo, n, ctx=gpu_ctx)


def test_lowercase_and_unknown_bytes(gpu_ctx):
    loci = [("cag", [("acgtACGT" * 3, "cagCAGcag" * 3, "ttgacc" * 4), ("ACGT?ACGT", "CAG-CAG", "TTGA.CC")])]
    b = LocusBatch.from_reads(loci)
    _check(b, np.array([0, 0], np.int32), np.array([14, 6], np.int32), ctx=gpu_ctx)


def test_long_windows_use_the_tiled_kernel(gpu_ctx):
    """|db| beyond the widest fast class (1 792 slots): k_dp_long tiles the columns; 1, 2 and 3 tiles."""
    rng = np.random.default_rng(15)
    loci = []
    for cn, m in ((450, 4), (900, 3), (700, 6), (2400, 1), (1300, 2)):
        motif = rand_seq(rng, m)
        from helpers import noisy_tract
        fl, fr = rand_seq(rng, 70), rand_seq(rng, 70)
        reads = [(fl, noisy_tract(rng, motif, cn + int(rng.integers(-2, 3)), int(rng.integers(0, 9)), ALPHA_WC), fr)
                 for _ in range(2)]
        loci.append((motif, reads))
    b = LocusBatch.from_reads(loci)
    assert (b.nfl + b.ntr + b.nfr).min() > 1792
    lo = np.maximum(0, b.est_cn - 2).astype(np.int32)
    n = np.full(b.n_reads, 5, np.int32)
    st = _check(b, lo, n, ctx=gpu_ctx)
    assert st["n_fallback"] == 0
    _check(b, lo, n, 0, ctx=gpu_ctx)
    _check(b, lo, n, 6, ctx=gpu_ctx)
