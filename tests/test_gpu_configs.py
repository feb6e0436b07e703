"""GPU: BASELINE.json's configurations at their STATED shapes, each against the oracle where the
scalar CPU restatement finishes in seconds and through size-independent properties beyond that.

* config 1: the 44 loci of the reference's catalog of disease-associated repeats (tests/golden/pathogenic_assoc.hg38.bed,
  taken from catalogs/pathogenic_assoc.hg38.tsv: IUPAC motifs such as AARRG, RAAAT, GCN), every read against the oracle;
* config 4: one shard of the whole-genome shape with its 70 % / 30 % motif-length mix (1-6 / 7-20 bp), 40 whole loci
  against the oracle;
* config 5: reads at the top of the stated range (1 900-2 000 copies of a 6-mer: windows of ~12 kb, within 200 bases of
  what the widest band class holds) against the oracle, and the full 40-reads-per-locus shape over the whole 50-2 000
  copy range with the banded path checked against the exact kernels read by read.
"""
import os

import numpy as np
import pytest

from helpers import oracle_count
from strkit_amd.synth import make_catalog_batch, make_config

pytestmark = pytest.mark.gpu
KEYS = ("cn", "score", "n_iters", "start")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _count(b, ctx, **kw):
    from strkit_amd.batch import count_loci
    return count_loci(b, ctx=ctx, with_stats=True, **kw)


def _assert_same(b, got, exp, what=""):
    for k in KEYS:
        bad = np.nonzero(got[k] != exp[k])[0]
        assert bad.size == 0, (what, k, int(bad.size), int(bad[0]), [int(got[x][bad[0]]) for x in KEYS],
                               [int(exp[x][bad[0]]) for x in KEYS])


def test_config1_pathogenic_catalog(gpu_ctx):
    from strkit_amd.frontend.loci import load_loci
    blocks = load_loci(os.path.join(GOLDEN, "pathogenic_assoc.hg38.bed"))
    loci = [(l.motif, l.right_coord - l.left_coord) for blk in blocks for l in blk]
    assert len(loci) == 44 and sum(any(ch not in "ACGT" for ch in m) for m, _ in loci) >= 10   # IUPAC motifs
    b = make_catalog_batch(loci, seed=0xC0FFEE + 1)
    assert b.n_reads == 44 * 30
    exp = oracle_count(b)
    for kw in (dict(), dict(band=False), dict(dedupe=False)):
        got, st = _count(b, gpu_ctx, **kw)
        _assert_same(b, got, exp, str(kw))
        assert st["n_fallback"] == 0
    # almost every read is counted as the allele it was drawn from (the rest carry an indel in the tract)
    assert (exp["cn"] == b.true_cn).mean() > 0.98


def test_config4_shard_with_motif_mix(gpu_ctx):
    b = make_config(4, n_loci=20000)
    mlen = np.diff(b.motif_off)
    assert 0.25 < (mlen >= 7).mean() < 0.35 and mlen.max() > 15       # 70 % 1-6 bp, 30 % 7-20 bp
    base, st = _count(b, gpu_ctx)
    assert st["n_fallback"] == 0 and st["n_dedup_reads"] > 0
    rng = np.random.default_rng(4)
    picks = rng.choice(b.n_loci, size=40, replace=False)
    assert (mlen[picks] >= 7).sum() >= 5
    for l in picks:
        one = b.locus_slice(int(l), int(l) + 1)
        exp = oracle_count(one)
        r0 = int(b.read_off[l])
        for k in KEYS:
            assert np.array_equal(base[k][r0:r0 + one.n_reads], exp[k]), (int(l), k, one.motif(0))
    exact, _ = _count(b, gpu_ctx, band=False)
    _assert_same(b, base, exact, "band vs exact kernels")


def test_config5_top_of_the_stated_range(gpu_ctx):
    """1 900-2 000 copies of a 6-mer (CN 50-2 000): ~5 s of scalar CPU per read."""
    b = make_config(5, n_loci=2, reads_per_locus=2, cn_range=(1900, 2000), motif_len=(6, 6))
    ndb = b.nfl + b.ntr + b.nfr
    assert ndb.min() > 11000 and ndb.max() < 12288
    exp = oracle_count(b)
    from strkit_amd import _lib
    fresh = _lib.Context(0)     # the shared test context may have switched its band off after a noisy batch
    got, st = _count(b, fresh)
    fresh.close()
    _assert_same(b, got, exp, "banded")
    assert st["n_band_reads"] == b.n_reads and st["n_fallback"] == 0
    got0, st0 = _count(b, gpu_ctx, band=False)
    _assert_same(b, got0, exp, "exact")
    assert st0["n_band_reads"] == 0


def test_config5_full_shape(gpu_ctx):
    """40 reads per locus, motif 1-6 bp, 50-2 000 copies: the banded first pass (512 / 1 024 diagonals, certificate)
    against the exact long-window kernel on every read, plus the size-independent properties."""
    b = make_config(5, n_loci=48)
    assert b.n_reads == 48 * 40
    ndb = (b.nfl + b.ntr + b.nfr).astype(np.int64)
    assert ndb.max() > 8000 and ndb.min() < 1500
    from strkit_amd import _lib
    fresh = _lib.Context(0)          # the shared test context may have switched its band off after a noisy batch
    _count(b, fresh)                 # (a new context tries the band on a sample of the reads first)
    base, st = _count(b, fresh)
    fresh.close()
    assert st["n_fallback"] == 0 and st["n_band_reads"] > b.n_reads // 2
    exact, st0 = _count(b, gpu_ctx, band=False)
    _assert_same(b, base, exact, "band vs exact kernels")
    again, _ = _count(b, gpu_ctx, dedupe=False)
    _assert_same(b, base, again, "dedupe")
    h = b.n_loci // 2
    lo_half, hi_half = _count(b.locus_slice(0, h), gpu_ctx)[0], _count(b.locus_slice(h, b.n_loci), gpu_ctx)[0]
    for k in KEYS:
        assert np.array_equal(base[k], np.concatenate([lo_half[k], hi_half[k]])), k
    assert (base["score"] <= 2 * ndb).all() and (base["n_iters"] >= 1).all()
    # HiFi reads: the counted size stays near the drawn allele (0.2 % indels move a 1-2 bp motif's count by a few units
    # per thousand copies)
    assert (np.abs(base["cn"] - b.true_cn) <= 2 + b.true_cn // 100).mean() > 0.97
    # oracle on the two loci with the shortest windows
    per_locus = np.add.reduceat(ndb ** 2, b.read_off[:-1].astype(np.int64))
    for l in np.argsort(per_locus)[:2]:
        one = b.locus_slice(int(l), int(l) + 1)
        exp = oracle_count(one)
        r0 = int(b.read_off[l])
        for k in KEYS:
            assert np.array_equal(base[k][r0:r0 + one.n_reads], exp[k]), (int(l), k)
