"""GPU parity of strk_realign (C ABI) against the CPU oracle's restatement of parasail sg_dx_trace
(strkit/call/realign.py:56-72): score, end position and CIGAR must be identical."""
import numpy as np
import pytest

import oracle
from helpers import ALPHA_IUPAC, ALPHA_WC, cigar_tuples, rand_seq, realign_pair, rescore_cigar
from strkit_amd import _lib
from strkit_amd.realign import (get_aligned_pair_matches, perform_realign, realign_pairs, realign_read,
                                realign_reads)

pytestmark = pytest.mark.gpu


def check_pairs(refs, reads, open_=7, ext=0, gap_pref=0):
    got = realign_pairs(refs, reads, open_, ext, gap_pref)
    for p, (r, q) in enumerate(zip(refs, reads)):
        sc, e2, cg = oracle.realign(r, q, open_, ext, gap_pref)
        assert got[p][0] == sc, (p, len(r), len(q), got[p][0], sc)
        assert got[p][1] == e2, (p, len(r), len(q), got[p][1], e2)
        assert cigar_tuples(got[p][2]) == cigar_tuples(cg), (p, len(r), len(q))


def test_known_answers(gpu_ctx):
    res = realign_pairs(["ACGTACGTAC", "ACGTACGGGGGTAC", "ACGTACGTAC"],
                        ["TTTTACGTACGTACTTT", "TTTTACGTACGTACTTT", "TTTTACGTAAAAAAAAAAAAAAAAAAAAAAAAACGTACTTT"])
    assert res[0][0] == 20 and res[0][1] == 13 and cigar_tuples(res[0][2]) == [(4, "D"), (10, "=")]
    assert res[1][0] == 13 and cigar_tuples(res[1][2]) == [(4, "D"), (6, "="), (4, "I"), (4, "=")]
    assert res[2][0] == 13 and res[2][1] == 37 and cigar_tuples(res[2][2]) == [(4, "D"), (4, "="), (24, "D"), (6, "=")]


@pytest.mark.parametrize("open_,ext", [(7, 0), (7, 1), (5, 5), (3, 2), (0, 0)])
@pytest.mark.parametrize("gap_pref", [0, 1])
def test_small_random_pairs(gpu_ctx, open_, ext, gap_pref):
    rng = np.random.default_rng(1000 + 17 * open_ + 3 * ext + gap_pref)
    refs, reads = [], []
    for _ in range(60):
        alpha = [ALPHA_WC, "AC", ALPHA_IUPAC, "ACGT"][int(rng.integers(4))]
        if rng.random() < 0.5:
            r, q = realign_pair(rng, int(rng.integers(1, 120)), int(rng.integers(1, 400)), ins=int(rng.integers(0, 30)),
                                dele=int(rng.integers(0, 10)), sub=0.05, indel=0.05, alpha=alpha)
        else:
            r, q = rand_seq(rng, int(rng.integers(1, 60)), alpha), rand_seq(rng, int(rng.integers(1, 200)), alpha)
        refs.append(r)
        reads.append(q.lower() if rng.random() < 0.2 else q)
    check_pairs(refs, reads, open_, ext, gap_pref)


@pytest.mark.parametrize("n_ref", [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048])
def test_window_length_classes(gpu_ctx, n_ref):
    rng = np.random.default_rng(n_ref)
    refs, reads = [], []
    for k in range(3):
        r, q = realign_pair(rng, n_ref, n_ref + int(rng.integers(0, 600)), ins=(0, 40, 0)[k], dele=(0, 0, 7)[k] if n_ref > 20 else 0,
                            sub=0.02, indel=0.02, wc=0.01)
        refs.append(r)
        reads.append(q)
    check_pairs(refs, reads)
    check_pairs(refs[:1], reads[:1], 6, 2, 1)


@pytest.mark.parametrize("n_ref", [2049, 3000, 4097, 5000])
def test_column_tiled_windows(gpu_ctx, n_ref):
    rng = np.random.default_rng(n_ref)
    r, q = realign_pair(rng, n_ref, n_ref + 900, ins=120, dele=33, sub=0.01, indel=0.01)
    check_pairs([r], [q])
    check_pairs([r], [q], 7, 3, 1)


def test_hifi_soft_clip_shape(gpu_ctx):
    """Reference window 2*70 + TR + 1 inside a 15 kb read carrying a large expansion (call_locus.py:860-865)."""
    rng = np.random.default_rng(77)
    refs, reads, lfc = [], [], []
    for k in range(6):
        tr = "CAG" * int(rng.integers(10, 60))
        ref = rand_seq(rng, 70) + tr + rand_seq(rng, 71)
        left = int(rng.integers(2000, 9000))
        body = ref[:70] + "CAG" * int(rng.integers(60, 300)) + ref[70 + len(tr):]
        refs.append(ref)
        reads.append(rand_seq(rng, left) + body + rand_seq(rng, 15000 - left - len(body)))
        lfc.append(1_000_000 + 10 * k)
    check_pairs(refs, reads)
    res, st = realign_pairs(refs, reads, with_stats=True)
    assert st["dp_cells"] == sum(len(a) * len(b) for a, b in zip(refs, reads))
    coords = realign_reads(refs, reads, lfc, 70)
    for k, ac in enumerate(coords):
        assert ac is not None
        sc, i_end, j_end = rescore_cigar(refs[k], reads[k], res[k][2])
        assert (sc, i_end, j_end - 1) == (res[k][0], len(refs[k]), res[k][1])
        # left flank start and right flank end of the window are matched to read bases
        assert ac.ref_coords[0] == lfc[k] and ac.ref_coords[-1] == lfc[k] + len(refs[k]) - 1
        assert np.all(np.diff(ac.query_coords) > 0) and np.all(np.diff(ac.ref_coords) > 0)
    # a read that does not hold the window scores under the gate (realign.py:65) -> None
    assert realign_read(refs[0], rand_seq(rng, 3000), lfc[0], 70) is None


def test_perform_realign_mirror(gpu_ctx):
    class Obj:
        pass
    rng = np.random.default_rng(5)
    ref = rand_seq(rng, 200)
    read = rand_seq(rng, 500) + ref[:100] + rand_seq(rng, 50) + ref[100:] + rand_seq(rng, 400)
    quals = np.full(len(read), 30)
    quals[520:524] = 2          # -> X (threshold 3), still aligned: X scores 0 against a base
    lw, seg, prm = Obj(), Obj(), Obj()
    lw.ref_total_seq, lw.locus_def = ref, Obj()
    lw.locus_def.left_flank_coord = 5000
    seg.query_sequence, seg.query_qualities, seg.name = read, quals, "r1"
    prm.flank_size, prm.log_level = 70, 0
    ac = perform_realign(lw, seg, prm, None)
    assert ac is not None and len(ac) == 200
    assert ac.pair_at_idx(0) == (500, 5000) and ac.pair_at_idx(199) == (500 + 249, 5199)


def test_trace_budget_chunks(gpu_ctx, monkeypatch):
    rng = np.random.default_rng(9)
    pairs = [realign_pair(rng, int(rng.integers(50, 700)), int(rng.integers(800, 4000)), ins=20) for _ in range(40)]
    refs, reads = [p[0] for p in pairs], [p[1] for p in pairs]
    monkeypatch.setenv("STRKIT_AMD_TRACE_BYTES", str(2 << 20))
    check_pairs(refs, reads)


def test_bad_input(gpu_ctx):
    with pytest.raises(_lib.StrkError):
        realign_pairs(["ACGT", ""], ["ACGT", "ACGT"])
    with pytest.raises(_lib.StrkError):
        realign_pairs(["ACGT"], ["ACGT"], open_penalty=3, extend_penalty=5)
    assert realign_pairs([], []) == []


def test_hifi_read_shape_rate_and_parity(gpu_ctx, capsys):
    """The shape tools/bench_realign.py times (441-base window, 15 kb read): parity on a sample and, for the record,
    the rate of the CPU restatement next to the device's (printed with -s; reported in profiles/README.md)."""
    import time
    rng = np.random.default_rng(12)
    refs, reads = [], []
    for _ in range(128):
        ref = rand_seq(rng, 441)
        read = list(rand_seq(rng, 15000))
        pos = int(rng.integers(0, 15000 - 441 - 400))
        body = ref[:70] + rand_seq(rng, int(rng.integers(0, 300))) + ref[70:]
        read[pos:pos + len(body)] = body
        refs.append(ref)
        reads.append("".join(read))
    realign_pairs(refs[:8], reads[:8])
    t = time.perf_counter()
    got, st = realign_pairs(refs, reads, with_stats=True)
    gpu_wall = time.perf_counter() - t
    t = time.perf_counter()
    exp = [oracle.realign(refs[i], reads[i]) for i in range(8)]
    cpu = time.perf_counter() - t
    for i in range(8):
        assert (got[i][0], got[i][1], got[i][2].tolist()) == (exp[i][0], exp[i][1], exp[i][2].tolist())
    with capsys.disabled():
        print(f"\n[realign 441 x 15000] CPU restatement {8 / cpu:.1f} reads/s on one core; device {128 / gpu_wall:.0f} reads/s "
              f"wall for 128 reads ({st['kernel_ms']:.2f} ms of kernels)")


def test_golden_realign_cases(gpu_ctx):
    import json
    import os
    from strkit_amd.realign import cigar_to_string
    with open(os.path.join(os.path.dirname(__file__), "golden", "realign_cases.json")) as f:
        cases = json.load(f)
    groups = {}
    for c in cases:
        groups.setdefault((c["open"], c["extend"], c["gap_pref"]), []).append(c)
    for (open_, ext, pref), cs in groups.items():
        got = realign_pairs([c["ref"] for c in cs], [c["read"] for c in cs], open_, ext, pref)
        for c, (sc, e2, cg) in zip(cs, got):
            assert (sc, e2, cigar_to_string(cg)) == (c["score"], c["end_ref"], c["cigar"])
