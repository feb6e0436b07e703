"""GPU parity of the whole per-read path (DP + search + caller feedback) vs the CPU oracle."""
import numpy as np
import pytest

from helpers import ALPHA_ACGT, ALPHA_IUPAC, ALPHA_WC, oracle_count, rand_seq, random_locus
from strkit_amd.synth import LocusBatch, make_config

pytestmark = pytest.mark.gpu
KEYS = ("cn", "score", "n_iters", "start")


def _compare(b, got, exp):
    for k in KEYS:
        bad = np.nonzero(got[k] != exp[k])[0]
        assert bad.size == 0, f"{k}: {bad.size} of {b.n_reads} reads differ; first read {bad[0]}: " \
                              f"got {[int(got[x][bad[0]]) for x in KEYS]} exp {[int(exp[x][bad[0]]) for x in KEYS]} " \
                              f"{b.read(int(bad[0]))}"


def _run(b, ctx, **kw):
    from strkit_amd.batch import count_loci
    from strkit_amd.repeat_count_params import RepeatCountParams
    rc = RepeatCountParams("repalign", kw.pop("max_iters", 50), kw.pop("lsr", 3), kw.pop("step", 1))
    return count_loci(b, rc, ctx=ctx, with_stats=True, **kw)


@pytest.mark.parametrize("cfg,n_loci", [(1, 44), (2, 150), (3, 60)])
def test_configs_match_oracle(gpu_ctx, cfg, n_loci):
    b = make_config(cfg, n_loci=n_loci)
    got, st = _run(b, gpu_ctx)
    _compare(b, got, oracle_count(b))
    assert st["dp_cells"] > 0


def test_window_miss_path(gpu_ctx):
    """A 1-wide speculative window forces almost every read through the host-driven miss rounds."""
    rng = np.random.default_rng(21)
    loci = [random_locus(rng, 8, motif_len=(2, 6), cn=(5, 30), flank=(30, 70), alpha=ALPHA_WC) for _ in range(40)]
    b = LocusBatch.from_reads(loci)
    b.est_cn = np.maximum(0, b.est_cn + rng.integers(-9, 10, size=b.n_reads)).astype(np.int32)  # bad estimates
    exp = oracle_count(b)
    for window in (1, 3, 8):
        got, st = _run(b, gpu_ctx, window=window)
        _compare(b, got, exp)
        if window == 1:
            assert st["n_miss_reads"] > 0


@pytest.mark.parametrize("kw", [dict(feedback=False), dict(tie_rule=1), dict(max_iters=5), dict(lsr=1, step=2),
                                dict(lsr=1, step=15, max_iters=50), dict(end_flags=0), dict(end_flags=10),
                                dict(narrowing=1), dict(narrowing=2, lsr=5), dict(narrowing=3), dict(narrowing=1, lsr=4, step=2, tie_rule=1),
                                dict(narrowing=3, lsr=5, feedback=False, band=False)])
def test_search_variants(gpu_ctx, kw):
    rng = np.random.default_rng(22)
    loci = [random_locus(rng, 6, motif_len=(1, 6), cn=(0, 25), flank=(10, 70), alpha=ALPHA_WC) for _ in range(50)]
    if "narrowing" in kw:   # estimates a few sizes off: the schedules differ only when the search has to walk
        loci += [random_locus(rng, 8, motif_len=(2, 6), cn=(5, 40), flank=(70, 70), edits=(0, 2)) for _ in range(60)]
    b = LocusBatch.from_reads(loci)
    if "narrowing" in kw:
        b.est_cn = np.maximum(0, b.est_cn + rng.integers(-7, 8, size=b.n_reads)).astype(np.int32)
    got, _ = _run(b, gpu_ctx, **dict(kw))
    okw = dict(max_iters=kw.get("max_iters", 50), lsr=kw.get("lsr", 3), step=kw.get("step", 1), narrowing=kw.get("narrowing", 0),
               tie_rule=kw.get("tie_rule", 0), flags=kw.get("end_flags", 15), feedback=kw.get("feedback", True))
    _compare(b, got, oracle_count(b, **okw))


def test_scalar_get_repeat_count(gpu_ctx):
    import oracle
    from strkit_amd.repeat_count_params import default_read_rc_params
    from strkit_amd.repeats import get_repeat_count
    rng = np.random.default_rng(23)
    rc = default_read_rc_params()
    for _ in range(25):
        motif, reads = random_locus(rng, 1, cn=(0, 30), flank=(20, 70), alpha=ALPHA_WC)
        fl, tr, fr = reads[0]
        start = max(0, round(len(tr) / len(motif)) + int(rng.integers(-12, 13)))
        assert get_repeat_count(start, tr, fl, fr, motif, rc) == oracle.repeat_count(start, tr, fl, fr, motif)


def test_empty_and_ragged_batches(gpu_ctx):
    from strkit_amd.batch import count_loci
    empty = LocusBatch.from_reads([])
    assert count_loci(empty, ctx=gpu_ctx)["cn"].size == 0
    b = LocusBatch.from_reads([("CAG", []), ("AT", [("ACGTTGCA", "ATATATAT", "GGCCTTAA")]), ("A", [])])
    got = count_loci(b, ctx=gpu_ctx)
    exp = oracle_count(b)
    for k in KEYS:
        assert np.array_equal(got[k], exp[k])


def test_identical_reads_share_one_table(gpu_ctx):
    """Dedupe (the device-side lru_cache): byte-identical reads of a locus are scored once; answers do
    not change, including when the start-count feedback gives the copies different starts."""
    rng = np.random.default_rng(24)
    loci = []
    for _ in range(30):
        motif, reads = random_locus(rng, 3, motif_len=(2, 6), cn=(4, 30), flank=(40, 70), alpha=ALPHA_WC)
        reads = [reads[i % 3] for i in rng.integers(0, 3, size=12)]  # many copies, shuffled
        loci.append((motif, reads))
    b = LocusBatch.from_reads(loci)
    b.est_cn[::7] += 2        # same bytes but another estimate: must NOT be merged
    b.est_cn[3::11] = np.maximum(0, b.est_cn[3::11] - 3)
    exp = oracle_count(b)
    got, st = _run(b, gpu_ctx)
    _compare(b, got, exp)
    assert st["n_dedup_reads"] > b.n_reads // 3
    got2, st2 = _run(b, gpu_ctx, dedupe=False)
    _compare(b, got2, exp)
    assert st2["n_dedup_reads"] == 0 and st2["dp_cells"] > st["dp_cells"]
    got3, _ = _run(b, gpu_ctx, window=1)  # copies through the miss path too
    _compare(b, got3, exp)


def test_expansion_stress_long_reads(fresh_ctx):
    """BASELINE config 5 shape (motif 1-6, hundreds of copies): windows of several kb through k_dp_long."""
    gpu_ctx = fresh_ctx   # (the shared context may be in a band cool-down after the noisy batches of earlier tests)
    b = make_config(5, n_loci=3, reads_per_locus=4, cn_range=(700, 1100), motif_len=(3, 6))
    assert (b.nfl + b.ntr + b.nfr).max() > 1792
    exp = oracle_count(b)
    got, st = _run(b, gpu_ctx)                 # banded: 512 / 1 024 diagonals, rows generated on the fly
    _compare(b, got, exp)
    assert st["n_fallback"] == 0 and st["n_band_reads"] > 0
    got0, st0 = _run(b, gpu_ctx, band=False)   # exact: k_dp_long
    _compare(b, got0, exp)
    assert st0["n_band_reads"] == 0 and st0["dp_cells"] > st["dp_cells"]


def test_band_kernel_certifies_hifi_and_falls_back_on_noise(fresh_ctx):
    """k_dp_band: lower bounds + certificate.  HiFi reads certify, noisy reads are re-scored exactly; the
    answers never differ from the exact kernels'."""
    gpu_ctx = fresh_ctx
    b = make_config(2, n_loci=120)
    exp = oracle_count(b)
    got, st = _run(b, gpu_ctx)
    _compare(b, got, exp)
    assert st["n_band_reads"] > b.n_reads // 4, st
    assert st["n_band_fallback"] < st["n_band_reads"] // 5, st
    got0, st0 = _run(b, gpu_ctx, band=False)
    _compare(b, got0, exp)
    assert st0["n_band_reads"] == 0
    # ONT-like noise: mostly uncertified -> exact path; results identical
    b3 = make_config(3, n_loci=60, motif_len=(2, 6), cn_range=(30, 60))
    got3, st3 = _run(b3, gpu_ctx)
    _compare(b3, got3, oracle_count(b3))
    # low-complexity windows (many off-band matches) with bad estimates and a narrow speculative window
    rng = np.random.default_rng(25)
    loci = []
    for _ in range(40):
        motif, reads = random_locus(rng, 6, motif_len=(1, 4), cn=(30, 70), flank=(60, 70), alpha="AC", edits=(0, 3))
        loci.append((motif, reads))
    bl = LocusBatch.from_reads(loci)
    bl.est_cn = np.maximum(0, bl.est_cn + rng.integers(-3, 4, size=bl.n_reads)).astype(np.int32)
    expl = oracle_count(bl)
    for window in (0, 4):
        gotl, stl = _run(bl, gpu_ctx, window=window)
        _compare(bl, gotl, expl)


def test_reads_with_more_than_eight_symbol_classes_are_searched_on_the_generic_table(gpu_ctx):
    """IUPAC motif + wildcarded read: nine distinct symbols exceed a v_perm word, the read is scored by the generic
    kernel, which does not search; k_replay must search its table even when the start equals the estimate (no
    feedback).  Found by tests/test_gpu_fuzz.py (seed 12695296)."""
    fl = "GAAAXCNNACAAATNNANXTXTGGCGTXNNTNGAAXCCAGTXCTGXCTXAGNAXGATGTTTXCCATCNTNGAXNXAANNXACGXNX"
    fr = "CATANCTAXAXANTATTCXNAGCGTXCGCGCATNNGNAXTACGCTCGTGTNGTGAXATX"
    b = LocusBatch.from_reads([("VKB", [(fl, "VKB" * 23, fr), (fl, "VKB" * 21 + "VK", fr)]), ("CAG", [("ACGTTGCA" * 5, "CAG" * 9, "TTGACCA" * 6)])])
    for feedback in (False, True):
        for band in (False, True):
            got, st = _run(b, gpu_ctx, max_iters=7, lsr=1, feedback=feedback, band=band)
            assert st["n_fallback"] == 2
            _compare(b, got, oracle_count(b, 7, 1, 1, 0, 15, feedback))


def test_generic_pool_grows_for_a_batch_of_flankless_reads(monkeypatch):
    """Reads without a left flank go to the generic kernel, whose H rows live in a pool that a context sizes once.  A call
    that asks for more must grow the pool and run again instead of failing with STRK_E_NOMEM — found by tools/fuzz_parity.py
    (seed 2), where a few hundred ragged loci failed a whole call (one row per thread then; one per wave now, so the contexts
    of this test start with a pool of 64 Ki ints: about eighty rows).  The first and the last loci are checked against the
    oracle; the window-miss rounds and the score-table entry point take the same way."""
    from strkit_amd import _lib
    from strkit_amd.batch import score_table
    monkeypatch.setenv("STRKIT_AMD_GENERIC_POOL_INTS", str(64 << 10))
    fresh_ctx = _lib.Context(0)
    rng = np.random.default_rng(20261006)
    loci = []
    for _ in range(250):
        motif = rand_seq(rng, int(rng.integers(2, 7)))
        fr = rand_seq(rng, 70)
        cn = int(rng.integers(90, 130))
        loci.append((motif, [("", motif * (cn + int(rng.integers(-2, 3))), fr) for _ in range(24)]))
    b = LocusBatch.from_reads(loci)
    got, st = _run(b, fresh_ctx, dedupe=False)
    assert st["n_fallback"] == b.n_reads
    for lo, hi in ((0, 3), (b.n_loci - 2, b.n_loci)):
        part = b.locus_slice(lo, hi)
        r0, r1 = int(b.read_off[lo]), int(b.read_off[hi])
        _compare(part, {k: got[k][r0:r1] for k in KEYS}, oracle_count(part))
    # the same in a window-miss round: the first read of each locus starts at three times its size, and the widened windows
    # (hundreds of candidate sizes, one row each) of 200 loci are re-scored together
    rng = np.random.default_rng(20261007)
    loci, est = [], []
    for _ in range(200):
        motif = rand_seq(rng, int(rng.integers(2, 7)))
        fr = rand_seq(rng, 70)
        cn = int(rng.integers(100, 120))
        loci.append((motif, [("", motif * cn, fr), ("", motif * (cn + 1), fr)]))
        est.append([3 * cn + 1, cn])
    bm = LocusBatch.from_reads(loci, est)
    ctx3 = _lib.Context(0)
    try:
        gotm, stm = _run(bm, ctx3, dedupe=False, step=5, lsr=2)
        assert stm["n_miss_reads"] >= 200
        for lo, hi in ((0, 1), (bm.n_loci - 1, bm.n_loci)):
            part = bm.locus_slice(lo, hi)
            r0, r1 = int(bm.read_off[lo]), int(bm.read_off[hi])
            _compare(part, {k: gotm[k][r0:r1] for k in KEYS}, oracle_count(part, 50, 2, 5))
    finally:
        ctx3.close()
    # the explicit-window entry point on a context of its own (a fresh pool): 31 candidates per read, every read generic
    ctx2 = _lib.Context(0)
    try:
        lo_w = np.maximum(b.est_cn - 15, 0).astype(np.int32)
        n_w = np.full(b.n_reads, 31, np.int32)
        tab = score_table(b, lo_w, n_w, ctx=ctx2)
        import oracle
        for r in (0, 1, b.n_reads - 1):
            fl, tr, fr = b.read(r)
            l = int(np.searchsorted(b.read_off, r, side="right") - 1)
            exp = [oracle.candidate_score(tr, fl, fr, b.motif(l), int(lo_w[r]) + k) for k in range(31)]
            assert tab[r].tolist() == exp
    finally:
        ctx2.close()
        fresh_ctx.close()


def test_long_kernel_slot_grows_for_a_start_far_above_the_tract(fresh_ctx):
    """A start count of 9 000 copies of an 18-base motif is a candidate of 162 000 rows against a window of 1 100 bases: the long
    kernel's slot (two boundary columns of one int per row) starts at 48 Ki ints.  In-locus feedback produces such starts from
    one wild estimate (tools/fuzz_parity.py, seed 2, round 138: the call failed with STRK_E_NOMEM); the context must grow the
    slots and run the call again."""
    rng = np.random.default_rng(20261008)
    motif = rand_seq(rng, 18)
    fl, fr = rand_seq(rng, 51), rand_seq(rng, 72)
    b = LocusBatch.from_reads([(motif, [(fl, motif * 55, fr), (fl, motif * 54 + motif[:7], fr)]), ("CAG", [(fl, "CAG" * 20, fr)])],
                              [[9000, 8000], [20]])
    got, st = _run(b, fresh_ctx, max_iters=5, feedback=False)
    assert st["n_long_reads"] >= 2
    _compare(b, got, oracle_count(b, 5, 3, 1, 0, 15, False))
    got2, _ = _run(b, fresh_ctx, max_iters=5, feedback=False)      # (the grown context: no second run needed, same answers)
    _compare(b, got2, got)


@pytest.mark.parametrize("flags", list(range(16)))
def test_band_kernel_under_every_end_flag_mode(gpu_ctx, flags):
    """The banded first pass carries the two free boundaries as table content (pad bytes of the row words, the "no row"
    symbol's word) and hands the boundary value in at one step per pass: every combination of the four end-gap flags, on
    clean and on lightly mutated reads, must give the oracle's answers with the band doing the work; reads whose band
    meets column 0 / row 0 at different steps share a wave (short and long left flanks, small and large copy numbers)."""
    rng = np.random.default_rng(100 + flags)
    loci = []
    for k in range(24):
        motif, reads = random_locus(rng, 5, motif_len=(1, 6), cn=(2, 45), flank=((10, 70) if k % 3 else (60, 110)), edits=(0, 2))
        loci.append((motif, reads))
    b = LocusBatch.from_reads(loci)
    exp = oracle_count(b, flags=flags)
    from strkit_amd import _lib
    ctx = _lib.Context(0)      # a context of its own: the band of a shared one may be in a cool-down after noisy batches
    try:
        got, st = _run(b, ctx, end_flags=flags, window=6)      # (a fixed window: the default one adapts to earlier calls)
        _compare(b, got, exp)
        assert st["n_band_reads"] >= b.n_reads // 4, st
        got0, st0 = _run(b, ctx, end_flags=flags, band=False, window=6)
        _compare(b, got0, exp)
        assert st0["n_band_reads"] == 0
        if flags == 15:
            assert st["n_band_fallback"] <= st["n_band_reads"] // 3, st
    finally:
        ctx.close()


def test_host_buffer_entry_point_is_pipelined_and_equal(monkeypatch):
    """strk_count_loci on host buffers cuts a large batch into sub-batches of whole loci that travel through three pinned
    slots (strk_host_pipe.inc): same answers as the oracle and as the one-piece path, empty loci and a window-miss round
    inside a sub-batch included; the statistics of the sub-batches add up."""
    from strkit_amd import _lib
    monkeypatch.setenv("STRKIT_AMD_PIPE_MB", "1")           # sub-batches of ~1 MB: a dozen of them for this batch
    b = LocusBatch.concat([make_config(2, n_loci=700, seed_shift=5), make_config(3, n_loci=40, seed_shift=6),
                           make_config(2, n_loci=600, seed_shift=7)])
    rng = np.random.default_rng(5)
    b.est_cn = b.est_cn.copy()
    b.est_cn[rng.integers(0, b.n_reads, size=30)] += 11        # bad estimates: some searches leave their window
    exp = oracle_count(b)
    ctx = _lib.Context(0)
    try:
        got, st = _run(b, ctx)
        _compare(b, got, exp)
        assert st["n_sub_batches"] >= 6 and st["n_dp_launches"] >= 2 * st["n_sub_batches"], st   # launch rounds add up over the sub-batches
        assert st["n_band_reads"] + st["n_dedup_reads"] > b.n_reads // 2 and st["kernel_ms"] > 0
        monkeypatch.setenv("STRKIT_AMD_PIPE_MB", "4096")     # too small to cut up: the direct path
        got1, st1 = _run(b, ctx)
        _compare(b, got1, exp)
        assert st1["n_dp_launches"] == 2 and st1["n_sub_batches"] == 0
        assert st1["band_cells"] + st1["wide_cells"] + st1["exact_cells"] + st1["long_cells"] <= st1["dp_cells"] and st1["band_cells"] > 0
    finally:
        ctx.close()


def test_narrow_band_class_takes_short_motifs(gpu_ctx):
    """Motifs of up to four bases run in the 96-diagonal class (8 lanes x 12 diagonals), longer ones in the 128-diagonal
    class; both certify HiFi-like reads and agree with the oracle (strk_search.h: band_geometry)."""
    rng = np.random.default_rng(77)
    for mlen in ((2, 4), (5, 6)):
        loci = [random_locus(rng, 12, motif_len=mlen, cn=(8, 50), flank=(70, 70), edits=(0, 2)) for _ in range(60)]
        b = LocusBatch.from_reads(loci)
        from strkit_amd import _lib
        ctx = _lib.Context(0)
        try:
            got, st = _run(b, ctx, window=6)
            _compare(b, got, oracle_count(b))
            assert st["n_band_reads"] >= b.n_reads // 3 and st["n_band_fallback"] <= st["n_band_reads"] // 5, st
        finally:
            ctx.close()


def test_staircase_fork_rows_in_the_wide_exact_classes(gpu_ctx):
    """Windows of 321 bases and more run with 16 / 32 / 64 lanes per read and fold their fork rows along a staircase
    (strk_dp_exact.h); items whose smallest candidate has fewer motif rows than lanes take the per-lane fold, alone or in
    one chunk with others.  Noisy reads, long and short motifs, every end-flag family that the fold touches."""
    rng = np.random.default_rng(78)
    loci = []
    for k in range(36):
        mlen = (2, 6) if k % 3 == 0 else ((7, 13) if k % 3 == 1 else (14, 20))
        hi = 1500 // mlen[1]
        loci.append(random_locus(rng, 4, motif_len=mlen, cn=(320 // mlen[0], max(330 // mlen[0], hi)), flank=(30, 70), edits=(0, 40), alpha=ALPHA_WC))
    loci.append(random_locus(rng, 4, motif_len=(19, 20), cn=(16, 18), flank=(70, 70), edits=(0, 6)))   # window reaches candidate 1
    for k in range(12):                                                                                  # 16 lanes per read: windows of 321-448
        mlen = (2, 5) if k % 2 else (8, 16)
        loci.append(random_locus(rng, 4, motif_len=mlen, cn=(200 // mlen[1], 290 // mlen[1]), flank=(60, 70), edits=(0, 25), alpha=ALPHA_WC))
    b = LocusBatch.from_reads(loci)
    for flags, fb in ((15, True), (0, False), (5, True), (10, True)):
        got, st = _run(b, gpu_ctx, band=False, end_flags=flags, feedback=fb, window=15)
        _compare(b, got, oracle_count(b, flags=flags, feedback=fb))
        assert st["n_fallback"] == 0


def test_unknown_narrowing_schedule_is_rejected(gpu_ctx):
    from strkit_amd import _lib
    from strkit_amd.batch import batch_struct, make_params
    import ctypes as C
    b = make_config(2, n_loci=3)
    s, keep = batch_struct(b)
    p = make_params(narrowing=4)
    outs = [np.zeros(b.n_reads, np.int32) for _ in range(4)]
    rc = _lib.load().strk_count_loci(gpu_ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], None)
    assert rc == -22 and b"narrowing" in _lib.load().strk_last_error()


@pytest.mark.gpu
def test_host_entry_point_reads_page_locked_arrays_in_place(gpu_ctx):
    """strk_host_register: the bases of a batch that are page-locked are uploaded by DMA from where they lie (no staging copy);
    the results are those of the pageable path, and unregistering gives the pageable path back."""
    import ctypes as C
    from strkit_amd import _lib
    from strkit_amd.batch import batch_struct, make_params, pin_batch, unpin_batch
    from strkit_amd.synth import LocusBatch, make_config
    b = LocusBatch.concat([make_config(2, n_loci=1000, seed_shift=70 + k) for k in range(3)])   # ~30 MB: several sub-batches
    L = _lib.load()
    p = make_params()

    def run():
        s, keep = batch_struct(b)
        outs = [np.zeros(b.n_reads, np.int32) for _ in range(4)]
        st = _lib.StrkStats()
        _lib.check(L.strk_count_loci(gpu_ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
        return outs, int(st.n_sub_batches), keep

    plain, n_sub, _ = run()
    assert n_sub > 1
    assert not _lib.host_is_pinned(b.seqs)
    pin_batch(b)
    try:
        assert _lib.host_is_pinned(b.seqs)
        locked, n_sub2, keep = run()
        assert keep["seqs"].ctypes.data == b.seqs.ctypes.data   # batch_struct hands the registered arrays on, not copies
        for x, y in zip(plain, locked):
            assert np.array_equal(x, y)
    finally:
        unpin_batch(b)
    assert not _lib.host_is_pinned(b.seqs)
    again, _, _ = run()
    for x, y in zip(plain, again):
        assert np.array_equal(x, y)


def test_pin_cases_fixture(gpu_ctx):
    """tests/golden/pin_cases.json through the C ABI: lower-case windows, empty tracts, start 0, IUPAC motifs against wildcards,
    constructed ties under both tie rules (read side) and the boundary-tie loci of the reference side."""
    import json
    import os
    from strkit_amd.batch import count_loci
    from strkit_amd.repeat_count_params import RepeatCountParams
    from strkit_amd.repeats import get_ref_repeat_count
    with open(os.path.join(os.path.dirname(__file__), "golden", "pin_cases.json")) as f:
        cases = json.load(f)
    for name, case in cases.items():
        if name == "ref_ties":
            continue
        b = LocusBatch.from_reads([(l["motif"], [tuple(r) for r in l["reads"]]) for l in case["loci"]], [l["est_cn"] for l in case["loci"]])
        for tag, tie in (("expected_first_max", 0), ("expected_last_max", 1)):
            got = count_loci(b, ctx=gpu_ctx, tie_rule=tie)
            for k, v in case[tag].items():
                assert got[k].tolist() == v, (name, tag, k)
    ref = cases["ref_ties"]
    rc = RepeatCountParams("repalign", 50, 3, 1)
    k = 0
    for l in ref["loci"]:
        for a, tr, c in l["reads"]:
            for respect in (False, True):
                res = get_ref_repeat_count(round(len(tr) / len(l["motif"])), tr, a, c, l["motif"], len(tr), 5, rc, respect_coords=respect,
                                           context=gpu_ctx)
                assert json.loads(json.dumps(res)) == ref["expected"][k], (l["motif"], respect)
                k += 1



def test_scalar_fast_path_equals_the_oracle_on_every_shape(gpu_ctx):
    """strk_repeat_count's short launch chain (one upload, k_scalar_plan + k_dp_all on one block, one download) and the
    general path it falls back to — long windows, empty flanks, IUPAC-rich reads, starts far from the tract, start 0, other
    search parameters — against the oracle; and what one call costs."""
    import time
    import oracle
    from strkit_amd.repeat_count_params import RepeatCountParams, default_read_rc_params
    from strkit_amd.repeats import get_repeat_count
    rng = np.random.default_rng(77)
    rc = default_read_rc_params()
    cases = []
    for it in range(60):
        alpha = (ALPHA_WC, ALPHA_ACGT, ALPHA_IUPAC)[it % 3]
        motif, reads = random_locus(rng, 1, motif_len=(1, 12), cn=(0, 60), flank=(1, 90), alpha=alpha)
        fl, tr, fr = reads[0]
        shift = int(rng.integers(-3, 4)) if it % 4 else int(rng.integers(-30, 31))
        cases.append((max(0, round(len(tr) / len(motif)) + shift), tr, fl, fr, motif, rc))
    fl, fr = rand_seq(rng, 70), rand_seq(rng, 70)
    cases += [(0, "", fl, fr, "CAG", rc), (3, "CAGCAGCAG", "", fr, "CAG", rc), (3, "CAGCAGCAG", fl, "", "CAG", rc),
              (400, "CAG" * 400, fl, fr, "CAG", rc),                                  # 1 340 bases: the widest fast class
              (700, "AT" * 700 + "A", fl, fr, "AT", rc),                                # beyond it: the general path (k_dp_long)
              (12, "CAG" * 12, fl, fr, "CAG", RepeatCountParams("repalign", 5, 3, 1)),  # max_iters cut-off
              (12, "CAG" * 20, fl, fr, "CAG", RepeatCountParams("repalign", 50, 5, 2)),
              (2, "CAG" * 25, fl, fr, "CAG", rc)]                                       # a chase that leaves the window: general path
    for start, tr, a, c, motif, p in cases:
        get_repeat_count.cache_clear()
        try:
            exp = oracle.repeat_count(start, tr, a, c, motif, p.max_iters, p.initial_local_search_range, p.initial_step_size)
        except ValueError:
            with pytest.raises(ValueError):
                get_repeat_count(start, tr, a, c, motif, p)
            continue
        assert get_repeat_count(start, tr, a, c, motif, p) == exp, (start, len(tr), len(a), len(c), motif)
    # latency of the drop-in (lru_cache bypassed): a 30-copy CAG tract with 70-base flanks
    tr = "CAG" * 30
    t0 = time.perf_counter()
    n = 300
    for k in range(n):
        get_repeat_count.__wrapped__(30 + (k % 3) - 1, tr, fl, fr, "CAG", rc)
    us = (time.perf_counter() - t0) / n * 1e6
    print(f"\n[scalar drop-in] {us:.1f} us per get_repeat_count call (round 3: about 100 us through the batched path)")
    assert us < 500
