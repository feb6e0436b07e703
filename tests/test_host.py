"""CPU: host logic and the C-ABI surface (no compute calls: there is no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import oracle_count
from strkit_amd import _build, _lib
from strkit_amd.repeat_count_params import RepeatCountParams, default_read_rc_params, get_reference_rc_params
from strkit_amd.synth import CONFIGS, LocusBatch, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "strkit_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(strk_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load(build=False)
    declared = _header_functions()
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    # the code object inside is gfx950 only
    blob = open(path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert b"amdhsa--" + other not in blob and other + b"\0" not in blob[:0]
    assert b"gfx950" in lib.strk_version()


def test_host_register_rejects_bad_arguments_before_touching_the_device():
    lib = _lib.load(build=False)
    assert lib.strk_host_register(None, 16) == _lib.STRK_E_INVALID
    assert b"strk_host_register" in lib.strk_last_error()
    buf = (C.c_uint8 * 16)()
    assert lib.strk_host_register(C.cast(buf, C.c_void_p), 0) == _lib.STRK_E_INVALID
    assert lib.strk_host_unregister(None) == _lib.STRK_E_INVALID
    assert lib.strk_host_is_pinned(None, 16) == 0


def test_init_fails_loudly_without_a_gpu_and_errors_are_reported():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    lib = _lib.load(build=False)
    h = C.c_void_p()
    rc = lib.strk_init(0, C.byref(h))
    assert rc != 0 and not h.value
    assert lib.strk_last_error()
    with pytest.raises(_lib.StrkError):
        _lib.Context(0)
    # NULL context / bad arguments are rejected, never dereferenced
    assert lib.strk_count_loci(None, None, None, None, None, None, None, None) == -22
    assert lib.strk_score_table(None, None, None, None, None, 15, 0, None, None) == -22
    from strkit_amd.repeats import get_repeat_count
    with pytest.raises(RuntimeError):  # product path has no CPU fallback
        get_repeat_count(3, "CAGCAGCAG", "ACGT", "TTGA", "CAG", default_read_rc_params())


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.StrkParams) == 40
    assert C.sizeof(_lib.StrkStats) == 168
    assert C.sizeof(_lib.StrkBatch) == 8 + 9 * 8


def test_repeat_count_params_mirror_the_reference():
    p = default_read_rc_params()
    assert (p.method, p.max_iters, p.initial_local_search_range, p.initial_step_size) == ("repalign", 50, 3, 1)
    assert hash(p) == hash(RepeatCountParams("repalign", 50, 3, 1))  # lru_cache key (repeats.py:47)
    with pytest.raises(Exception):
        p.max_iters = 3  # frozen
    # strkit/call/repeat_count_params.py:17-42
    g = lambda cn: get_reference_rc_params("repalign", cn, 250)
    assert (g(10).max_iters, g(10).initial_step_size, g(10).initial_local_search_range) == (250, 1, 3)
    assert (g(199).max_iters, g(199).initial_step_size) == (250, 1)
    assert (g(200).max_iters, g(200).initial_step_size, g(200).initial_local_search_range) == (200, 3, 3)
    assert (g(999).max_iters, g(999).initial_step_size) == (200, 3)
    assert (g(1000).max_iters, g(1000).initial_step_size) == (150, 5)
    assert (g(1999).max_iters, g(1999).initial_step_size) == (150, 5)
    assert (g(2000).max_iters, g(2000).initial_step_size, g(2000).initial_local_search_range) == (50, 15, 1)


def test_synthetic_generator_is_deterministic_and_shaped():
    a, b = make_config(2, n_loci=20), make_config(2, n_loci=20)
    assert np.array_equal(a.seqs, b.seqs) and np.array_equal(a.est_cn, b.est_cn)
    assert not np.array_equal(a.seqs[:1000], make_config(2, n_loci=20, seed_shift=1).seqs[:1000])
    assert a.n_loci == 20 and a.n_reads == 20 * CONFIGS[2]["reads_per_locus"]
    assert (a.seq_off[1:] - a.seq_off[:-1] == a.nfl + a.ntr + a.nfr).all()
    for l in range(a.n_loci):
        assert 3 <= len(a.motif(l)) <= 6
    fl, tr, fr = a.read(0)
    assert 60 <= len(fl) <= 80 and 60 <= len(fr) <= 80
    assert a.algorithmic_bytes() == int(a.seq_off[-1]) + 16 * a.n_reads + int(a.motif_off[-1]) + 8 * a.n_loci
    s = a.locus_slice(5, 9)
    assert s.n_loci == 4 and s.read(0) == a.read(int(a.read_off[5])) and s.motif(3) == a.motif(8)


def test_sharding_partitions_loci_and_is_balanced():
    from strkit_amd.sharding import deal_blocks, select_loci
    b = make_config(3, n_loci=90)
    for world in (1, 2, 3, 8):
        shares = deal_blocks(b, world, block=7)
        allv = np.sort(np.concatenate(shares))
        assert np.array_equal(allv, np.arange(b.n_loci))  # every locus exactly once
        if world > 1:
            loads = [sum(int(b.read_off[l + 1] - b.read_off[l]) for l in s) for s in shares]
            assert max(loads) <= 1.5 * (sum(loads) / world) + 7 * 20
    sub, reads = select_loci(b, deal_blocks(b, 2, block=7)[1])
    for i, r in enumerate(reads[:50]):
        assert sub.read(i) == b.read(int(r))
    assert sub.n_reads == len(reads) and sub.n_loci == len(deal_blocks(b, 2, block=7)[1])


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from strkit_amd.sharding import count_loci_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    b = make_config(1, n_loci=30)
    full = count_loci_sharded(b, oracle_count, device=None, block=4)  # the oracle stands in for the GPU on CPU
    q.put((rank, {k: v.tolist() for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    """world_size-2 gloo: sharding + one all-gather reproduce the 1-process table bit for bit."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = oracle_count(make_config(1, n_loci=30))
    for rank in (0, 1):
        for k, v in exp.items():
            assert got[rank][k] == v.tolist()


def test_aligned_pair_matches_from_cigar():
    from strkit_amd.realign import cigar_to_string, get_aligned_pair_matches
    enc = {"M": 0, "I": 1, "D": 2, "=": 7, "X": 8}
    cig = np.array([(n << 4) | enc[o] for n, o in [(3, "D"), (2, "="), (1, "I"), (1, "X"), (2, "D"), (2, "=")]], np.uint32)
    assert cigar_to_string(cig) == "3D2=1I1X2D2="
    ac = get_aligned_pair_matches(cig, 100, 0, swap=True)      # "query" = ref window at 100, "ref" = read at 0
    assert ac.ref_coords.tolist() == [100, 101, 103, 104, 105]
    assert ac.query_coords.tolist() == [3, 4, 5, 8, 9]
    ac2 = get_aligned_pair_matches(cig, 100, 0)
    assert ac2.query_coords.tolist() == ac.ref_coords.tolist() and len(ac2) == 5 and ac2.pair_at_idx(2) == (103, 5)


def test_adjusted_score_and_read_filters_follow_the_callers_loop():
    """call_locus.py:1172,1222-1252 restated as a plain loop vs the vectorised host helper."""
    from strkit_amd.batch import calc_adj_score, filter_reads
    from strkit_amd.synth import make_config
    b = make_config(2, n_loci=40)
    rng = np.random.default_rng(4)
    total = (b.nfl + b.ntr + b.nfr).astype(np.int64)
    score = (2 * total).astype(np.int32)                       # perfect reads: adj 2.0 (docs/output_formats.md:102)
    assert np.allclose(calc_adj_score(score, b.nfl, b.ntr, b.nfr), 2.0)
    bad = rng.random(b.n_reads) < 0.12
    score[bad] = (total[bad] * rng.uniform(-1.0, 0.15, int(bad.sum()))).astype(np.int32)
    got = filter_reads(b, {"score": score})
    for l in range(b.n_loci):
        poor, ok = 0, True
        for r in range(int(b.read_off[l]), int(b.read_off[l + 1])):
            adj = score[r] / total[r]
            if not ok:
                assert not got["keep"][r]
                continue
            if adj < 0.1:
                if adj < 0.1:
                    poor += 1
                    if poor > 3:
                        ok = False
                assert not got["keep"][r]
                continue
            assert got["keep"][r] and abs(got["sc"][r] - adj) < 1e-12
        assert bool(got["locus_ok"][l]) == ok
    assert (~got["locus_ok"]).any() and got["locus_ok"].any()


def _fake_rows(mine, ref):
    """Stands in for the device path in the CPU test: rows of the real shape with made-up read records."""
    from strkit_amd.frontend.call import CallOptions, _locus_dict, _locus_row
    rows, errors = [], []
    for blk in mine:
        for l in blk:
            if l.t_idx % 7 == 3:
                rows.append(_locus_dict(l))                    # a skipped locus (no reference data)
                continue
            if l.t_idx % 11 == 5:
                errors.append({"locus_index": l.t_idx, "error": "boom"})
                continue
            s_adj, e_adj = l.left_coord - (l.t_idx % 3), l.right_coord + (l.t_idx % 2)
            rd = {"ref_cn": 5 + l.t_idx, "left_coord_adj": s_adj, "right_coord_adj": e_adj, "ref_seq": ref.fetch(l.contig, s_adj, e_adj),
                  "ref_left_flank_seq": ref.fetch(l.contig, s_adj - 5, s_adj)}
            n = l.t_idx % 4
            reads = {f"read_{l.t_idx}_{k}" + "x" * (70 if k == 2 else 0): {"s": "+-"[k % 2], "cn": l.t_idx + k, "w": 1.0 / n,
                                                                       "sc": None if k == 1 else 1.5 + 0.125 * k, "sl": 30 + k,
                                                                       **({"realn": True} if k == 3 else {})} for k in range(n)}
            rows.append(_locus_row(l, rd, reads, CallOptions()))
    return rows, sum(len(r.get("reads") or {}) for r in rows), {"count_s": 0.1, "errors": errors}


def _gloo_blocks():
    from strkit_amd.frontend.loci import Locus
    return [[Locus(10 * k + i + 1, f"l{10 * k + i}", "chr1", 200 + 100 * (10 * k + i), 200 + 100 * (10 * k + i) + 6 * (1 + (k * 7 + i) % 9), "CAG")
             for i in range(1 + k % 4)] for k in range(11)]


class _Ref:
    seq = "ACGTTGCA" * 2000

    def fetch(self, contig, a, b):
        return self.seq[a:b]


def _gloo_call_worker(rank, world, port, q):
    import torch.distributed as dist
    from strkit_amd.frontend.call import call_blocks_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ref = _Ref()

    def fake_call(mine):
        rows, n, tm = _fake_rows(mine, ref)
        return rows, n, {**tm, "count_s": 0.1 * (rank + 1)}

    merged, n, tm = call_blocks_sharded(_gloo_blocks(), fake_call, ref)
    q.put((rank, merged, n, tm))
    dist.barrier()
    dist.destroy_process_group()


def test_call_driver_shards_locus_blocks_over_two_ranks():
    """world_size-2 gloo: blocks are dealt to ranks, fixed-size per-locus / per-read records (names as a fixed-width field)
    are all-gathered, and every rank ends with the rows a single process builds, in catalog order."""
    import socket

    import torch.multiprocessing as mp
    from strkit_amd.frontend.call import deal_locus_blocks
    from strkit_amd.frontend.loci import Locus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_call_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_rows, want_n, want_tm = _fake_rows(_gloo_blocks(), _Ref())
    want_rows.sort(key=lambda r: r["locus_index"])
    assert any(len(nm) > 64 for r in want_rows for nm in (r.get("reads") or {}))       # a name longer than the default field
    for rank, merged, n, tm in got:
        assert merged == want_rows and n == want_n
        assert [e["locus_index"] for e in tm["errors"]] == [e["locus_index"] for e in want_tm["errors"]] != []
        assert abs(tm["count_s"] - 0.2) < 1e-9
    blocks = [[Locus(i + 1, "x", "chr1", 0, 100 * (i + 1), "CAG")] for i in range(7)]
    shares = deal_locus_blocks(blocks, 3)
    assert sorted(k for s_ in shares for k in s_) == list(range(7)) and shares == deal_locus_blocks(blocks, 3)
    assert shares[0][-1] == 6     # the heaviest block goes to the first rank


def _gloo_stage_worker(rank, world, port, q):
    """The staging / gather layout of `bench.py --strong`, on CPU tensors over gloo (the oracle stands in for the GPU)."""
    import torch
    import torch.distributed as dist
    from strkit_amd.sharding import NF, deal_blocks, gathered_step_table, select_loci, share_sizes, step_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    catalog = make_config(2, n_loci=26)
    shares = deal_blocks(catalog, world, block=3)
    mine, my_reads = select_loci(catalog, shares[rank])
    rows, G = max(share_sizes(catalog, shares)), 3
    stage = torch.full((G * NF, rows), -1, dtype=torch.int32)
    stage[0::NF, :len(my_reads)] = torch.from_numpy(my_reads.astype(np.int32))
    res = oracle_count(mine)
    for j in range(G):                                     # G steps of one round (the same share every step)
        o = step_rows(stage, j)
        for i, k in enumerate(("cn", "score", "n_iters", "start")):
            o[1 + i, :mine.n_reads] = torch.from_numpy(res[k] + j * (k == "n_iters"))     # steps differ: the layout must not mix them
    gathered = torch.zeros((world * G * NF, rows), dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, stage)
    q.put((rank, [gathered_step_table(gathered.numpy(), world, G, j, catalog.n_reads).tolist() for j in range(G)]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_strong_staging_layout_over_two_ranks():
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_stage_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = oracle_count(make_config(2, n_loci=26))
    for rank in (0, 1):
        for j in range(3):
            t = np.array(got[rank][j], np.int32)
            assert np.array_equal(t[0], exp["cn"]) and np.array_equal(t[1], exp["score"])
            assert np.array_equal(t[2], exp["n_iters"] + j) and np.array_equal(t[3], exp["start"])


def test_realign_i16_saturation_flags():
    """What parasail's fixed 16-bit kernel (realign.py:56) would have done: flags from lengths and 32-bit scores."""
    lib = _lib.load(build=False)
    s1 = np.array([0, 400, 400 + 16383, 400 + 16383 + 17000, 400 + 16383 + 17000 + 20000], np.int64)
    s2 = np.array([0, 15000, 40000, 70000, 70010], np.int64)      # the last read is 10 bases long
    score = np.array([790, 32000, 32766, 12], np.int32)
    out = np.full(4, -1, np.int32)
    assert lib.strk_realign_i16_flags(4, s1.ctypes.data, s2.ctypes.data, score.ctypes.data, out.ctypes.data) == 0
    assert out.tolist() == [0, _lib.STRK_I16_CELL_MAY_SATURATE,
                            _lib.STRK_I16_CELL_MAY_SATURATE | _lib.STRK_I16_SCORE_SATURATES, 0]
    assert lib.strk_realign_i16_flags(1, None, s2.ctypes.data, score.ctypes.data, out.ctypes.data) == -22


def test_contiguous_block_dealing_gives_every_rank_one_balanced_run():
    """The file path deals ONE run of consecutive catalog blocks to every rank (a rank then loads only its own byte range of the
    alignment file), balanced by the same cost estimate as the scatter of the counting path."""
    from strkit_amd.frontend.call import deal_locus_blocks
    from strkit_amd.frontend.loci import Locus
    rng = np.random.default_rng(9)
    blocks = []
    pos = 1000
    for b in range(57):
        blk = []
        for _ in range(int(rng.integers(1, 200))):
            n = int(rng.integers(6, 400))
            blk.append(Locus(len(blk), f"l{b}_{len(blk)}", "chr1", pos, pos + n, "CAG", 70))
            pos += n + 500
        blocks.append(blk)
    cost = [sum((l.right_coord - l.left_coord + 140) ** 2 for l in blk) for blk in blocks]
    for world in (1, 2, 3, 8):
        runs = deal_locus_blocks(blocks, world, contiguous=True)
        assert sorted(k for r in runs for k in r) == list(range(len(blocks)))
        for r in runs:
            assert r == list(range(r[0], r[-1] + 1)) if r else True          # one run of consecutive blocks
        assert [r[0] for r in runs if r] == sorted(r[0] for r in runs if r)  # in catalog order over the ranks
        share = sum(cost) / world
        assert max(sum(cost[k] for k in r) for r in runs) < share + max(cost), world
    # the scatter (counting path) stays what it was: every block exactly once
    assert sorted(k for r in deal_locus_blocks(blocks, 3) for k in r) == list(range(len(blocks)))


def test_class_known_band_geometry_equals_the_search_over_the_classes(tmp_path):
    """band_geometry_of_class (what the band kernels recompute per item from its class) must give exactly what band_geometry
    (k_plan: the search over the eight classes) gave for every eligible item: same band, same column range.  Host build of
    strk_search.h, 3 million random shapes incl. long windows, every window half-width and every band placement (BandTune)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "geo_check.cpp"
    src.write_text(r"""
#include <cstdio>
#include <random>
#include <algorithm>
#include "%s/strkit_amd/csrc/strk_search.h"
int main() {
    std::mt19937 rng(7);
    long n_ok = 0, bad = 0, fly_limits = 0;
    for (long it = 0; it < 3000000; ++it) {
        int nfl = 1 + rng() %% 260, nfr = 1 + rng() %% 127, m = 1 + rng() %% ((rng() %% 8 == 0) ? 200 : 24);
        int est = rng() %% ((rng() %% 4 == 0) ? 2100 : 60);
        int W = 3 + rng() %% 13;
        int lo = std::max(0, est - W), n = std::min(32, est + W - lo + 1);
        int ntr = std::max(0, est * m + (int)(rng() %% 41) - 20);
        strk::BandTune tune = {(int)(rng() %% 3 == 0 ? 64 : 3 + rng() %% 5), (int)(rng() %% 2 ? 0 : rng() %% 33)};
        strk::BandGeo a = strk::band_geometry(nfl, ntr, nfr, m, lo, n, tune);
        if (!a.ok) continue;
        ++n_ok;
        if (strk::band_class_fly(a.cls) && (nfl > strk::kBandFlyMaxFlank || m > strk::kBandFlyMaxMotif)) ++fly_limits;
        strk::BandGeo b = strk::band_geometry_of_class(a.cls, nfl, ntr, m, lo, n, tune);
        if (a.cls != b.cls || a.G != b.G || a.wd != b.wd || a.dlo != b.dlo || a.bwd != b.bwd || a.bdlo != b.bdlo ||
            a.cmin != b.cmin || a.ncol != b.ncol) ++bad;
    }
    printf("%%ld %%ld %%ld\n", n_ok, bad, fly_limits);
    return 0;
}
""" % root)
    exe = tmp_path / "geo_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(src)], check=True)
    n_ok, bad, fly_limits = (int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert n_ok > 1_000_000 and bad == 0
    assert fly_limits == 0      # the on-the-fly classes never get a flank or a motif their staged 2 x 256 bytes cannot hold


def test_search_replay_with_every_narrowing_schedule_equals_a_plain_python_search(tmp_path):
    """Host build of strk_search.h::search_replay (the function the kernels and the window-miss path run) against the plain
    Python loop above on 20 000 random score tables: every schedule of local_search_range (STRK_NARROW_*), both tie rules,
    steps 1-5, ranges 0-5, small max_iters, starts inside and outside the table (window misses), flat tables (ties)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "search_check.cpp"
    src.write_text(r"""
#include <cstdio>
#include <vector>
#include "%s/strkit_amd/csrc/strk_search.h"
struct Seen { std::vector<char> v; bool test(int k) const { return v[k]; } void set(int k) { v[k] = 1; } };
int main() {
    int start, step, lsr, max_iters, tie, narrow, lo, n;
    while (scanf("%%d %%d %%d %%d %%d %%d %%d %%d", &start, &step, &lsr, &max_iters, &tie, &narrow, &lo, &n) == 8) {
        std::vector<int32_t> sc(n);
        for (int k = 0; k < n; ++k) scanf("%%d", &sc[k]);
        Seen seen; seen.v.assign(n, 0);
        strk::SearchResult r = strk::search_replay(start, step, lsr, max_iters, tie, sc.data(), lo, n, seen, narrow);
        if (r.miss) printf("miss\n"); else if (r.empty) printf("empty\n"); else printf("%%d %%d %%d\n", r.cn, r.score, r.n_explored);
    }
    return 0;
}
""" % root)
    exe = tmp_path / "search_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(src)], check=True)
    from helpers import py_search
    rng = np.random.default_rng(20261009)
    lines, want = [], []
    for _ in range(20000):
        lo, n = int(rng.integers(0, 30)), int(rng.integers(1, 48))
        kind = int(rng.integers(4))
        peak = lo + int(rng.integers(-3, n + 3))
        if kind == 0:
            sc = rng.integers(-50, 400, size=n)
        elif kind == 1:                                        # one hill, noisy
            sc = 300 - 7 * np.abs(np.arange(lo, lo + n) - peak) + rng.integers(-6, 7, size=n)
        elif kind == 2:                                        # plateaus: ties decide
            sc = 300 - 10 * (np.abs(np.arange(lo, lo + n) - peak) // 3)
        else:
            sc = np.full(n, 17)
        start = lo + int(rng.integers(-4, n + 4))
        step, lsr = int(rng.integers(1, 6)), int(rng.integers(0, 6))
        max_iters, tie, narrow = int(rng.choice((1, 3, 7, 20, 50))), int(rng.integers(2)), int(rng.integers(4))
        lines.append(" ".join(str(int(x)) for x in (start, step, lsr, max_iters, tie, narrow, lo, n, *sc)))
        want.append(py_search(start, step, lsr, max_iters, tie, narrow, {lo + k: int(sc[k]) for k in range(n)}))
    out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", check=True, capture_output=True, text=True).stdout.split("\n")
    seen_kinds = set()
    for k, (line, w) in enumerate(zip(out, want)):
        g = line if line in ("miss", "empty") else tuple(int(x) for x in line.split())
        assert g == w, (k, lines[k][:80], g, w)
        seen_kinds.add(w if isinstance(w, str) else "found")
    assert seen_kinds == {"miss", "found"} or seen_kinds == {"miss", "found", "empty"}
