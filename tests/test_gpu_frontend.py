"""End to end on the device: synthetic BAM + FASTA + catalog -> call_sample -> per-read copy numbers."""
import json
import os

import numpy as np
import pytest

import oracle
from strkit_amd.frontend import (Fasta, call_sample, get_aligned_pairs, get_read_coords_from_matched_pairs,
                                 get_sequence_data_for_locus, load_loci, read_bam)
from strkit_amd.frontend.synth_dataset import make_dataset

pytestmark = pytest.mark.gpu


def test_error_free_reads_give_their_allele(gpu_ctx, tmp_path):
    t = make_dataset(str(tmp_path), n_loci=25, reads_per_locus=10, read_len=2500, seed=11)
    rep = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])
    assert len(rep["results"]) == 25
    bam, (block,) = read_bam(t["paths"]["bam"]), load_loci(t["paths"]["loci"])
    segs_of = {l.t_idx: bam.fetch(l.contig, l.left_flank_coord, l.right_flank_coord) for l in block}
    for res, truth in zip(rep["results"], t["loci"]):
        assert res["call"] is None and res["read_peaks_called"] is False and res["ref_cn"] == truth["ref_cn"] and res["motif"] == truth["motif"]
        assert set(res["reads"]) == set(truth["reads"])
        for name, rd in res["reads"].items():
            assert rd["cn"] == truth["reads"][name] and rd["sc"] == 2.0 and rd["s"] in "+-"
            assert rd["sl"] == truth["reads"][name] * len(truth["motif"])
        # read weights (call_locus.py:1254-1259): (L + t - 2) / (L - t + 1), L = mean length of the locus' reads, t = flanks + tract
        lens = np.array([s.length for s in segs_of[res["locus_index"]]], np.float64)
        for name, rd in res["reads"].items():
            t_ = rd["sl"] + 140
            assert abs(rd["w"] - (lens.mean() + t_ - 2) / (lens.mean() - t_ + 1)) < 1e-12


def test_noisy_reads_match_the_oracle_on_the_extracted_triples(gpu_ctx, tmp_path):
    t = make_dataset(str(tmp_path), n_loci=12, reads_per_locus=8, read_len=2000, seed=5, sub=0.01, indel=0.015, low_qual=0.01)
    rep = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], respect_ref=True)
    bam, (block,) = read_bam(t["paths"]["bam"]), load_loci(t["paths"]["loci"])
    n_reads = 0
    for locus, res in zip(block, rep["results"]):
        triples, names = [], []
        for s in bam.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord):
            q, r = get_aligned_pairs(s)
            c = get_read_coords_from_matched_pairs(locus.left_flank_coord, locus.left_coord, locus.right_coord,
                                                   locus.right_flank_coord, q, r)
            sd = get_sequence_data_for_locus(s, c, 70)
            triples.append((sd.flank_left_seq_wc[-70:], sd.tr_seq_wc, sd.flank_right_seq_wc[:70]))
            names.append(s.name)
        seqs = np.frombuffer("".join(a + b + c for a, b, c in triples).encode(), np.uint8)
        off = np.concatenate(([0], np.cumsum([len(a) + len(b) + len(c) for a, b, c in triples]))).astype(np.int64)
        arr = lambda k: np.array([len(x[k]) for x in triples], np.int32)  # noqa: E731
        est = np.array([round(len(x[1]) / len(locus.motif)) for x in triples], np.int32)
        exp = oracle.count_locus(seqs, off, arr(0), arr(1), arr(2), est, locus.motif)
        for i, name in enumerate(names):
            total = len(triples[i][0]) + len(triples[i][1]) + len(triples[i][2])
            if exp["score"][i] / total < 0.1:
                assert name not in res["reads"]
                continue
            assert res["reads"][name]["cn"] == exp["cn"][i]
            assert abs(res["reads"][name]["sc"] - exp["score"][i] / total) < 1e-12
            n_reads += 1
    assert n_reads >= 90


def test_soft_clipped_expansions_come_back_with_realign(gpu_ctx, tmp_path):
    t = make_dataset(str(tmp_path), n_loci=10, reads_per_locus=8, read_len=2500, seed=2, soft_clip_frac=0.7, expansion=40)
    plain = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])
    realn = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], realign=True)
    lost = recovered = 0
    for a, b, truth in zip(plain["results"], realn["results"], t["loci"]):
        lost += len(truth["reads"]) - len(a["reads"])
        for name, rd in b["reads"].items():
            assert rd["cn"] == truth["reads"][name]
            recovered += bool(rd.get("realn"))
        assert set(b["reads"]) == set(truth["reads"])
    assert lost > 10 and recovered >= lost
    # the device front end (the default) and the host one realign the same reads to the same answers
    assert realn["stage_times"]["front_end"] == "device"
    assert call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], realign=True, front_end="host")["results"] == realn["results"]


def test_cli_writes_json(gpu_ctx, tmp_path):
    from strkit_amd.__main__ import main
    t = make_dataset(str(tmp_path), n_loci=4, reads_per_locus=4, read_len=1500, seed=9)
    out = str(tmp_path / "out.json")
    assert main(["call", t["paths"]["bam"], "--ref", t["paths"]["ref"], "--loci", t["paths"]["loci"], "--json", out]) == 0
    rep = json.load(open(out))
    assert [r["ref_cn"] for r in rep["results"]] == [x["ref_cn"] for x in t["loci"]]
    # the per-locus entry point gives the same record as the block path
    from strkit_amd.frontend import Fasta, call_locus, load_loci, read_bam
    bam, ref = read_bam(t["paths"]["bam"]), Fasta(t["paths"]["ref"])
    (block,) = load_loci(t["paths"]["loci"])
    one = call_locus(block[2], bam, ref)
    assert json.loads(json.dumps(one)) == rep["results"][2]
    assert main(["call", t["paths"]["bam"], "--ref", t["paths"]["ref"], "--loci", t["paths"]["loci"], "--json", out,
                 "--processes", "2", "--max-rcn-iters", "30", "--min-read-align-score", "0.2", "--sample-id", "s1", "--seed", "7"]) == 0
    assert json.load(open(out))["sample_id"] == "s1"


def test_compare_tool_finds_the_generating_switches(gpu_ctx, tmp_path):
    """tools/compare_strkit_json.py on a report that plays STRkit's: a report made with tie_rule = last-maximum, a
    one-sided end-gap mode and a halving search range is reproduced by that combination of the sweep (16 end-gap modes x 2 tie
    rules x 4 schedules), and the plain diff against the default switches sees the differing reads."""
    import importlib.util
    import os
    from strkit_amd.frontend.compare import diff_reports
    t = make_dataset(str(tmp_path), n_loci=14, reads_per_locus=8, read_len=1500, seed=21, sub=0.02, indel=0.03, motif_len=(1, 3))
    theirs = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], tie_rule=1, end_flags=5, narrowing=2)
    theirs["parameters"]["rc_method"] = "repalign"
    path = str(tmp_path / "strkit.json")
    json.dump(theirs, open(path, "w"))
    default = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])
    d = diff_reports(theirs, default)
    assert d["loci_common"] == 14 and d["locus_fields_equal"] == d["locus_fields_compared"]   # the reference side has no switch
    assert not d["identical"] and d["cn_equal"] + d["sc_equal"] < d["reads_common"] + d["sc_compared"]
    spec = importlib.util.spec_from_file_location("cmp_tool", os.path.join(os.path.dirname(__file__), "..", "tools", "compare_strkit_json.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = str(tmp_path / "sweep.json")
    rc = tool.main([path, "--bam", t["paths"]["bam"], "--ref", t["paths"]["ref"], "--loci", t["paths"]["loci"], "--sweep", "--json", out])
    rows = json.load(open(out))
    assert rc == 0 and len(rows) == 128 and rows[0]["identical"]
    assert (5, 1, 2) in {(r["end_flags"], r["tie_rule"], r["narrowing"]) for r in rows if r["identical"]}
    assert tool.main([path, "--ours", path]) == 0
    ours_path = str(tmp_path / "ours.json")
    json.dump(default, open(ours_path, "w"))
    assert tool.main([path, "--ours", ours_path]) == 1


def test_large_expansion_reads_are_called(gpu_ctx, tmp_path):
    """One sparse locus whose reads carry a 1 000-copy insertion (a hundred times the reference window): round 1's
    extraction buffer bound made the library fail and the whole sample abort."""
    from test_frontend import _expansion_bam
    from strkit_amd.frontend.fasta import write_fasta
    path, ref = _expansion_bam(tmp_path)
    write_fasta(str(tmp_path / "ref.fa"), {"chr1": ref})
    (tmp_path / "loci.bed").write_text("chr1\t600\t630\tCAG\n")
    rep = call_sample(path, str(tmp_path / "ref.fa"), str(tmp_path / "loci.bed"))
    (row,) = rep["results"]
    assert rep["errors"] == [] and row["ref_cn"] == 10 and len(row["reads"]) == 30
    assert sorted({rd["cn"] for rd in row["reads"].values()}) == [10, 1010]
    assert all(rd["sc"] == 2.0 for rd in row["reads"].values())


def test_chr_prefix_is_normalised_between_catalog_and_files(gpu_ctx, tmp_path):
    """call_locus.py:758 normalize_contig: a catalog written with "chr1" against files that say "1" (and the reverse)."""
    t = make_dataset(str(tmp_path), n_loci=5, reads_per_locus=4, read_len=1200, seed=6)
    base = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])
    assert len(base["results"]) == 5
    rows = [r for r in open(t["paths"]["loci"]).read().splitlines() if r and not r.startswith("#")]
    contig = rows[0].split("\t")[0]
    other = contig[3:] if contig.startswith("chr") else "chr" + contig
    alt = str(tmp_path / "alt.bed")
    open(alt, "w").write("\n".join(other + r[len(contig):] for r in rows) + "\nchrNope\t10\t40\tCAG\n")
    rep = call_sample(t["paths"]["bam"], t["paths"]["ref"], alt)
    assert rep["catalog"] == {"num_loci": 5, "num_loci_unknown_contig": 1}
    for a, b in zip(base["results"], rep["results"]):
        assert {**a, "contig": None} == {**b, "contig": None} and b["contig"] == other


def test_packed_reference_side_equals_the_per_locus_path(gpu_ctx, tmp_path):
    """get_loci_with_ref_data over a block (windows gathered as packed arrays, one library call) against one locus at a
    time through Fasta.fetch strings — soft-masked lower case, a locus too close to the contig start / end (skipped:
    "reference flank size too small"), a tract flanked by N (skipped), a tract that continues into its flank."""
    from strkit_amd.frontend import Fasta
    from strkit_amd.frontend.call import get_loci_with_ref_data
    from strkit_amd.frontend.fasta import write_fasta
    from strkit_amd.frontend.loci import Locus
    rng = np.random.default_rng(8)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(4, size=n))  # noqa: E731
    pieces, loci, pos = [], [], 0
    specs = [("CAG", 12, "plain"), ("AT", 30, "lower"), ("GGC", 9, "extend"), ("TTTA", 15, "nflank"), ("A", 25, "plain"),
             ("CCG", 20, "lower"), ("AAG", 14, "plain")]
    for k, (m, cn, kind) in enumerate(specs):
        gap = rnd(30 if k == 0 else 400)                       # the first locus sits too close to the contig start
        tract = m * cn
        if kind == "lower":
            gap, tract = gap[:-80] + gap[-80:].lower(), tract.lower()
        if kind == "extend":
            gap = gap[:-6] + m * 2                                # two more copies hide in the left flank
        if kind == "nflank":
            gap = gap[:-4] + "NNNN"
        pieces += [gap, tract]
        pos += len(gap)
        loci.append(Locus(k + 1, f"l{k}", "chr1", pos, pos + len(tract), m, 70))
        pos += len(tract)
    pieces.append(rnd(40))                                     # ... and the last one too close to its end
    write_fasta(str(tmp_path / "r.fa"), {"chr1": "".join(pieces), "chr2": rnd(500)})
    ref = Fasta(str(tmp_path / "r.fa"))
    loci.append(Locus(9, "other", "chr2", 200, 230, "AC", 70))
    loci.append(Locus(10, "nowhere", "chrX", 200, 230, "AC", 70))
    block = get_loci_with_ref_data(loci, ref, False, gpu_ctx)
    single = [get_loci_with_ref_data([l], ref, False, gpu_ctx)[0] for l in loci]
    assert block == single
    assert [r is None for r in block] == [True, False, False, True, False, False, True, False, True]
    assert block[2]["left_coord_adj"] == loci[2].left_coord - 6 and block[1]["ref_seq"] == "at" * 30
    assert get_loci_with_ref_data(loci, ref, True, gpu_ctx) == [get_loci_with_ref_data([l], ref, True, gpu_ctx)[0] for l in loci]


def test_indexed_and_whole_file_access_give_the_same_report(gpu_ctx, tmp_path):
    from strkit_amd.frontend import IndexedBam, NativeBam
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=450, depth=9, read_len=3000, seed=9, spacing=9000, procs=4)
    a = call_sample(NativeBam(t["paths"]["bam"]), t["paths"]["ref"], t["paths"]["loci"])
    b = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], front_end="host")   # a path with a .bai: IndexedBam
    assert "load_s" in b["stage_times"] and "load_s" not in a["stage_times"]
    assert a["results"] == b["results"] and len(a["results"]) == 450
    truth = {(int(x), int(y)): int(z) for x, y, z in t["truth"]}
    n = hit = 0
    # (a random flank may continue the repeat by a copy or two: the boundary extension then counts them)
    assert sum(row["ref_cn"] == int(t["ref_cn"][row["locus_index"] - 1]) for row in b["results"]) > 0.9 * 450
    for row in b["results"]:
        for name, rd in row["reads"].items():
            l_, r_ = name[1:].split("_r")
            n += 1
            hit += rd["cn"] == truth[(int(l_), int(r_))]
    assert n == 450 * 9 and hit > 0.95 * n


def test_report_rows_and_vcf_match_the_golden_fixture(gpu_ctx, tmp_path):
    """tests/golden/report_rows.json / report.vcf (tests/golden/make_report_golden.py): the rows `python -m strkit_amd call`
    writes - read records with the keys of call_locus.py:1279-1288, locus fields of :1040-1047,1340-1352 - and their VCF."""
    import os
    from strkit_amd.frontend.output import allele_calling_inputs, write_vcf
    g = os.path.join(os.path.dirname(__file__), "golden")
    gold = json.load(open(os.path.join(g, "report_rows.json")))
    t = make_dataset(str(tmp_path), **gold["params"])
    rep = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], realign=True, sample_id="golden")
    assert json.loads(json.dumps(rep["results"])) == gold["results"]
    write_vcf(rep, str(tmp_path / "o.vcf"), Fasta(t["paths"]["ref"]), date="20261004")
    assert open(tmp_path / "o.vcf").read() == open(os.path.join(g, "report.vcf")).read()
    row = rep["results"][0]
    assert set(next(iter(row["reads"].values()))) <= {"s", "cn", "w", "sc", "sl", "realn"}
    cns, w = allele_calling_inputs(row)             # what allele.call_alleles receives (call_locus.py:188-204)
    assert cns.dtype == np.int32 and len(cns) == len(row["reads"]) and abs(w.sum() - 1.0) < 1e-12


def test_device_inflater_equals_the_host_one(gpu_ctx, tmp_path):
    """strk_dbam_inflate (one GPU lane per BGZF block, CRC checked on the device) against strk_bgzf_inflate: a whole file, a
    stretch from a block boundary with a size limit, a corrupted payload."""
    import ctypes as C
    from strkit_amd import _lib
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=300, depth=10, read_len=4000, seed=13, spacing=9000, procs=4)
    L = _lib.load()
    comp = np.fromfile(t["paths"]["bam"], np.uint8)
    n = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, None, 0, 0)
    want = np.empty(int(n), np.uint8)
    assert L.strk_bgzf_inflate(comp.ctypes.data, comp.size, want.ctypes.data, want.size, 0) == n
    h = C.c_void_p()
    _lib.check(L.strk_dbam_open(0, C.byref(h)))
    try:
        nxt = C.c_int64(0)
        assert L.strk_dbam_inflate(h, comp.ctypes.data, comp.size, 0, 1 << 40, C.byref(nxt)) == n and nxt.value == comp.size
        got = np.empty(int(n), np.uint8)
        _lib.check(L.strk_dbam_download(h, 0, int(n), got.ctypes.data))
        assert np.array_equal(got, want)
        # a stretch: from the third block on, at most 200 000 bytes -> whole blocks only, the next offset is a block start
        out3 = np.empty(1 << 20, np.uint8)
        nx = C.c_int64(0)
        first = L.strk_bgzf_inflate_range(comp.ctypes.data, comp.size, 0, out3.ctypes.data, 140000, C.byref(nx), 1)
        coff = nx.value
        m = L.strk_dbam_inflate(h, comp.ctypes.data, comp.size, coff, 200000, C.byref(nxt))
        assert 0 < m <= 200000 and coff < nxt.value < comp.size
        part = np.empty(int(m), np.uint8)
        _lib.check(L.strk_dbam_download(h, 0, int(m), part.ctypes.data))
        assert np.array_equal(part, want[first:first + m])
        bad = comp.copy()
        bad[bad.size // 3] ^= 0x44
        assert L.strk_dbam_inflate(h, bad.ctypes.data, bad.size, 0, 1 << 40, C.byref(nxt)) < 0 and b"BGZF" in L.strk_last_error()
    finally:
        L.strk_dbam_close(h)


def test_file_upload_equals_the_buffer_upload(gpu_ctx, tmp_path):
    """strk_dbam_inflate_file (the library reads the file: ring of pinned pieces, copies as they come in, headers walked
    meanwhile) against the bytes that went in: a file of several pieces whose blocks straddle the piece ends, block lengths
    that do not divide the piece size, an empty file, a truncated one, a missing one."""
    import ctypes as C
    import os
    from strkit_amd import _lib
    from strkit_amd.frontend.bam import _bgzf_blocks
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 24, size=44_000_000, dtype=np.uint8)         # ~1.7x compressible: ~26 MB file, four pieces
    raw[1_000_000:3_000_000] = 7                                         # a stretch of tiny blocks' worth of repeats
    comp = _bgzf_blocks(raw.tobytes())
    assert len(comp) > 3 * (8 << 20)
    path = str(tmp_path / "big.bgzf")
    with open(path, "wb") as fh:
        fh.write(comp)
    L = _lib.load()
    h = C.c_void_p()
    _lib.check(L.strk_dbam_open(0, C.byref(h)))
    try:
        for threads in (0, 1, 3):
            nc = C.c_int64(0)
            n = L.strk_dbam_inflate_file(h, os.fsencode(path), threads, C.byref(nc))
            assert n == raw.size and nc.value == len(comp), (n, L.strk_last_error())
            got = np.empty(raw.size, np.uint8)
            _lib.check(L.strk_dbam_download(h, 0, raw.size, got.ctypes.data))
            assert np.array_equal(got, raw)
        # virtual offsets resolve as after the buffer upload
        arr = np.frombuffer(comp, np.uint8)
        nxt = C.c_int64(0)
        voff = np.array([0, 5, (len(comp) - 28) << 16], np.uint64)
        a, b = np.empty(3, np.int64), np.empty(3, np.int64)
        _lib.check(L.strk_dbam_voffsets(h, voff.ctypes.data, 3, a.ctypes.data))
        assert L.strk_dbam_inflate(h, arr.ctypes.data, arr.size, 0, 1 << 40, C.byref(nxt)) == raw.size
        _lib.check(L.strk_dbam_voffsets(h, voff.ctypes.data, 3, b.ctypes.data))
        assert np.array_equal(a, b) and a[0] == 0 and a[1] == 5 and a[2] == raw.size
        empty = str(tmp_path / "empty.bgzf")
        open(empty, "wb").close()
        assert L.strk_dbam_inflate_file(h, os.fsencode(empty), 0, None) == 0
        cut = str(tmp_path / "cut.bgzf")
        with open(cut, "wb") as fh:
            fh.write(comp[:(8 << 20) + 1000])
        assert L.strk_dbam_inflate_file(h, os.fsencode(cut), 0, None) < 0 and b"truncated" in L.strk_last_error()
        assert L.strk_dbam_inflate_file(h, os.fsencode(str(tmp_path / "none.bgzf")), 0, None) < 0
        # and the object still works afterwards
        assert L.strk_dbam_inflate_file(h, os.fsencode(path), 2, None) == raw.size
    finally:
        L.strk_dbam_close(h)


def test_device_front_end_gives_the_host_report(gpu_ctx, tmp_path):
    """DeviceBam (inflate, record scan, read extraction and names on the GPU; the bases never leave it) against the host
    readers: the same records, the same extracted triples, the same report; noisy reads with low-quality bases and
    quality-gated reads included."""
    from strkit_amd import _lib
    from strkit_amd.frontend import DeviceBam, IndexedBam, NativeBam, extract_reads
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=260, depth=9, read_len=3000, seed=19, spacing=9000, procs=4)
    nb, db = NativeBam(t["paths"]["bam"]), DeviceBam(t["paths"]["bam"])
    try:
        assert db.n_records == nb.n_records and db.references == nb.references
        for k in ("rec_off", "tid", "pos", "end", "flag", "l_seq", "clip_l", "clip_r"):
            assert np.array_equal(getattr(db, k), getattr(nb, k)), k
        some = np.arange(0, nb.n_records, 37)
        assert db.names(some) == nb.names(some)
        a, b = db.segment(int(some[3])), nb.segment(int(some[3]))
        assert (a.name, a.start, a.end, a.query_sequence) == (b.name, b.start, b.end, b.query_sequence) and np.array_equal(a.cigar, b.cigar)
        # extraction: every locus of the first blocks, all overlapping records (spanning and not)
        blocks = load_loci(t["paths"]["loci"], max_block_size=40)
        for blk in blocks[:3]:
            lfc = np.array([l.left_flank_coord for l in blk]); rfc = np.array([l.right_flank_coord for l in blk])
            rec, n_per = nb.fetch_many("chr1", lfc, rfc, 250)
            rec_d, n_per_d = db.fetch_many("chr1", lfc, rfc, 250)
            assert np.array_equal(rec, rec_d) and np.array_equal(n_per, n_per_d)
            owner = np.repeat(np.arange(len(blk)), n_per)
            coords = np.stack((lfc, np.array([l.left_coord for l in blk]), np.array([l.right_coord for l in blk]), rfc), axis=1)[owner]
            for phred in (13, 39):
                eh = extract_reads(nb, rec, coords, 70, phred)
                ed = extract_reads(db, rec, coords, 70, phred)
                for k in ("status", "nfl", "ntr", "nfr", "seq_off"):
                    assert np.array_equal(eh[k], ed[k]), (k, phred)
                got = np.empty(max(int(ed["seq_off"][-1]), 1), np.uint8)
                _lib.check(_lib.load().strk_dbam_download_seqs(db._h, int(ed["seq_off"][-1]), got.ctypes.data))
                assert np.array_equal(got[:int(ed["seq_off"][-1])], eh["seqs"]) and ed["d_seqs"]
                assert (eh["status"] == 0).sum() > 0
        host = call_sample(IndexedBam(t["paths"]["bam"]), t["paths"]["ref"], t["paths"]["loci"])
        dev = call_sample(db, t["paths"]["ref"], t["paths"]["loci"])
        auto = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])               # auto: the device front end
        assert host["results"] == dev["results"] == auto["results"] and len(dev["results"]) == 260
        assert "load_s" not in auto["stage_times"]
        with pytest.raises(ValueError):
            call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], front_end="gpu")
    finally:
        db.close()


def test_streamed_device_front_end_gives_the_resident_report(gpu_ctx, tmp_path):
    """DeviceBam in streamed mode (a file larger than device memory: spans of the file, chosen through the .bai, go through HBM
    one after the other) against the same file held whole: the same report from many small spans and from one span; a region
    outside the resident span loads its own."""
    from strkit_amd.frontend import DeviceBam, Fasta, call_sample, load_loci
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=700, depth=8, read_len=8000, seed=23, spacing=9000, procs=4)
    p = t["paths"]
    size = os.path.getsize(p["bam"])
    assert size > (6 << 20)
    whole = call_sample(p["bam"], p["ref"], p["loci"], front_end="device")
    assert whole["stage_times"]["front_end"] == "device" and len(whole["results"]) == 700
    ref = Fasta(p["ref"])
    for span in (1 << 20, 3 << 20, 1 << 40):
        db = DeviceBam(p["bam"], span_bytes=span)
        try:
            assert db.streamed and db.n_records == 0
            blocks = load_loci(p["loci"], contigs=db.references)
            plan = db.plan(blocks)
            assert sum(len(g[3]) for g in plan) == len(blocks) and (len(plan) > 3 if span < size else len(plan) == 1)
            rep = call_sample(db, ref, p["loci"])
            assert db.open_stage_s["spans"] == len(plan)
        finally:
            db.close()
        assert rep["results"] == whole["results"], span
    # on demand: a region outside the resident span loads its own
    db = DeviceBam(p["bam"], span_bytes=1 << 20)
    try:
        blocks = load_loci(p["loci"], contigs=db.references)
        last = blocks[-1]
        lo, hi = last[0].left_flank_coord, last[-1].right_flank_coord + 1
        r = db.region(last[0].contig, lo, hi)
        n1 = len(r.fetch_indices(last[0].contig, lo, hi))
        first = blocks[0]
        r2 = db.region(first[0].contig, first[0].left_flank_coord, first[0].right_flank_coord + 1)
        assert n1 > 0 and len(r2.fetch_indices(first[0].contig, first[0].left_flank_coord, first[0].right_flank_coord + 1)) > 0
        assert db.open_stage_s["spans"] == 2
    finally:
        db.close()


def _two_rank_call_worker(rank, world, port, paths, q):
    os.environ["STRKIT_AMD_DEVICE"] = "0"            # both ranks share the one GPU of the test box
    import torch.distributed as dist
    from strkit_amd.frontend import call_sample
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rep = call_sample(paths["bam"], paths["ref"], paths["loci"])
    q.put((rank, rep["results"], rep["stage_times"].get("front_end"), rep["catalog"], rep["stage_times"].get("front_end_compressed_mb")))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_from_files_give_the_single_process_report(gpu_ctx, tmp_path):
    """`call_sample` under torch.distributed (world size 2, gloo, both ranks on this box's GPU): every rank opens the file with
    the device front end in spans, calls its own run of consecutive locus blocks — loading less than 60 % of the file for it —
    and all ranks end with the report of a single process: the fixed-size record gather of call_blocks_sharded on real rows."""
    import socket

    import torch.multiprocessing as mp
    from strkit_amd.frontend.synth_large import make_dataset_large
    # (32 Mb of reference: the megabase of margin a span keeps behind its last locus is small against a rank's half)
    t = make_dataset_large(str(tmp_path), n_loci=1600, depth=5, read_len=2000, seed=31, spacing=20000, procs=4)
    want = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"])
    assert want["stage_times"]["front_end"] == "device" and len(want["results"]) == 1600
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_call_worker, args=(r, 2, port, t["paths"], q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    file_mb = os.path.getsize(t["paths"]["bam"]) / 1e6
    for rank, rows, fe, cat, comp_mb in got:
        assert fe == "device" and cat == want["catalog"]
        assert rows == want["results"], rank
        # one run of consecutive catalog blocks per rank: a rank reads, uploads and inflates its own byte range of the file only
        assert comp_mb is not None and 0 < comp_mb < 0.6 * file_mb, (rank, comp_mb, file_mb)
