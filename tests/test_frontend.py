"""CPU tests of the minimal front end (catalog loader, BAM/FASTA, read-coordinate extraction).  The parameter tables
of the first four tests are the reference's own vectors (tests/test_caller_loci_fns.py:8-35,
tests/test_caller_utils.py:1-11, tests/data/test_loci.bed) used as data."""
import os

import numpy as np
import pytest

from strkit_amd.frontend import (Fasta, Locus, LocusValidationError, find_pair_by_ref_pos, get_aligned_pairs,
                                 get_read_coords_from_matched_pairs, get_sequence_data_for_locus, load_loci,
                                 parse_last_column, read_bam, valid_motif, validate_locus, write_bam)
from strkit_amd.frontend.extract import LowMeanBaseQual, get_read_coords_from_cigar
from strkit_amd.frontend.synth_dataset import make_dataset

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("motif,valid", [("CAG", True), ("CAGN", True), ("CAGX", False), ("(CAG)n", False), ("XX", False)])
def test_valid_motif(motif, valid):
    assert valid_motif(motif) == valid


def test_validate_locus():
    with pytest.raises(LocusValidationError):
        validate_locus(Locus(1, "locus1", "1", 1000, 1000, "CAG", 70))      # start == end
    with pytest.raises(LocusValidationError):
        validate_locus(Locus(1, "locus1", "1", 1000, 1200, "(CAG)n", 70))   # invalid motif
    validate_locus(Locus(1, "locus1", "1", 1000, 1200, "CAG", 70))


@pytest.mark.parametrize("t_idx,val,parsed", [
    (0, "CAG", {"id": "locus0", "motif": "CAG"}),
    (0, "MOTIF=CAG", {"id": "locus0", "motif": "CAG"}),
    (0, "Motif = CAG", {"id": "locus0", "motif": "CAG"}),
    (0, "ID=HTT; MOTIF = CAG", {"id": "HTT", "motif": "CAG"}),
    (0, "ID = HTT ;motif = CAG", {"id": "HTT", "motif": "CAG"}),
    (0, "id=HTT ; motif=cag", {"id": "HTT", "motif": "CAG"}),
])
def test_parse_last_column(t_idx, val, parsed):
    assert parse_last_column(t_idx, val) == parsed


def test_find_pair_by_ref_pos_vectors():
    pairs_r = [1000, 1001, 1003, 1004, 1005, 1006, 1008, 1009]
    assert find_pair_by_ref_pos(pairs_r, 1004) == (3, True)
    assert find_pair_by_ref_pos(pairs_r, 1007) == (6, False)


@pytest.mark.parametrize("val", ["ID=HTT", "MOTIF=", "FOO=1;MOTIF=CAG", "ID=a=b;MOTIF=CAG", "MOTIF=CAG;"])
def test_parse_last_column_rejects(val):
    with pytest.raises(LocusValidationError):
        parse_last_column(3, val)


def test_reference_catalog_file_and_blocks(tmp_path):
    blocks = load_loci(os.path.join(HERE, "golden", "ref_test_loci.bed"))
    flat = [l for b in blocks for l in b]
    assert [(l.contig, l.left_coord, l.right_coord, l.motif) for l in flat] == [
        ("chr1", 200, 300, "ACAA"), ("chr1", 300, 400, "GA"), ("chr1", 350, 450, "GAGA"), ("chr2", 100, 200, "CAG")]
    assert [l.contig for l in blocks[-1]] == ["chr2"] and all(len({l.contig for l in b}) == 1 for b in blocks)
    assert flat[0].left_flank_coord == 130 and flat[0].right_flank_coord == 370 and flat[3].locus_id == "locus4"
    # block rules: at most 200 loci, split at gaps of more than 20 000 bases, unknown contigs dropped
    p = tmp_path / "many.bed"
    rows = [f"chr1\t{1000 + 300 * i}\t{1100 + 300 * i}\tCAG" for i in range(450)]
    rows += ["chr1\t900000\t900100\tID=far;MOTIF=ac", "chrUn\t5\t50\tCAG", "# comment", ""]
    p.write_text("\n".join(rows) + "\n")
    blocks = load_loci(str(p), contigs={"chr1"})
    assert [len(b) for b in blocks] == [200, 200, 50, 1] and blocks[-1][0].motif == "AC" and blocks[-1][0].locus_id == "far"
    assert [len(b) for b in load_loci(str(p), contigs={"chr1"}, processes=8)] == [56] * 8 + [2, 1]


def test_bam_round_trip_and_fetch(tmp_path):
    recs = [dict(name="r1", flag=0, contig="chr1", pos=100, cigar=[(5, "S"), (10, "M"), (2, "I"), (8, "M"), (3, "D"), (5, "=")],
                 seq="ACGTN" * 6, qual=np.arange(30)),
            dict(name="r2", flag=16, contig="chr1", pos=400, cigar=[(7, "M")], seq="ACGTACG", qual=None),
            dict(name="r3", flag=0, contig="chr2", pos=0, cigar=[(70000, "M")], seq="A" * 70000, qual=np.full(70000, 30))]
    path = str(tmp_path / "t.bam")
    write_bam(path, [("chr1", 1000), ("chr2", 80000)], recs)
    b = read_bam(path)
    assert b.references == ["chr1", "chr2"] and [s.name for s in b.segments] == ["r1", "r2", "r3"]
    s = b.segments[0]
    assert (s.start, s.end, s.length, s.is_reverse, s.soft_clips()) == (100, 126, 30, False, (5, 0))
    assert s.query_sequence == "ACGTN" * 6 and s.query_qualities.tolist() == list(range(30))
    assert b.segments[1].query_qualities is None and b.segments[1].is_reverse and b.segments[2].length == 70000
    assert [x.name for x in b.fetch("chr1", 120, 130)] == ["r1"] and b.fetch("chr1", 126, 400) == []
    assert [x.name for x in b.fetch("chr1", 126, 401)] == ["r2"] and b.fetch("chrX", 0, 10) == []
    q, r = get_aligned_pairs(s)
    assert q.tolist() == list(range(5, 15)) + list(range(17, 30)) and r.tolist() == list(range(100, 118)) + list(range(121, 126))


def test_read_coordinates_put_boundary_insertions_into_the_tract():
    # ref: flank [0,10) tract [10,16) flank [16,26); read has 6 extra bases inserted right after the tract's last base
    class Seg:
        start = 0
        cigar = np.array([(16 << 4) | 0, (6 << 4) | 1, (11 << 4) | 0], np.uint32)
        query_sequence = "A" * 10 + "CAGCAG" + "CAGCAG" + "T" * 11
        query_qualities = np.full(33, 30)
    q, r = get_aligned_pairs(Seg)
    c = get_read_coords_from_matched_pairs(0, 10, 16, 26, q, r)
    assert (c.left_flank_start, c.left_flank_end, c.right_flank_start, c.right_flank_end) == (0, 10, 22, 32)
    sd = get_sequence_data_for_locus(Seg, c, 10)
    assert (sd.flank_left_seq_wc, sd.tr_seq_wc, sd.flank_right_seq_wc, sd.tr_len_with_flank) == ("A" * 10, "CAG" * 4, "T" * 10, 32)
    assert sd.get_est_copy_num(3) == 4
    # ... and right before its first base; a read that stops short of a flank end is incomplete
    Seg.cigar = np.array([(10 << 4) | 0, (6 << 4) | 1, (17 << 4) | 0], np.uint32)
    q, r = get_aligned_pairs(Seg)
    c = get_read_coords_from_matched_pairs(0, 10, 16, 26, q, r)
    assert (c.left_flank_end, c.right_flank_start) == (10, 22)
    assert get_read_coords_from_matched_pairs(0, 10, 16, 40, q, r).is_incomplete()
    # call_locus.py:907-909: a read is skipped when right_flank_coord >= segment.end - here the alignment ends at
    # reference position 27 (exclusive), so right_flank_coord 26 is the last one that still counts as spanned
    assert not get_read_coords_from_matched_pairs(0, 10, 16, 26, q, r).is_incomplete()
    assert get_read_coords_from_matched_pairs(0, 10, 16, 27, q, r).is_incomplete()
    assert get_read_coords_from_matched_pairs(1, 10, 16, 26, q, r).full_left_flank
    assert not get_read_coords_from_matched_pairs(-1, 10, 16, 26, q, r).full_left_flank
    # low-quality tract bases: wildcards at <= 3, LowMeanBaseQual under the mean threshold
    Seg.query_qualities = np.full(33, 30)
    Seg.query_qualities[12:14] = 2
    assert get_sequence_data_for_locus(Seg, c, 10).tr_seq_wc == "CAXXAGCAGCAG"
    Seg.query_qualities[10:22] = 5
    with pytest.raises(LowMeanBaseQual):
        get_sequence_data_for_locus(Seg, c, 10)


def test_right_flank_boundary_agrees_in_all_four_implementations():
    """call_locus.py:907-909 skips a read when right_flank_coord >= segment.end: with segment.end == right_flank_coord the
    read is dropped, with segment.end == right_flank_coord + 1 it is kept — in the pair-list walk, the run index
    (extract.py), the host C++ run index (strk_frontend.h) and the one-pass walk the device runs (strk_bamrec.h)."""
    import ctypes as C
    from strkit_amd import _lib
    L = _lib.load()

    class Seg:
        start = 100
        cigar = np.array([(30 << 4) | 0, (3 << 4) | 1, (20 << 4) | 0], np.uint32)      # 30M 3I 20M: reference [100, 150)
        query_sequence = "A" * 53
        query_qualities = np.full(53, 30)
    seg_end = 150
    q, r = get_aligned_pairs(Seg)
    for rfc, spans in ((seg_end - 1, True), (seg_end, False), (seg_end + 1, False)):
        args = (100, 110, 130, rfc)
        a = get_read_coords_from_matched_pairs(*args, q, r)
        b = get_read_coords_from_cigar(*args, Seg)
        assert a.full_right_flank == b.full_right_flank == spans, (rfc, a, b)
        assert a.is_incomplete() == b.is_incomplete() == (not spans)
        assert (a.left_flank_start, a.left_flank_end, a.right_flank_start, a.right_flank_end) == \
               (b.left_flank_start, b.left_flank_end, b.right_flank_start, b.right_flank_end)
        c = np.array(args, np.int64)
        o1, o2 = np.zeros(4, np.int64), np.zeros(4, np.int64)
        rc = L.strk_read_coords_both(Seg.cigar.ctypes.data, len(Seg.cigar), Seg.start, c.ctypes.data, o1.ctypes.data, o2.ctypes.data)
        assert rc == (3 if spans else 0), (rfc, rc)
        if spans:
            assert o1.tolist() == o2.tolist() == [a.left_flank_start, a.left_flank_end, a.right_flank_start, a.right_flank_end]


def test_synthetic_dataset_extraction_recovers_the_alleles(tmp_path):
    t = make_dataset(str(tmp_path), n_loci=6, reads_per_locus=6, read_len=1200, seed=3)
    bam, ref = read_bam(t["paths"]["bam"]), Fasta(t["paths"]["ref"])
    (block,) = load_loci(t["paths"]["loci"])
    for locus, truth in zip(block, t["loci"]):
        assert (locus.left_coord, locus.right_coord, locus.motif) == (truth["start"], truth["end"], truth["motif"])
        assert ref.fetch(locus.contig, locus.left_coord, locus.right_coord) == truth["motif"] * truth["ref_cn"]
        segs = bam.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        assert len(segs) == 6
        for s in segs:
            q, r = get_aligned_pairs(s)
            c = get_read_coords_from_matched_pairs(locus.left_flank_coord, locus.left_coord, locus.right_coord,
                                                   locus.right_flank_coord, q, r)
            sd = get_sequence_data_for_locus(s, c, 70)
            assert sd.tr_seq == truth["motif"] * truth["reads"][s.name]
            assert sd.flank_left_seq_wc[-70:] == ref.fetch(locus.contig, locus.left_flank_coord, locus.left_coord)
            assert sd.flank_right_seq_wc[:70] == ref.fetch(locus.contig, locus.right_coord, locus.right_flank_coord)


def test_cigar_run_lookup_equals_expanded_pairs(tmp_path):
    from strkit_amd.frontend import get_read_coords_from_cigar
    t = make_dataset(str(tmp_path), n_loci=8, reads_per_locus=8, read_len=1500, seed=21, sub=0.02, indel=0.03,
                     soft_clip_frac=0.4, expansion=12)
    bam = read_bam(t["paths"]["bam"])
    (block,) = load_loci(t["paths"]["loci"])
    n = 0
    for locus in block:
        for s in bam.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord):
            q, r = get_aligned_pairs(s)
            for shift in (0, -3, 5):
                args = (locus.left_flank_coord + shift, locus.left_coord + shift, locus.right_coord - shift, locus.right_flank_coord - shift)
                a = get_read_coords_from_matched_pairs(*args, q, r)
                b = get_read_coords_from_cigar(*args, s)
                assert a == b, (s.name, shift)
                n += not a.is_incomplete()
    assert n > 100


def test_native_scan_and_extraction_equal_the_python_statement(tmp_path):
    """strk_bam_scan / strk_extract_reads (C++, host only) against bam.py / extract.py, record by record."""
    from strkit_amd.frontend import NativeBam, extract_reads, get_read_coords_from_cigar
    t = make_dataset(str(tmp_path), n_loci=12, reads_per_locus=10, read_len=1500, seed=4, sub=0.01, indel=0.02,
                     low_qual=0.02, soft_clip_frac=0.3, expansion=15)
    nb, pb = NativeBam(t["paths"]["bam"]), read_bam(t["paths"]["bam"])
    assert nb.n_records == len(pb.segments) == 120 and nb.references == pb.references
    (block,) = load_loci(t["paths"]["loci"])
    seen = np.zeros(3, int)
    for locus in block:
        idx = nb.fetch_indices(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        segs = pb.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        assert [nb.name(i) for i in idx] == [s.name for s in segs]
        assert nb.soft_clip_overlaps(idx, locus.left_flank_coord, locus.right_flank_coord).tolist() == \
            [s.soft_clip_overlaps_locus(locus) for s in segs]
        coords = np.tile([locus.left_flank_coord, locus.left_coord, locus.right_coord, locus.right_flank_coord], (len(idx), 1))
        ex = extract_reads(nb, idx, coords, 70, 13)
        for k, s in enumerate(segs):
            s2 = nb.segment(int(idx[k]))
            assert (s2.query_sequence, s2.start, s2.end, s2.flag) == (s.query_sequence, s.start, s.end, s.flag)
            assert np.array_equal(s2.cigar, s.cigar) and np.array_equal(s2.query_qualities, s.query_qualities)
            c = get_read_coords_from_cigar(locus.left_flank_coord, locus.left_coord, locus.right_coord, locus.right_flank_coord, s)
            seen[ex["status"][k]] += 1
            if c.is_incomplete():
                assert ex["status"][k] == 1
                continue
            try:
                sd = get_sequence_data_for_locus(s, c, 70)
            except LowMeanBaseQual:
                assert ex["status"][k] == 2
                continue
            fl, tr, fr = sd.flank_left_seq_wc[-70:], sd.tr_seq_wc, sd.flank_right_seq_wc[:70]
            assert ex["status"][k] == 0 and (ex["nfl"][k], ex["ntr"][k], ex["nfr"][k]) == (len(fl), len(tr), len(fr))
            assert ex["seqs"][ex["seq_off"][k]:ex["seq_off"][k + 1]].tobytes().decode() == fl + tr + fr
    assert seen[0] > 50 and seen[1] > 5


def test_realign_cigar_becomes_a_read_alignment():
    from strkit_amd.frontend.native import realign_cigar_to_read_alignment
    enc = {"M": 0, "I": 1, "D": 2, "S": 4, "=": 7, "X": 8}
    mk = lambda runs: np.array([(n << 4) | enc[o] for n, o in runs], np.uint32)  # noqa: E731
    got = realign_cigar_to_read_alignment(mk([(500, "D"), (70, "="), (12, "D"), (30, "="), (2, "I"), (40, "=")]))
    assert got.tolist() == mk([(500, "S"), (70, "="), (12, "I"), (30, "="), (2, "D"), (40, "=")]).tolist()


@pytest.mark.parametrize("seed,kw", [
    (101, dict(read_len=900, sub=0.0, indel=0.0)),
    (102, dict(read_len=2500, sub=0.03, indel=0.05, low_qual=0.05)),
    (103, dict(read_len=1200, sub=0.01, indel=0.01, soft_clip_frac=0.8, expansion=25, motif_len=(1, 3))),
    (104, dict(read_len=1500, sub=0.02, indel=0.02, low_qual=0.3, motif_len=(5, 12), cn_range=(3, 12))),
])
def test_native_extraction_on_varied_datasets(tmp_path, seed, kw):
    """Whole-catalog comparison of strk_extract_reads with extract.py under different error, clipping and quality
    regimes, with shifted boundaries (as the reference-side offsets produce them) and two flank sizes."""
    from strkit_amd.frontend import NativeBam, extract_reads, get_read_coords_from_cigar
    t = make_dataset(str(tmp_path), n_loci=10, reads_per_locus=8, seed=seed, **kw)
    nb, pb = NativeBam(t["paths"]["bam"]), read_bam(t["paths"]["bam"])
    (block,) = load_loci(t["paths"]["loci"])
    rng = np.random.default_rng(seed)
    n_ok = 0
    for locus in block:
        idx = nb.fetch_indices(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        segs = pb.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        for flank, phred in ((70, 13), (25, 30)):
            dl, dr = int(rng.integers(0, 6)), int(rng.integers(0, 6))
            c4 = (locus.left_coord - flank, locus.left_coord - dl, locus.right_coord + dr, locus.right_coord + flank)
            ex = extract_reads(nb, idx, np.tile(c4, (len(idx), 1)), flank, phred)
            for k, s in enumerate(segs):
                c = get_read_coords_from_cigar(*c4, s)
                if c.is_incomplete():
                    assert ex["status"][k] == 1
                    continue
                try:
                    sd = get_sequence_data_for_locus(s, c, flank, phred)
                except LowMeanBaseQual:
                    assert ex["status"][k] == 2
                    continue
                exp = sd.flank_left_seq_wc[-flank:] + sd.tr_seq_wc + sd.flank_right_seq_wc[:flank]
                assert ex["status"][k] == 0 and ex["seqs"][ex["seq_off"][k]:ex["seq_off"][k + 1]].tobytes().decode() == exp
                n_ok += 1
    assert n_ok > 20


def test_parallel_bgzf_inflate(tmp_path):
    import ctypes as C
    import gzip
    from strkit_amd import _lib
    from strkit_amd.frontend.native import bgzf_read
    t = make_dataset(str(tmp_path), n_loci=20, reads_per_locus=10, read_len=4000, seed=8, sub=0.01, indel=0.01)
    path = t["paths"]["bam"]
    with gzip.open(path, "rb") as fh:
        want = np.frombuffer(fh.read(), np.uint8)
    for threads in (0, 1, 3):
        assert np.array_equal(bgzf_read(path, threads), want)
    # a flipped payload byte is caught by the block's CRC (or by inflate itself)
    comp = np.fromfile(path, np.uint8).copy()
    L = _lib.load()
    n = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, None, 0, 0)
    assert n == want.size
    comp[comp.size // 2] ^= 0x5A
    out = np.empty(int(n), np.uint8)
    assert L.strk_bgzf_inflate(comp.ctypes.data, comp.size, out.ctypes.data, out.size, 2) < 0
    assert b"BGZF" in L.strk_last_error()
    # a plain gzip file is not BGZF: the reader falls back to Python's gzip
    plain = str(tmp_path / "plain.gz")
    with gzip.open(plain, "wb") as fh:
        fh.write(want[:5000].tobytes())
    raw = np.fromfile(plain, np.uint8)
    assert L.strk_bgzf_inflate(raw.ctypes.data, raw.size, None, 0, 0) < 0
    assert np.array_equal(bgzf_read(plain), want[:5000])


def _expansion_bam(tmp_path, n_reads=30, n_expanded=15, ins=3000, long_cigar=False):
    """One sparse locus (a block of its own): 30 bp reference tract, flank 70; half of the reads carry an insertion a
    hundred times the reference window (the flagship large-expansion case)."""
    from strkit_amd.frontend.bam import write_bam
    rng = np.random.default_rng(17)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(4, size=n))  # noqa: E731
    left, right = rnd(600), rnd(600)
    ref = left + "CAG" * 10 + right
    recs = []
    for i in range(n_reads):
        if i < n_expanded:
            seq = left + "CAG" * 10 + "CAG" * (ins // 3) + right
            cigar = [(630, "="), (ins, "I"), (600, "=")]
        else:
            seq, cigar = ref, [(1230, "=")]
        recs.append(dict(name=f"r{i}", flag=0, contig="chr1", pos=0, cigar=cigar, seq=seq, qual=np.full(len(seq), 40),
                         long_cigar=long_cigar and i % 2 == 0))
    path = str(tmp_path / "exp.bam")
    write_bam(path, [("chr1", len(ref))], recs)
    return path, ref


@pytest.mark.parametrize("long_cigar", [False, True])
def test_large_expansions_do_not_overflow_the_extraction_buffer(tmp_path, long_cigar):
    """A read may hold many times the reference window (a 4x bound made strk_extract_reads fail with
    STRK_E_NOMEM and aborted the sample); the buffer is now sized by a size query.  With long_cigar every other record
    stores its CIGAR the way alignments with more than 65 535 operations do (placeholder + CG:B,I tag)."""
    from strkit_amd.frontend import NativeBam, extract_reads, get_read_coords_from_cigar
    path, _ = _expansion_bam(tmp_path, long_cigar=long_cigar)
    nb, pb = NativeBam(path), read_bam(path)
    idx = nb.fetch_indices("chr1", 530, 700)
    assert len(idx) == 30
    c4 = (530, 600, 630, 700)
    ex = extract_reads(nb, idx, np.tile(c4, (30, 1)), 70, 13)
    assert (ex["status"] == 0).all() and sorted(set(ex["ntr"].tolist())) == [30, 3030]
    for k, s in enumerate(pb.fetch("chr1", 530, 700)):
        s2 = nb.segment(int(idx[k]))
        assert np.array_equal(s2.cigar, s.cigar) and (s.end, nb.end[idx[k]]) == (1230, 1230) and len(s.cigar) in (1, 3)
        sd = get_sequence_data_for_locus(s, get_read_coords_from_cigar(*c4, s), 70)
        assert ex["seqs"][ex["seq_off"][k]:ex["seq_off"][k + 1]].tobytes().decode() == \
            sd.flank_left_seq_wc[-70:] + sd.tr_seq_wc + sd.flank_right_seq_wc[:70]
    # size query (seqs == NULL) and a buffer one byte short
    import ctypes as C
    from strkit_amd import _lib
    L = _lib.load()
    n = 30
    rec_off = np.ascontiguousarray(nb.rec_off[idx], np.int64)
    coords = np.ascontiguousarray(np.tile(c4, (n, 1)), np.int64)
    st, a, b, c = (np.zeros(n, np.int32) for _ in range(4))
    off = np.zeros(n + 1, np.int64)
    args = (nb.data.ctypes.data, nb.data.size, n, rec_off.ctypes.data, coords.ctypes.data, None, None, None, 70, 13, 3,
            st.ctypes.data, a.ctypes.data, b.ctypes.data, c.ctypes.data)
    assert L.strk_extract_reads(*args, None, 0, off.ctypes.data) == 0 and off[-1] == ex["seq_off"][-1]
    buf = np.zeros(int(off[-1]), np.uint8)
    assert L.strk_extract_reads(*args, buf.ctypes.data, buf.size - 1, off.ctypes.data) == -12   # STRK_E_NOMEM
    assert L.strk_extract_reads(*args, buf.ctypes.data, buf.size, off.ctypes.data) == 0 and np.array_equal(buf, ex["seqs"])


def test_threaded_extraction_equals_the_serial_one(tmp_path):
    """More than two thousand items take the multi-threaded path of strk_extract_reads: same bytes as piecewise calls."""
    from strkit_amd.frontend import NativeBam, extract_reads
    t = make_dataset(str(tmp_path), n_loci=60, reads_per_locus=40, read_len=1000, seed=12, sub=0.01, indel=0.01, low_qual=0.02)
    nb = NativeBam(t["paths"]["bam"])
    (block,) = load_loci(t["paths"]["loci"])
    rec, coords = [], []
    for locus in block:
        idx = nb.fetch_indices(locus.contig, locus.left_flank_coord, locus.right_flank_coord)
        rec.append(idx)
        coords.append(np.tile([locus.left_flank_coord, locus.left_coord, locus.right_coord, locus.right_flank_coord], (len(idx), 1)))
    rec, coords = np.concatenate(rec), np.concatenate(coords)
    assert len(rec) > 2100
    whole = extract_reads(nb, rec, coords, 70, 13)
    parts = [extract_reads(nb, rec[i:i + 100], coords[i:i + 100], 70, 13) for i in range(0, len(rec), 100)]
    assert np.array_equal(whole["seqs"], np.concatenate([p["seqs"] for p in parts]))
    for k in ("status", "nfl", "ntr", "nfr"):
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts]))


def test_report_diff_logic():
    """strkit_amd/frontend/compare.py: the STRkit-JSON diff that tools/compare_strkit_json.py prints."""
    from strkit_amd.frontend.compare import diff_reports, format_diff
    theirs = {"parameters": {"rc_method": "repalign"}, "results": [
        {"locus_index": 1, "contig": "chr4", "start": 96617, "end": 96648, "motif": "AC", "ref_cn": 16, "start_adj": 96617,
         "end_adj": 96648, "reads": {"r1": {"s": "-", "sc": 2.0, "cn": 15, "w": 1.02}, "r2": {"s": "+", "sc": 1.9, "cn": 16},
                                     "r3": {"s": "+", "sc": None, "cn": 0}}},
        {"locus_index": 2, "contig": "chr4", "start": 200, "end": 230, "motif": "CAG", "ref_cn": 10, "reads": {}}]}
    ours = {"results": [
        {"locus_index": 7, "contig": "4", "start": 96617, "end": 96648, "motif": "ac", "ref_cn": 16, "start_adj": 96617,
         "end_adj": 96650, "reads": {"r1": {"s": "-", "sc": 2.0, "cn": 15, "w": 0.5}, "r2": {"s": "+", "sc": 1.8, "cn": 17},
                                     "r4": {"s": "+", "sc": 2.0, "cn": 15}}},
        {"locus_index": 9, "contig": "chr5", "start": 1, "end": 9, "motif": "A", "ref_cn": 8, "reads": {}}]}
    d = diff_reports(theirs, ours)
    assert (d["loci_common"], d["loci_only_theirs"], d["loci_only_ours"]) == (1, 1, 1)           # "chr4" == "4", motif case
    assert (d["locus_fields_compared"], d["locus_fields_equal"]) == (3, 2)                        # end_adj differs
    assert (d["reads_common"], d["reads_only_theirs"], d["reads_only_ours"], d["cn_equal"]) == (2, 1, 1, 1)
    assert (d["sc_compared"], d["sc_equal"]) == (2, 1) and not d["identical"]
    assert {x["kind"] for x in d["diffs"]} == {"end_adj", "cn", "sc", "read missing in ours", "locus missing in ours"}
    assert "DIFFERENT" in format_diff(d)
    same = diff_reports(theirs, theirs)
    assert same["identical"] and same["diffs"] == [] and same["sc_compared"] == 2


def test_report_sweep_orders_the_switch_combinations():
    """strkit_amd/frontend/compare.py::sweep over end-gap mode x tie rule x search-range schedule with a made-up `run`: the
    generating combination comes first and is the only identical one; without `narrowings` the callback keeps its two-argument
    form (tools written against the round-3 signature)."""
    from strkit_amd.frontend.compare import sweep

    def report(ef, tie, nw=0):
        cn = 10 + (ef == 5) + 2 * (tie == 1) + 4 * (nw == 2)
        return {"results": [{"locus_index": 1, "contig": "chr1", "start": 10, "end": 40, "motif": "CAG", "ref_cn": 10,
                             "reads": {"r1": {"s": "+", "sc": 1.5 + 0.1 * ef, "cn": cn}, "r2": {"s": "-", "sc": 2.0, "cn": 10 + (tie == 1)}}}]}
    theirs = report(5, 1, 2)
    rows = sweep(theirs, report, end_flags=(15, 5), tie_rules=(0, 1), narrowings=(0, 1, 2, 3))
    assert len(rows) == 16 and rows[0]["identical"] and (rows[0]["end_flags"], rows[0]["tie_rule"], rows[0]["narrowing"]) == (5, 1, 2)
    assert sum(r["identical"] for r in rows) == 1 and rows[0]["cn_equal"] == 2 and rows[-1]["cn_equal"] < 2
    calls = []

    def two_args(ef, tie):
        calls.append((ef, tie))
        return report(ef, tie)
    rows2 = sweep(report(15, 1), two_args, end_flags=(15, 5))
    assert len(rows2) == 4 and len(calls) == 4 and rows2[0]["identical"] and rows2[0]["narrowing"] == 0


def test_indexed_bam_regions_equal_the_whole_file(tmp_path):
    """Block-wise access through the .bai linear index (IndexedBam.region: virtual offset -> strk_bgzf_inflate_range ->
    strk_bam_scan_piece) returns, for every block of loci, the records the whole-file reader finds."""
    from strkit_amd.frontend import IndexedBam, NativeBam
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=90, depth=7, read_len=2500, seed=5, spacing=9000, procs=2)
    nb, ib = NativeBam(t["paths"]["bam"]), IndexedBam(t["paths"]["bam"])
    assert nb.n_records == t["n_reads"] == 630 and ib.references == nb.references == ["chr1"]
    blocks = load_loci(t["paths"]["loci"], max_block_size=25)
    assert len(blocks) == 4
    seen = 0
    for blk in blocks:
        reg = ib.region("chr1", min(l.left_flank_coord for l in blk), max(l.right_flank_coord for l in blk) + 1)
        assert reg.data.size < nb.data.size // 2                     # a region, not the file
        starts = np.array([l.left_flank_coord for l in blk]); ends = np.array([l.right_flank_coord for l in blk])
        rec, n_per = reg.fetch_many("chr1", starts, ends, 250)
        want = [nb.fetch_indices("chr1", int(s), int(e)) for s, e in zip(starts, ends)]
        assert n_per.tolist() == [len(w) for w in want]
        assert reg.names(rec) == [nb.name(int(i)) for w in want for i in w]
        k = 0
        for w in want[::6]:
            pass
        for li, w in enumerate(want):
            for i in w[:2]:
                a, b = reg.segment(int(rec[k])), nb.segment(int(i))
                assert (a.name, a.start, a.end, a.query_sequence) == (b.name, b.start, b.end, b.query_sequence) and np.array_equal(a.cigar, b.cigar)
                k += 1
            k += len(w) - min(2, len(w))
        seen += int(n_per.sum())
        capped, n_cap = reg.fetch_many("chr1", starts, ends, 3)
        assert n_cap.max() == 3 and np.array_equal(capped[:3], rec[:3])
    assert seen == 630
    # an interval without reads, a contig the file does not have, the "1" spelling of "chr1"
    assert ib.region("chr1", 10, 500).fetch_indices("chr1", 10, 500).size == 0
    assert ib.region("chrNope", 0, 1000).n_records == 0
    assert ib.region("1", int(starts[0]), int(ends[0])).n_records > 0
    # truth: every read extracted from its region gives its allele's size estimate (HiFi error rates)
    truth = {(int(a), int(b)): int(c) for a, b, c in t["truth"]}
    reg = ib.region("chr1", blocks[0][0].left_flank_coord, blocks[0][-1].right_flank_coord + 1)
    l0 = blocks[0][3]
    idx = reg.fetch_indices("chr1", l0.left_flank_coord, l0.right_flank_coord)
    from strkit_amd.frontend import extract_reads
    ex = extract_reads(reg, idx, np.tile([l0.left_flank_coord, l0.left_coord, l0.right_coord, l0.right_flank_coord], (len(idx), 1)), 70, 13)
    for k, name in enumerate(reg.names(idx)):
        l_, r_ = name[1:].split("_r")
        assert ex["status"][k] == 0 and abs(round(int(ex["ntr"][k]) / len(l0.motif)) - truth[(int(l_), int(r_))]) <= 1


def test_read_weights_reproduce_the_documented_example():
    """docs/output_formats.md:92-104: a read whose tract has 31 bases (flank 70: 171 with flanks) carries w = 1.0217145751733625
    in a sample of HiFi reads.  The formula of output.read_weights, (L + t - 2) / (L - t + 1), gives exactly that number for
    one L (15 790.2), a plausible mean HiFi read length - the one anchor the reference's tree holds for this quantity."""
    from strkit_amd.frontend.output import allele_calling_inputs, read_weights
    w, t = 1.0217145751733625, 171.0
    L = (t - 2 + w * (t - 1)) / (w - 1)
    assert 15000 < L < 17000
    assert abs(read_weights(np.array([L]), np.array([t]))[0] - w) < 1e-12
    # partition: only reads long enough to contain flank + tract + flank enter the mean
    lens = np.array([100, 200, 1000, 3000])
    got = read_weights(lens, np.array([150, 250, 90, 5000]))
    exp = [((200 + 1000 + 3000) / 3 + 148) / ((200 + 1000 + 3000) / 3 - 149), (2000 + 248) / (2000 - 249), (4300 / 4 + 88) / (4300 / 4 - 89)]
    assert np.allclose(got[:3], exp) and np.isnan(got[3])
    assert np.allclose(read_weights(lens, np.array([150.0]), read_length=np.array([900.0]), targeted=True), [(900 + 148) / (900 - 149)])
    cns, wn = allele_calling_inputs({"reads": {"a": {"cn": 8, "w": 1.0}, "b": {"cn": 9, "w": 3.0}}})
    assert cns.dtype == np.int32 and cns.tolist() == [8, 9] and wn.dtype == np.float64 and wn.tolist() == [0.25, 0.75]


def test_mcrl_slr_and_vcf_rows(tmp_path):
    from strkit_amd.frontend.output import mcrl_field, slr_field, write_vcf
    # the example of output/vcf.py:320-321: two alleles with 8 and 9 copies -> 7x1|8x10|9x1 , 8x2|9x12
    reads = {}
    for p, hist in ((0, {7: 1, 8: 10, 9: 1}), (1, {8: 2, 9: 12})):
        for cn, k in hist.items():
            for i in range(k):
                reads[f"r{p}_{cn}_{i}"] = {"s": "+", "cn": cn, "w": 1.0, "sc": 2.0, "sl": 3 * cn + (i == 0), "p": p}
    assert mcrl_field(reads, 2) == ("7x1|8x10|9x1", "8x2|9x12")
    assert slr_field(reads, 2) == ("21x0|22x1|24x9|25x1|27x0|28x1".replace("21x0|", "").replace("27x0|", ""), "24x1|25x1|27x11|28x1")
    assert mcrl_field(reads) == ("7x1|8x12|9x13",)
    rep = {"sample_id": "s1", "catalog": {"num_loci": 3}, "results": [
        {"locus_index": 2, "locus_id": "HTT", "contig": "chr4", "start": 3074876, "end": 3074940, "start_adj": 3074874, "end_adj": 3074940,
         "motif": "CAG", "ref_cn": 22, "ref_start_anchor": "ccatg", "ref_seq": "cag" * 22, "reads": reads, "call": [8, 9], "assign_method": "dist"},
        {"locus_index": 1, "locus_id": "l1", "contig": "chr4", "start": 100, "end": 130, "motif": "AC", "ref_cn": 15,
         "ref_start_anchor": "TTTTT", "ref_seq": "AC" * 15, "reads": {"x": {"s": "-", "cn": 15, "w": 1.0, "sc": 2.0, "sl": 30}}, "call": None},
        {"locus_index": 3, "locus_id": "skipped", "contig": "chr4", "start": 500, "end": 530, "motif": "AC", "call": None}]}
    path = str(tmp_path / "o.vcf")
    assert write_vcf(rep, path, date="20261004") == 2
    body = [l for l in open(path).read().splitlines() if not l.startswith("##")]
    assert body[0].split("\t")[-1] == "s1" and len(body) == 3
    f = body[1].split("\t")
    assert f[:5] == ["chr4", "96", "l1", "TTTTT" + "AC" * 15, "."] and f[7] == "VT=str;MOTIF=AC;REFMC=15;BED_START=100;BED_END=130;ANCH=5"
    assert f[8:] == ["GT:DP:MCRL:SLR", "./.:1:15x1:30x1"]
    f = body[2].split("\t")
    assert f[1] == str(3074874 - 5 + 1) and f[3] == "CCATG" + "CAG" * 22 and f[8] == "GT:DP:PM:MC:MCRL:SLR"
    assert f[9].split(":")[2:5] == ["dist", "8,9", "7x1|8x10|9x1,8x2|9x12"]


def test_device_inflater_code_on_the_host_equals_zlib(tmp_path):
    """strk_inflate.h (the decoder k_bgzf_inflate runs, one GPU lane per BGZF block) compiled for the host,
    strk_bgzf_inflate_sw: every block of a synthetic BAM, streams of all three block types (stored, fixed, dynamic codes),
    a corrupted payload."""
    import zlib
    from strkit_amd import _lib
    from strkit_amd.frontend.bam import bgzf_block
    t = make_dataset(str(tmp_path), n_loci=12, reads_per_locus=8, read_len=3000, seed=21, sub=0.01, indel=0.01)
    L = _lib.load()
    comp = np.fromfile(t["paths"]["bam"], np.uint8)
    n = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, None, 0, 0)
    a, b = np.empty(int(n), np.uint8), np.empty(int(n), np.uint8)
    assert L.strk_bgzf_inflate(comp.ctypes.data, comp.size, a.ctypes.data, a.size, 1) == n
    assert L.strk_bgzf_inflate_sw(comp.ctypes.data, comp.size, b.ctypes.data, b.size) == n
    assert np.array_equal(a, b)
    rng = np.random.default_rng(4)
    payloads = [b"", b"A", bytes(rng.integers(256, size=40000, dtype=np.uint8)), b"ACGT" * 9000, bytes(60000),
                bytes(rng.integers(33, 74, size=65000, dtype=np.uint8))]
    # runs of every short period and matches of every length at short and long distances, up to the last byte of the block
    # (the copy paths of the decoder: pattern fill from registers, 8 / 32 / 128 / 264 bytes per trip, byte-wise tail)
    runs = bytearray()
    for period in range(1, 41):
        pat = bytes(rng.integers(256, size=period, dtype=np.uint8))
        for reps in (3, 11, 40, 300 // period + 2):
            runs += pat * reps + bytes(rng.integers(256, size=int(rng.integers(1, 9)), dtype=np.uint8))
    far = bytearray(bytes(rng.integers(256, size=3000, dtype=np.uint8)))
    for ln in list(range(3, 40)) + [63, 64, 65, 127, 128, 129, 130, 200, 257, 258, 259, 300, 600]:
        for back in (ln, ln + 1, 31, 32, 33, 127, 128, 129, 263, 264, 265, 2000):
            if back <= len(far) and back >= 1:
                far += bytes(far[len(far) - back + i % back] if i >= back else far[len(far) - back + i] for i in range(ln)) if back < ln \
                    else far[len(far) - back:len(far) - back + ln]
                far += bytes(rng.integers(256, size=int(rng.integers(0, 4)), dtype=np.uint8))
    payloads += [bytes(runs[:65000]), bytes(far[:65000]), bytes(far[:60000]) + b"\x07" * 300, b"ab" * 150 + b"xyz" * 100]
    for raw in payloads:
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_HUFFMAN_ONLY),
                                (4, zlib.Z_RLE)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            body = co.compress(raw) + co.flush()
            if len(body) + 26 > 65536:
                continue
            blk = np.frombuffer(bgzf_block(raw, body), np.uint8)
            out = np.full(len(raw) + 8, 0xEE, np.uint8)
            assert L.strk_bgzf_inflate_sw(blk.ctypes.data, blk.size, out.ctypes.data, len(raw)) == len(raw), (len(raw), level, strategy)
            assert out[:len(raw)].tobytes() == raw and (out[len(raw):] == 0xEE).all()
    bad = comp.copy()
    bad[bad.size // 2] ^= 0x21
    assert L.strk_bgzf_inflate_sw(bad.ctypes.data, bad.size, b.ctypes.data, b.size) < 0 and b"BGZF" in L.strk_last_error()


def test_inflater_code_is_clean_under_the_sanitizers(tmp_path):
    """tools/inflate_asan.sh: strk_inflate.h compiled for the host with AddressSanitizer + UBSan inflates every block of a
    synthetic BAM and of streams for every copy path into heap buffers of exactly the block's size (the decoder stores whole
    words past the end of a match and reads its input up to 16 bytes ahead: never outside the block / the padded payload)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(HERE)
    r = subprocess.run(["bash", "tools/inflate_asan.sh"], cwd=root, env={**os.environ, "TMPDIR": str(tmp_path)}, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("the sanitizer run-time libraries are not installed")
    assert r.returncode == 0 and "clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_indexed_bam_slots_reuse_their_buffers(tmp_path):
    """IndexedBam.region(slot=k): the same records as without a slot, in a buffer that the next region of the slot takes over."""
    from strkit_amd.frontend import IndexedBam
    from strkit_amd.frontend.synth_large import make_dataset_large
    t = make_dataset_large(str(tmp_path), n_loci=60, depth=6, read_len=2500, seed=6, spacing=9000, procs=2)
    ib = IndexedBam(t["paths"]["bam"])
    spans = [(0, 150000), (150000, 300000), (300000, 540000), (20000, 90000)]
    plain = [ib.region("chr1", a, b) for a, b in spans]
    for k, (a, b) in enumerate(spans):
        reg = ib.region("chr1", a, b, slot=k % 2)
        assert reg.n_records == plain[k].n_records and np.array_equal(reg.rec_off, plain[k].rec_off)
        assert np.array_equal(reg.data, plain[k].data) and reg.names(np.arange(min(5, reg.n_records))) == plain[k].names(np.arange(min(5, reg.n_records)))
        if k >= 2:
            assert np.shares_memory(reg.data, ib._pool[k % 2])
    assert set(ib._pool) == {0, 1}
    # an index that is OLDER than its file (plain cp, rsync without -t) only warns, as htslib does: the same records come back
    bai = t["paths"]["bam"] + ".bai"
    old = os.path.getmtime(t["paths"]["bam"]) - 3600
    os.utime(bai, (old, old))
    with pytest.warns(RuntimeWarning, match="older than"):
        ib2 = IndexedBam(t["paths"]["bam"])
    assert ib2.region("chr1", 0, 150000).n_records == plain[0].n_records


def test_fasta_loader_handles_line_shapes(tmp_path):
    """Fixed-width lines take the strided path, anything else the masked one; case is kept (soft-masked references,
    docs/output_formats.md:96); contigs are matched with or without the chr prefix."""
    from strkit_amd.frontend.fasta import write_fasta
    rng = np.random.default_rng(12)
    seqs = {"chr1": "".join("ACGTacgtN"[i] for i in rng.integers(9, size=100_003)), "chr2": "ACGT" * 15, "chrEmptyTail": "A" * 60}
    p = str(tmp_path / "a.fa")
    write_fasta(p, seqs, width=60)
    f = Fasta(p)
    assert f.references == list(seqs) and all(f.fetch(k, 0, len(v) + 5) == v for k, v in seqs.items())
    assert f.get_reference_length("chr1") == 100_003 and f.fetch("1", 59, 62) == seqs["chr1"][59:62] and f.array("chr2").tobytes() == seqs["chr2"].encode()
    odd = str(tmp_path / "odd.fa")
    with open(odd, "w", newline="") as fh:
        fh.write(">a some description\nACGT\nAC\n\n>b\r\nAC GT\r\nTT\r\n>c\nACGTACGTAC\nACGTACGTAC\nACG")
    g = Fasta(odd)
    assert {k: g.fetch(k, 0, 100) for k in g.references} == {"a": "ACGTAC", "b": "ACGTTT", "c": "ACGTACGTACACGTACGTACACG"}
    with pytest.raises(KeyError):
        g.fetch("nope", 0, 1)
    with pytest.raises(IndexError):
        g.fetch("a", 7, 9)
    # lines of one width are addressed in place (no stripped copy), LF and CR LF alike, with or without a .fai
    from strkit_amd.frontend.fasta import _Contig
    assert isinstance(f.array("chr1"), _Contig) and isinstance(g.array("c"), _Contig) and not isinstance(g.array("a"), _Contig)
    crlf = str(tmp_path / "crlf.fa")
    with open(crlf, "w", newline="") as fh:
        fh.write(">x\r\n" + "".join(seqs["chr1"][i:i + 70] + "\r\n" for i in range(0, 100_003, 70)) + ">y\r\nACGT\r\n")
    c = Fasta(crlf)
    assert isinstance(c.array("x"), _Contig) and c.get_reference_length("x") == 100_003 and c.fetch("y", 0, 9) == "ACGT"
    pi = str(tmp_path / "indexed.fa")
    write_fasta(pi, seqs, width=60, index=True)
    fi = Fasta(pi)
    want = np.frombuffer(seqs["chr1"].encode(), np.uint8)
    idx = rng.integers(0, 100_003, size=5000)
    for reader in (f, c, fi):
        a = reader.array("x" if reader is c else "chr1")
        assert len(a) == 100_003 and np.array_equal(a[idx], want[idx]) and a.tobytes() == want.tobytes()
        for lo, hi in ((0, 0), (0, 1), (59, 61), (60, 120), (100_000, 100_003), (100_003, 100_010), (5, 99_999)):
            assert a[lo:hi].tobytes() == want[lo:hi].tobytes(), (lo, hi)
        with pytest.raises(IndexError):
            a[np.array([100_003])]
    assert fi.references == list(seqs) and all(fi.fetch(k, 0, len(v) + 5) == v for k, v in seqs.items())
    # an index that does not fit the file (or is older than it) is not trusted
    with open(pi + ".fai", "w") as fh:
        fh.write("chr1\t999999999\t6\t60\t61\n")
    assert Fasta(pi).fetch("chr2", 0, 100) == seqs["chr2"]


def test_one_pass_boundary_walk_equals_the_run_index(tmp_path):
    """strk_bamrec.h::read_coords_linear (what a GPU lane runs: one pass over the CIGAR, no arrays) against
    strk_frontend.h's Runs + read_coords on random alignments and boundaries, including boundaries inside insertions,
    deletions, soft clips and outside the alignment."""
    import ctypes as C
    from strkit_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(33)
    ops = np.array([0, 7, 8, 1, 2, 4, 3, 5], np.uint32)          # M = X I D S N H
    n_span = 0
    for case in range(4000):
        n = int(rng.integers(1, 14))
        op = ops[rng.choice(len(ops), size=n, p=[0.25, 0.25, 0.1, 0.12, 0.12, 0.06, 0.05, 0.05])]
        ln = rng.integers(0 if case % 7 == 0 else 1, 40, size=n).astype(np.uint32)
        cigar = np.ascontiguousarray((ln << 4) | op, np.uint32)
        start = int(rng.integers(0, 500))
        ref_len = int(ln[np.isin(op, [0, 2, 3, 7, 8])].sum())
        c = np.sort(rng.integers(start - 10, start + ref_len + 12, size=4)).astype(np.int64)
        a, b = np.zeros(4, np.int64), np.zeros(4, np.int64)
        rc = L.strk_read_coords_both(cigar.ctypes.data, n, start, c.ctypes.data, a.ctypes.data, b.ctypes.data)
        assert rc in (0, 3), (case, rc, cigar.tolist(), start, c.tolist())
        if rc == 3:
            n_span += 1
            assert np.array_equal(a, b), (case, cigar.tolist(), start, c.tolist(), a.tolist(), b.tolist())
    assert n_span > 300
