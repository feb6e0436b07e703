"""End to end on the device: synthetic BAM + FASTA + catalog -> call_sample -> per-read copy numbers."""
import json

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
    for res, truth in zip(rep["results"], t["loci"]):
        assert res["call"] is None and res["read_peaks_called"] is False and res["ref_cn"] == truth["ref_cn"] and res["motif"] == truth["motif"]
        assert set(res["reads"]) == set(truth["reads"])
        for name, rd in res["reads"].items():
            assert rd["cn"] == truth["reads"][name] and rd["sc"] == 2.0 and rd["s"] in "+-"
            assert abs(rd["w"] - 0.1) < 1e-12 and rd["sl"] == truth["reads"][name] * len(truth["motif"])


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
