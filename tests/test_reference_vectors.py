"""Consumes tests/golden/reference_vectors.json — input -> output vectors of STRkit's own functions, written by
tools/make_reference_vectors.py on a machine that has STRkit installed.  This is the only route from "parity unpinned"
 to a pinned oracle: STRkit's read-side arithmetic is in two packages that are not in the reference
tree and cannot be imported in the build container.

Absent file: every test here skips and says so.  Present: the CPU oracle (and, with a GPU, the HIP library) must reproduce
every vector under the DEFAULT switches; if they do not, the failure names the combination of end-gap flags and tie rule
that does — which is then what the defaults have to become."""
import json
import os

import numpy as np
import pytest

import oracle

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")
# parasail's semi-global variants -> the oracle's end-gap flags (s1 = parasail's query = the read window, s2 = its database =
# the candidate; "b" = begin, "e" = end, "x" = both ends of that sequence free)
PARASAIL_FLAGS = {"sg": 15, "sg_qb": 1, "sg_qe": 2, "sg_qx": 3, "sg_db": 4, "sg_de": 8, "sg_dx": 12, "sg_qb_de": 9, "sg_qe_db": 6,
                  "sg_qb_db": 5, "sg_qe_de": 10}


def _vectors():
    if not os.path.exists(PATH):
        pytest.skip("tests/golden/reference_vectors.json is absent: run tools/make_reference_vectors.py where STRkit is installed "
                    "and commit its output (parity stays unpinned until then)")
    with open(PATH) as f:
        v = json.load(f)
    assert v.get("schema") == 1, "unknown schema of reference_vectors.json"
    return v


def _count_mismatches(vectors, fn):
    bad = []
    for i, rec in enumerate(vectors):
        try:
            got = fn(rec)
        except ValueError:
            got = "raises"
        exp = "raises" if "raises" in rec else rec["result"]
        if got != exp:
            bad.append((i, got, exp))
    return bad


def _oracle_repeat_count(rec, flags=15, tie=0, narrowing=0):
    (cn, sc), n, off = oracle.repeat_count(rec["start"], rec["tr"], rec["fl"], rec["fr"], rec["motif"], rec["max_iters"], rec["lsr"],
                                           rec["step"], tie_rule=tie, flags=flags, narrowing=narrowing)
    return [[cn, sc], n, off]


def test_parasail_semi_global_scores_pin_the_recurrence_and_the_flags():
    v = _vectors()["parasail_scores"]
    if not v:
        pytest.skip("the vector file holds no parasail scores")
    bad = []
    for rec in v:
        sc, e1, e2 = oracle.sg_align(rec["query"], rec["db"], rec["open"], rec["extend"], PARASAIL_FLAGS[rec["fn"]])
        if sc != rec["score"]:
            bad.append((rec["fn"], sc, rec["score"]))
    assert not bad, f"{len(bad)} of {len(v)} parasail scores differ, e.g. {bad[:5]}"


def test_oracle_reproduces_get_repeat_count():
    v = _vectors()["repeat_count"]
    bad = _count_mismatches(v, _oracle_repeat_count)
    if bad:
        fits = [(fl, tie, nw) for fl in range(16) for tie in (0, 1) for nw in range(4)
                if not _count_mismatches(v, lambda r, fl=fl, tie=tie, nw=nw: _oracle_repeat_count(r, fl, tie, nw))]
        # scores alone (schedule-independent whenever the searches end on the same size): which end-gap modes fit at all
        pytest.fail(f"{len(bad)} of {len(v)} get_repeat_count vectors differ under the defaults (end_flags 15, first maximum, fixed search "
                    f"range), e.g. {bad[:3]}; combinations of (end_flags, tie_rule, narrowing) that reproduce all of them: {fits or 'none'}")


def test_oracle_reproduces_get_ref_repeat_count():
    v = _vectors()["ref_repeat_count"]
    if not v:
        pytest.skip("the vector file holds no reference-side vectors")

    def run(rec):
        res, lo, ro, (n1, n2), (fl, tr, fr) = oracle.ref_repeat_count(rec["start"], rec["tr"], rec["fl"], rec["fr"], rec["motif"],
                                                                      rec["ref_size"], rec["vcf_anchor_size"], rec["max_iters"],
                                                                      rec["lsr"], rec["step"], rec["respect_coords"])
        return [[[res[0][0], res[0][1]], res[1], res[2]], lo, ro, [n1, n2], [fl, tr, fr]]
    bad = _count_mismatches(v, run)
    assert not bad, f"{len(bad)} of {len(v)} get_ref_repeat_count vectors differ, e.g. {bad[:2]}"


def test_oracle_reproduces_the_realignment_call():
    v = _vectors()["realign"]
    if not v:
        pytest.skip("the vector file holds no realignment vectors")
    from helpers import cigar_tuples
    bad = []
    for rec in v:
        fits = []
        for pref in (0, 1):
            sc, e2, cg = oracle.realign(rec["ref"], rec["read"], 7, 0, pref)
            fits.append((sc, e2, "".join(f"{n}{o}" for n, o in cigar_tuples(cg))))
        if rec["score"] != fits[0][0] or rec["cigar"] != fits[0][2]:
            bad.append((rec["score"], rec["cigar"], fits))
    assert not bad, f"{len(bad)} of {len(v)} realignments differ under gap_pref 0 (both preferences shown), e.g. {bad[:2]}"


@pytest.mark.gpu
def test_library_reproduces_get_repeat_count(gpu_ctx):
    from strkit_amd.repeat_count_params import RepeatCountParams
    from strkit_amd.repeats import get_repeat_count
    v = _vectors()["repeat_count"]

    def run(rec):
        get_repeat_count.cache_clear()
        (cn, sc), n, off = get_repeat_count(rec["start"], rec["tr"], rec["fl"], rec["fr"], rec["motif"],
                                            RepeatCountParams("repalign", rec["max_iters"], rec["lsr"], rec["step"]))
        return [[cn, sc], n, off]
    bad = _count_mismatches(v, run)
    assert not bad, f"{len(bad)} of {len(v)} get_repeat_count vectors differ on the device, e.g. {bad[:3]}"


@pytest.mark.gpu
def test_library_reproduces_get_ref_repeat_count(gpu_ctx):
    from strkit_amd.repeat_count_params import RepeatCountParams
    from strkit_amd.repeats import get_ref_repeat_count
    v = _vectors()["ref_repeat_count"]
    if not v:
        pytest.skip("the vector file holds no reference-side vectors")

    def run(rec):
        res, lo, ro, (n1, n2), (fl, tr, fr) = get_ref_repeat_count(rec["start"], rec["tr"], rec["fl"], rec["fr"], rec["motif"], rec["ref_size"],
                                                                   rec["vcf_anchor_size"], RepeatCountParams("repalign", rec["max_iters"], rec["lsr"], rec["step"]),
                                                                   respect_coords=rec["respect_coords"])
        return [[[res[0][0], res[0][1]], res[1], res[2]], lo, ro, [n1, n2], [fl, tr, fr]]
    bad = _count_mismatches(v, run)
    assert not bad, f"{len(bad)} of {len(v)} get_ref_repeat_count vectors differ on the device, e.g. {bad[:2]}"


def test_the_kit_runs_against_a_stand_in_of_the_vector_file(tmp_path, monkeypatch):
    """The consumer side itself is tested: a vector file written FROM THE ORACLE (not from STRkit: it pins nothing) goes
    through the same checks, and a corrupted vector is caught and reported with the switches that would fit."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "count_cases.json")) as f:
        cases = json.load(f)
    recs = []
    for locus in cases["cfg1_hifi"]["loci"][:4]:
        for (fl, tr, fr), est in zip(locus["reads"][:3], locus["est_cn"][:3]):
            rec = {"start": int(est), "tr": tr, "fl": fl, "fr": fr, "motif": locus["motif"], "max_iters": 50, "lsr": 3, "step": 1}
            rec["result"] = _oracle_repeat_count(rec)
            recs.append(rec)
    stand_in = {"schema": 1, "repeat_count": recs, "ref_repeat_count": [], "realign": [], "parasail_scores": []}
    p = tmp_path / "reference_vectors.json"
    p.write_text(json.dumps(stand_in))
    monkeypatch.setattr(sys.modules[__name__], "PATH", str(p))
    test_oracle_reproduces_get_repeat_count()
    recs[0]["result"][0][1] += 1
    p.write_text(json.dumps(stand_in))
    with pytest.raises(pytest.fail.Exception, match="get_repeat_count vectors differ"):
        test_oracle_reproduces_get_repeat_count()
