#!/usr/bin/env python3
"""Pinning kit: function-level input -> output vectors of STRkit's OWN repeat-count path.

Run this ONCE on a machine that has STRkit installed (`pip install strkit`, which brings `strkit_rust_ext` and `parasail`;
Python >= 3.11).  It does not need a GPU or this backend's library, only this repository's `tests/golden/*.json` for the
inputs:

    python tools/make_reference_vectors.py [--out tests/golden/reference_vectors.json]

It calls, with the seeded inputs already committed under tests/golden/ (read windows cut into flank | tract | flank,
motifs, start estimates):

  * `strkit.call.repeats.get_repeat_count(start_count, tr_seq, flank_left_seq, flank_right_seq, motif, rc_params)`
    (strkit/call/repeats.py:47-70 -> strkit_rust_ext.get_repeat_count) at several start counts per read window;
  * `strkit.call.repeats.get_ref_repeat_count(...)` (repeats.py:73-192) on the first window of every locus;
  * `strkit.call.realign.realign_read(...)` (realign.py:34-72) and the bare parasail call behind it
    (`sg_dx_trace_scan_16`, realign.py:56) on the committed realignment pairs;
  * every semi-global parasail variant the installed version has (`sg`, `sg_qb`, `sg_qe`, `sg_qx`, `sg_db`, `sg_de`,
    `sg_dx`, `sg_qb_de`, `sg_qe_db`, ...: `<name>_scan_sat`) on candidate / window pairs with the counting gap model
    (open = extend = 5, align_matrix.py:17), which pins the recurrence and the meaning of the end-gap flags;

and writes ONE JSON file of inputs and outputs.  Commit that file: `tests/test_reference_vectors.py` then checks the CPU
oracle against it (`-m "not gpu"`) and the HIP library against it (`-m gpu`), and names, if the defaults do not reproduce
it, which combinations of the open switches (end-gap flags x tie rule x search-range schedule) do.  Without the file those
tests skip; parity stays "unpinned" until it exists.  Nothing of STRkit's source is read or copied by this script: it only
imports the installed package and records what its functions return.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
SCHEMA = 1
START_SHIFTS = (0, -1, 2, -5, 7)          # start counts tried per read window, relative to its estimate
PARASAIL_VARIANTS = ("sg", "sg_qb", "sg_qe", "sg_qx", "sg_db", "sg_de", "sg_dx", "sg_qb_de", "sg_qe_db", "sg_qb_db", "sg_qe_de")


def _version(mod) -> str:
    try:
        from importlib.metadata import version
        return version(mod)
    except Exception:  # noqa: BLE001
        return "unknown"


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(GOLDEN, "reference_vectors.json"))
    ap.add_argument("--max-windows", type=int, default=400, help="read windows taken from count_cases.json")
    a = ap.parse_args(argv)

    try:
        from strkit.call.repeats import get_ref_repeat_count, get_repeat_count
        from strkit.call.repeat_count_params import RepeatCountParams
    except Exception as e:  # noqa: BLE001
        print(f"this script needs an installed STRkit (pip install strkit): {e!r}", file=sys.stderr)
        return 2
    notes: list[str] = []
    out: dict = {"schema": SCHEMA, "versions": {m: _version(m) for m in ("strkit", "strkit_rust_ext", "parasail")},
                 "python": sys.version.split()[0], "repeat_count": [], "ref_repeat_count": [], "realign": [],
                 "parasail_scores": [], "notes": notes}

    with open(os.path.join(GOLDEN, "count_cases.json")) as f:
        cases = json.load(f)
    rc = RepeatCountParams(method="repalign", max_iters=50, initial_local_search_range=3, initial_step_size=1)
    rc_kw = {"max_iters": 50, "lsr": 3, "step": 1}
    windows = []            # (motif, fl, tr, fr, est)
    for name in sorted(cases):
        for locus in cases[name]["loci"]:
            for (fl, tr, fr), est in zip(locus["reads"], locus["est_cn"]):
                windows.append((locus["motif"], fl, tr, fr, int(est), name))
    windows = windows[:: max(1, len(windows) // a.max_windows)][:a.max_windows]
    # the inputs the oracle's open assumptions hinge on (tests/golden/make_pin_cases.py): every one of them, in front
    pin_path = os.path.join(GOLDEN, "pin_cases.json")
    ref_pin = []
    if os.path.exists(pin_path):
        with open(pin_path) as f:
            pin = json.load(f)
        extra = []
        for name in sorted(pin):
            for locus in pin[name]["loci"]:
                for (fl, tr, fr), est in zip(locus["reads"], locus["est_cn"]):
                    (ref_pin if name == "ref_ties" else extra).append((locus["motif"], fl, tr, fr, int(est), "pin_" + name))
        windows = extra + ref_pin + windows

    # ---- read side: get_repeat_count ------------------------------------------------------------------------------
    for motif, fl, tr, fr, est, name in windows:
        for shift in START_SHIFTS:
            start = max(0, est + shift)
            rec = {"case": name, "start": start, "tr": tr, "fl": fl, "fr": fr, "motif": motif, **rc_kw}
            try:
                (cn, score), n, off = get_repeat_count(start, tr, fl, fr, motif, rc)
                rec["result"] = [[int(cn), int(score)], int(n), int(off)]
            except Exception as e:  # noqa: BLE001  (e.g. max() of an empty sequence)
                rec["raises"] = type(e).__name__
            out["repeat_count"].append(rec)
            get_repeat_count.cache_clear() if hasattr(get_repeat_count, "cache_clear") else None

    # ---- reference side: get_ref_repeat_count on the first window of each locus ----------------------------------------
    seen = set()
    for motif, fl, tr, fr, est, name in windows:
        key = (motif, fl, tr, fr)
        if key in seen or not tr:
            continue
        seen.add(key)
        for respect in (False, True):
            rec = {"case": name, "start": est, "tr": tr, "fl": fl, "fr": fr, "motif": motif, "ref_size": len(tr),
                   "vcf_anchor_size": 5, "respect_coords": respect, **rc_kw}
            try:
                res, l_off, r_off, (n_off, n_it), (fl2, tr2, fr2) = get_ref_repeat_count(
                    est, tr, fl, fr, motif, len(tr), 5, rc, respect_coords=respect)
                rec["result"] = [[[int(res[0][0]), int(res[0][1])], int(res[1]), int(res[2])], int(l_off), int(r_off),
                                 [int(n_off), int(n_it)], [fl2, tr2, fr2]]
            except Exception as e:  # noqa: BLE001
                rec["raises"] = type(e).__name__
            out["ref_repeat_count"].append(rec)
        if len(out["ref_repeat_count"]) >= 240 + 2 * len(ref_pin):
            break

    # ---- parasail: semi-global variants with the counting gap model ---------------------------------------------------
    try:
        import parasail
        from strkit.call.align_matrix import dna_matrix, indel_penalty
        for motif, fl, tr, fr, est, name in windows[:120 + len(ref_pin) + 80]:
            db = fl + tr + fr
            for i in (max(0, est - 2), est, est + 3):
                cand = fl + motif * i + fr
                if not db or not cand:
                    continue
                for fn in PARASAIL_VARIANTS:
                    f = getattr(parasail, fn + "_scan_sat", None)
                    if f is None:
                        continue
                    # parasail's "query" is its first sequence argument: the read window, as in repeats.py:92-93
                    r = f(db, cand, indel_penalty, indel_penalty, dna_matrix)
                    out["parasail_scores"].append({"fn": fn, "query": db, "db": cand, "open": indel_penalty, "extend": indel_penalty,
                                                   "score": int(r.score), "end_query": int(r.end_query), "end_ref": int(r.end_ref)})
    except Exception as e:  # noqa: BLE001
        notes.append(f"parasail score vectors skipped: {e!r}")

    # ---- realignment ---------------------------------------------------------------------------------------------------
    try:
        import parasail
        from strkit.call.align_matrix import dna_matrix
        from strkit.call.realign import realign_read
        with open(os.path.join(GOLDEN, "realign_cases.json")) as f:
            rcases = [c for c in json.load(f) if c["open"] == 7 and c["extend"] == 0 and c["gap_pref"] == 0]
        for c in rcases:
            ref, read = c["ref"], c["read"]
            rec = {"ref": ref, "read": read, "left_flank_coord": 1000, "flank_size": max(1, len(ref) // 4)}
            pr = parasail.sg_dx_trace_scan_16(ref, read, 7, 0, dna_matrix)
            rec.update({"score": int(pr.score), "end_query": int(pr.end_query), "end_ref": int(pr.end_ref),
                        "cigar": pr.cigar.decode.decode() if isinstance(pr.cigar.decode, bytes) else str(pr.cigar.decode),
                        "saturated": bool(getattr(pr, "saturated", False))})
            try:
                pairs = realign_read(ref, read, rec["left_flank_coord"], rec["flank_size"], None, "vector", 0)
                if pairs is None:
                    rec["pairs"] = None
                else:   # STRkitAlignedCoords: query_coords / ref_coords
                    rec["pairs"] = [[int(x) for x in pairs.query_coords], [int(x) for x in pairs.ref_coords]]
            except Exception as e:  # noqa: BLE001
                rec["realign_read_raises"] = type(e).__name__
            out["realign"].append(rec)
    except Exception as e:  # noqa: BLE001
        notes.append(f"realignment vectors skipped: {e!r}")

    notes.append("Rust methods on alignment segments (get_est_copy_num, calc_adj_score, get_read_weight) need BAM-backed objects; "
                 "they are pinned at report level with tools/compare_strkit_json.py instead")
    with open(a.out, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {a.out}: {len(out['repeat_count'])} get_repeat_count, {len(out['ref_repeat_count'])} get_ref_repeat_count, "
          f"{len(out['parasail_scores'])} parasail scores, {len(out['realign'])} realignments")
    return 0


if __name__ == "__main__":
    sys.exit(main())
