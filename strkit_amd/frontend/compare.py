"""Diff of two `strkit call` JSON reports — one written by STRkit itself (elsewhere: the reference cannot run next to
this backend), one by `python -m strkit_amd call` on the same alignment file, reference and catalog.

This is the tool that can PIN the parity of this backend: the reference's tree holds no result vectors for the
repeat-count path, and the read-side counter lives in an un-vendored crate, so the switches it leaves open
(`end_flags`, `tie_rule`) can only be decided by comparing with a real report.  `sweep()` runs the
backend under every combination and ranks them by the number of per-read copy numbers and scores reproduced.

Fields compared (layout: strkit/call/output/json_report.py:37-154, call_locus.py:1040-1047,1279-1288,1340-1352):
per locus `ref_cn`, `start_adj`, `end_adj` (reference side: repalign always, call_locus.py:799-810), per read `cn` and
`sc` (the read side).  Loci are matched by (contig without "chr", start, end, motif), reads by name.
"""
from __future__ import annotations

import json
from typing import Callable, Iterable

__all__ = ["load_report", "diff_reports", "sweep", "format_diff"]

LOCUS_FIELDS = ("ref_cn", "start_adj", "end_adj")


def load_report(path: str) -> dict:
    with open(path) as fh:
        return json.load(fh)


def _key(row: dict) -> tuple:
    contig = str(row.get("contig", ""))
    return (contig[3:] if contig.startswith("chr") else contig, row.get("start"), row.get("end"), str(row.get("motif", "")).upper())


def diff_reports(theirs: dict, ours: dict, sc_tol: float = 1e-9, max_diffs: int = 50) -> dict:
    """Counts of compared / equal items and the first `max_diffs` differences.  `theirs` is the STRkit report."""
    a = {_key(r): r for r in theirs.get("results", [])}
    b = {_key(r): r for r in ours.get("results", [])}
    out = {"loci_theirs": len(a), "loci_ours": len(b), "loci_common": 0, "loci_only_theirs": 0, "loci_only_ours": 0,
           "locus_fields_compared": 0, "locus_fields_equal": 0, "reads_common": 0, "reads_only_theirs": 0,
           "reads_only_ours": 0, "cn_equal": 0, "sc_compared": 0, "sc_equal": 0, "diffs": []}

    def note(kind, key, **kw):
        if len(out["diffs"]) < max_diffs:
            out["diffs"].append({"kind": kind, "locus": "%s:%s-%s[%s]" % key, **kw})

    for key, ra in a.items():
        rb = b.get(key)
        if rb is None:
            out["loci_only_theirs"] += 1
            note("locus missing in ours", key)
            continue
        out["loci_common"] += 1
        for f in LOCUS_FIELDS:
            if f in ra or f in rb:
                out["locus_fields_compared"] += 1
                if ra.get(f) == rb.get(f):
                    out["locus_fields_equal"] += 1
                else:
                    note(f, key, theirs=ra.get(f), ours=rb.get(f))
        reads_a, reads_b = ra.get("reads") or {}, rb.get("reads") or {}
        for name, xa in reads_a.items():
            xb = reads_b.get(name)
            if xb is None:
                out["reads_only_theirs"] += 1
                note("read missing in ours", key, read=name, theirs={k: xa.get(k) for k in ("cn", "sc")})
                continue
            out["reads_common"] += 1
            if xa.get("cn") == xb.get("cn"):
                out["cn_equal"] += 1
            else:
                note("cn", key, read=name, theirs=xa.get("cn"), ours=xb.get("cn"))
            sa, sb = xa.get("sc"), xb.get("sc")
            if sa is not None or sb is not None:
                out["sc_compared"] += 1
                if sa is not None and sb is not None and abs(float(sa) - float(sb)) <= sc_tol:
                    out["sc_equal"] += 1
                else:
                    note("sc", key, read=name, theirs=sa, ours=sb)
        out["reads_only_ours"] += sum(1 for name in reads_b if name not in reads_a)
    out["loci_only_ours"] = sum(1 for key in b if key not in a)
    out["identical"] = (out["loci_only_theirs"] == out["loci_only_ours"] == out["reads_only_theirs"] == out["reads_only_ours"] == 0
                        and out["locus_fields_equal"] == out["locus_fields_compared"]
                        and out["cn_equal"] == out["reads_common"] and out["sc_equal"] == out["sc_compared"])
    return out


def sweep(theirs: dict, run: Callable[..., dict], end_flags: Iterable[int] = range(16),
          tie_rules: Iterable[int] = (0, 1), narrowings: Iterable[int] = (0,)) -> list[dict]:
    """`run(end_flags, tie_rule[, narrowing]) -> our report`; one row per combination, best first (most per-read copy numbers
    reproduced, then most scores).  The read side's switches do not touch ref_cn / start_adj / end_adj.  `narrowings`: the
    schedules of local_search_range to try (STRK_NARROW_*; `run` gets the third argument only when more than the default 0
    is asked for) — a report holds no iteration counts, so a schedule shows only where it ends the search on another size."""
    rows = []
    narrowings = tuple(narrowings)
    for ef in end_flags:
        for tr in tie_rules:
            for nw in narrowings:
                d = diff_reports(theirs, run(ef, tr) if narrowings == (0,) else run(ef, tr, nw), max_diffs=0)
                rows.append({"end_flags": ef, "tie_rule": tr, "narrowing": nw, "cn_equal": d["cn_equal"], "sc_equal": d["sc_equal"],
                             "reads_common": d["reads_common"], "reads_only_theirs": d["reads_only_theirs"],
                             "reads_only_ours": d["reads_only_ours"], "identical": d["identical"]})
    rows.sort(key=lambda r: (-r["cn_equal"], -r["sc_equal"], r["reads_only_theirs"] + r["reads_only_ours"], r["end_flags"], r["tie_rule"],
                             r["narrowing"]))
    return rows


def format_diff(d: dict) -> str:
    lines = [f"loci: {d['loci_common']} common, {d['loci_only_theirs']} only in theirs, {d['loci_only_ours']} only in ours",
             f"reference side (ref_cn, start_adj, end_adj): {d['locus_fields_equal']} / {d['locus_fields_compared']} equal",
             f"reads: {d['reads_common']} common, {d['reads_only_theirs']} only in theirs, {d['reads_only_ours']} only in ours",
             f"per-read cn: {d['cn_equal']} / {d['reads_common']} equal;  sc: {d['sc_equal']} / {d['sc_compared']} equal",
             "IDENTICAL on every compared field" if d["identical"] else "DIFFERENT"]
    for x in d["diffs"]:
        lines.append("  " + json.dumps(x))
    return "\n".join(lines)
