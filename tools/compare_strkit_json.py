#!/usr/bin/env python3
"""Compare a JSON report written by STRkit (`strkit call ... --json`) with this backend's on the same inputs.

    python tools/compare_strkit_json.py strkit_report.json --bam reads.bam --ref ref.fa --loci catalog.bed [--realign]
    python tools/compare_strkit_json.py strkit_report.json --ours our_report.json          # no GPU needed
    python tools/compare_strkit_json.py strkit_report.json --bam ... --ref ... --loci ... --sweep

Without --sweep: one run with the default switches (end_flags 15 = all four ends free, tie_rule 0 = first maximum) and a
field-by-field diff (exit code 1 when anything differs).  With --sweep: the backend runs under all 16 x 2 combinations
of the two read-side switches that the reference's tree leaves open and prints which of them reproduces STRkit's
per-read `cn` / `sc` — the evidence that pins (or refutes) the defaults recorded in DESIGN.md §2.  The call parameters
(flank size, realign, max reads, ...) are taken from the STRkit report's own `parameters` block unless given."""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from strkit_amd.frontend.compare import diff_reports, format_diff, load_report, sweep  # noqa: E402


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("strkit_json")
    ap.add_argument("--ours", help="a report of this backend that was written earlier (skips the run)")
    ap.add_argument("--bam")
    ap.add_argument("--ref")
    ap.add_argument("--loci")
    ap.add_argument("--realign", action="store_true", default=None)
    ap.add_argument("--flank-size", type=int, default=None)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--max-diffs", type=int, default=30)
    ap.add_argument("--json", help="write the diff (or the sweep table) as JSON here")
    a = ap.parse_args(argv)
    theirs = load_report(a.strkit_json)
    tp = theirs.get("parameters", {})
    if tp.get("rc_method", "repalign") != "repalign":
        print(f"note: the STRkit report was made with rc_method={tp['rc_method']!r}; its per-read cn are not repalign "
              f"answers (ref_cn / start_adj / end_adj still are, call_locus.py:799-810)", file=sys.stderr)
    if a.ours:
        d = diff_reports(theirs, load_report(a.ours), max_diffs=a.max_diffs)
        print(format_diff(d))
        if a.json:
            json.dump(d, open(a.json, "w"), indent=1)
        return 0 if d["identical"] else 1
    if not (a.bam and a.ref and a.loci):
        ap.error("--bam, --ref and --loci are needed to run the backend (or pass --ours)")
    from strkit_amd.frontend import Fasta, NativeBam, call_sample
    from strkit_amd.repeat_count_params import RepeatCountParams
    bam, ref = NativeBam(a.bam), Fasta(a.ref)
    kw = dict(flank_size=a.flank_size or tp.get("flank_size", 70),
              realign=tp.get("realign", False) if a.realign is None else a.realign,
              min_avg_phred=tp.get("min_avg_phred", 13), max_reads=tp.get("max_reads", 250),
              respect_ref=tp.get("respect_ref", False), min_read_align_score=tp.get("min_read_align_score", 0.1),
              rc_params=RepeatCountParams("repalign", tp.get("max_rcn_iters", 50), 3, 1))

    def run(end_flags=15, tie_rule=0, narrowing=0):
        return call_sample(bam, ref, a.loci, end_flags=end_flags, tie_rule=tie_rule, narrowing=narrowing, **kw)

    if a.sweep:
        rows = sweep(theirs, run, narrowings=(0, 1, 2, 3))
        print("end_flags tie_rule narrowing  cn_equal  sc_equal  reads_common  only_theirs  only_ours")
        for r in rows:
            print(f"{r['end_flags']:9d} {r['tie_rule']:8d} {r['narrowing']:9d} {r['cn_equal']:9d} {r['sc_equal']:9d} {r['reads_common']:13d} "
                  f"{r['reads_only_theirs']:12d} {r['reads_only_ours']:10d}{'   <- identical' if r['identical'] else ''}")
        if a.json:
            json.dump(rows, open(a.json, "w"), indent=1)
        return 0 if rows and rows[0]["identical"] else 1
    d = diff_reports(theirs, run(), max_diffs=a.max_diffs)
    print(format_diff(d))
    if a.json:
        json.dump(d, open(a.json, "w"), indent=1)
    return 0 if d["identical"] else 1


if __name__ == "__main__":
    sys.exit(main())
