#!/usr/bin/env python3
"""Writes tests/golden/report_rows.json + report.vcf: the result rows (`reads` records as call_locus.py:1279-1288, the locus
fields of :1040-1047,1340-1352) and the VCF text of `python -m strkit_amd call` on a fixed synthetic data set
(synth_dataset.make_dataset with the parameters below).  Needs the GPU:  gpurun -- python tests/golden/make_report_golden.py
(writes under gpurun_out/golden/, copy from there).  These are outputs of THIS backend (a regression fixture of the
report layout), not STRkit outputs."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from strkit_amd.frontend import Fasta, call_sample  # noqa: E402
from strkit_amd.frontend.output import write_vcf  # noqa: E402
from strkit_amd.frontend.synth_dataset import make_dataset  # noqa: E402

PARAMS = dict(n_loci=8, reads_per_locus=7, read_len=1800, seed=42, sub=0.004, indel=0.006, low_qual=0.002, soft_clip_frac=0.5, expansion=12)

if __name__ == "__main__":
    out = os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(out, exist_ok=True)
    d = tempfile.mkdtemp()
    t = make_dataset(d, **PARAMS)
    rep = call_sample(t["paths"]["bam"], t["paths"]["ref"], t["paths"]["loci"], realign=True, sample_id="golden")
    json.dump({"params": PARAMS, "results": rep["results"]}, open(os.path.join(out, "report_rows.json"), "w"), indent=1)
    write_vcf(rep, os.path.join(out, "report.vcf"), Fasta(t["paths"]["ref"]), date="20261004")
    print(len(rep["results"]), "rows")
