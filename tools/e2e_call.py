#!/usr/bin/env python3
"""End to end from files: synthetic FASTA + catalog + BAM -> python -m strkit_amd call path, with stage times.
Usage: python tools/e2e_call.py [n_loci] [reads_per_locus] [read_len]"""
import json
import sys
import tempfile
import time

sys.path.insert(0, ".")
from strkit_amd.frontend import Fasta, NativeBam, call_sample  # noqa: E402
from strkit_amd.frontend.synth_dataset import make_dataset  # noqa: E402

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rpl = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rlen = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
d = tempfile.mkdtemp()
t0 = time.perf_counter()
truth = make_dataset(d, n_loci=n_loci, reads_per_locus=rpl, read_len=rlen, seed=1, sub=0.001, indel=0.002, low_qual=0.0005)
t1 = time.perf_counter()
bam = NativeBam(truth["paths"]["bam"])
ref = Fasta(truth["paths"]["ref"])
t2 = time.perf_counter()
call_sample(bam, ref, truth["paths"]["loci"])          # warm-up (library load, workspaces)
t3 = time.perf_counter()
rep = call_sample(bam, ref, truth["paths"]["loci"])
t4 = time.perf_counter()
n_reads = sum(len(r.get("reads", {})) for r in rep["results"])
ok = sum(rd["cn"] == t["reads"][name] for r, t in zip(rep["results"], truth["loci"]) for name, rd in r["reads"].items())
print(json.dumps({"loci": n_loci, "reads": n_reads, "read_len": rlen, "make_dataset_s": round(t1 - t0, 2),
                  "parse_bam_fasta_s": round(t2 - t1, 2), "call_sample_s": round(t4 - t3, 3),
                  "loci_per_s_end_to_end": round(n_loci / (t4 - t3), 1), "reads_per_s_end_to_end": round(n_reads / (t4 - t3)),
                  "reads_with_true_allele_cn": ok, "stage_times": rep["stage_times"]}))
