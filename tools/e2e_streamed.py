#!/usr/bin/env python3
"""The bench's end-to-end data set through the streamed device front end (spans of the file through HBM) against the resident
one: wall time and identical reports.  python tools/e2e_streamed.py [span_mb ...]   (on the GPU box)"""
import sys
import time

sys.path.insert(0, ".")
from strkit_amd.frontend import DeviceBam, Fasta, call_sample
from strkit_amd.frontend.synth_large import make_dataset_large

spans = [int(x) for x in sys.argv[1:]] or [128, 512, 4096]
d = make_dataset_large("/tmp/e2e_streamed", n_loci=10000, depth=30, read_len=15000, seed=11, procs=16)
p = d["paths"]
call_sample(p["bam"], p["ref"], p["loci"], front_end="device")
t = time.perf_counter()
whole = call_sample(p["bam"], p["ref"], p["loci"], front_end="device")
print("resident:", round(time.perf_counter() - t, 4), "s", whole["stage_times"], flush=True)
for mb in spans:
    t = time.perf_counter()
    db = DeviceBam(p["bam"], span_bytes=mb << 20)
    try:
        rep = call_sample(db, Fasta(p["ref"]), p["loci"])
        st = dict(db.open_stage_s)
    finally:
        db.close()
    print(f"streamed, spans of {mb} MB:", round(time.perf_counter() - t, 4), "s", st, "same report:", rep["results"] == whole["results"],
          {k: v for k, v in rep["stage_times"].items() if k in ("load_s", "extract_s", "count_s", "report_s")}, flush=True)
