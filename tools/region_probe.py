#!/usr/bin/env python3
"""Where IndexedBam.region spends its time on a synthetic 30x data set (run on the GPU box's host): python tools/region_probe.py [n_loci]"""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, ".")
from strkit_amd.frontend.native import IndexedBam
from strkit_amd.frontend.synth_large import make_dataset_large

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
d = make_dataset_large("/tmp/region_probe", n_loci=n_loci, depth=30, read_len=15000, seed=11, procs=16)
b = IndexedBam(d["paths"]["bam"])
name, length = b.contigs[0]
step = 4_000_000
pr = cProfile.Profile()
tot = nbytes = 0.0
for k, beg in enumerate(range(0, length, step)):
    t = time.perf_counter()
    if k >= 3:
        pr.enable()
    r = b.region(name, beg, beg + step, slot=k % 3)
    pr.disable()
    dt = time.perf_counter() - t
    if k >= 3:
        tot += dt; nbytes += r.data.size
    print(k, r.n_records, round(r.data.size / 1e6), "MB", round(dt, 3), "s", round(r.data.size / dt / 1e9, 2), "GB/s", flush=True)
print("steady state:", round(nbytes / tot / 1e9, 2), "GB/s")
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
