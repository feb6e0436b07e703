#!/usr/bin/env python3
"""cProfile of one whole `call_sample` on the bench's end-to-end data set (10 000 loci x 30 reads x 15 kb), device front end:
where the wall time outside the kernels goes.  python tools/e2e_wall_profile.py [n_loci] [front_end]   (on the GPU box)"""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, ".")
from strkit_amd.frontend import call_sample
from strkit_amd.frontend.synth_large import make_dataset_large

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fe = sys.argv[2] if len(sys.argv) > 2 else "device"
d = make_dataset_large("/tmp/e2e_wall_profile", n_loci=n_loci, depth=30, read_len=15000, seed=11, procs=16)
p = d["paths"]
warm = "/tmp/e2e_wall_profile/warm.bed"
with open(p["loci"]) as fh, open(warm, "w") as out:
    out.writelines(fh.readlines()[:200])
call_sample(p["bam"], p["ref"], warm, front_end=fe)
t = time.perf_counter()
rep = call_sample(p["bam"], p["ref"], p["loci"], front_end=fe)
print("plain wall", round(time.perf_counter() - t, 4), rep["stage_times"], flush=True)
pr = cProfile.Profile()
pr.enable()
rep = call_sample(p["bam"], p["ref"], p["loci"], front_end=fe)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
