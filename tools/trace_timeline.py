#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per kernel launch start / duration / queue for the last calls of a run, and how
much launches of different queues overlap.  usage: trace_timeline.py <dir with *kernel_trace.csv> [n_last_band_launches]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("strk::", "").replace("void ", ""),
                     r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
rows.sort()
band = [r for r in rows if r[2] == "k_dp_band"]
if not band:
    sys.exit("no k_dp_band launches")
t0 = band[-n_last][0] - 300_000
sel = [r for r in rows if r[0] >= t0]
print(f"{'start_us':>10} {'dur_us':>9} {'queue':>6} {'grid':>8} {'lds':>6} {'vgpr':>5}  kernel")
for s, e, k, q, g, w, lds, vg in sel:
    if (e - s) > 3000 or k.startswith("k_dp_band") or k == "k_replay" or k == "k_plan":
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {q:>6} {g:>8} {lds:>6} {vg:>5}  {k}")
# overlap of consecutive k_dp_band launches
b = [r for r in sel if r[2] == "k_dp_band"]
tot = sum(e - s for s, e, *_ in b)
span = b[-1][1] - b[0][0]
print(f"k_dp_band: {len(b)} launches, summed duration {tot / 1e3:.1f} us over a span of {span / 1e3:.1f} us -> {tot / span:.2f} launches in flight on average")
