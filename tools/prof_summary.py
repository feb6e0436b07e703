#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/prof_bench.sh: per-kernel launch statistics (both traces) and the PMC
counters averaged per launch of each kernel.  Only FULL-SIZE launches count (a launch whose reading is at least half of
that kernel's largest one for the same counter; duration for the traces): the priming / parity-check calls of bench.py
are smaller than a timed step and would otherwise dilute the per-launch figures.
usage: prof_summary.py gpurun_out/prof_<tag> [summary.json [pmc_summary.json]]
pmc_summary.json is the file bench.py reads (profiles/<round>_pmc_summary.json): kernel -> counter -> value per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}


def full(v, frac=0.5):
    """the full-size launches of a kernel: readings within [frac, 1 / frac] of the median of the larger half of all readings
    (the priming / parity-check calls of bench.py are smaller than a timed step; the first call after band probation runs some
    kernels on a one-block grid and is an outlier the other way)"""
    s = sorted(v, reverse=True)
    h = s[:max(1, len(s) // 2)]
    med = h[len(h) // 2]
    return [x for x in v if frac * med <= x <= med / frac] if med > 0 else list(v)


def short(name):
    n = name.split("(")[0]
    return n.replace("strk::", "").replace("void ", "").strip()


for trace in ("trace", "trace_p1"):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(root, trace, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in rows.values()) or 1.0
    out[trace] = {}
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        f = full(v, 0.4)   # durations also vary with what the launch overlaps
        out[trace][k] = {"calls": len(v), "full_size_calls": len(f), "avg_us": sum(f) / len(f), "min_us": min(f), "max_us": max(f),
                         "total_ms": sum(v) / 1e3, "share": sum(v) / tot}
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_per_launch"] = {k: {c: sum(full(v)) / len(full(v)) for c, v in cs.items()} |
                         {"launches": max(len(v) for v in cs.values()),
                          "full_size_launches": len(full(cs["SQ_INSTS_VALU"])) if "SQ_INSTS_VALU" in cs else min(len(full(v)) for v in cs.values())}
                         for k, cs in pmc.items()}
if len(sys.argv) > 3:
    keep = {k: v for k, v in out["pmc_per_launch"].items() if k.startswith("k_")}
    for k, v in keep.items():
        if k in out.get("trace_p1", {}):
            v["unoverlapped_avg_us"] = out["trace_p1"][k]["avg_us"]
            # effective clock of the launch: GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            if v.get("GRBM_GUI_ACTIVE") and v["unoverlapped_avg_us"] > 300:
                v["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / (v["unoverlapped_avg_us"] * 1e3)
    json.dump(keep, open(sys.argv[3], "w"), indent=1)
json.dump(out, open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "summary.json"), "w"), indent=1)
for trace in ("trace", "trace_p1"):
    print(f"== {trace} (us per launch) ==")
    for k, v in out[trace].items():
        if v["share"] > 0.002:
            print(f"{k:28s} calls {v['calls']:5d} full {v['full_size_calls']:4d} avg {v['avg_us']:10.1f} min {v['min_us']:10.1f} total_ms {v['total_ms']:9.2f} share {v['share']:.3f}")
print("== PMC per launch ==")
for k, cs in out["pmc_per_launch"].items():
    if cs.get("SQ_INSTS_VALU", 0) > 1e5 or cs.get("FETCH_SIZE", 0) > 100:
        print(k, {c: round(v, 1) for c, v in cs.items()})
