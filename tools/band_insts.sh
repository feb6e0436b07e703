#!/bin/bash
# VALU wave-instructions of k_dp_band per 10 000-locus call with parts of the kernel switched off (STRKIT_AMD_DBG bits: 1 no
# forward pass, 2 no backward pass, 4 no in-kernel search, 8 no fork rows): where the instructions that are not DP steps go.
# Run on the GPU box from the repo root; writes gpurun_out/band_insts/summary.txt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/band_insts
mkdir -p $OUT
for dbg in 0 4 12 5 6 7 15; do
  export STRKIT_AMD_DBG=$dbg
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/d$dbg -o b -- python3 tools/band_one_call.py 3 6 > $OUT/d$dbg.log 2>&1
  f=$(find $OUT/d$dbg -name "*counter_collection.csv" | head -1)
  python3 - "$f" $dbg <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(dict)
for r in rows:
    if r["Kernel_Name"].startswith("strk::k_dp_band(") or r["Kernel_Name"] == "strk::k_dp_band" or "k_dp_band(" in r["Kernel_Name"] and "wide" not in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
last = acc[sorted(acc, key=int)[-1]] if acc else {}
print("dbg=%2d" % int(sys.argv[2]), {k: round(v / 1e6, 2) for k, v in last.items()}, "M per launch")
PY
  tail -1 $OUT/d$dbg.log | tee -a $OUT/summary.txt
  rm -rf $OUT/d$dbg
done
