#!/bin/bash
# VALU / SALU / LDS instruction counts of k_dp_band with parts of it switched off (STRKIT_AMD_DBG, see KArgs::dbg):
# tools/prof_dbg.sh "0 4 5 6 12 15"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --no-cpu-baseline --no-extras --no-e2e --steps 6 --warmup 2 --pipeline 1"
for dbg in ${1:-0 15}; do
  OUT=gpurun_out/prof_dbg/$dbg
  mkdir -p $OUT
  STRKIT_AMD_DBG=$dbg rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT -o b -- python3 $ARGS > $OUT.log 2>&1
  python3 - "$OUT" "$dbg" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dp_band(" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in rows.items():
    top = max(v); f = [x for x in v if x >= 0.5 * top]
    out[k] = sum(f) / len(f)
print("dbg", sys.argv[2], {k: round(v / 1e6, 1) for k, v in sorted(out.items())})
PY
done
