#!/bin/bash
# PMC counters of k_dp_long by itself (tools/long_probe.py: config 5's shape, band off), one pass per counter set; run from the repo root on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_long; mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o b -- python3 tools/long_probe.py 500 3 0 > $OUT/pmc_$tag.log 2>&1 || echo "pmc $set failed"
  f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('strk::','')
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'k_dp_long' in k or 'k_dp_all' in k:
        print(k, {c:(round(sum(x)/len(x)), len(x)) for c,x in v.items()})
PY
  rm -rf $OUT/pmc_$tag
done
