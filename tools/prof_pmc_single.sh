#!/bin/bash
# PMC evidence for a configuration whose combined passes did not come back in round 3 (config 5): ONE counter per pass, a short
# run (few calls), the program directly behind `--`, never combined with a trace domain; plus one kernel trace of the same command
# for the un-overlapped durations.  usage: EXTRA="--config 5 --loci 2000" tools/prof_pmc_single.sh <tag>  -> gpurun_out/prof_<tag>/
TAG=${1:-cfg5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-extras --no-e2e --steps ${STEPS:-4} --warmup 1 --prime ${PRIME:-4} --pipeline 1 $EXTRA"
echo "[prof $TAG] kernel trace, one call at a time"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p1 -o b -- python3 $ARGS > $OUT/trace_p1.log 2>&1 || echo "trace failed"
for set in ${COUNTERS:-SQ_INSTS_VALU GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES}; do
  echo "[prof $TAG] pmc $set"
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$set -o b -- python3 $ARGS > $OUT/pmc_$set.log 2>&1 || echo "pmc $set failed"
done
python3 tools/prof_summary.py $OUT $OUT/summary.json $OUT/pmc_summary.json > $OUT/summary.txt 2>&1 || true
head -40 $OUT/summary.txt | cut -c1-400
mkdir -p $OUT/keep
cp $OUT/summary.txt $OUT/summary.json $OUT/pmc_summary.json $OUT/keep/ 2>/dev/null || true
f=$(find $OUT/trace_p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/keep/trace_p1_kernel_stats.csv
grep -h "^{\"metric" $OUT/trace_p1.log > $OUT/keep/trace_p1_bench.json || true
rm -rf $OUT/trace_p1 $OUT/pmc_*/
