#!/bin/bash
# rocprofv3 evidence for the default bench (run from the repo root on the GPU box): kernel trace + stats for the
# pipelined and the one-call-at-a-time form, then separate PMC passes (never combined with traces).
# usage: tools/prof_bench.sh <tag>   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-bench}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-extras --no-e2e --steps 16 --warmup 4 $EXTRA"   # EXTRA: e.g. "--config 3"
echo "[prof $TAG] kernel trace, pipelined"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- python3 $ARGS > $OUT/trace.log 2>&1
echo "[prof $TAG] kernel trace, one call at a time"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p1 -o b -- python3 $ARGS --pipeline 1 > $OUT/trace_p1.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "[prof $TAG] pmc $set"
  # (a pass that does not come back — seen twice on this pool, after the program had finished — is cut off, the others still count)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o b -- python3 $ARGS --pipeline 1 > $OUT/pmc_$tag.log 2>&1 || echo "pmc $set failed"
done
grep -h "^{\"metric" $OUT/trace.log | cut -c1-300
python3 tools/prof_summary.py $OUT $OUT/summary.json $OUT/pmc_summary.json > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt | head -60
# keep what gets committed (stats, summaries, the bench lines), drop the raw per-launch tables
mkdir -p $OUT/keep
cp $OUT/summary.txt $OUT/summary.json $OUT/pmc_summary.json $OUT/keep/ 2>/dev/null || true
for t in trace trace_p1; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/keep/${t}_kernel_stats.csv; grep -h "^{\"metric" $OUT/$t.log > $OUT/keep/${t}_bench.json || true; done
rm -rf $OUT/trace $OUT/trace_p1 $OUT/pmc_*
