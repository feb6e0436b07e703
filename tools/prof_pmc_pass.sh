#!/bin/bash
# One more PMC pass of the bench (for when a pass of tools/prof_bench.sh did not come back): tools/prof_pmc_pass.sh <tag> "<counters>"
# with EXTRA as there; writes gpurun_out/prof_<tag>/pmc_<counters>/ and re-runs the summary.
TAG=$1; set=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-extras --no-e2e --steps 16 --warmup 4 $EXTRA"
tag=$(echo $set | tr ' ' '_' | cut -c1-40)
echo "[prof $TAG] pmc $set"
timeout -k 10 500 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o b -- python3 $ARGS --pipeline 1 > $OUT/pmc_$tag.log 2>&1 || echo "pmc $set failed"
f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
echo "csv: $f"
python3 tools/prof_summary.py $OUT $OUT/summary.json $OUT/pmc_summary_$tag.json > $OUT/summary_$tag.txt 2>&1 || true
tail -12 $OUT/summary_$tag.txt | cut -c1-300
rm -rf $OUT/pmc_$tag
