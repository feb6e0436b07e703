#!/bin/bash
# one PMC pass (VALU / SALU / wait counters) of the one-call-at-a-time bench + its kernel trace: tools/prof_quick.sh <tag>
set -e
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-extras --no-e2e --steps 16 --warmup 4 --pipeline 1 $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p1 -o b -- python3 $ARGS > $OUT/trace_p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -o b -- python3 $ARGS > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_wait -o b -- python3 $ARGS > $OUT/pmc_wait.log 2>&1
python3 tools/prof_summary.py $OUT $OUT/summary.json $OUT/pmc_summary.json 2>&1 | grep -v "at::\|rocclr" | cut -c1-420
