#!/bin/bash
# rocprofv3 evidence for the realignment kernels (run from the repo root on the GPU box).
# Kernel trace + stats, then separate PMC passes (never combined with traces).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_rl
mkdir -p $OUT
ARGS="tools/bench_realign.py 2048 300 15000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o rl -- python3 $ARGS > $OUT/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o rl -- python3 $ARGS > $OUT/pmc_$tag.log 2>&1 || echo "pmc $set failed"
done
find $OUT -name "*.csv" | head -30
