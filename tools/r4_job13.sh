#!/bin/bash
# round-4 GPU job 13: rocprofv3 evidence, configs 4 and 5
cd "$GRAFT_REPO_ROOT"
EXTRA="--config 4" tools/prof_bench.sh r04_cfg4 > gpurun_out/prof_r04_cfg4.log 2>&1; tail -6 gpurun_out/prof_r04_cfg4.log | cut -c1-300
EXTRA="--config 5 --loci 2000" STEPS=3 PRIME=3 COUNTERS="SQ_INSTS_VALU GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/prof_pmc_single.sh r04_cfg5 > gpurun_out/prof_r04_cfg5.log 2>&1; tail -12 gpurun_out/prof_r04_cfg5.log | cut -c1-400
