#!/bin/bash
# round-4 GPU job 12: rocprofv3 evidence of the final library: configs 2, 3, 4 (prof_bench.sh) and config 5 (one counter per pass)
cd "$GRAFT_REPO_ROOT"
tools/prof_bench.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -25 gpurun_out/prof_r04.log | cut -c1-300
EXTRA="--config 3" tools/prof_bench.sh r04_cfg3 > gpurun_out/prof_r04_cfg3.log 2>&1; tail -8 gpurun_out/prof_r04_cfg3.log | cut -c1-300
