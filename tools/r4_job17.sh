#!/bin/bash
# round-4 GPU job 17: scalar fast path (parity + latency), PMC passes of config 5 at one GPU's share
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4n
timeout -k 10 300 python -m pytest tests/test_gpu_count.py -x -q -s -k "scalar or pin" > gpurun_out/r4n/tests.log 2>&1; echo "tests rc $?"; grep -a "scalar drop-in\|passed\|failed\|Error" gpurun_out/r4n/tests.log | tail -5
STRKIT_AMD_NO_SCALAR_FAST=1 timeout -k 10 200 python -m pytest tests/test_gpu_count.py -x -q -s -k "scalar_fast" > gpurun_out/r4n/tests_slow.log 2>&1; grep -a "scalar drop-in\|passed\|failed" gpurun_out/r4n/tests_slow.log | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_ref.py tests/test_gpu_frontend.py -x -q > gpurun_out/r4n/tests2.log 2>&1; echo "tests2 rc $?"; tail -2 gpurun_out/r4n/tests2.log
EXTRA="--config 5" STEPS=4 PRIME=10 COUNTERS="SQ_INSTS_VALU GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE" tools/prof_pmc_single.sh r04_cfg5e > gpurun_out/prof_r04_cfg5e.log 2>&1; tail -8 gpurun_out/prof_r04_cfg5e.log | cut -c1-300
