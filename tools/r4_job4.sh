#!/bin/bash
# round-4 GPU job 4: shared backward pass (k_dp_bwd): parity tests, then configs 2 / 4 / 5 one call at a time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4d
timeout -k 10 300 python -m pytest tests/test_gpu_count.py tests/test_gpu_configs.py tests/test_gpu_score_table.py -x -q > gpurun_out/r4d/tests1.log 2>&1; echo "tests1 rc $?"; tail -15 gpurun_out/r4d/tests1.log
for c in "2 10000" "4 21250" "5 250"; do
  set -- $c
  timeout -k 10 150 python tools/cfg_probe.py $1 $2 16 6 > gpurun_out/r4d/cfg$1.log 2>&1; tail -4 gpurun_out/r4d/cfg$1.log
done
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r4d/tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r4d/tests.log
