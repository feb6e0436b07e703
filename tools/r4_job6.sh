#!/bin/bash
# round-4 GPU job 6: wave priority by rows (config 5), exact kernel without scalar selects (config 3), pin-case tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
for n in 250 2000; do
  timeout -k 10 120 python tools/cfg_probe.py 5 $n 12 1 > gpurun_out/r4f/cfg5_$n.log 2>&1; tail -3 gpurun_out/r4f/cfg5_$n.log | head -1
  STRKIT_AMD_DBG=32 timeout -k 10 120 python tools/cfg_probe.py 5 $n 12 0 > gpurun_out/r4f/cfg5_${n}_noprio.log 2>&1; tail -2 gpurun_out/r4f/cfg5_${n}_noprio.log | head -1
done
timeout -k 10 120 python tools/cfg_probe.py 4 21250 14 2 > gpurun_out/r4f/cfg4.log 2>&1; tail -3 gpurun_out/r4f/cfg4.log | head -1
timeout -k 10 120 python tools/cfg_probe.py 3 10000 10 2 > gpurun_out/r4f/cfg3.log 2>&1; tail -3 gpurun_out/r4f/cfg3.log
timeout -k 10 300 python -m pytest tests/test_gpu_count.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4f/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4f/tests.log
