#!/bin/bash
# round-4 GPU job 7: longest-first order of the wide classes (k_sort_wide) + priorities: parity, config 5 / 4 timing, chunk census
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4g
timeout -k 10 200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4g/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4g/tests.log
for n in 250 2000; do
  timeout -k 10 120 python tools/cfg_probe.py 5 $n 12 2 > gpurun_out/r4g/cfg5_$n.log 2>&1; tail -3 gpurun_out/r4g/cfg5_$n.log
  STRKIT_AMD_LIB=$PWD/strkit_amd/lib/exp/phase.so timeout -k 10 120 python tools/cfg_probe.py 5 $n 5 0 > gpurun_out/r4g/phase_cfg5_$n.log 2>&1; grep "phase" gpurun_out/r4g/phase_cfg5_$n.log | tail -2
done
timeout -k 10 120 python tools/cfg_probe.py 4 21250 14 2 > gpurun_out/r4g/cfg4.log 2>&1; tail -3 gpurun_out/r4g/cfg4.log
timeout -k 10 120 python tools/cfg_probe.py 2 10000 14 2 > gpurun_out/r4g/cfg2.log 2>&1; tail -3 gpurun_out/r4g/cfg2.log
timeout -k 10 120 python tools/cfg_probe.py 3 10000 10 2 > gpurun_out/r4g/cfg3.log 2>&1; tail -3 gpurun_out/r4g/cfg3.log | head -1
