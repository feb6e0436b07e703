#!/bin/bash
# round-4 GPU job 5: microbenchmark 3 (with the scalar-VCC selects), config 5 with the select-free wide steps, chunk census
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4e
tools/valu_rate3 > gpurun_out/r4e/valu_rate3.txt 2>&1; head -9 gpurun_out/r4e/valu_rate3.txt
for n in 250 2000; do
  timeout -k 10 120 python tools/cfg_probe.py 5 $n 12 1 > gpurun_out/r4e/cfg5_$n.log 2>&1; tail -3 gpurun_out/r4e/cfg5_$n.log
  STRKIT_AMD_LIB=$PWD/strkit_amd/lib/exp/phase.so timeout -k 10 120 python tools/cfg_probe.py 5 $n 5 0 > gpurun_out/r4e/phase_cfg5_$n.log 2>&1; grep "phase" gpurun_out/r4e/phase_cfg5_$n.log | tail -2
done
timeout -k 10 120 python tools/cfg_probe.py 4 21250 14 2 > gpurun_out/r4e/cfg4.log 2>&1; tail -3 gpurun_out/r4e/cfg4.log
STRKIT_AMD_LIB=$PWD/strkit_amd/lib/exp/phase.so timeout -k 10 120 python tools/cfg_probe.py 2 10000 5 0 > gpurun_out/r4e/phase_cfg2.log 2>&1; grep "phase" gpurun_out/r4e/phase_cfg2.log | tail -2
