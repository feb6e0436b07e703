#!/bin/bash
# round-4 GPU job 1: GPU tests, then config 4 / 5 probes with pinned windows per motif-length bucket
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4a
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/r4a/tests.log
tail -3 gpurun_out/r4a/tests.log
timeout -k 10 200 python tools/cfg_probe.py 4 21250 36 4 > gpurun_out/r4a/cfg4_adaptive.log 2>&1; tail -4 gpurun_out/r4a/cfg4_adaptive.log
for w in 8,8,8,8,8 6,6,6,6,6 6,6,6,5,5 6,6,6,4,4 6,6,5,4,4 6,5,5,4,4; do
  STRKIT_AMD_WINDOW_B=$w timeout -k 10 120 python tools/cfg_probe.py 4 21250 8 2 > gpurun_out/r4a/cfg4_w$w.log 2>&1; echo "== $w"; tail -3 gpurun_out/r4a/cfg4_w$w.log
done
timeout -k 10 120 python tools/cfg_probe.py 5 250 12 1 > gpurun_out/r4a/cfg5_250.log 2>&1; tail -3 gpurun_out/r4a/cfg5_250.log
timeout -k 10 120 python tools/cfg_probe.py 5 2000 8 1 > gpurun_out/r4a/cfg5_2000.log 2>&1; tail -3 gpurun_out/r4a/cfg5_2000.log
timeout -k 10 120 python tools/cfg_probe.py 2 10000 16 4 > gpurun_out/r4a/cfg2.log 2>&1; tail -3 gpurun_out/r4a/cfg2.log
