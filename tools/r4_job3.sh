#!/bin/bash
# round-4 GPU job 3: wave-busy census of config 5 (phase-timing build), the default bench, config 5 PMC passes (one counter each)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4c
for n in 250 2000; do
  STRKIT_AMD_LIB=$PWD/strkit_amd/lib/exp/phase.so timeout -k 10 120 python tools/cfg_probe.py 5 $n 6 0 > gpurun_out/r4c/phase_cfg5_$n.log 2>&1; tail -4 gpurun_out/r4c/phase_cfg5_$n.log
done
timeout -k 10 400 python bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; echo "bench rc $?"; cut -c1-600 gpurun_out/r4c/bench.json
EXTRA="--config 5 --loci 2000" STEPS=3 PRIME=3 tools/prof_pmc_single.sh r04_cfg5
