#!/bin/bash
# Where k_dp_band spends its time: builds a private copy of the library with -DSTRK_PHASE_TIMING and runs a few
# calls of the bench workload (run on the GPU box from the repo root).
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-parameter -DSTRK_PHASE_TIMING \
  -o gpurun_out/libstrkit_amd_phase.so strkit_amd/csrc/strk_api.hip -lz -lpthread
STRKIT_AMD_LIB=$PWD/gpurun_out/libstrkit_amd_phase.so python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from strkit_amd.synth import make_config, LocusBatch
from strkit_amd.batch import count_loci
b = LocusBatch.concat([make_config(2, seed_shift=k) for k in range(10)])   # one bench step: 10 000 loci
for _ in range(4):
    count_loci(b)
PY
