#!/bin/bash
# Builds experiment variants of the library (cross-compiles here, no GPU needed): tools/exp_build.sh NAME "-DFLAG=1 ..." [NAME2 "..."]...
# -> strkit_amd/lib/exp/NAME.so (git-ignored, travels to the GPU box); tools/exp_band.py measures every variant found there.
set -e
cd "$(dirname "$0")/.."
mkdir -p strkit_amd/lib/exp
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-parameter $flags \
      -o strkit_amd/lib/exp/$name.so strkit_amd/csrc/strk_api.hip -lz -lpthread && echo "built $name ($flags)" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
