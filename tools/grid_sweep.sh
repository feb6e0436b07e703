#!/bin/bash
# pipelined bench with the default grids and with full-size DP grids (STRKIT_AMD_DP_BLOCKS), at two pipeline depths
mkdir -p gpurun_out/sweep
for p in 2 4; do
  for blk in 0 512; do
    STRKIT_AMD_DP_BLOCKS=$blk python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --pipeline $p > gpurun_out/sweep/b_${p}_${blk}.json 2>/dev/null
    python -c "
import json;d=json.load(open('gpurun_out/sweep/b_${p}_${blk}.json'));print('pipeline',$p,'blocks',$blk,d['value']/1e6,d['ms_per_step'])"
  done
done
