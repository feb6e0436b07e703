#!/bin/bash
# pipelined bench with the default grids and with larger DP grids (STRKIT_AMD_DP_BLOCKS), at three pipeline depths
mkdir -p gpurun_out/sweep
for p in 2 3 4; do
  for blk in 0 448 512; do
    STRKIT_AMD_DP_BLOCKS=$blk python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-extras --no-e2e --pipeline $p > gpurun_out/sweep/b_${p}_${blk}.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/sweep/b_${p}_${blk}.json').read().strip().split(chr(10))[-1]);print('pipeline',$p,'blocks',$blk,round(d['value']/1e6,1),round(d['ms_per_step'],3))"
  done
done
