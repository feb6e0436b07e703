#!/bin/bash
# Round 3: two calls in flight; band / exact grids of 448..512 blocks (STRKIT_AMD_DP_BLOCKS) with the product library and with
# the k_replay variant held to 64 VGPRs (tools/exp_build.sh replay8 "-DSTRK_REPLAY_WAVES=8"), which fits beside two band waves.
mkdir -p gpurun_out/sweep2
for lib in product replay8; do
  path=strkit_amd/lib/libstrkit_amd.so; [ $lib != product ] && path=strkit_amd/lib/exp/$lib.so
  [ -f $path ] || continue
  for blk in 0 480 496 512; do
    for p in 2 3; do
      STRKIT_AMD_LIB=$PWD/$path STRKIT_AMD_DP_BLOCKS=$blk python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-e2e --no-configs --pipeline $p > gpurun_out/sweep2/b_${lib}_${p}_${blk}.json 2>/dev/null
      python -c "
import json;d=json.loads(open('gpurun_out/sweep2/b_${lib}_${p}_${blk}.json').read().strip().split(chr(10))[-1]);print('$lib','pipeline',$p,'blocks',$blk,round(d['value']/1e6,1),round(d['ms_per_step'],3),d.get('parity_check'))" | tee -a gpurun_out/sweep2/sweep.txt
    done
  done
done
