#!/bin/bash
# round-4 GPU job 15: config 4's shard with two calls in flight against the grid fractions of the three persistent kernels; fuzz soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4l
for g in 16,16,16 15,15,15 15,16,16 16,15,16 16,16,15 15,15,16 14,14,16; do
  STRKIT_AMD_GRID16=$g timeout -k 10 200 python bench.py --config 4 --pipeline 2 --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4l/cfg4_g$g.json 2> gpurun_out/r4l/cfg4_g$g.err
  python3 -c "import json;j=json.load(open('gpurun_out/r4l/cfg4_g$g.json'));print('cfg4 shard, grids $g/16:', round(j['value']/1e6,2),'M reads/s', round(j['ms_per_step'],3),'ms/step', j['parity_check'])"
done
STRK_FUZZ_SECONDS=240 timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/r4l/fuzz.log 2>&1; echo "fuzz rc $?"; tail -3 gpurun_out/r4l/fuzz.log
