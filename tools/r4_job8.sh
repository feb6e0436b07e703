#!/bin/bash
# round-4 GPU job 8: config 5 (one GPU's share) against calls in flight; config 4 after the sort rule
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4h
for d in 2 3 4 6; do
  timeout -k 10 200 python bench.py --config 5 --pipeline $d --steps 24 --warmup 6 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4h/cfg5_d$d.json 2> gpurun_out/r4h/cfg5_d$d.err
  python3 -c "import json;j=json.load(open('gpurun_out/r4h/cfg5_d$d.json'));print('cfg5 eighth, calls in flight $d:', round(j['value']/1e6,3),'M reads/s', round(j['ms_per_step'],3),'ms/step', j['config']['window_by_motif_bucket'], j['parity_check'], 'miss', j['window_miss_reads_per_step'])"
done
timeout -k 10 120 python tools/cfg_probe.py 4 21250 14 2 > gpurun_out/r4h/cfg4.log 2>&1; tail -3 gpurun_out/r4h/cfg4.log | head -1
for d in 2 3; do
  timeout -k 10 200 python bench.py --config 4 --pipeline $d --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4h/cfg4_d$d.json 2> gpurun_out/r4h/cfg4_d$d.err
  python3 -c "import json;j=json.load(open('gpurun_out/r4h/cfg4_d$d.json'));print('cfg4 shard, calls in flight $d:', round(j['value']/1e6,3),'M reads/s', round(j['ms_per_step'],3),'ms/step', j['config']['window_by_motif_bucket'], j['parity_check'])"
done
