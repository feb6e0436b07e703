#!/bin/bash
# round-4 GPU job 9: the default bench (headline + sub-configs) and config 4 with two calls in flight after the grid rule
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4i
for d in 2 3; do
  timeout -k 10 200 python bench.py --config 4 --pipeline $d --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4i/cfg4_d$d.json 2> gpurun_out/r4i/cfg4_d$d.err
  python3 -c "import json;j=json.load(open('gpurun_out/r4i/cfg4_d$d.json'));print('cfg4 shard, calls in flight $d:', round(j['value']/1e6,3),'M reads/s', round(j['ms_per_step'],3),'ms/step', j['parity_check'])"
done
timeout -k 10 200 python bench.py --config 3 --pipeline 2 --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4i/cfg3_d2.json 2> gpurun_out/r4i/cfg3_d2.err
python3 -c "import json;j=json.load(open('gpurun_out/r4i/cfg3_d2.json'));print('cfg3, calls in flight 2:', round(j['value']/1e6,3),'M reads/s', round(j['ms_per_step'],3),'ms/step', j['parity_check'])"
timeout -k 10 500 python bench.py > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r4i/bench.json'))
print('headline', round(j['value']/1e6,2), 'M reads/s', round(j['ms_per_step'],4), 'ms/step', j['config']['window_by_motif_bucket'], j['parity_check'])
r=j['roofline']; print('roofline', {k:r.get(k) for k in ('kernel','kernel_ms','frac','frac_valu','floor_ms','insts_per_cell','frac_of_cell_floor','valu_from_profile')})
for k,v in j['configs'].items(): print(k, round(v['value']/1e6,3), 'M reads/s', round(v['ms_per_step'],3), 'ms/step; one at a time', round(v['one_call_at_a_time']['value']/1e6,3), v['window_by_motif_bucket'], v['parity_check'], {kk:round(vv,3) for kk,vv in v['roofline']['dp_kernels_ms'].items()})
print('h2d', {k:(round(v['value']/1e6,1) if isinstance(v,dict) and 'value' in v else None) for k,v in j['h2d_inclusive'].items() if isinstance(v,dict)}, round(j['h2d_inclusive']['value']/1e6,1))
print('e2e', j['e2e']['wall_s'], j['e2e']['host_front_end']['wall_s'], j['e2e']['front_ends_agree'])
print('cpu', j['cpu_baseline']['value'])
PY
