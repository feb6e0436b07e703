#!/bin/bash
# round-4 GPU job 16: the whole GPU test suite, smoke, and the default bench (the round's final line)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4m
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r4m/tests.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4m/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r4m/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r4m/bench.json'))
print('headline', round(j['value']/1e6,2), 'M reads/s', round(j['ms_per_step'],4), 'ms/step', j['config']['window_by_motif_bucket'], j['parity_check'])
r=j['roofline']; print('roofline', {k:r.get(k) for k in ('kernel','kernel_ms','achieved','frac','traffic','frac_valu','floor_ms','insts_per_cell','frac_of_cell_floor','valu_from_profile','profile_kernel_ms')})
for k,v in j['configs'].items():
    rr=v['roofline']
    print(k, round(v['value']/1e6,3), 'M reads/s', round(v['ms_per_step'],3), 'ms/step; one at a time', round(v['one_call_at_a_time']['value']/1e6,3), v['window_by_motif_bucket'], v['parity_check'], 'miss', v['window_miss_reads_per_step'], {kk:round(vv,3) for kk,vv in rr['dp_kernels_ms'].items()}, 'frac_valu', rr.get('frac_valu'), 'ipc', rr.get('insts_per_cell'), 'traffic', rr.get('traffic'), 'others', list((rr.get('other_dp_kernels') or {}).keys()))
print('h2d', {k:(round(v['value']/1e6,1) if isinstance(v,dict) and 'value' in v else None) for k,v in j['h2d_inclusive'].items() if isinstance(v,dict)}, round(j['h2d_inclusive']['value']/1e6,1))
print('e2e', j['e2e']['wall_s'], j['e2e']['host_front_end']['wall_s'], j['e2e']['front_ends_agree'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
PY
