#!/bin/bash
# round-4 GPU job 18: scalar path with the lowest-latency class; final default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4o
timeout -k 10 300 python -m pytest tests/test_gpu_count.py -x -q -s -k "scalar" > gpurun_out/r4o/tests.log 2>&1; echo "tests rc $?"; grep -a "scalar drop-in\|passed\|failed\|Error" gpurun_out/r4o/tests.log | tail -5
timeout -k 10 500 python bench.py > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r4o/bench.json'))
print('headline', round(j['value']/1e6,2), 'M reads/s', round(j['ms_per_step'],4), 'ms/step', j['parity_check'])
r=j['roofline']; print('roofline', {k:r.get(k) for k in ('kernel_ms','frac','traffic','frac_valu','cycles_per_inst','floor_ms','insts_per_cell','frac_of_cell_floor','valu_from_profile')})
for k,v in j['configs'].items():
    rr=v['roofline']
    print(k, round(v['value']/1e6,3), 'M reads/s', round(v['ms_per_step'],3), 'ms/step; one at a time', round(v['one_call_at_a_time']['value']/1e6,3), v['parity_check'], 'frac_valu', rr.get('frac_valu'), 'cyc', rr.get('cycles_per_inst'), 'ipc', rr.get('insts_per_cell'), 'traffic', rr.get('traffic'), 'from', rr.get('valu_from_profile'), 'others', {kk:(vv.get('frac_valu'),vv.get('insts_per_cell')) for kk,vv in (rr.get('other_dp_kernels') or {}).items()})
PY
