#!/bin/bash
# round-4 GPU job 10: the default bench after the deterministic grid rule and the +-8 floor for 1-2-base motifs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4j
timeout -k 10 500 python bench.py > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r4j/bench.json'))
print('headline', round(j['value']/1e6,2), 'M reads/s', round(j['ms_per_step'],4), 'ms/step', j['config']['window_by_motif_bucket'], j['parity_check'])
r=j['roofline']; print('roofline', {k:r.get(k) for k in ('kernel','kernel_ms','frac','frac_valu','floor_ms','insts_per_cell','frac_of_cell_floor','valu_from_profile')})
for k,v in j['configs'].items(): print(k, round(v['value']/1e6,3), 'M reads/s', round(v['ms_per_step'],3), 'ms/step; one at a time', round(v['one_call_at_a_time']['value']/1e6,3), v['window_by_motif_bucket'], v['parity_check'], 'miss', v['window_miss_reads_per_step'], {kk:round(vv,3) for kk,vv in v['roofline']['dp_kernels_ms'].items()})
print('h2d', {k:(round(v['value']/1e6,1) if isinstance(v,dict) and 'value' in v else None) for k,v in j['h2d_inclusive'].items() if isinstance(v,dict)}, round(j['h2d_inclusive']['value']/1e6,1))
print('e2e', j['e2e']['wall_s'], j['e2e']['host_front_end']['wall_s'], j['e2e']['front_ends_agree'])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_count.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4j/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4j/tests.log
