#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4p/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4p/tests.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4p/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r4p/smoke.log
timeout -k 10 200 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4p/dist.json 2> gpurun_out/r4p/dist.err; echo "dist rc $?"; cut -c1-200 gpurun_out/r4p/dist.json
timeout -k 10 300 python bench.py --strong --config 4 --loci 20000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4p/strong4.json 2> gpurun_out/r4p/strong4.err; echo "strong rc $?"; python3 -c "import json;j=json.load(open('gpurun_out/r4p/strong4.json'));print(round(j['value']/1e6,2), j.get('strong_scaling_check'), j['parity_check'])"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r4p/bench_driver_form.json 2> gpurun_out/r4p/bench_driver_form.err; echo "driver-form bench rc $?"; python3 -c "import json;j=json.load(open('gpurun_out/r4p/bench_driver_form.json'));print(round(j['value']/1e6,2), j['ms_per_step'], j['parity_check'], {k:round(v['value']/1e6,2) for k,v in j['configs'].items()})"
