#!/bin/bash
# round-4 GPU job 11: timeline of config 4's shard with two calls in flight (why does the overlap cost 10 %?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4k
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4k/trace -o b -- python3 bench.py --config 4 --pipeline 2 --steps 8 --warmup 2 --prime 12 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/r4k/trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/r4k/trace 6 > gpurun_out/r4k/timeline.txt 2>&1; head -70 gpurun_out/r4k/timeline.txt
rm -rf gpurun_out/r4k/trace
