#!/bin/bash
# rocprofv3 kernel trace + stats of the end-to-end sub-result alone (file front end on the device + calling): which kernels
# the 0.4 s contain.  usage: tools/prof_e2e.sh <tag>   -> gpurun_out/prof_e2e_<tag>/   (from the repo root on the GPU box)
set -e
TAG=${1:-e2e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_e2e_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- python3 tools/e2e_wall_profile.py 10000 device > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(ls $OUT/trace/*kernel_stats.csv | head -1)
cp "$f" $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-200
grep "plain wall" $OUT/trace.log | cut -c1-400
