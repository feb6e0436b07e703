#!/bin/bash
# round-4 GPU job 2: band placement sweep (span_w, slack_m8) on configs 2 and 4, VALU microbenchmark 3, config 5 timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b
tools/valu_rate3 > gpurun_out/r4b/valu_rate3.txt 2>&1; cat gpurun_out/r4b/valu_rate3.txt
for t in 64,0 4,0 5,0 4,16 4,24 4,28 5,24 3,28; do
  sw=${t%,*}; sm=${t#*,}
  echo "== span_w $sw slack_m8 $sm"
  STRKIT_AMD_SPAN_W=$sw STRKIT_AMD_SLACK_M8=$sm timeout -k 10 120 python tools/cfg_probe.py 2 10000 14 2 > gpurun_out/r4b/cfg2_s$t.log 2>&1; tail -3 gpurun_out/r4b/cfg2_s$t.log | head -2
  STRKIT_AMD_SPAN_W=$sw STRKIT_AMD_SLACK_M8=$sm timeout -k 10 120 python tools/cfg_probe.py 4 21250 14 2 > gpurun_out/r4b/cfg4_s$t.log 2>&1; tail -3 gpurun_out/r4b/cfg4_s$t.log | head -2
done
