#!/bin/bash
# default bench under different persistent-grid sizes (STRKIT_AMD_DP_BLOCKS) and pipeline depths
for cfg in "$@"; do
  blocks=${cfg%%:*}; depth=${cfg##*:}
  STRKIT_AMD_DP_BLOCKS=$blocks python bench.py --steps 200 --warmup 24 --no-cpu-baseline --pipeline $depth 2>&1 | tail -1 > /tmp/gs.json
  python3 - "$blocks" "$depth" <<'PY'
import sys, json
j = json.load(open("/tmp/gs.json"))
print("blocks", sys.argv[1], "depth", sys.argv[2], round(j["value"] / 1e6, 1), "M reads/s", round(j["ms_per_step"], 3), "ms")
PY
done
