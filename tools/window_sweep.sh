#!/bin/bash
# miss rate and throughput of the default bench as a function of the speculative window half-width
for w in "$@"; do
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --window $w 2>&1 | tail -1 > /tmp/ws_$w.json
  python3 - $w <<'PY'
import sys, json
w = sys.argv[1]
j = json.load(open(f"/tmp/ws_{w}.json"))
print("W", w, round(j["value"] / 1e6, 1), "M reads/s", round(j["ms_per_step"], 3), "ms",
      {k: v for k, v in j.items() if "miss" in k or "generic" in k or "band" in k})
PY
done
