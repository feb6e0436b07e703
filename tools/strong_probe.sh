#!/bin/bash
# one-rank runs of bench.py --strong for the 8-GPU configurations, with the fields that explain their cost
mkdir -p gpurun_out/strong
for args in "--config 4 --loci 20000" "--config 4" "--config 5"; do
  tag=$(echo $args | tr -d ' -')
  python bench.py --strong $args --force-dist --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/strong/$tag.json 2> gpurun_out/strong/$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/strong/$tag.json").read().strip().splitlines()[-1])
print("$args", {k: d[k] for k in ("value", "ms_per_step", "device_ms_per_step", "band_reads_per_step", "band_fallback_per_step", "window_miss_reads_per_step",
      "generic_kernel_items_per_step", "dedup_reads_per_step", "strong_scaling_check", "parity_check")}, d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["valu"])
PY
done
