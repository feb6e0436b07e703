#!/bin/bash
# Stage times of opening the alignment file on the device (upload + inflation, header + index, record scan, record index),
# twice (second run: page cache warm for sure): bash tools/e2e_open.sh  [on the GPU box]
mkdir -p gpurun_out/open
python3 - <<'PY' > gpurun_out/open/open.txt 2>&1
import sys, time
sys.path.insert(0, ".")
from strkit_amd.frontend import DeviceBam
from strkit_amd.frontend.synth_large import make_dataset_large
d = make_dataset_large("/tmp/e2e_open", n_loci=10000, depth=30, read_len=15000, seed=11, procs=16)
for it in range(3):
    t = time.perf_counter()
    b = DeviceBam(d["paths"]["bam"])
    dt = time.perf_counter() - t
    print(it, round(dt, 4), b.open_stage_s, "kernels", round(b.kernel_s(), 4), "records", b.n_records, flush=True)
    b.close()
PY
cat gpurun_out/open/open.txt
