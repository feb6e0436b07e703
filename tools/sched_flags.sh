#!/bin/bash
# Un-overlapped k_dp_band time of private builds with other instruction-scheduling strategies of the AMDGPU back end
# (run on the GPU box from the repo root): tools/sched_flags.sh "max-ilp max-memory-clause iterative-ilp"
mkdir -p gpurun_out/sched
CODE='
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from strkit_amd import _lib
from strkit_amd.batch import batch_struct, make_params
from strkit_amd.synth import make_config, LocusBatch
b = LocusBatch.concat([make_config(2, seed_shift=k) for k in range(4)])
L = _lib.load(); ctx = _lib.default_context(0)
s, keep = batch_struct(b); p = make_params(window=8); st = _lib.StrkStats()
outs = [np.zeros(b.n_reads, np.int32) for _ in range(4)]
t = []
for i in range(6):
    L.strk_count_loci(ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st))
    if i >= 2: t.append(st.band_kernel_ms)
print("%.4f ms, band reads %d, cn checksum %d" % (sum(t) / len(t), st.n_band_reads, int(outs[0].astype(np.int64).sum())))
'
echo -n "default build: "; python3 -c "$CODE"
for s in ${1:-max-ilp max-memory-clause}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-parameter -mllvm -amdgpu-sched-strategy=$s \
    -o gpurun_out/sched/lib_$s.so strkit_amd/csrc/strk_api.hip -lz -lpthread > gpurun_out/sched/build_$s.log 2>&1 || { echo "$s: build failed"; continue; }
  echo -n "$s: "; STRKIT_AMD_LIB=$PWD/gpurun_out/sched/lib_$s.so python3 -c "$CODE"
done
