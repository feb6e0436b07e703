#!/usr/bin/env python3
"""Diagnostic (GPU box): strk_count_loci on host buffers, the pipelined entry point, with its per-phase host times.
usage: [STRKIT_AMD_PIPE_MB=12] [STRKIT_AMD_COPY_THREADS=7] python tools/pipe_probe.py"""
import ctypes as C
import os
import sys
import time

os.environ["STRKIT_AMD_PIPE_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strkit_amd import _lib
from strkit_amd.batch import batch_struct, make_params
from strkit_amd.synth import LocusBatch, make_config

batches = [LocusBatch.concat([make_config(2, seed_shift=b * 1024 + j) for j in range(10)]) for b in range(3)]
L = _lib.load()
ctx = _lib.default_context(0)
p = make_params()
st = _lib.StrkStats()
hb = [batch_struct(b) for b in batches]
outs = [np.zeros(batches[0].n_reads + 1000, np.int32) for _ in range(4)]
print("cpus", len(os.sched_getaffinity(0)), "bytes per call", batches[0].seqs.nbytes + batches[0].n_reads * 24, flush=True)
for w in range(10):
    t = time.perf_counter()
    s, _k = hb[w % 3]
    _lib.check(L.strk_count_loci(ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
    dt = time.perf_counter() - t
    print(f"call {w}: {dt * 1e3:.3f} ms  ({batches[w % 3].n_reads / dt / 1e6:.1f} M reads/s)  sub-batches {st.n_dp_launches}  device ms (summed) {st.kernel_ms:.3f}", flush=True)
