#!/usr/bin/env python3
"""Profiling aid: N un-pipelined strk_count_loci calls on one 10 000-locus batch of the bench workload (run under rocprofv3 by
tools/band_insts.sh; STRKIT_AMD_DBG switches parts of k_dp_band off — results are then wrong, only counters are read)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STRKIT_AMD_NO_PIPE", "1")
from strkit_amd import _lib  # noqa: E402
from strkit_amd.batch import batch_struct, make_params  # noqa: E402
from strkit_amd.synth import LocusBatch, make_config  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
window = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = LocusBatch.concat([make_config(2, seed_shift=k) for k in range(10)])
L = _lib.load()
ctx = _lib.default_context(0)
s, keep = batch_struct(b)
p = make_params(window=window)
st = _lib.StrkStats()
outs = [np.zeros(b.n_reads, np.int32) for _ in range(4)]
for _ in range(n_calls):
    _lib.check(L.strk_count_loci(ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
print("band %.4f ms, band reads %d, fallbacks %d, cells %.4g" % (st.band_kernel_ms, st.n_band_reads, st.n_band_fallback, st.dp_cells))
