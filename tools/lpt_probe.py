#!/usr/bin/env python3
"""Probe (GPU box): would the wide band kernel gain from taking its items longest first?  Config 5's loci are fed in arrival
order, sorted by tract length descending (the class lists then hold the long reads first: k_plan appends block by block) and
ascending; the un-overlapped duration of k_dp_band_wide is what is read.  usage: python tools/lpt_probe.py [config] [n_loci]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STRKIT_AMD_NO_PIPE", "1")
from strkit_amd import _lib  # noqa: E402
from strkit_amd.batch import batch_struct, make_params  # noqa: E402
from strkit_amd.synth import LocusBatch, make_config  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else 250
b = make_config(cfg, n_loci=n_loci, seed_shift=0)
mean_tr = np.array([b.ntr[b.read_off[l]:b.read_off[l + 1]].mean() for l in range(b.n_loci)])
L = _lib.load()
ctx = _lib.default_context(0)
p = make_params()
for name, order in (("arrival", np.arange(b.n_loci)), ("longest first", np.argsort(-mean_tr)), ("shortest first", np.argsort(mean_tr))):
    bb = LocusBatch.concat([b.locus_slice(int(l), int(l) + 1) for l in order])
    s, keep = batch_struct(bb)
    st = _lib.StrkStats()
    outs = [np.zeros(bb.n_reads, np.int32) for _ in range(4)]
    tw, tb = [], []
    for i in range(16):
        _lib.check(L.strk_count_loci(ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
        if i >= 12:
            tw.append(st.band_wide_kernel_ms); tb.append(st.band_kernel_ms)
    print(f"{name:15s} k_dp_band_wide {sum(tw) / len(tw):.3f} ms  k_dp_band {sum(tb) / len(tb):.3f} ms  window {st.window_used}  cells {st.dp_cells:.3g}", flush=True)
