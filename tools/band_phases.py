#!/usr/bin/env python3
"""Profiling aid: un-overlapped duration of k_dp_band on one 10 000-locus batch of the bench workload with parts of the
kernel switched off (STRKIT_AMD_DBG; the results of such runs are wrong, only the time is read).
Run on the GPU box: python tools/band_phases.py"""
import os
import subprocess
import sys

CODE = r'''
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
print("%.4f %d %d" % (sum(t) / len(t), st.n_band_reads, st.n_band_fallback))
'''
names = {0: "all", 4: "no in-kernel search", 5: "no search, no forward pass", 6: "no search, no backward pass",
         12: "no search, no fork rows", 15: "nothing but staging", 16: "all, items in arrival order", 20: "no search, arrival order"}
for dbg, name in names.items():
    env = dict(os.environ, STRKIT_AMD_DBG=str(dbg), STRKIT_AMD_NO_PIPE="1")
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"dbg={dbg:2d} {name:28s} k_dp_band ms, band reads, fallbacks: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
