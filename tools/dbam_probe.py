#!/usr/bin/env python3
"""Device inflater against the host one on a synthetic BAM (run on the GPU box): python tools/dbam_probe.py [n_loci]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from strkit_amd import _lib
from strkit_amd.frontend.synth_large import make_dataset_large

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
d = make_dataset_large("/tmp/dbam_probe", n_loci=n_loci, depth=30, read_len=15000, seed=11, procs=16)
L = _lib.load()
comp = np.fromfile(d["paths"]["bam"], np.uint8)
t0 = time.perf_counter()
n = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, None, 0, 0)
ref = np.empty(int(n), np.uint8)
got = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, ref.ctypes.data, ref.size, 0)
t_host = time.perf_counter() - t0
print(f"compressed {comp.size / 1e6:.1f} MB, decompressed {got / 1e6:.1f} MB, host inflate {t_host:.3f} s = {got / t_host / 1e9:.2f} GB/s", flush=True)
h = C.c_void_p()
_lib.check(L.strk_dbam_open(0, C.byref(h)))
nxt = C.c_int64(0)
for it in range(3):
    t0 = time.perf_counter()
    tot = L.strk_dbam_inflate(h, comp.ctypes.data, comp.size, 0, 1 << 40, C.byref(nxt))
    t_dev = time.perf_counter() - t0
    if tot < 0:
        _lib.check(int(tot))
    print(f"device inflate (H2D of the compressed bytes + kernel + CRC): {t_dev:.3f} s = {tot / t_dev / 1e9:.2f} GB/s of output", flush=True)
assert tot == got and nxt.value == comp.size, (tot, got, nxt.value)
out = np.empty(int(tot), np.uint8)
_lib.check(L.strk_dbam_download(h, 0, int(tot), out.ctypes.data))
print("identical to the host inflater:", bool(np.array_equal(out, ref)))
bad = comp.copy()
bad[comp.size // 2] ^= 0x10
r = L.strk_dbam_inflate(h, bad.ctypes.data, bad.size, 0, 1 << 40, C.byref(nxt))
print("corrupted byte ->", r, L.strk_last_error().decode() if r < 0 else "(not detected)")
L.strk_dbam_close(h)
