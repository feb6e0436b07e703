#!/usr/bin/env python3
"""Kernel time of the device inflater against the number of blocks in the launch (is a lane's time per block constant, i.e. is
the kernel bound by latency at three waves per CU?): python tools/inflate_rounds.py   (on the GPU box)"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from strkit_amd import _lib
from strkit_amd.frontend.synth_large import make_dataset_large

d = make_dataset_large("/tmp/inflate_rounds", n_loci=10000, depth=30, read_len=15000, seed=11, procs=16)
L = _lib.load()
comp = np.fromfile(d["paths"]["bam"], np.uint8)
h = C.c_void_p()
_lib.check(L.strk_dbam_open(0, C.byref(h)))
nxt = C.c_int64(0)
L.strk_dbam_inflate(h, comp.ctypes.data, comp.size, 0, 1 << 40, C.byref(nxt))
for wgs in (64, 256, 512, 768, 1024, 1536, 1647, 100000):
    k0 = L.strk_dbam_kernel_ms(h)
    tot = L.strk_dbam_inflate(h, comp.ctypes.data, comp.size, 0, wgs * 64 * 65280, C.byref(nxt))
    ms = L.strk_dbam_kernel_ms(h) - k0
    print(f"{wgs:7d} workgroups' worth: {tot / 1e6:9.1f} MB out, kernel {ms:7.2f} ms, {tot / ms / 1e6:6.1f} GB/s", flush=True)
L.strk_dbam_close(h)
