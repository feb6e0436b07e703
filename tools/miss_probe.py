#!/usr/bin/env python3
"""Diagnostic (GPU box): the bench loop (several calls in flight, eight rotating batches of the config-2 workload) with per-call
miss statistics and the host time spent inside strk_finish.  usage: STRKIT_AMD_LIB=... python tools/miss_probe.py [window]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from strkit_amd import _lib
from strkit_amd.batch import make_params
from strkit_amd.synth import LocusBatch, make_config

window = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
batches = [LocusBatch.concat([make_config(2, seed_shift=b * 1024 + j) for j in range(10)]) for b in range(8)]
L = _lib.load()


def resident(b):
    t = {k: torch.from_numpy(getattr(b, k)).to(dev) for k in ("seqs", "seq_off", "nfl", "ntr", "nfr", "est_cn", "read_off", "motifs", "motif_off")}
    return t, _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})


res = [resident(b) for b in batches]
D = 3
ctxs = [_lib.Context(0) for _ in range(D)]
streams = [torch.cuda.Stream(dev) for _ in range(D)]
outs = [torch.zeros((4, batches[0].n_reads + 1000), dtype=torch.int32, device=dev) for _ in range(D)]
p = make_params(window=window)
st = _lib.StrkStats()


def submit(i):
    k = i % D
    o = outs[k]
    _lib.check(L.strk_submit_loci_device(ctxs[k].handle, C.byref(res[(i + i // D) % 8][1]), C.byref(p), o[0].data_ptr(), o[1].data_ptr(),
                                         o[2].data_ptr(), o[3].data_ptr(), C.c_void_p(streams[k].cuda_stream)))


N = 64
for i in range(D):
    submit(i)
rows = []
t0 = time.perf_counter()
for i in range(N):
    if i == 24:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    t = time.perf_counter()
    _lib.check(L.strk_finish(ctxs[i % D].handle, C.byref(st)))
    rows.append((time.perf_counter() - t, st.n_miss_reads, st.n_miss_rounds, st.n_band_fallback, st.window_used, st.kernel_ms))
    if i + D < N:
        submit(i + D)
torch.cuda.synchronize()
el = time.perf_counter() - t0
r = np.array(rows[24:])
print(f"lib {os.environ.get('STRKIT_AMD_LIB', 'product')}: {el / (N - 24) * 1e3:.3f} ms/step over {N - 24} steps; finish() host ms mean {r[:, 0].mean() * 1e3:.3f} "
      f"max {r[:, 0].max() * 1e3:.3f}; calls with misses {int((r[:, 1] > 0).sum())}/{len(r)}; miss reads/call {r[:, 1].mean():.2f}; rounds/call {r[:, 2].mean():.2f}; "
      f"fallback/call {r[:, 3].mean():.2f}; windows {sorted(set(int(x) for x in r[:, 4]))}; device ms/call {r[:, 5].mean():.3f}")
with_m, without = r[r[:, 1] > 0], r[r[:, 1] == 0]
if len(with_m) and len(without):
    print(f"   finish() host ms: calls with misses {with_m[:, 0].mean() * 1e3:.3f}, without {without[:, 0].mean() * 1e3:.3f}")
