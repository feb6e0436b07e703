#!/usr/bin/env python3
"""One configuration, resident in HBM, one call at a time: per-kernel un-overlapped durations, cells per kernel, band reads /
fall-backs / window misses and the windows per motif-length bucket — and the first loci against the oracle (checker).

    python tools/cfg_probe.py <config> [n_loci] [n_calls] [check_loci]
    STRKIT_AMD_WINDOW_B="6,6,6,4,4" python tools/cfg_probe.py 4 21250        # pinned windows per motif-length bucket
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from strkit_amd.synth import LocusBatch, make_config  # noqa: E402

cfg = int(sys.argv[1])
n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else {2: 10000, 3: 10000, 4: 21250, 5: 250}[cfg]
n_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 30
n_check = int(sys.argv[4]) if len(sys.argv) > 4 else 6
parts = [make_config(cfg, n_loci=min(1000, n_loci - k), seed_shift=7 * 1024 + j) for j, k in enumerate(range(0, n_loci, 1000))]
b = LocusBatch.concat(parts)

import torch  # noqa: E402
from strkit_amd import _lib  # noqa: E402
from strkit_amd.batch import make_params  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.load()
t = {k: torch.from_numpy(getattr(b, k)).to(dev) for k in ("seqs", "seq_off", "nfl", "ntr", "nfr", "est_cn", "read_off", "motifs", "motif_off")}
sb = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})
out = torch.zeros((4, b.n_reads), dtype=torch.int32, device=dev)
ctx = _lib.Context(0)
p = make_params()
st = _lib.StrkStats()
keys = ("head_ms", "band_kernel_ms", "band_wide_kernel_ms", "dp_kernel_ms", "long_kernel_ms", "replay_ms", "kernel_ms")
hist = []
for i in range(n_calls):
    t0 = time.perf_counter()
    _lib.check(L.strk_count_loci_device(ctx.handle, C.byref(sb), C.byref(p), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                        out[3].data_ptr(), None, C.byref(st)))
    wall = (time.perf_counter() - t0) * 1e3
    d = st.as_dict()
    hist.append(d)
    if i < 3 or i >= n_calls - 3 or (i and hist[-2]["window_bucket"] != d["window_bucket"]):
        print(f"call {i:3d} wall {wall:7.3f} ms | " + " ".join(f"{k[:-3]} {d[k]:.3f}" for k in keys) +
              f" | win {d['window_bucket']} band {d['n_band_reads']} fb {d['n_band_fallback']} miss {d['n_miss_reads']}/{d['n_miss_rounds']} "
              f"dedup {d['n_dedup_reads']} long {d['n_long_reads']} generic {d['n_fallback']}", flush=True)
last = hist[-4:]
avg = {k: sum(h[k] for h in last) / len(last) for k in keys}
d = hist[-1]
print(f"cfg {cfg}: {b.n_loci} loci, {b.n_reads} reads; steady (last 4 calls): " + " ".join(f"{k[:-3]} {v:.3f}" for k, v in avg.items()) +
      f" -> {b.n_reads / avg['kernel_ms'] / 1e3:.2f} M reads/s of device time")
print("cells: band %.3g wide %.3g exact %.3g long %.3g total %.3g" % (d["band_cells"], d["wide_cells"], d["exact_cells"], d["long_cells"], d["dp_cells"]))
if n_check:
    import oracle
    oracle.build()
    got = out.cpu().numpy()
    bad = 0
    for l in range(min(n_check, b.n_loci)):
        r0, r1 = int(b.read_off[l]), int(b.read_off[l + 1])
        s0 = int(b.seq_off[r0])
        o = oracle.count_locus(b.seqs[s0:int(b.seq_off[r1])], b.seq_off[r0:r1 + 1] - s0, b.nfl[r0:r1], b.ntr[r0:r1], b.nfr[r0:r1],
                               b.est_cn[r0:r1], b.motif(l))
        for i, k in enumerate(("cn", "score", "n_iters", "start")):
            if not np.array_equal(got[i, r0:r1], o[k]):
                bad += 1
                print("MISMATCH locus", l, k)
    print(f"oracle check on {min(n_check, b.n_loci)} loci: {'ok' if not bad else 'MISMATCH'}")
ctx.close()
