#!/usr/bin/env python3
"""k_dp_long by itself: BASELINE config 5's shape (expansions to 12 kb) with the banded first pass switched off, resident in
HBM, one call at a time — durations, cells and G cell updates/s of the long-read kernel, and the first loci against the oracle.

    python tools/long_probe.py [n_loci] [n_calls] [check_loci] [sub indel]      # e.g. 0.03 0.04: ONT-like noise
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from strkit_amd.synth import make_config  # noqa: E402

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
over = {}
if len(sys.argv) > 5:
    over = dict(sub=float(sys.argv[4]), indel=float(sys.argv[5]))
b = make_config(5, n_loci=n_loci, seed_shift=3, **over)

import torch  # noqa: E402
from strkit_amd import _lib  # noqa: E402
from strkit_amd.batch import make_params  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.load()
t = {k: torch.from_numpy(getattr(b, k)).to(dev) for k in ("seqs", "seq_off", "nfl", "ntr", "nfr", "est_cn", "read_off", "motifs", "motif_off")}
sb = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})
out = torch.zeros((4, b.n_reads), dtype=torch.int32, device=dev)
ctx = _lib.Context(0)
band_on = os.environ.get("LONG_PROBE_BAND", "0") == "1"      # LONG_PROBE_BAND=1: the banded first pass stays on
p = make_params(band=band_on)
st = _lib.StrkStats()
import time  # noqa: E402
for i in range(n_calls):
    t0 = time.perf_counter()
    _lib.check(L.strk_count_loci_device(ctx.handle, C.byref(sb), C.byref(p), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                        out[3].data_ptr(), None, C.byref(st)))
    d = st.as_dict()
    d["wall"] = (time.perf_counter() - t0) * 1e3
    print(f"call {i}: wall {d['wall']:.1f} ms, miss rounds {d['n_miss_rounds']}, window {d['window_bucket']} | kernel {d['kernel_ms']:.2f} ms | exact {d['dp_kernel_ms']:.2f} ms ({d['exact_cells']:.3g} cells) | long {d['long_kernel_ms']:.2f} ms "
          f"({d['long_cells']:.3g} cells, {d['n_long_reads']} items) -> long {d['long_cells'] / max(d['long_kernel_ms'], 1e-9) / 1e6:.0f} G cells/s, "
          f"exact {d['exact_cells'] / max(d['dp_kernel_ms'], 1e-9) / 1e6:.0f} G cells/s | miss {d['n_miss_reads']} dedup {d['n_dedup_reads']}", flush=True)
if n_check:
    from helpers import oracle_count
    import oracle
    oracle.build()
    oracle.set_simd(True)
    part = b.locus_slice(0, n_check)
    exp = oracle_count(part)
    got = out.cpu().numpy()
    ok = all(np.array_equal(got[j, :part.n_reads], exp[k]) for j, k in enumerate(("cn", "score", "n_iters", "start")))
    print(f"oracle check on {n_check} loci ({part.n_reads} reads): {'ok' if ok else 'MISMATCH'}")
ctx.close()
