#!/usr/bin/env python3
"""CPU probe (oracle = checker; nothing of the product path): how wide must the speculative candidate window be, per motif length?

For every read of a sample of BASELINE config `cfg`, the oracle's per-locus protocol gives (cn, n_iters, start).  The search from
`start` scores [start - 4, start + 4] when it converges at once (n_iters = 9) and one more size per chase step, so the half-width
the table needs around the ESTIMATE is at most |start - est| + 4 + (n_iters - 9).  Prints, per motif-length bucket, the share of
LOCI with at least one read that needs more than W for W = 4, 5, 6, 8.
    python tools/window_need.py [cfg] [n_loci]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from strkit_amd.synth import make_config

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
oracle.build()
oracle.set_simd(True)
b = make_config(cfg, n_loci=n_loci)
need_by_m = {}
for l in range(b.n_loci):
    r0, r1 = int(b.read_off[l]), int(b.read_off[l + 1])
    s0 = int(b.seq_off[r0])
    o = oracle.count_locus(b.seqs[s0:int(b.seq_off[r1])], b.seq_off[r0:r1 + 1] - s0, b.nfl[r0:r1], b.ntr[r0:r1], b.nfr[r0:r1],
                           b.est_cn[r0:r1], b.motif(l), memo=True)
    est = b.est_cn[r0:r1].astype(np.int64)
    need = np.abs(o["start"] - est) + 4 + np.maximum(o["n_iters"] - 9, 0)
    need_by_m.setdefault(len(b.motif(l)), []).append(int(need.max()))
print(f"config {cfg}, {b.n_loci} loci: share of loci with a read that needs a half-width > W")
print(" motif  loci   W=4     W=5     W=6     W=8")
for lo, hi in ((1, 2), (3, 4), (5, 6), (7, 10), (11, 20)):
    v = np.array([x for m, xs in need_by_m.items() if lo <= m <= hi for x in xs])
    if len(v) == 0:
        continue
    print(f" {lo:2d}-{hi:2d} {len(v):5d}  " + "  ".join(f"{(v > w).mean():6.4f}" for w in (4, 5, 6, 8)))
