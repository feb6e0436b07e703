#!/usr/bin/env python3
"""CPU count for VERDICT r3 item 2: how much banded DP work do the NON-duplicate reads of a locus share with an earlier read?

Forward pass: the rows of a read's candidates are fl + motif * i — the same for every read of the locus with the same left flank —
and the band cells of row r touch window columns <= r + dhi only.  So a read whose window agrees with an earlier read's in its first
c columns could start from that read's DP state at row R0 = c - dhi - G (checkpointed every `ckpt` rows).  Backward pass (right flank
rows, reversed): shared entirely when the last nfr + wd/2 + G columns agree.
Prints, per configuration, the share of the forward / backward band steps of the unique reads that a follower would skip.
    python tools/share_count.py [cfg] [n_loci]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strkit_amd.synth import make_config

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ckpt = 16
b = make_config(cfg, n_loci=n_loci)
tot_f = tot_b = sav_f = sav_b = 0
n_unique = n_follow_f = n_follow_b = 0
for l in range(b.n_loci):
    m = int(b.motif_off[l + 1] - b.motif_off[l])
    r0, r1 = int(b.read_off[l]), int(b.read_off[l + 1])
    seen = []
    for r in range(r0, r1):
        s = b.seqs[b.seq_off[r]:b.seq_off[r + 1]]
        nfl, ntr, nfr, est = int(b.nfl[r]), int(b.ntr[r]), int(b.nfr[r]), int(b.est_cn[r])
        key = (s.tobytes(), nfl, ntr, nfr, est)
        if any(k == key for k, *_ in seen):
            continue        # exact copy: deduped already
        span = 2 * W * m + 1
        wd = 96 if span + 44 <= 96 else (128 if span + 24 <= 128 else (192 if span + 44 <= 192 else 256))
        G = 8 if wd <= 128 else 16
        dhi = wd // 2 + W * m // 2 + 2
        lo = max(est - W, 0)
        rows_f = nfl + (lo + 2 * W) * m
        first_fork = nfl + lo * m
        tot_f += rows_f + G
        tot_b += nfr + G
        n_unique += 1
        best_f = best_b = 0
        for (_k, s2, nfl2, nfr2) in seen:
            n = min(len(s), len(s2))
            if nfl2 == nfl:
                neq = np.flatnonzero(s[:n] != s2[:n])
                lcp = int(neq[0]) if len(neq) else n
                R0 = min(lcp - dhi - G, first_fork - 1)
                R0 = (R0 // ckpt) * ckpt
                best_f = max(best_f, R0)
            if nfr2 == nfr:
                neq = np.flatnonzero(s[::-1][:n] != s2[::-1][:n])
                lcs = int(neq[0]) if len(neq) else n
                if lcs >= nfr + wd // 2 + G:
                    best_b = nfr + G
        if best_f > 0:
            n_follow_f += 1
        if best_b > 0:
            n_follow_b += 1
        sav_f += max(best_f, 0)
        sav_b += best_b
        seen.append((key, s, nfl, nfr))
print(f"config {cfg} ({b.n_loci} loci, W = {W}): {n_unique} unique reads of {b.n_reads}")
print(f"  forward steps: {tot_f}, skipped by starting from an earlier read's checkpoint: {sav_f} = {sav_f / tot_f:.1%}  ({n_follow_f / n_unique:.1%} of the unique reads follow)")
print(f"  backward steps: {tot_b}, shared whole: {sav_b} = {sav_b / tot_b:.1%}  ({n_follow_b / n_unique:.1%} of the unique reads follow)")
fw, bw = 0.61 + 0.106, 0.18
print(f"  of k_dp_band's instructions (forward steps + fork rows 72 %, backward 18 %): {fw * sav_f / tot_f * 0.85 + bw * sav_b / tot_b:.1%} (fork rows are never skipped: x 0.85)")
