#!/usr/bin/env python3
"""Model of the tighter band certificate VERDICT r2 item 5 proposes, against the CPU oracle (no GPU needed).

Proposal: keep the bound 2 * len(diagonal) for alignments that START outside the band, and bound alignments that LEAVE it by
what the banded pass already knows at its edge cells (first-exit decomposition).  Whatever the second part achieves, the first
part alone is a NECESSARY condition: with all four end gaps free (the default, parasail `sg`) an alignment may start on the top
row right of the band, run down one diagonal and end in the last column without ever entering the band, and the only thing a
sequence-blind bound can say about it is 2 * (cells on that diagonal).  The best size's score S* must beat that bound for the
search to certify:

    S* >= 2 * len(first diagonal outside the band)   <=>   deficit(S*) := 2 * min(rows, columns) - S* <= 2 * (slack + 1)

where slack = (band - span) / 2 is what the band has left on each side once it holds diagonal 0 and the end-corner diagonals of
every candidate size of the window (span = 2 W |motif| + 1).  This script draws BASELINE config 3 reads (ONT error model,
motif 2-20 bp), scores them exactly with the oracle and counts how many satisfy the necessary condition in a band of at most
256 diagonals, for the search windows the library uses on such reads (W = 8 and 15).

    python tools/cert_model.py [n_loci]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # the checker: this is a model, not product code
from strkit_amd.synth import make_config

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b = make_config(3, n_loci=n_loci)
rows = []
for l in range(b.n_loci):
    motif = b.motif(l)
    m = len(motif)
    for r in range(int(b.read_off[l]), int(b.read_off[l + 1])):
        fl, tr, fr = b.read(r)
        (cn, score), _n, _off = oracle.repeat_count(int(b.est_cn[r]), tr, fl, fr, motif)
        ndb = len(fl) + len(tr) + len(fr)
        nc = len(fl) + cn * m + len(fr)
        deficit = 2 * min(nc, ndb) - score
        rows.append((m, ndb, deficit, int(b.est_cn[r]), cn))
a = np.array(rows, np.int64)
m, ndb, deficit = a[:, 0], a[:, 1], a[:, 2]
print(f"{len(a)} reads of config 3 ({n_loci} loci): |window| median {int(np.median(ndb))}, score deficit of the best size: median "
      f"{int(np.median(deficit))}, 10th-90th percentile {int(np.percentile(deficit, 10))}-{int(np.percentile(deficit, 90))} "
      f"(a substitution costs 9, an inserted or deleted base 5-7)")
for W in (8, 15):
    span = 2 * W * m + 1
    for band in (128, 256):
        slack = (band - span) // 2
        fits = slack >= 0
        ok = fits & (deficit <= 2 * (slack + 1))
        print(f"W = {W:2d}, band {band:3d}: the band holds the window's candidate sizes for {fits.mean() * 100:5.1f} % of the reads; "
              f"the necessary condition S* >= 2 * len(diagonal outside) holds for {ok.mean() * 100:5.1f} % (of all reads)")
# a counter-example: the band of 256 holds the window's sizes at W = 8 with room to spare, and still the bound wins
W, band = 8, 256
slack = (band - (2 * W * m + 1)) // 2
bad = np.flatnonzero((slack >= 0) & (deficit > 2 * (slack + 1)))
if bad.size:
    i = int(bad[np.argmax(slack[bad])])          # the one with the MOST slack
    s_star = 2 * min(ndb[i], a[i, 1] - 0) - deficit[i]
    print(f"counter-example: motif of {m[i]} bases, window of {ndb[i]} bases, best size {a[i, 4]} with exact score {2 * min(ndb[i] - (ndb[i] - (ndb[i])), ndb[i]) - deficit[i]} "
          f"({deficit[i]} under a perfect alignment: ONT errors).  At W = 8 the candidate sizes span {2 * W * m[i] + 1} diagonals, a band of 256 leaves "
          f"{slack[i]} on each side; an alignment that starts on the top row {slack[i] + 1} columns right of the band's edge diagonal, runs down that diagonal and "
          f"ends in the last column never enters the band, and all a sequence-blind bound knows is <= {2 * (ndb[i] - slack[i] - 1)}: above the exact score, "
          f"so the read cannot be certified whatever is known about the alignments that LEAVE the band.")
