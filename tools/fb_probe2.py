import sys, numpy as np
sys.path.insert(0, ".")
from strkit_amd.synth import make_batch
from strkit_amd.batch import count_loci
from strkit_amd import _lib
for m in (3, 4, 5, 6, 2, 1):
    for err in (0.0, 0.003):
        b = make_batch(7, 200, 10, (m, m), (5, 60), err / 3, err * 2 / 3, 0.0)
        ctx = _lib.Context(0)
        for w in (6, 8):
            for _ in range(2):
                got, st = count_loci(b, ctx=ctx, with_stats=True, window=w)
            print("m", m, "err", err, "W", w, {k: st[k] for k in ("n_band_reads", "n_band_fallback", "n_miss_reads", "n_dedup_reads")})
        ctx.close()
