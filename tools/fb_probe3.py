import sys, numpy as np
sys.path.insert(0, ".")
from strkit_amd.synth import make_batch
from strkit_amd.batch import count_loci
from strkit_amd import _lib
sys.path.insert(0, "tests")
for m in (5, 6):
    b = make_batch(7, 40, 10, (m, m), (5, 60), 0.0, 0.0, 0.0)
    ctx = _lib.Context(0)
    got, st = count_loci(b, ctx=ctx, with_stats=True, window=6, dedupe=False)
    print("m", m, {k: st[k] for k in ("n_band_reads", "n_band_fallback", "n_miss_reads", "n_dedup_reads")})
    ctx.close()
    # which reads fell back? run read by read
    from strkit_amd.sharding import select_loci
    bad = []
    for l in range(b.n_loci):
        sub, _ = select_loci(b, np.array([l]))
        c2 = _lib.Context(0)
        g, s2 = count_loci(sub, ctx=c2, with_stats=True, window=6, dedupe=False)
        c2.close()
        if s2["n_band_fallback"]:
            bad.append((l, s2["n_band_reads"], s2["n_band_fallback"], int(sub.ntr[0]), int(sub.est_cn[0]), int(sub.nfl[0]), int(sub.nfr[0])))
    print(len(bad), bad[:12])
