import sys, numpy as np
sys.path.insert(0, ".")
from strkit_amd.synth import make_config
from strkit_amd.batch import count_loci
from strkit_amd import _lib
b = make_config(4, n_loci=3000)
ctx = _lib.Context(0)
for _ in range(3):
    got, st = count_loci(b, ctx=ctx, with_stats=True)
print({k: st[k] for k in ("n_band_reads", "n_band_fallback", "n_dedup_reads", "n_miss_reads")})
# which reads differ between band and exact classification? use per-locus runs to find fallback counts by motif length
mlen = np.diff(b.motif_off)
out = {}
for m in range(1, 21):
    loci = np.nonzero(mlen == m)[0][:60]
    if len(loci) == 0: continue
    from strkit_amd.sharding import select_loci
    sub, _ = select_loci(b, loci)
    c2 = _lib.Context(0)
    for _ in range(2): g, s2 = count_loci(sub, ctx=c2, with_stats=True)
    c2.close()
    ndb = (sub.nfl + sub.ntr + sub.nfr)
    out[m] = (sub.n_reads, s2["n_band_reads"], s2["n_band_fallback"], int(ndb.mean()))
for m, v in out.items(): print(m, v)
