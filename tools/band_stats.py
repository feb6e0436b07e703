import sys; sys.path.insert(0,".")
from strkit_amd.synth import make_config
from strkit_amd.batch import count_loci
import numpy as np, time
for cfg, nl, kw in ((4, 3000, {}), (2, 1000, {}), (3, 2000, {})):
    b = make_config(cfg, n_loci=nl, **kw)
    for band in (True, False):
        r, st = count_loci(b, with_stats=True, band=band)
        t=time.perf_counter(); r, st = count_loci(b, with_stats=True, band=band); dt=time.perf_counter()-t
        print("cfg", cfg, "band", band, {k: st[k] for k in ("n_band_reads","n_band_fallback","n_dedup_reads","n_miss_reads","n_fallback")}, "kernel_ms %.2f band %.2f exact %.2f wall_ms %.1f" % (st["kernel_ms"], st["band_kernel_ms"], st["dp_kernel_ms"], dt*1e3), b.n_reads)
