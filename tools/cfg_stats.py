"""Per-call statistics of one configuration: python tools/cfg_stats.py <config> [n_loci] [window]"""
import sys, time
sys.path.insert(0, ".")
from strkit_amd.synth import make_config
from strkit_amd.batch import count_loci
cfg = int(sys.argv[1]); nl = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = make_config(cfg, n_loci=nl)
for rep in range(4):
    t = time.perf_counter(); r, st = count_loci(b, with_stats=True, window=w); dt = time.perf_counter() - t
    print(f"cfg {cfg} reads {b.n_reads} wall {dt*1e3:.1f} ms kernel {st['kernel_ms']:.2f} band {st['band_kernel_ms']:.2f} exact {st['dp_kernel_ms']:.2f} "
          f"miss_reads {st['n_miss_reads']} rounds {st['n_miss_rounds']} band_reads {st['n_band_reads']} fb {st['n_band_fallback']} dedup {st['n_dedup_reads']} cells {st['dp_cells']/1e9:.1f}G")
