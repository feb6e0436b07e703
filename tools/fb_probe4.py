import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["STRKIT_AMD_DUMP"] = "/tmp/strk_dump.bin"
os.environ["STRKIT_AMD_DBG"] = "4"     # no in-kernel search: every band read counts as certified, tables stay as the band wrote them
import oracle
from strkit_amd.synth import make_batch
from strkit_amd.batch import count_loci
from strkit_amd import _lib
for m in (5, 4):
    b = make_batch(7, 6, 2, (m, m), (5, 60), 0.0, 0.0, 0.0)
    ctx = _lib.Context(0)
    try:
        got, st = count_loci(b, ctx=ctx, with_stats=True, window=6, dedupe=False)
    except Exception as e:
        print("err", e)
    raw = np.fromfile("/tmp/strk_dump.bin", np.int32)
    n, ts = int(raw[0]), int(raw[1])
    tab = raw[2:2 + n * ts].reshape(n, ts); wl = raw[2 + n * ts:2 + n * ts + n]; wn = raw[2 + n * ts + n:2 + n * ts + 2 * n]
    for r in range(min(n, 4)):
        fl, tr, fr = b.read(r)
        l = int(np.searchsorted(b.read_off, r, side="right") - 1)
        exp = [oracle.candidate_score(tr, fl, fr, b.motif(l), int(wl[r]) + k) for k in range(int(wn[r]))]
        print("m", m, "read", r, "ntr", len(tr), "est", int(b.est_cn[r]), "lo", int(wl[r]))
        print("  band ", tab[r, :int(wn[r])].tolist())
        print("  exact", exp)
    ctx.close()
