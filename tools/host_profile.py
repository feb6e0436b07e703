#!/usr/bin/env python3
"""Profiling aid for the HOST side of `python -m strkit_amd call` without a GPU: the two device entry points are replaced by
stand-ins that return the estimate as the answer (no parity meaning), so that fetch / extraction / report assembly can be
timed where no device exists.  Never part of the product or the tests.
usage: python tools/host_profile.py <dataset dir made by synth_large> [--profile]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import strkit_amd.frontend.call as call  # noqa: E402
from strkit_amd import _lib  # noqa: E402


def fake_count_loci(batch, rc_params=None, ctx=None, **kw):
    ndb = (batch.nfl + batch.ntr + batch.nfr).astype(np.int32)
    return {"cn": batch.est_cn.copy(), "score": 2 * ndb, "n_iters": np.full(batch.n_reads, 9, np.int32), "start": batch.est_cn.copy()}


def fake_ref_counts(jobs, vcf_anchor_size, respect, context=None):
    return [((est, 2 * (len(fl) + len(tr) + len(fr))), 0, 0, (9, 9), (fl, tr, fr)) for est, tr, fl, fr, *_ in jobs]


def fake_ref_packed(start, seqs, seq_off, nfl, ntr, nfr, motifs, motif_off, ref_size, max_iters, lsr, step, anchor, respect=False, context=None):
    n = len(start)
    out = np.zeros((n, 9), np.int32)
    out[:, 0] = start; out[:, 4] = 9; out[:, 5] = 9; out[:, 6] = nfl; out[:, 7] = ntr; out[:, 8] = nfr
    return out


call.get_ref_repeat_counts_packed = fake_ref_packed
call.count_loci = fake_count_loci
call.get_ref_repeat_counts = fake_ref_counts
_lib.default_context = lambda *a, **k: None
d = sys.argv[1]
t0 = time.perf_counter()
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
rep = call.call_sample(d + "/reads.bam", d + "/ref.fa", d + "/loci.bed", front_end="host")
if "--profile" in sys.argv:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
print("total", round(time.perf_counter() - t0, 3), "loci", len(rep["results"]), "reads", sum(len(r.get("reads", {})) for r in rep["results"]),
      rep["stage_times"])
