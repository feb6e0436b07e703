#!/usr/bin/env python3
"""Time-bounded differential run of the counting path against the oracle (the checker): random shapes x random
search parameters x the library's switches, every read's (cn, score, n_iters, start) compared bit for bit.

    python tools/fuzz_parity.py [--seconds 420] [--seed 1] [--workers 12] [--out gpurun_out/fuzz.txt]

Every round is made from (seed, round index) alone, so a failure is reproduced with `--only ROUND`.  The oracle runs in
worker processes that are forked BEFORE this process touches the GPU; the GPU side runs here, through `strk_count_loci`
(host buffers, pipelined), `strk_count_loci_device` (resident buffers) and, for a few reads per round, the scalar drop-in
`strk_repeat_count`.  What a round draws:
  shape   — short HiFi-like loci; noisy (ONT-like) loci; long expansions (wide band classes, the long kernel); ragged
            flanks (0..80 bases) and empty tracts; IUPAC motifs; reads with X / N wildcards; lower-case stretches;
            start estimates from exact to wildly off (0, three times the size);
  search  — max_iters, local_search_range, step_size, tie rule, the four free-end flags, in-locus feedback on / off, the
            schedule of local_search_range (STRK_NARROW_*);
  library — adaptive or pinned candidate window, dedupe on / off, banded first pass on / off.
Exit code 1 on the first mismatch (its round, read and the two answers are printed and written to --out).
"""
from __future__ import annotations

import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from strkit_amd.synth import LocusBatch  # noqa: E402  (no GPU involved in importing it)

KEYS = ("cn", "score", "n_iters", "start")
IUPAC_CODES = "RYSWKMBDHVN"


def _seq(rng, n, alpha="ACGT"):
    return "".join(alpha[i] for i in rng.integers(len(alpha), size=n))


def _noise(rng, s, sub, indel, xrate, alpha="ACGT"):
    out = []
    for ch in s:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.append(alpha[rng.integers(len(alpha))])
            out.append(ch)
        elif u < indel + sub:
            out.append(alpha[rng.integers(len(alpha))])
        elif u < indel + sub + xrate:
            out.append("XN"[int(rng.random() < 0.3)])
        else:
            out.append(ch)
    return "".join(out)


def _expand(rng, motif):
    from strkit_amd.synth import _IUPAC
    return "".join((lambda c: c[rng.integers(len(c))])(_IUPAC.get(ch, ch)) for ch in motif)


def make_round(seed: int, i: int):
    """(batch, search/library parameters) of round i."""
    rng = np.random.default_rng([seed, i])
    shape = ("short", "short", "noisy", "noisy", "long", "ragged", "iupac", "tiny")[int(rng.integers(8))]
    if shape == "long":
        n_loci, rpl, mlen, cn = int(rng.integers(1, 5)), (1, 6), (1, 6), (80, int(rng.choice((300, 300, 700, 1600))))
        sub, indel, xr = (0.001, 0.002, 0.0005) if rng.random() < 0.7 else (0.02, 0.03, 0.005)
    elif shape == "noisy":
        n_loci, rpl, mlen, cn = int(rng.integers(20, 150)), (1, 25), (1, 20), (0, 70)
        sub, indel, xr = float(rng.choice((0.01, 0.03, 0.08))), float(rng.choice((0.01, 0.04, 0.1))), 0.01
    elif shape == "tiny":
        n_loci, rpl, mlen, cn = int(rng.integers(1, 40)), (0, 3), (1, 8), (0, 6)
        sub, indel, xr = 0.01, 0.02, 0.0
    else:
        n_loci, rpl, mlen, cn = int(rng.integers(20, 250)), (1, 40), (1, 20 if shape != "short" else 6), (0, 60)
        sub, indel, xr = 0.001, 0.002, 0.0005
    flank_rng = (0, 80) if shape in ("ragged", "tiny") else (70, 70)
    loci, ests = [], []
    for _ in range(n_loci):
        m = int(rng.integers(mlen[0], mlen[1] + 1))
        motif = _seq(rng, m)
        if shape == "iupac" and rng.random() < 0.6:
            motif = "".join(IUPAC_CODES[rng.integers(len(IUPAC_CODES))] if rng.random() < 0.3 else ch for ch in motif)
        fl0 = _seq(rng, int(rng.integers(flank_rng[0], flank_rng[1] + 1)))
        fr0 = _seq(rng, int(rng.integers(flank_rng[0], flank_rng[1] + 1)))
        if rng.random() < 0.15 and m <= len(fl0):          # flanks that run into the tract
            fl0 = fl0[:len(fl0) - m] + _expand(rng, motif)
        base = int(rng.integers(cn[0], cn[1] + 1))
        alleles = (base, max(0, base + int(rng.integers(-6, 7))))
        reads, est = [], []
        n_reads = int(rng.integers(rpl[0], rpl[1] + 1))
        dup = rng.random() < 0.5                             # HiFi: many reads of a locus are identical
        for r in range(n_reads):
            c = alleles[r & 1]
            if dup and r >= 2 and rng.random() < 0.6:
                reads.append(reads[r - 2])
            else:
                tr = _noise(rng, "".join(_expand(rng, motif) for _ in range(c)), sub, indel, xr)
                fl, fr = _noise(rng, fl0, sub, indel, xr), _noise(rng, fr0, sub, indel, xr)
                if rng.random() < 0.1:
                    k = int(rng.integers(0, len(tr) + 1))
                    tr = tr[:k].lower() + tr[k:]
                if rng.random() < 0.05:
                    fl, fr = fl.lower(), fr.lower()
                reads.append((fl, tr, fr))
            true = round(len(reads[-1][1]) / m)
            u = rng.random()
            wild = shape != "long"                         # (a 10 kb window against three times its size: minutes of oracle time)
            est.append(true if u < 0.5 else max(0, true + int(rng.integers(-9, 10))) if u < 0.9 or not wild else
                       0 if u < 0.95 else 3 * true + 1)
        loci.append((motif, reads))
        ests.append(est)
    p = dict(
        max_iters=int(rng.choice((5, 20, 50, 50, 50, 60))), lsr=int(rng.choice((1, 2, 3, 3, 3, 4, 5))),
        step=int(rng.choice((1, 1, 1, 2, 3, 5))), tie_rule=int(rng.integers(2)),
        flags=15 if rng.random() < 0.6 else int(rng.integers(16)), feedback=bool(rng.random() < 0.7),
        window=0 if rng.random() < 0.6 else int(rng.integers(4, 16)), dedupe=bool(rng.random() < 0.8),
        band=bool(rng.random() < 0.8), narrowing=0 if rng.random() < 0.6 else int(rng.integers(1, 4)), shape=shape)
    # bound the ORACLE's time per round to seconds of one core: about 9 G cells/s in its AVX2 pass, 0.5 G in the scalar code that
    # the non-default end flags and the IUPAC motifs take; a wild estimate runs the search to max_iters
    budget = 1e10 if p["flags"] == 15 and shape != "iupac" else 6e8
    cost, keep = 0.0, 0
    for (motif, reads), est in zip(loci, ests):
        for (fl, tr, fr), e in zip(reads, est):
            L = len(fl) + len(fr) + max(len(tr), e * len(motif))
            cost += float(L) * L * min(p["max_iters"] + 2 * p["lsr"], 14 if abs(e * len(motif) - len(tr)) <= 9 * len(motif) else 70)
        if cost > budget and keep:
            break
        keep += 1
    b = LocusBatch.from_reads(loci[:keep], ests[:keep])
    return b, p


def oracle_round(args):
    seed, i = args
    import oracle
    from helpers import oracle_count
    oracle.build()
    oracle.set_simd(True)       # (tests/test_oracle.py: identical to the scalar restatement)
    b, p = make_round(seed, i)
    t0 = time.perf_counter()
    try:
        exp = oracle_count(b, p["max_iters"], p["lsr"], p["step"], p["tie_rule"], p["flags"], p["feedback"], p["narrowing"])
    except ValueError as e:                                  # max() of nothing: the library answers STRK_E_EMPTY
        return i, None, str(e), time.perf_counter() - t0
    return i, exp, None, time.perf_counter() - t0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=420)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workers", type=int, default=min(12, (os.cpu_count() or 4) - 2))
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    log = open(a.out, "w") if a.out else None

    def say(*x):
        s = " ".join(str(v) for v in x)
        print(s, flush=True)
        if log:
            log.write(s + "\n")
            log.flush()

    pool = mp.get_context("fork").Pool(max(1, a.workers))    # before anything touches the GPU
    rounds = [a.only] if a.only >= 0 else range(10 ** 9)
    it = pool.imap_unordered(oracle_round, ((a.seed, i) for i in rounds), chunksize=1)

    import ctypes as C
    import torch
    from strkit_amd import _lib
    from strkit_amd.batch import count_loci, make_params
    from strkit_amd.repeat_count_params import RepeatCountParams
    L = _lib.load()
    dev = torch.device("cuda", 0)
    shared = _lib.Context(0)
    t_end = time.time() + a.seconds
    n_rounds = n_reads = n_empty = n_scalar = 0
    by_shape: dict[str, int] = {}
    tot = dict(band=0, fb=0, long=0, generic=0, miss=0, dedup=0)
    last_say = time.time()
    bad = 0
    import faulthandler
    for i, exp, err, t_or in it:
        b, p = make_round(a.seed, i)
        faulthandler.dump_traceback_later(120, exit=False)   # a call that takes minutes: say where (stderr)
        if a.only >= 0:
            say(f"round {i}: {b.n_loci} loci, {b.n_reads} reads, oracle {t_or:.1f} s; {p}")
        # a context of its own for two rounds in three: the band pass of a long-lived context cools down for many calls after a
        # batch whose certificates failed, and most of these batches are such (the shared one covers that state)
        own = (i % 3) != 0
        ctx = _lib.Context(0) if own else shared
        rc_params = RepeatCountParams("repalign", p["max_iters"], p["lsr"], p["step"])
        kw = dict(feedback=p["feedback"], window=p["window"], tie_rule=p["tie_rule"], end_flags=p["flags"], dedupe=p["dedupe"],
                  band=p["band"], narrowing=p["narrowing"])
        got = {}
        t_call = time.perf_counter()
        try:
            got["host"], st = count_loci(b, rc_params, ctx=ctx, with_stats=True, **kw)
        except _lib.StrkError as e:
            say(f"ROUND {i} ({p}; {b.n_loci} loci, {b.n_reads} reads): library error {e}")
            bad = 1
            break
        except ValueError as e:
            got["host"] = None
            if exp is not None:
                say(f"ROUND {i} ({p}): library raised {e!r}, the oracle did not")
                bad = 1
                break
        t_call = time.perf_counter() - t_call
        if a.only >= 0 or t_call > 20:
            say(f"round {i}: the call on host buffers took {t_call:.1f} s (oracle, one core: {t_or:.1f} s); {p['shape']}, {b.n_reads} reads, "
                f"missed reads {st['n_miss_reads'] if got['host'] is not None else '-'}, generic items {st['n_fallback'] if got['host'] is not None else '-'}")
        if exp is None:
            if got["host"] is not None:
                say(f"ROUND {i} ({p}): the oracle raised {err!r}, the library did not")
                bad = 1
                break
            n_empty += 1
            continue
        # resident buffers
        if b.n_reads:
            t = {k: torch.from_numpy(np.ascontiguousarray(getattr(b, k))).to(dev)
                 for k in ("seqs", "seq_off", "nfl", "ntr", "nfr", "est_cn", "read_off", "motifs", "motif_off")}
            if t["seqs"].numel() == 0:
                t["seqs"] = torch.zeros(1, dtype=torch.uint8, device=dev)
            sb = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})
            out = torch.zeros((4, b.n_reads), dtype=torch.int32, device=dev)
            pp = make_params(rc_params, **kw)
            st2 = _lib.StrkStats()
            _lib.check(L.strk_count_loci_device(ctx.handle, C.byref(sb), C.byref(pp), out[0].data_ptr(), out[1].data_ptr(),
                                                out[2].data_ptr(), out[3].data_ptr(), None, C.byref(st2)))
            o = out.cpu().numpy()
            got["device"] = {k: o[j] for j, k in enumerate(KEYS)}
        for name, g in got.items():
            for k in KEYS:
                if not np.array_equal(g[k], exp[k]):
                    r = int(np.flatnonzero(g[k] != exp[k])[0])
                    l = int(np.searchsorted(b.read_off, r, side="right") - 1)
                    say(f"MISMATCH round {i} path {name} key {k} read {r} (locus {l}, motif {b.motif(l)}): got "
                        f"{[int(g[x][r]) for x in KEYS]} expected {[int(exp[x][r]) for x in KEYS]}; params {p}; "
                        f"read lengths {int(b.nfl[r])}/{int(b.ntr[r])}/{int(b.nfr[r])} est {int(b.est_cn[r])}")
                    bad = 1
                    break
            if bad:
                break
        if bad:
            break
        # the scalar drop-in (default switches) on the first read of a few loci — what feedback cannot have touched
        if p["flags"] == 15 and p["tie_rule"] == 0 and p["narrowing"] == 0 and b.n_reads:
            from strkit_amd.repeats import get_repeat_count
            for l in range(min(3, b.n_loci)):
                r = int(b.read_off[l])
                if r == int(b.read_off[l + 1]):
                    continue
                fl, tr, fr = b.read(r)
                res = get_repeat_count(int(b.est_cn[r]), tr, fl, fr, b.motif(l), rc_params)
                want = ((int(exp["cn"][r]), int(exp["score"][r])), int(exp["n_iters"][r]), int(exp["cn"][r]) - int(b.est_cn[r]))
                if res != want:
                    say(f"MISMATCH round {i} scalar read {r}: got {res} expected {want}; params {p}")
                    bad = 1
                    break
                n_scalar += 1
            if bad:
                break
        faulthandler.cancel_dump_traceback_later()
        if own:
            ctx.close()
        n_rounds += 1
        n_reads += b.n_reads
        by_shape[p["shape"]] = by_shape.get(p["shape"], 0) + 1
        tot["band"] += st["n_band_reads"]; tot["fb"] += st["n_band_fallback"]; tot["long"] += st["n_long_reads"]
        tot["generic"] += st["n_fallback"]; tot["miss"] += st["n_miss_reads"]; tot["dedup"] += st["n_dedup_reads"]
        if time.time() - last_say > 30:
            say(f"[{time.strftime('%H:%M:%S')}] {n_rounds} rounds, {n_reads} reads identical ({n_empty} empty rounds); {by_shape}; {tot}")
            last_say = time.time()
        if time.time() > t_end:
            break
    pool.terminate()
    say(f"{'FAILED' if bad else 'ok'}: {n_rounds} rounds, {n_reads} reads x 2 paths and {n_scalar} scalar calls identical to the oracle ({n_empty} rounds in which both raised); "
        f"shapes {by_shape}; kernels {tot}; seed {a.seed}")
    shared.close()
    return bad


if __name__ == "__main__":
    sys.exit(main())
