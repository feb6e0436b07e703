#!/usr/bin/env python3
"""bench.py — reads/s through the per-read repeat-count hot path on MI355X.

One "step" = one pass of the hot path (plan -> DP kernels -> search replay) over one batch of
synthetic reads that is already resident in HBM: BASELINE.json configs[1]
(1 000 loci x 30 HiFi reads, motif 3-6 bp, 70 bp flanks).  With --gpus N every rank owns its own
shard of N x 1 000 loci (weak scaling, loci are independent: strkit/call/call_sample.py:414) and
the per-read results are collected with one RCCL all-gather per step.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement; extra keys documented in DESIGN.md §6).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# Four calls in flight use four HIP streams besides the default one; the runtime's default of 4 hardware queues would
# make two of them share a queue (and serialise).  Read by the HIP runtime at its initialisation, so set before it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# int32 VALU peak used for the companion figure: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tops/s
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
WORKLOAD = ("cfg2: 1000 loci x 30 HiFi reads per GPU, motif 3-6 bp, flank 70; exact scores "
            "(128/256-diagonal banded pass with exactness certificate, exact fall-back)")


def _cpu_worker(args):
    """One CPU process of the baseline: the oracle's per-locus loop over a slice of loci."""
    cfg, lo, hi, seed_shift = args
    import oracle  # CPU baseline leg only (test infrastructure, never on the product path)
    from strkit_amd.synth import make_config
    b = make_config(cfg, n_loci=hi, seed_shift=seed_shift).locus_slice(lo, hi)
    t0 = time.perf_counter()
    cells = 0
    for l in range(b.n_loci):
        r0, r1 = int(b.read_off[l]), int(b.read_off[l + 1])
        s0 = int(b.seq_off[r0])
        o = oracle.count_locus(b.seqs[s0:int(b.seq_off[r1])], b.seq_off[r0:r1 + 1] - s0, b.nfl[r0:r1], b.ntr[r0:r1],
                               b.nfr[r0:r1], b.est_cn[r0:r1], b.motif(l), memo=True)  # lru_cache, repeats.py:47
        cells += o["cells"]
    return b.n_reads, time.perf_counter() - t0, cells


def cpu_baseline(cfg: int, sample_loci: int) -> dict:
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample, all host cores,
    loci sharded over processes as strkit/call/call_sample.py:414 does.  Runs BEFORE HIP is
    initialised so the forked workers never see a GPU context."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    per = max(1, sample_loci // cores)
    jobs = [(cfg, i * per, (i + 1) * per, 0) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    reads = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": reads / busy, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {per * cores} loci ({reads} reads) of the same workload, scalar C oracle with the reference's "
                      f"per-locus memoisation, {cores} processes, {busy:.1f} s busy each "
                      f"({busy * cores:.0f} core-seconds; {wall:.1f} s wall incl. input generation)",
            "reads_per_s_per_core": reads / busy / cores, "gcups": sum(r[2] for r in res) / busy / 1e9}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--loci", type=int, default=None, help="loci per GPU (default: the config's own count)")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--no-dedupe", action="store_true", help="score identical reads of a locus separately")
    ap.add_argument("--no-band", action="store_true", help="exact kernels only (no banded first pass)")
    ap.add_argument("--gather-every", type=int, default=8, help="steps per result all-gather when several ranks run")
    ap.add_argument("--pipeline", type=int, default=4, help="batched calls in flight (contexts/streams)")
    ap.add_argument("--cpu-sample-loci", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even with one rank (self-test)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    cpu = None
    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.config, a.cpu_sample_loci)

    import torch
    import torch.distributed as dist
    from strkit_amd import _lib
    from strkit_amd.batch import make_params
    from strkit_amd.synth import make_config

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- this rank's shard: its own loci (weak scaling), inputs made resident in HBM -----------
    b = make_config(a.config, n_loci=a.loci, seed_shift=rank)
    L = _lib.load()
    t = dict(seqs=torch.from_numpy(b.seqs).to(dev), seq_off=torch.from_numpy(b.seq_off).to(dev),
             nfl=torch.from_numpy(b.nfl).to(dev), ntr=torch.from_numpy(b.ntr).to(dev),
             nfr=torch.from_numpy(b.nfr).to(dev), est_cn=torch.from_numpy(b.est_cn).to(dev),
             read_off=torch.from_numpy(b.read_off).to(dev), motifs=torch.from_numpy(b.motifs).to(dev),
             motif_off=torch.from_numpy(b.motif_off).to(dev))
    sb = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})
    p = make_params(window=a.window, dedupe=not a.no_dedupe, band=not a.no_band)
    st = _lib.StrkStats()
    # D steps in flight: one context (workspace) + one HIP stream + one output buffer per slot, so
    # the tail of one batch overlaps the head of the next (successive locus blocks of a real run).
    D = max(1, a.pipeline)
    ctxs = [_lib.Context(local_rank) for _ in range(D)]
    streams = [torch.cuda.Stream(dev) for _ in range(D)]
    outs = [torch.zeros((4, b.n_reads), dtype=torch.int32, device=dev) for _ in range(D)]  # cn | score | n_iters | start
    # Results of every step are collected on every rank (RCCL all-gather over xGMI), G steps per collective: the
    # reference merges its workers' results once per contig (call_sample.py:195-197,420), not once per locus block.
    # The kernels write straight into the staging buffer of their round (two buffers: steps of the next round are in
    # flight while a round is being gathered), so collecting costs no extra copy.
    G = max(D, a.gather_every)
    stage = torch.zeros((2, G * 4, b.n_reads), dtype=torch.int32, device=dev) if use_dist else None
    gathered = torch.zeros((world * G * 4, b.n_reads), dtype=torch.int32, device=dev) if use_dist else None

    def out_of(i):
        if not use_dist:
            return outs[i % D]
        j = i % G
        return stage[(i // G) % 2, 4 * j:4 * j + 4]

    acc = dict(dp_ms=0.0, band_ms=0.0, all_ms=0.0, misses=0, fallback=0, dedup=0, band=0, band_fb=0, n=0)

    def submit(i):
        k = i % D
        if use_dist and i % G < D:   # first use of this stream in a round: the collective that last read the round's buffer is done
            streams[k].wait_stream(torch.cuda.current_stream(dev))
        o = out_of(i)
        _lib.check(L.strk_submit_loci_device(ctxs[k].handle, C.byref(sb), C.byref(p), o[0].data_ptr(), o[1].data_ptr(),
                                             o[2].data_ptr(), o[3].data_ptr(), C.c_void_p(streams[k].cuda_stream)))

    def finish(i, timed):
        k = i % D
        _lib.check(L.strk_finish(ctxs[k].handle, C.byref(st)))
        if timed:
            acc["dp_ms"] += st.dp_kernel_ms; acc["band_ms"] += st.band_kernel_ms; acc["all_ms"] += st.kernel_ms
            acc["band"] += st.n_band_reads; acc["band_fb"] += st.n_band_fallback
            acc["misses"] += st.n_miss_reads; acc["fallback"] += st.n_fallback; acc["dedup"] += st.n_dedup_reads
            acc["n"] += 1
        if use_dist and (i + 1) % G == 0:  # a round is complete: collect it from every shard
            dist.all_gather_into_tensor(gathered, stage[(i // G) % 2])

    def run(n_steps, timed):
        for i in range(min(D, n_steps)):
            submit(i)
        for i in range(n_steps):
            finish(i, timed)
            if i + D < n_steps:
                submit(i + D)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def flush(n_steps):
        if use_dist and n_steps % G:   # the last, partial round (the collective always moves the full buffer)
            dist.all_gather_into_tensor(gathered, stage[((n_steps - 1) // G) % 2])

    # One-time set-up, outside the W warm-up steps the caller asked for: the first call of every context allocates its
    # workspace (0.4 GB of scratch) and tries the band on a sample of the reads, and the library settles the default
    # candidate window after eight calls without a miss -- none of which belongs to a steady-state step.
    prime = 4 * D
    run(prime, False)
    flush(prime)
    fence()
    run(a.warmup, False)
    flush(a.warmup)
    fence()
    t0 = time.perf_counter()
    run(a.steps, True)
    flush(a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    dp_ms, all_ms, misses, fallback = acc["dp_ms"], acc["all_ms"], acc["misses"], acc["fallback"]
    out = out_of(a.steps - 1).clone()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([b.n_reads, b.n_loci], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        n_reads_all, n_loci_all = int(tot[0]), int(tot[1])
    else:
        n_reads_all, n_loci_all = b.n_reads, b.n_loci
    # un-overlapped duration of one call, for reference (outside the timed region)
    iso_dp, iso_band, iso_all = 0.0, 0.0, 0.0
    for _ in range(5):
        submit(0); finish(0, False)
        iso_dp += st.dp_kernel_ms / 5; iso_band += st.band_kernel_ms / 5; iso_all += st.kernel_ms / 5
    band_bytes, exact_bytes = int(st.band_bytes), int(st.exact_bytes)
    fence()

    def pmc_traffic(kernel):
        """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
        (profiles/README.md): FETCH_SIZE and WRITE_SIZE are in KiB and were collected in separate passes;
        FETCH_SIZE is doubled, the guide's gfx950 correction (calibrated here on k_hash, which reads every
        input byte exactly once: raw FETCH_SIZE = 0.49 x bytes)."""
        try:
            with open(os.path.join(ROOT, "profiles", "r01_f_pmc_summary.json")) as f:
                k = json.load(f)[kernel]
            return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        except Exception:  # noqa: BLE001
            return None

    if rank == 0:
        # sanity: the timed path's answers on the first loci agree with the oracle (checker only)
        import oracle
        chk = b.locus_slice(0, min(8, b.n_loci))
        got = out[:, :chk.n_reads].cpu().numpy()
        parity = "ok"
        for l in range(chk.n_loci):
            r0, r1 = int(chk.read_off[l]), int(chk.read_off[l + 1])
            s0 = int(chk.seq_off[r0])
            o = oracle.count_locus(chk.seqs[s0:int(chk.seq_off[r1])], chk.seq_off[r0:r1 + 1] - s0, chk.nfl[r0:r1],
                                   chk.ntr[r0:r1], chk.nfr[r0:r1], chk.est_cn[r0:r1], chk.motif(l))
            for i, k in enumerate(("cn", "score", "n_iters", "start")):
                if not np.array_equal(got[i, r0:r1], o[k]):
                    parity = f"MISMATCH locus {l} field {k}"
        # the dominant kernel of the timed region: the banded kernel when most reads certify, else k_dp_all
        band_ms = acc["band_ms"]
        if band_ms > dp_ms:
            kname, k_ms, alg_bytes = "k_dp_band", band_ms / a.steps, band_bytes
        else:
            kname, k_ms, alg_bytes = "k_dp_all", dp_ms / a.steps, exact_bytes
        dp_s = max(k_ms, 1e-9) / 1e3
        all_dp_s = max(dp_ms + band_ms, 1e-9) / a.steps / 1e3
        cells = int(st.dp_cells)
        line = {
            "metric": "reads/sec realigned", "value": n_reads_all * a.steps / elapsed, "unit": "reads/s",
            "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": WORKLOAD if a.config == 2 and a.loci is None else f"cfg{a.config}, {b.n_loci} loci per GPU",
                       "loci_per_gpu": b.n_loci, "reads_per_gpu": b.n_reads, "window": int(p.window) or 8,
                       "parallelism": f"loci-sharded x{a.gpus}" + (f" + all_gather every {G} steps" if use_dist else ""),
                       "calls_in_flight": D},
            "loci_per_s": n_loci_all * a.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": alg_bytes / dp_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / dp_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(kname) if (a.config == 2 and a.loci is None and not a.no_dedupe and not a.no_band) else None,
                         "kernel": "strk::" + kname,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "whole_path_algorithmic_bytes_per_step": b.algorithmic_bytes(),
                         "note": "integer max-plus DP: the binding unit is VALU issue (see valu); kernel_ms is the HIP-event "
                                 "duration inside the timed region, where calls_in_flight launches overlap; "
                                 "algorithmic bytes = (|window| + 16) per read this kernel scored"},
            "valu": {"gcups": cells / all_dp_s / 1e9, "cells_per_step": cells,
                     "peak_int32_tops": VALU_PEAK_TOPS, "unit": "G cell updates/s"},
            "device_ms_per_step": all_ms / a.steps, "isolated_call": {"k_dp_all_ms": iso_dp, "k_dp_band_ms": iso_band, "device_ms": iso_all},
            "band_reads_per_step": acc["band"] / a.steps, "band_fallback_per_step": acc["band_fb"] / a.steps, "band": not a.no_band, "window_miss_reads_per_step": misses / a.steps,
            "generic_kernel_items_per_step": fallback / a.steps,
            "dedup_reads_per_step": acc["dedup"] / a.steps, "dedupe": not a.no_dedupe,
            "parity_check": parity,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        j = (a.steps - 1) % G   # rank 0's block of the last collective, slot of the last step
        if rank == 0 and not torch.equal(gathered[4 * j:4 * j + 4], out):
            sys.exit("all_gather self-check failed")
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
