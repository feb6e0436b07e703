#!/usr/bin/env python3
"""bench.py — reads/s through the per-read repeat-count hot path on MI355X.

One "step" = one pass of the hot path (hash -> plan -> banded / exact DP kernels -> search replay with the caller's
start-count feedback) over one batch of synthetic reads that is already resident in HBM.  The default workload is
BASELINE.json configs[1]'s shape at north_star's size: 10 000 loci x 30 HiFi reads (motif 3-6 bp, 70 bp flanks) per
step and GPU, built from ten independent instances of the 1 000-locus config.  EIGHT distinct batches (different
seeds) rotate through the timed region, so that nothing adaptive inside the library (candidate-window level, band
probation, history-sized grids) is replaying one input.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--strong --config 4|5]

Default (weak scaling): every rank owns its own batches; the per-read results of every step are collected on every
rank with one RCCL all-gather per `--gather-every` steps.  --strong: ONE catalog (the named config at its full size)
is dealt to the ranks in blocks of <= 200 loci balanced by estimated DP cells (strkit/call/loci.py:193,
call_sample.py:414), every rank counts its share each step, fixed-size per-read records are all-gathered and rank 0
checks the gathered table against the one-rank table bit for bit.

Rank 0 prints ONE JSON line (contract in the task statement; extra keys documented in DESIGN.md §6).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# Calls in flight use one HIP stream each besides the default one; the runtime's default of 4 hardware queues would make
# two of them share a queue (and serialise).  Read by the HIP runtime at its initialisation, so set before it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
N_SIMD = 256 * 4
# Measured on MI355X (profiles/r04_valu_rate3.txt, tools/valu_rate3.hip): the band kernel's ACTUAL step — 16 x {v_add_u32_sdwa,
# v_max3_i32} in a dependent chain, 4 v_perm_b32, 4 v_alignbyte_b32, two DPP edge exchanges, the LDS reads — issues in 205 SIMD
# cycles per 44 instructions at two waves per SIMD (the kernels' occupancy): 4.65 cycles per wave-instruction (independent
# instructions of the same opcodes: 4.2-4.3, profiles/r03_valu_rate2.txt; one wave alone 5.7; only plain VOP2 adds reach 2.6).
VALU_CYCLES_PER_INST = 4.65
# ... per kernel: the two band kernels run that step; the exact kernel's stream (24-40 columns per lane, the adds independent of the
# max chain) issues at the 4.3 of independent instructions (config 3: 4 077 M instructions in 7.4 ms at 2.3 GHz = 4.3 cycles each)
VALU_CYCLES_BY_KERNEL = {"k_dp_band": 4.65, "k_dp_band_wide": 4.65}
VALU_CYCLES_DEFAULT = 4.3
DP_INSTS_PER_CELL_FLOOR = 2.25      # v_add_u32_sdwa + v_max3_i32 per cell + one v_perm_b32 per four cells
PROFILE_TAG_PREV = "r03"
PROFILE_TAG = "r04"                 # profiles/<tag>[_cfgN]_pmc_summary.json: the PMC passes of this same command (config N)
UNIT_LOCI = 1000                    # one instance of a BASELINE config 2 / 3 batch
# the five DP kernels of a call: (name, strk_stats field with its HIP-event duration)
DP_KERNELS = (("k_dp_band", "band_kernel_ms"), ("k_dp_band_wide", "band_wide_kernel_ms"), ("k_dp_all", "dp_kernel_ms"),
              ("k_dp_long", "long_kernel_ms"), ("k_dp_generic", "generic_kernel_ms"))
SUB_CONFIGS = {"cfg3": (3, 10000, "cfg3 shape: 10 000 loci x 20 ONT-error reads, motif 2-20 bp"),
               "cfg4": (4, 21250, "cfg4 shape, one GPU's eighth of the whole-genome catalog: 21 250 loci x 30 HiFi reads, 70 % motifs 1-6 bp / 30 % 7-20 bp"),
               "cfg5": (5, 250, "cfg5 shape, expansion stress, one GPU's eighth: 250 loci x 40 reads, motif 1-6 bp, 50-2 000 copies"),
               "cfg5_all": (5, 2000, "cfg5, the whole configuration on this GPU: 2 000 loci x 40 reads, motif 1-6 bp, 50-2 000 copies")}


def _gen_worker(args):
    cfg, n_loci, seed_shift = args
    from strkit_amd.synth import make_config
    return make_config(cfg, n_loci=n_loci, seed_shift=seed_shift)


def make_batches(cfg: int, n_loci: int, n_batches: int, rank: int, pool) -> list:
    """`n_batches` distinct batches of `n_loci` loci of config `cfg`: each is the concatenation of independent
    instances of <= 1 000 loci (seed shifts unique per rank, batch and instance), generated on all host cores."""
    from strkit_amd.synth import LocusBatch
    per = [min(UNIT_LOCI, n_loci - k) for k in range(0, n_loci, UNIT_LOCI)]
    jobs = [(cfg, n, (rank * 64 + b) * 1024 + j) for b in range(n_batches) for j, n in enumerate(per)]
    parts = pool.map(_gen_worker, jobs) if pool is not None else [_gen_worker(j) for j in jobs]
    return [LocusBatch.concat(parts[b * len(per):(b + 1) * len(per)]) for b in range(n_batches)]


def _cpu_worker(args):
    """One CPU process of the baseline: the oracle's per-locus loop over a slice of loci."""
    cfg, lo, hi, seed_shift, simd = args
    import oracle  # CPU baseline leg only (test infrastructure, never on the product path)
    from strkit_amd.synth import make_config
    oracle.set_simd(bool(simd))
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


def cpu_baseline(cfg: int, sample_loci: int, pool, cores: int) -> dict:
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample, all host cores, loci sharded over
    processes as strkit/call/call_sample.py:414 does — once with the scalar code and once with the inter-sequence AVX2
    variant (oracle/strk_simd.c: sixteen candidate sizes of a read per pass, the class of code parasail is), whose rate is
    the reported baseline.  Runs BEFORE HIP is initialised so the forked workers never see a GPU context."""
    import oracle
    out = {}
    for name, simd, loci in (("scalar", 0, sample_loci), ("simd", 1, 4 * sample_loci)):
        if simd and not oracle.set_simd(True):
            continue
        per = max(1, min(UNIT_LOCI, loci // cores))
        # every process takes its own slice; beyond one instance of the config the processes use further instances
        jobs = [(cfg, (i * per) % UNIT_LOCI, (i * per) % UNIT_LOCI + per, (i * per) // UNIT_LOCI, simd) for i in range(max(cores, loci // per))]
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        reads = sum(r[0] for r in res)
        busy = sum(r[1] for r in res) / cores            # jobs are dealt to `cores` processes
        out[name] = {"value": reads / busy, "unit": "reads/s", "cores": cores, "reads": reads, "busy_s_per_core": busy, "wall_s": wall,
                     "reads_per_s_per_core": reads / busy / cores, "gcups": sum(r[2] for r in res) / busy / 1e9}
    oracle.set_simd(False)
    best = out.get("simd", out["scalar"])
    return {"value": best["value"], "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"{best['reads']} reads ({best['reads'] // 30} loci) of the same workload, C restatement of the reference algorithm with its "
                      f"per-locus memoisation, {'AVX2 inter-sequence int16 scoring (16 candidate sizes per pass)' if 'simd' in out else 'scalar scoring'}, "
                      f"{cores} processes, {best['busy_s_per_core']:.1f} s busy each ({best['busy_s_per_core'] * cores:.0f} core-seconds)",
            "reads_per_s_per_core": best["reads_per_s_per_core"], "gcups": best["gcups"], "scalar": out["scalar"], "simd": out.get("simd")}


def pmc_file(cfg=2) -> str | None:
    """profiles/<tag>[_cfgN]_pmc_summary.json of this round, else the previous round's (named in the output).  `cfg`: a
    configuration number, or a key such as "5e" (config 5 at one GPU's share: its own profile, `r04_cfg5e_*`)."""
    for tag in (PROFILE_TAG, PROFILE_TAG_PREV):
        name = f"{tag}_pmc_summary.json" if cfg == 2 else f"{tag}_cfg{cfg}_pmc_summary.json"
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            return name
    return None


def pmc_summary(kernel: str, cfg=2) -> dict | None:
    """Per-launch PMC readings of `kernel` from the committed rocprofv3 passes of this same command on config `cfg`
    (profiles/README.md; separate --pmc passes, FETCH_SIZE / WRITE_SIZE in KiB)."""
    try:
        with open(os.path.join(ROOT, "profiles", pmc_file(cfg))) as f:
            return json.load(f)[kernel]
    except Exception:  # noqa: BLE001
        return None


def kernel_bytes(acc: dict) -> dict:
    """Algorithmic bytes ((|window| + 16) per read) of the reads each DP kernel was handed."""
    return {"k_dp_band": acc["band_bytes"] - acc["wide_bytes"], "k_dp_band_wide": acc["wide_bytes"],
            "k_dp_all": acc["exact_bytes"] - acc["long_bytes"], "k_dp_long": acc["long_bytes"], "k_dp_generic": 0}


def kernel_cells(acc: dict) -> dict:
    """DP cells by the kernel that executed them (strk_stats.band_cells ...; the generic kernel's are the remainder)."""
    named = acc["band_cells"] + acc["wide_cells"] + acc["exact_cells"] + acc["long_cells"]
    return {"k_dp_band": acc["band_cells"], "k_dp_band_wide": acc["wide_cells"], "k_dp_all": acc["exact_cells"],
            "k_dp_long": acc["long_cells"], "k_dp_generic": max(0, acc["cells"] - named)}


def valu_block(kname: str, cfg, k_ms: float, cells_launch: float) -> dict | None:
    """VALU-issue figures of ONE kernel: instruction count from the committed PMC pass of this command on config `cfg`, this
    run's un-overlapped duration and this run's cells OF THAT KERNEL."""
    pmc = pmc_summary(kname, cfg)
    if not pmc or "SQ_INSTS_VALU" not in pmc:
        return None
    clock_ghz = pmc.get("clock_ghz", 2.3)
    cyc = VALU_CYCLES_BY_KERNEL.get(kname, VALU_CYCLES_DEFAULT)
    floor_ms = pmc["SQ_INSTS_VALU"] * cyc / N_SIMD / (clock_ghz * 1e9) * 1e3
    return {"valu_from_profile": f"profiles/{pmc_file(cfg)}",
            "valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "cycles_per_inst": cyc,
            "valu_clock_ghz": clock_ghz, "floor_ms": floor_ms,
            "profile_kernel_ms": pmc.get("unoverlapped_avg_us", 0.0) / 1e3,
            "frac_valu": floor_ms / k_ms, "cells_per_launch": cells_launch,
            "insts_per_cell": pmc["SQ_INSTS_VALU"] * 64.0 / max(cells_launch, 1.0),
            "insts_per_cell_floor": DP_INSTS_PER_CELL_FLOOR,
            "frac_of_cell_floor": (cells_launch / 64.0 * DP_INSTS_PER_CELL_FLOOR * cyc / N_SIMD
                                   / (clock_ghz * 1e9) * 1e3) / k_ms}


def roofline_block(cfg, acc_timed: dict, iso: dict | None, n_timed: int, plain: bool) -> dict:
    """`roofline` of the dominant kernel.  iso = per-kernel durations of one call at a time (un-overlapped), or None;
    acc_timed = the same sums over the timed region, where calls_in_flight launches share the device."""
    n = max(1, n_timed)
    src = {k: iso[k] for k, _ in DP_KERNELS} if iso else {k: acc_timed[f] / n for k, f in DP_KERNELS}
    kname = max(src, key=lambda k: src[k])
    k_ms = max(src[kname], 1e-9)
    alg = (iso["bytes"] if iso else {k: v / n for k, v in kernel_bytes(acc_timed).items()})[kname]
    over_ms = acc_timed[dict(DP_KERNELS)[kname]] / n
    roof = {"bound": "hbm", "achieved": alg / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "kernel": "strk::" + kname, "kernel_ms": k_ms, "kernel_ms_is": "un-overlapped (one call at a time)" if iso else "overlapped",
            "kernel_ms_overlapped": over_ms, "algorithmic_bytes_per_launch": alg,
            "dp_kernels_ms": {k: src[k] for k, _ in DP_KERNELS},
            "note": "integer max-plus DP: what binds is VALU issue, not HBM (valu_* keys); achieved / frac divide the kernel's "
                    "algorithmic bytes ((|window| + 16) per read it scored) by its HIP-event duration with ONE call on the device; "
                    "kernel_ms_overlapped is the same kernel's event duration inside the timed region, where calls_in_flight "
                    "launches share the CUs (not a throughput denominator)"}
    pmc = pmc_summary(kname, cfg) if plain else None
    cells_k = iso["cells_k"] if iso else {k: v / n for k, v in kernel_cells(acc_timed).items()}
    roof["dp_kernels_cells"] = cells_k
    if pmc:
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            roof["traffic"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        vb = valu_block(kname, cfg, k_ms, cells_k[kname])
        if vb:
            roof.update(vb)
            roof["valu_note"] = ("instruction counts come from the committed rocprofv3 PMC pass of this command (a live run cannot "
                                 "count instructions); cells (of THIS kernel) and kernel_ms are this run's: profile_kernel_ms far "
                                 "from kernel_ms means the profile is stale")
        # every other DP kernel that takes more than a fifth of the configuration's DP time gets the same figures
        dp_total = sum(src.values())
        others = {}
        for k, _f in DP_KERNELS:
            if k != kname and src[k] > 0.2 * dp_total:
                ob = valu_block(k, cfg, max(src[k], 1e-9), cells_k[k])
                if ob:
                    pk = pmc_summary(k, cfg)
                    ob["kernel_ms"] = src[k]
                    ob["traffic"] = ((2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024.0) if "FETCH_SIZE" in pk and "WRITE_SIZE" in pk else None
                    others["strk::" + k] = ob
        if others:
            roof["other_dp_kernels"] = others
    return roof


def run_e2e(data: dict) -> dict:
    """`python -m strkit_amd call` on files: alignment file + reference + catalog -> per-read copy numbers, every stage timed,
    with the device front end (the file inflated, scanned and cut on the GPU: the default without --realign) and with the host
    one (block-wise through the .bai on the host cores).  calling_s = reference side + read extraction + counting + filtering;
    report_s = building the per-read report rows; open_s = opening the alignment file (device: upload, inflation and record
    scan of the whole file)."""
    import shutil
    from strkit_amd.frontend import call_sample
    paths = data["paths"]
    warm = os.path.join(data["dir"], "warm.bed")
    with open(paths["loci"]) as fh, open(warm, "w") as out:
        out.writelines(fh.readlines()[:200])
    truth = {(int(l), int(r)): int(c) for l, r, c in data["truth"]}

    def one(front_end):
        call_sample(paths["bam"], paths["ref"], warm, front_end=front_end)      # workspace allocation, library warm-up
        t0 = time.perf_counter()
        rep = call_sample(paths["bam"], paths["ref"], paths["loci"], front_end=front_end)
        wall = time.perf_counter() - t0
        st = rep["stage_times"]
        n_reads = n_true = 0
        for row in rep["results"]:
            for name, rd in row.get("reads", {}).items():
                l, r = name[1:].split("_r")
                n_reads += 1
                n_true += rd["cn"] == truth[(int(l), int(r))]
        calling = sum(st.get(k, 0.0) for k in ("ref_side_s", "realign_s", "extract_s", "count_s"))
        return {"loci": len(rep["results"]), "reads": n_reads, "wall_s": wall, "loci_per_s": len(rep["results"]) / wall,
                "reads_per_s": n_reads / wall, "calling_s": calling, "stage_s": st,
                "device_share_of_calling": (st.get("count_device_s", 0.0)) / max(calling, 1e-9),
                # device front end: kernels (inflation, scan, extraction, counting) over opening the file + calling
                "device_share_of_open_and_calling": ((st.get("count_device_s", 0.0) + st.get("front_end_device_s", 0.0))
                                                     / max(calling + st.get("open_s", 0.0), 1e-9)) if "front_end_device_s" in st else None,
                "reads_with_true_allele_cn": n_true}, rep["results"]

    # both front ends by the same protocol: two runs each, the FIRST one reported, both walls listed (opening the file depends
    # on where the box has the file's pages and the process's pinned buffers: open_stage_s says where a run lost its time)
    runs = [one("device") for _ in range(2)]
    dev, rows_dev = runs[0]
    hruns = [one("host") for _ in range(2)]
    host, rows_host = hruns[0]
    host["wall_s_runs"] = [r[0]["wall_s"] for r in hruns]
    out = dict(dev)
    out["wall_s_runs"] = [r[0]["wall_s"] for r in runs]
    out.update({"read_len": int(data.get("read_len", 0)) or None, "bam_mb": os.path.getsize(paths["bam"]) >> 20,
                "host_front_end": host, "front_ends_agree": rows_dev == rows_host, "dataset_gen_s": data["gen_s"],
                "note": "wall_s = opening the alignment file + FASTA + catalog + calling + report rows.  Device front end: open_s = "
                        "read + upload + inflation + record scan of the whole file on the GPU (a thread of its own, beside catalog and "
                        "reference side: open_wait_s is what the caller still waited), the bases never leave the device.  Host front end: load_s "
                        "(BGZF inflate of a block's records, all cores) runs in a second thread and overlaps calling, load_wait_s is what "
                        "the caller waited for it"})
    shutil.rmtree(data["dir"], ignore_errors=True)
    return out


def bucket_windows(acc: dict) -> dict:
    """candidate-window half-widths the timed calls ran with, per motif-length bucket that had loci (strk_stats.window_bucket)"""
    names = ("1-2", "3-4", "5-6", "7-10", "11+")
    return {names[k]: sorted(v) for k, v in enumerate(acc["window_bucket"]) if v}


def progress(msg: str) -> None:
    """Stage marker on stderr (stdout carries the one JSON line only)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--loci", type=int, default=None, help="loci per step (weak: per GPU; strong: of the whole catalog)")
    ap.add_argument("--batches", type=int, default=8, help="distinct resident batches that rotate through the steps")
    ap.add_argument("--strong", action="store_true", help="one catalog dealt to the ranks (strong scaling), gathered table checked")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--no-dedupe", action="store_true", help="score identical reads of a locus separately")
    ap.add_argument("--no-band", action="store_true", help="exact kernels only (no banded first pass)")
    ap.add_argument("--gather-every", type=int, default=4, help="steps per result all-gather when several ranks run")
    ap.add_argument("--pipeline", type=int, default=2, help="batched calls in flight (contexts/streams): with two, the band kernel of one call and "
                    "everything else of the other share the device and the persistent band blocks of a THIRD call never hold the LDS that the small "
                    "kernels behind a band pass wait for (profiles/README.md round 3: 221 M reads/s at 2, 174 M at 3, 198 M at 4-6)")
    ap.add_argument("--cpu-sample-loci", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the pipeline-1 and host-buffer (PCIe-inclusive) sub-results")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even with one rank (self-test)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end sub-result (BAM + FASTA + BED -> per-read copy numbers)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` sub-results (BASELINE configs 3, 4-shard, 5 timed on this GPU)")
    ap.add_argument("--prime", type=int, default=0, help="untimed set-up calls before the warm-up (0: enough for the adaptive state to settle)")
    ap.add_argument("--e2e-loci", type=int, default=10000)
    ap.add_argument("--e2e-depth", type=int, default=30)
    ap.add_argument("--e2e-read-len", type=int, default=15000)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    default_loci = {1: 44, 2: 10000, 3: 10000, 4: 170000 if a.strong else 21250, 5: 2000 if a.strong else 250}
    n_loci = a.loci or default_loci[a.config]

    # ---- host-side preparation in forked workers, BEFORE HIP is initialised --------------------------------------
    import multiprocessing as mp
    cores = max(1, min(16, len(os.sched_getaffinity(0)) // max(1, min(world, 8))))
    cpu = None
    with mp.get_context("fork").Pool(cores) as pool:
        if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
            import oracle
            oracle.build()
            progress("cpu baseline")
            cpu = cpu_baseline(a.config, a.cpu_sample_loci, pool, cores)
        progress("generating batches")
        if a.strong:
            catalog = make_batches(a.config, n_loci, 1, 0, pool)[0]      # the same catalog on every rank
        else:
            batches = make_batches(a.config, n_loci, max(1, a.batches), rank, pool)
        sub_batches = {}
        if rank == 0 and a.gpus == 1 and not a.no_extras and not a.no_configs and not a.strong and a.config == 2 and a.loci is None:
            for name, (c, nl, _) in SUB_CONFIGS.items():     # the other single-GPU configurations of BASELINE.json, two batches each
                sub_batches[name] = make_batches(c, nl, 2, 7, pool)
    e2e_data = None
    if rank == 0 and a.gpus == 1 and not a.no_e2e and not a.strong:
        # files for the end-to-end sub-result: north_star's "10 000 loci x 30x HiFi reads genotyped end-to-end", ~15 kb reads
        import tempfile
        from strkit_amd.frontend.synth_large import make_dataset_large
        e2e_dir = tempfile.mkdtemp(prefix="strk_e2e_")
        t_gen = time.perf_counter()
        e2e_data = make_dataset_large(e2e_dir, n_loci=a.e2e_loci, depth=a.e2e_depth, read_len=a.e2e_read_len, seed=11, procs=cores)
        e2e_data["gen_s"] = time.perf_counter() - t_gen
        e2e_data["dir"] = e2e_dir
        e2e_data["read_len"] = a.e2e_read_len

    import torch
    import torch.distributed as dist
    from strkit_amd import _lib
    from strkit_amd.batch import batch_struct, make_params
    from strkit_amd.sharding import NF, deal_blocks, gathered_step_table, select_loci, share_sizes, step_rows

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    if a.strong:    # this rank's share of the one catalog, and where its reads sit in the whole table
        shares = deal_blocks(catalog, world)
        mine, my_reads = select_loci(catalog, shares[rank])
        batches = [mine]
        n_pad = max(share_sizes(catalog, shares))
    L = _lib.load()

    def resident(b):
        t = dict(seqs=torch.from_numpy(b.seqs).to(dev), seq_off=torch.from_numpy(b.seq_off).to(dev),
                 nfl=torch.from_numpy(b.nfl).to(dev), ntr=torch.from_numpy(b.ntr).to(dev),
                 nfr=torch.from_numpy(b.nfr).to(dev), est_cn=torch.from_numpy(b.est_cn).to(dev),
                 read_off=torch.from_numpy(b.read_off).to(dev), motifs=torch.from_numpy(b.motifs).to(dev),
                 motif_off=torch.from_numpy(b.motif_off).to(dev))
        return t, _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})

    res_b = [resident(b) for b in batches]
    NB = len(batches)
    n_reads_max = max(b.n_reads for b in batches)
    rows = n_pad if a.strong else n_reads_max       # result columns per step (strong: padded to the largest share)
    p = make_params(window=a.window, dedupe=not a.no_dedupe, band=not a.no_band)
    st = _lib.StrkStats()
    # D steps in flight: one context (workspace) + one HIP stream per slot, so the head of one batch (hash, plan: latency
    # bound) overlaps the tail of the previous one (successive locus blocks of a real run).
    D = max(1, a.pipeline)
    ctxs = [_lib.Context(local_rank) for _ in range(D)]
    streams = [torch.cuda.Stream(dev) for _ in range(D)]
    # Results: per step five int32 rows (read index | cn | score | n_iters | start).  With several ranks the kernels
    # write straight into the staging buffer of their gather round (two buffers: the next round is in flight while one
    # is being gathered) and ONE RCCL all-gather per G steps collects the fixed-size records on every rank (the
    # reference merges its workers' results once per contig, call_sample.py:195-197,420, not once per locus block).
    G = max(D, a.gather_every) if use_dist else D
    stage = torch.full((2, G * NF, rows), -1, dtype=torch.int32, device=dev)
    gathered = torch.zeros((world * G * NF, rows), dtype=torch.int32, device=dev) if use_dist else None
    if a.strong:
        stage[:, 0::NF, :len(my_reads)] = torch.from_numpy(my_reads.astype(np.int32)).to(dev)
    else:
        stage[:, 0::NF, :] = torch.arange(rows, dtype=torch.int32, device=dev)

    def batch_of(i):
        return (i + i // D) % NB        # every context sees every batch

    def out_of(i):
        return step_rows(stage[(i // G) % 2], i % G)

    def new_acc():
        return dict(dp_kernel_ms=0.0, band_kernel_ms=0.0, band_wide_kernel_ms=0.0, long_kernel_ms=0.0, generic_kernel_ms=0.0,
                    head_ms=0.0, replay_ms=0.0, all_ms=0.0, misses=0, fallback=0, dedup=0, band=0, band_fb=0, n=0, reads=0, loci=0,
                    band_bytes=0, exact_bytes=0, wide_bytes=0, long_bytes=0, cells=0, band_cells=0, wide_cells=0, exact_cells=0,
                    long_cells=0, windows=set(), window_bucket=[set() for _ in range(5)])

    def add_stats(ac, st_, n_reads, n_loci_):
        for _k, f in DP_KERNELS:
            ac[f] += getattr(st_, f)
        ac["head_ms"] += st_.head_ms; ac["replay_ms"] += st_.replay_ms; ac["all_ms"] += st_.kernel_ms
        ac["band"] += st_.n_band_reads; ac["band_fb"] += st_.n_band_fallback
        ac["misses"] += st_.n_miss_reads; ac["fallback"] += st_.n_fallback; ac["dedup"] += st_.n_dedup_reads
        ac["band_bytes"] += st_.band_bytes; ac["exact_bytes"] += st_.exact_bytes
        ac["wide_bytes"] += st_.wide_bytes; ac["long_bytes"] += st_.long_bytes; ac["cells"] += st_.dp_cells
        for _k in ("band_cells", "wide_cells", "exact_cells", "long_cells"):
            ac[_k] += getattr(st_, _k)
        ac["windows"].add(int(st_.window_used))
        for _k in range(5):
            if st_.window_bucket[_k]:
                ac["window_bucket"][_k].add(int(st_.window_bucket[_k]))
        ac["n"] += 1; ac["reads"] += n_reads; ac["loci"] += n_loci_

    acc = new_acc()

    def submit(i):
        k = i % D
        if use_dist and i % G < D:   # first use of this stream in a round: the collective that last read the round's buffer is done
            streams[k].wait_stream(torch.cuda.current_stream(dev))
        o = out_of(i)
        _lib.check(L.strk_submit_loci_device(ctxs[k].handle, C.byref(res_b[batch_of(i)][1]), C.byref(p), o[1].data_ptr(),
                                             o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), C.c_void_p(streams[k].cuda_stream)))

    def finish(i, timed):
        k = i % D
        _lib.check(L.strk_finish(ctxs[k].handle, C.byref(st)))
        if timed:
            add_stats(acc, st, batches[batch_of(i)].n_reads, batches[batch_of(i)].n_loci)
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
    # workspace and tries the band on a sample of the reads, and the library settles the default candidate window after
    # eight calls without a miss -- none of which belongs to a steady-state step.
    # (the window level of a motif-length bucket drops after eight quiet calls, one level at a time: 8 -> 6 takes 8 calls, the
    # long-motif buckets' 8 -> 6 -> 5 -> 4 takes 24; band probation ends with the first call)
    prime = a.prime if a.prime > 0 else (max(3 * D, 12) if a.config in (2, 3) else 28)
    progress("priming + warm-up + timed region")
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
    last = a.steps - 1
    out_last = out_of(last).clone()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([acc["reads"], acc["loci"]], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        reads_all, loci_all = int(tot[0]), int(tot[1])
    else:
        reads_all, loci_all = acc["reads"], acc["loci"]

    # ---- sub-results outside the timed region (rank 0 of a one-GPU run) -------------------------------------------
    def one_at_a_time(res_list, blist, n_calls, out_rows):
        """n_calls calls, one at a time on context 0: (reads/s, ms per call, per-kernel un-overlapped durations + bytes + cells)."""
        ac = new_acc()
        fence()
        t1 = time.perf_counter()
        for i in range(n_calls):
            _lib.check(L.strk_submit_loci_device(ctxs[0].handle, C.byref(res_list[i % len(res_list)][1]), C.byref(p), out_rows[1].data_ptr(),
                                                 out_rows[2].data_ptr(), out_rows[3].data_ptr(), out_rows[4].data_ptr(),
                                                 C.c_void_p(streams[0].cuda_stream)))
            _lib.check(L.strk_finish(ctxs[0].handle, C.byref(st)))
            add_stats(ac, st, blist[i % len(blist)].n_reads, blist[i % len(blist)].n_loci)
        fence()
        e1 = time.perf_counter() - t1
        iso_ = {k: ac[f] / n_calls for k, f in DP_KERNELS}
        iso_["bytes"] = {k: v / n_calls for k, v in kernel_bytes(ac).items()}
        iso_["cells"] = ac["cells"] / n_calls
        iso_["cells_k"] = {k: v / n_calls for k, v in kernel_cells(ac).items()}
        iso_["head_ms"] = ac["head_ms"] / n_calls; iso_["replay_ms"] = ac["replay_ms"] / n_calls; iso_["device_ms"] = ac["all_ms"] / n_calls
        return ac["reads"] / e1, e1 / n_calls * 1e3, iso_, ac

    extras = {}
    iso = None
    if rank == 0 and world == 1 and not a.no_extras:
        # (a) one call at a time on one context: no overlap between calls; its per-kernel durations are un-overlapped
        progress("extras: one call at a time, host-buffer entry point")
        n1 = max(4, min(a.steps, 16))
        v1, ms1, iso, _ac1 = one_at_a_time(res_b, batches, n1, out_of(0))
        extras["pipeline1"] = {"value": v1, "unit": "reads/s", "ms_per_step": ms1, "steps": n1,
                               "k_dp_band_ms": iso["k_dp_band"], "k_dp_all_ms": iso["k_dp_all"], "k_dp_band_wide_ms": iso["k_dp_band_wide"],
                               "k_dp_long_ms": iso["k_dp_long"], "head_ms": iso["head_ms"], "replay_ms": iso["replay_ms"],
                               "device_ms": iso["device_ms"],
                               "note": "one call at a time: kernel durations are un-overlapped"}
        # (b) the host-buffer entry point strk_count_loci: the batch starts in pageable host memory and the results end
        # there, every step (sub-batches through four pinned slots and two compute contexts: staging copy, H2D, kernels, D2H overlap)
        nh = max(3, min(a.steps, 8))
        hb = [batch_struct(b) for b in batches[:min(NB, 3)]]
        outs = [np.zeros(n_reads_max, np.int32) for _ in range(4)]
        for w in range(2 + nh):
            if w == 2:
                t2 = time.perf_counter()
                r2 = 0
            s, _keep = hb[w % len(hb)]
            _lib.check(L.strk_count_loci(ctxs[0].handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
            if w >= 2:
                r2 += batches[w % len(hb)].n_reads
        e2 = time.perf_counter() - t2
        bl_h = batches[(1 + nh) % len(hb)]
        extras["h2d_inclusive"] = {"value": r2 / e2, "unit": "reads/s", "ms_per_step": e2 / nh * 1e3, "steps": nh,
                                   "sub_batches_per_step": int(st.n_sub_batches),
                                   "bytes_h2d_per_step": int(bl_h.seqs.nbytes + bl_h.n_reads * 24),
                                   "note": "strk_count_loci with pageable host buffers in and out, one call at a time; inside a call "
                                           "sub-batches of whole loci travel through four pinned slots and two compute contexts (strk_host_pipe.inc); never "
                                           "the headline value"}
        # (c) the same with the caller's bases page-locked (strk_host_register): DMA from where they lie, no staging copy
        progress("extras: host-buffer entry point, page-locked caller arrays")
        t_pin = time.perf_counter()
        pinned = []
        for _s, keep in hb:
            _lib.host_register(keep["seqs"])
            pinned.append(keep["seqs"])
        pin_ms = (time.perf_counter() - t_pin) * 1e3
        for w in range(2 + nh):
            if w == 2:
                t3 = time.perf_counter()
                r3 = 0
            s, _keep = hb[w % len(hb)]
            _lib.check(L.strk_count_loci(ctxs[0].handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st)))
            if w >= 2:
                r3 += batches[w % len(hb)].n_reads
        e3 = time.perf_counter() - t3
        # (d) two callers at once, a context each (the reference runs N worker processes, call_sample.py:414): the tail of one
        # caller's batch overlaps the head of the other's, PCIe is shared
        import threading

        def two_callers(n_calls):
            outs2 = [[np.zeros(n_reads_max, np.int32) for _ in range(4)] for _ in range(2)]
            stats2 = [_lib.StrkStats(), _lib.StrkStats()]
            errs = []

            def work(t, n):
                try:
                    for w in range(n):
                        s_, _k = hb[(w + t) % len(hb)]
                        _lib.check(L.strk_count_loci(ctxs[t].handle, C.byref(s_), C.byref(p), *[o.ctypes.data for o in outs2[t]], C.byref(stats2[t])))
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            for n in (2, n_calls):     # (first round: the second context's pipeline is created)
                th = [threading.Thread(target=work, args=(t, n)) for t in range(2)]
                t0 = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                el = time.perf_counter() - t0
            if errs:
                raise errs[0]
            reads = sum(batches[(w + t) % len(hb)].n_reads for t in range(2) for w in range(n_calls))
            return {"value": reads / el, "unit": "reads/s", "ms_per_call": el / n_calls * 1e3 / 2, "calls": 2 * n_calls}
        if len(ctxs) >= 2:
            progress("extras: host-buffer entry point, two callers")
            extras["h2d_inclusive"]["two_callers_page_locked"] = two_callers(nh)
        for arr in pinned:
            _lib.host_unregister(arr)
        if len(ctxs) >= 2:
            extras["h2d_inclusive"]["two_callers"] = two_callers(nh)
            extras["h2d_inclusive"]["two_callers"]["note"] = ("two threads, a context each, calling strk_count_loci at the same time (pageable "
                                                             "buffers; two_callers_page_locked: registered bases)")
        extras["h2d_inclusive"]["page_locked"] = {
            "value": r3 / e3, "unit": "reads/s", "ms_per_step": e3 / nh * 1e3, "steps": nh,
            "register_ms_per_batch": pin_ms / len(hb),
            "note": "the caller registered its array of bases once (strk_host_register: a reused buffer); the bases of a sub-batch are then "
                    "copied by DMA straight from it (the 24 bytes per read of the other arrays and the results still go through the "
                    "library's pinned blocks)"}
    if e2e_data is not None:
        progress("extras: end to end from files")
        extras["e2e"] = run_e2e(e2e_data)
    fence()

    strong_check = None
    if a.strong and rank == 0:
        # the gathered table of the last step against this rank counting the WHOLE catalog alone, bit for bit
        j = last % G
        if use_dist:
            table = gathered_step_table(gathered.cpu().numpy(), world, G, j, catalog.n_reads)
        else:
            table = gathered_step_table(out_last.cpu().numpy(), 1, 1, 0, catalog.n_reads)
        _t, sb_all = resident(catalog)
        one = torch.zeros((4, catalog.n_reads), dtype=torch.int32, device=dev)
        _lib.check(L.strk_count_loci_device(ctxs[0].handle, C.byref(sb_all), C.byref(p), one[0].data_ptr(), one[1].data_ptr(),
                                            one[2].data_ptr(), one[3].data_ptr(), None, C.byref(st)))
        strong_check = "identical" if np.array_equal(table, one.cpu().numpy()) else "MISMATCH"

    def parity_check(bl, got4):
        """The device's answers (int32[4, >= n]: cn, score, n_iters, start) on the first loci of batch `bl` against the oracle
        (checker only): up to eight loci, within ~1e9 scalar DP cells (long windows: config 5)."""
        import oracle
        n_chk, cells_chk = 0, 0
        while n_chk < min(8, bl.n_loci) and cells_chk < 1e9:
            r0c, r1c = int(bl.read_off[n_chk]), int(bl.read_off[n_chk + 1])
            cells_chk += 9 * float(((bl.nfl[r0c:r1c] + bl.ntr[r0c:r1c] + bl.nfr[r0c:r1c]).astype(np.float64) ** 2).sum())
            n_chk += 1
        chk = bl.locus_slice(0, max(1, n_chk - (1 if cells_chk > 3e9 and n_chk > 1 else 0)))
        verdict = "ok"
        for l in range(chk.n_loci):
            r0, r1_ = int(chk.read_off[l]), int(chk.read_off[l + 1])
            s0 = int(chk.seq_off[r0])
            o = oracle.count_locus(chk.seqs[s0:int(chk.seq_off[r1_])], chk.seq_off[r0:r1_ + 1] - s0, chk.nfl[r0:r1_],
                                   chk.ntr[r0:r1_], chk.nfr[r0:r1_], chk.est_cn[r0:r1_], chk.motif(l))
            for i, k in enumerate(("cn", "score", "n_iters", "start")):
                if not np.array_equal(got4[i, r0:r1_], o[k]):
                    verdict = f"MISMATCH locus {l} field {k}"
        return verdict

    # ---- the other single-GPU configurations of BASELINE.json, timed by this same run (rank 0 of a one-GPU default run) ----
    configs_out = {}
    for name in sorted(sub_batches):
        c = SUB_CONFIGS[name][0]
        progress(f"extras: {name}")
        bl_c = sub_batches[name]
        # contexts of its own: the band of a shared context may be in a cool-down after another configuration's noisy reads
        for cx in ctxs:
            cx.close()
        ctxs = [_lib.Context(local_rank) for _ in range(D)]
        res_c = [resident(b) for b in bl_c]
        rows_c = max(b.n_reads for b in bl_c)
        out_c = torch.zeros((D + 1, NF, rows_c), dtype=torch.int32, device=dev)
        # a new sample: the library forgets the previous configuration's window levels, then primes until its own are quiet
        # (band probation, history-sized grids; 8 -> 6 takes 8 quiet calls, the long-motif buckets' 8 -> 6 -> 5 -> 4 takes 24)
        L.strk_adaptive_reset()
        n_prime = 28 if c in (4, 5) else 12
        for i in range(n_prime):
            _lib.check(L.strk_count_loci_device(ctxs[i % D].handle, C.byref(res_c[i % 2][1]), C.byref(p), out_c[0, 1].data_ptr(),
                                                out_c[0, 2].data_ptr(), out_c[0, 3].data_ptr(), out_c[0, 4].data_ptr(),
                                                C.c_void_p(streams[i % D].cuda_stream), C.byref(st)))
        # (two calls in flight: the first and the last step of the timed region overlap with nothing — enough steps that this
        # ramp is a few per cent: about 100-200 ms of device time per configuration)
        k_steps = {"cfg3": 16, "cfg4": 20, "cfg5": 32, "cfg5_all": 8}.get(name, 12)
        ac_c = new_acc()
        fence()
        t_c = time.perf_counter()
        for i in range(min(D, k_steps)):
            _lib.check(L.strk_submit_loci_device(ctxs[i % D].handle, C.byref(res_c[i % 2][1]), C.byref(p), out_c[i % D, 1].data_ptr(),
                                                 out_c[i % D, 2].data_ptr(), out_c[i % D, 3].data_ptr(), out_c[i % D, 4].data_ptr(),
                                                 C.c_void_p(streams[i % D].cuda_stream)))
        for i in range(k_steps):
            _lib.check(L.strk_finish(ctxs[i % D].handle, C.byref(st)))
            add_stats(ac_c, st, bl_c[i % 2].n_reads, bl_c[i % 2].n_loci)
            j = i + D
            if j < k_steps:
                _lib.check(L.strk_submit_loci_device(ctxs[j % D].handle, C.byref(res_c[j % 2][1]), C.byref(p), out_c[j % D, 1].data_ptr(),
                                                     out_c[j % D, 2].data_ptr(), out_c[j % D, 3].data_ptr(), out_c[j % D, 4].data_ptr(),
                                                     C.c_void_p(streams[j % D].cuda_stream)))
        fence()
        el_c = time.perf_counter() - t_c
        v1c, ms1c, iso_c, _ = one_at_a_time(res_c, bl_c, 2, out_c[D])
        # (the one-at-a-time leg's last call ran batch 1 into out_c[D])
        roof_c = roofline_block("5e" if name == "cfg5" else c, ac_c, iso_c, ac_c["n"], True)
        dp_ms_c = sum(iso_c[k] for k, _ in DP_KERNELS)
        configs_out[name] = {
            "workload": SUB_CONFIGS[name][2], "loci_per_step": bl_c[0].n_loci, "reads_per_step": bl_c[0].n_reads,
            "value": ac_c["reads"] / el_c, "unit": "reads/s", "steps": k_steps, "ms_per_step": el_c / k_steps * 1e3, "calls_in_flight": D,
            "one_call_at_a_time": {"value": v1c, "ms_per_step": ms1c},
            "window": sorted(ac_c["windows"]), "window_by_motif_bucket": bucket_windows(ac_c), "band_reads_per_step": ac_c["band"] / ac_c["n"],
            "band_fallback_per_step": ac_c["band_fb"] / ac_c["n"], "window_miss_reads_per_step": ac_c["misses"] / ac_c["n"],
            "generic_kernel_items_per_step": ac_c["fallback"] / ac_c["n"], "dedup_reads_per_step": ac_c["dedup"] / ac_c["n"],
            "gcups": iso_c["cells"] / max(dp_ms_c, 1e-9) / 1e6, "roofline": roof_c,
            "overlapped_ms": {**{k: ac_c[f] / ac_c["n"] for k, f in DP_KERNELS}, "head": ac_c["head_ms"] / ac_c["n"],
                              "replay": ac_c["replay_ms"] / ac_c["n"], "device": ac_c["all_ms"] / ac_c["n"]},
            "parity_check": parity_check(bl_c[1], out_c[D, 1:5].cpu().numpy()),
        }
        del res_c, out_c
    if configs_out:
        extras["configs"] = configs_out

    if rank == 0:
        # sanity: the timed path's answers on the first loci of the last step agree with the oracle (checker only)
        parity = parity_check(batches[batch_of(last)], out_last[1:5].cpu().numpy())
        n = max(1, acc["n"])
        plain = a.loci is None and not a.no_dedupe and not a.no_band and not a.strong and a.window == 0
        roof = roofline_block(a.config, acc, iso, acc["n"], plain)
        dp_ms_un = sum(iso[k] for k, _ in DP_KERNELS) if iso else None
        b0 = batches[0]
        wl = (f"cfg{a.config} shape" + (" (1000 loci x 30 HiFi reads, motif 3-6 bp, flank 70)" if a.config == 2 else "") +
              (f": ONE catalog of {catalog.n_loci} loci dealt to {world} rank(s) in blocks of <= 200 loci by estimated cells" if a.strong else
               f", {b0.n_loci} loci x {b0.n_reads // max(1, b0.n_loci)} reads per step and GPU, {NB} distinct resident batches in rotation") +
              "; exact scores (banded first pass with exactness certificate, exact fall-back)")
        line = {
            "metric": "reads/sec realigned", "value": reads_all / elapsed, "unit": "reads/s",
            "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": wl, "loci_per_step_per_gpu": b0.n_loci, "reads_per_step_per_gpu": b0.n_reads,
                       "distinct_batches": NB, "window": sorted(acc["windows"]), "window_by_motif_bucket": bucket_windows(acc),
                       "parallelism": f"loci-sharded x{a.gpus}" + (f" + all_gather every {G} steps" if use_dist else ""),
                       "calls_in_flight": D},
            "loci_per_s": loci_all / elapsed,
            "roofline": roof,
            "valu": {"gcups": (iso["cells"] / max(dp_ms_un, 1e-9) / 1e6) if iso else None, "cells_per_step": acc["cells"] / n,
                     "unit": "G cell updates/s: cells the DP kernels executed per call / the summed un-overlapped durations of all five "
                             "DP kernels (one call at a time)"},
            "device_ms_per_step": acc["all_ms"] / n,
            # event spans inside the timed region, where calls_in_flight calls share the device (a span includes the wait for
            # the other call's kernels): where a call's time goes when calls overlap
            "overlapped_ms": {**{k: acc[f] / n for k, f in DP_KERNELS}, "head": acc["head_ms"] / n, "replay": acc["replay_ms"] / n},
            "band_reads_per_step": acc["band"] / n, "band_fallback_per_step": acc["band_fb"] / n, "band": not a.no_band,
            "window_miss_reads_per_step": acc["misses"] / n, "generic_kernel_items_per_step": acc["fallback"] / n,
            "dedup_reads_per_step": acc["dedup"] / n, "dedupe": not a.no_dedupe,
            **extras,
            "parity_check": parity,
            **({"strong_scaling_check": strong_check} if a.strong else {}),
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        if rank == 0 and not a.strong:   # rank 0's block of the last collective, slot of the last step
            j = last % G
            if not torch.equal(gathered[NF * j:NF * j + NF], out_last):
                sys.exit("all_gather self-check failed")
        dist.destroy_process_group()
    if rank == 0 and a.strong and strong_check != "identical":
        sys.exit("strong-scaling table check failed")
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
