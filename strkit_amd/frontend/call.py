"""`strkit call`-shaped driver over the device backend: alignment file + reference + catalog -> per-locus read copy
numbers.  Follows the worker loop of strkit/call/call_sample.py:81-197 (blocks of loci, segments fetched once per
block) and the per-locus path of strkit/call/call_locus.py:700-835 (reference window, reference copy number, adjusted
boundaries), :837-958 (read coordinates, optional realignment), :1082-1161 (triples, start estimates) and
:1172-1283 (adjusted score, filters, read_dict).  Allele calling, SNV phasing and VCF output stay with the reference
(SURVEY.md §8f rank 4): the JSON written here carries the `reads` records they start from.
"""
from __future__ import annotations

import json
import sys
import threading
import time
from dataclasses import dataclass

import numpy as np

from .. import _lib
from ..batch import MIN_READ_ALIGN_SCORE, count_loci, filter_reads
from ..realign import _gate as realign_gate, realign_pairs, realign_reads
from ..repeat_count_params import RepeatCountParams, get_reference_rc_params
from ..repeats import get_ref_repeat_counts, get_ref_repeat_counts_packed
from ..segment import calculate_seq_with_wildcards
from ..synth import LocusBatch
from .bam import BamFile, read_bam
from .extract import (LowMeanBaseQual, MIN_AVG_PHRED, get_read_coords_from_cigar, get_read_coords_from_matched_pairs,
                      get_sequence_data_for_locus)
from .fasta import Fasta
from .loci import Locus, load_loci, parse_loci_bed, resolve_contig
from .native import DeviceBam, IndexedBam, NativeBam, extract_reads, host_header, realign_cigar_to_read_alignment
from .output import read_weights

__all__ = ["CallOptions", "call_sample", "call_locus", "call_blocks", "call_blocks_sharded", "deal_locus_blocks", "write_json", "get_locus_with_ref_data", "get_loci_with_ref_data", "MAX_READS"]

@dataclass
class CallOptions:
    """The knobs of `strkit call` this path honours (strkit/call/params.py:20-50) plus the two semantic switches of the
    read-side counter that the reference's tree does not pin: tools/compare_strkit_json.py sweeps them."""
    flank_size: int = 70
    realign: bool = False
    min_avg_phred: int = MIN_AVG_PHRED
    max_reads: int = 250
    respect_ref: bool = False
    rc_params: RepeatCountParams | None = None
    min_read_align_score: float = MIN_READ_ALIGN_SCORE
    tie_rule: int = _lib.STRK_TIE_FIRST
    end_flags: int = _lib.STRK_SG_ALL
    narrowing: int = _lib.STRK_NARROW_NONE


MAX_READS = 250                 # params.max_reads default (strkit/call/params.py:21)
RESIDENT_FACTOR = 7.5           # device bytes per byte of a BGZF alignment file kept whole in HBM: the compressed bytes + ~6x
                                # decompressed (measured 5.8x on 30x HiFi data) + scan / extraction work buffers
STRK_E_NOMEM = -12
DEFAULT_REF_MAX_ITERS = 250     # call_locus.py:71 default_ref_max_iters (100 there is only the "slow" warning level, :72)
VCF_ANCHOR_SIZE = 5             # params.vcf_anchor_size default


def _ref_window(locus: Locus, ref: Fasta):
    """Reference window of a locus split into flank / tract / flank, or None where the reference raises SkipLocus /
    InvalidLocus (call_locus.py:765-787)."""
    try:
        total = ref.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord + 1)
    except (IndexError, KeyError):
        return None
    off_l, off_r = locus.left_coord - locus.left_flank_coord, locus.right_coord - locus.left_flank_coord
    fl, fr, tr = total[:off_l], total[off_r:-1], total[off_l:off_r]
    if len(fl) < locus.flank_size or len(fr) < locus.flank_size:
        return None                                           # "reference flank size too small"
    n_run = "N" * locus.motif_size
    if fl.endswith(n_run) or fr.startswith(n_run):
        return None                                           # "reference has flanking N[...] sequence"
    return total, fl, tr, fr


def get_loci_with_ref_data(block: list[Locus], ref: Fasta, respect_ref: bool = False, context=None) -> list[dict | None]:
    """call_locus.py:736-835 for a block of loci: reference windows, reference copy numbers by the same counter (all
    loci in one batched library call), boundaries widened by the offsets it found.  None per skipped locus.
    With a `Fasta` the windows of all loci are gathered and handed over as packed arrays (no Python per locus before
    the result records); any other reference object goes through its `fetch`."""
    if isinstance(ref, Fasta) and len(block) > 1:
        return _loci_with_ref_data_packed(block, ref, respect_ref, context)
    windows = [_ref_window(locus, ref) for locus in block]
    jobs, idx = [], []
    for i, (locus, w) in enumerate(zip(block, windows)):
        if w is None:
            continue
        _, fl, tr, fr = w
        est = round(len(tr) / locus.motif_size)
        jobs.append((est, tr, fl, fr, locus.motif, locus.right_coord - locus.left_coord,
                     get_reference_rc_params("repalign", est, DEFAULT_REF_MAX_ITERS)))
        idx.append(i)
    out: list[dict | None] = [None] * len(block)
    for i, ((ref_cn, _), l_off, r_off, _n_is, (fl2, tr2, fr2)) in zip(
            idx, get_ref_repeat_counts(jobs, VCF_ANCHOR_SIZE, respect_ref, context)):
        locus = block[i]
        out[i] = {"ref_cn": ref_cn, "ref_total_seq": windows[i][0], "ref_seq": tr2, "ref_left_flank_seq": fl2,
                  "ref_right_flank_seq": fr2,
                  "left_coord_adj": locus.left_coord if respect_ref else locus.left_coord - max(0, l_off),
                  "right_coord_adj": locus.right_coord if respect_ref else locus.right_coord + max(0, r_off)}
    return out


def _loci_with_ref_data_packed(block: list[Locus], ref: Fasta, respect_ref: bool, context) -> list[dict | None]:
    n = len(block)
    out: list[dict | None] = [None] * n
    lc = np.array([l.left_coord for l in block], np.int64)
    rc = np.array([l.right_coord for l in block], np.int64)
    fs = np.array([l.flank_size for l in block], np.int64)
    mlen = np.array([l.motif_size for l in block], np.int64)
    lfc = np.maximum(0, lc - fs)
    rfc = rc + fs
    contig_of = [l.contig for l in block]
    ok = np.zeros(n, bool)
    clen = np.zeros(n, np.int64)
    arrays = {}
    for c in set(contig_of):
        try:
            arrays[c] = ref.array(c)
        except KeyError:
            arrays[c] = None                                   # "invalid region" (InvalidLocus)
    for i, c in enumerate(contig_of):
        clen[i] = len(arrays[c]) if arrays[c] is not None else -1
    end = np.minimum(rfc + 1, clen)                            # Python slicing of the fetch (call_locus.py:772)
    nfl = lc - lfc
    nfr = (end - 1) - rc                                       # ref_total_seq[off_r:-1]
    ok = (clen >= 0) & (lfc <= clen) & (nfl >= fs) & (nfr >= fs) & (rc > lc)       # "reference flank size too small"
    idx = np.flatnonzero(ok)
    if idx.size == 0:
        return out
    # gather [lfc, end - 1) of every live locus into one flat array (the trailing +1 base is only used by realign)
    lens = (end - 1 - lfc)[idx]
    seq_off = np.concatenate(([0], np.cumsum(lens)))
    seqs = np.empty(int(seq_off[-1]), np.uint8)
    last_base = np.empty(idx.size, np.uint8)                     # the base after each window (ref_total_seq has it)
    by_contig: dict[str, list[int]] = {}
    for k, i in enumerate(idx.tolist()):
        by_contig.setdefault(contig_of[i], []).append(k)
    for c, ks in by_contig.items():
        ks = np.array(ks, np.int64)
        ln = lens[ks]
        src0 = lfc[idx[ks]]
        owner = np.repeat(np.arange(len(ks)), ln)
        within = np.arange(int(ln.sum())) - (np.cumsum(ln) - ln)[owner]
        seqs[seq_off[ks][owner] + within] = arrays[c][src0[owner] + within]
        last_base[ks] = arrays[c][end[idx[ks]] - 1]
    nfl_i, ntr_i, nfr_i = nfl[idx], (rc - lc)[idx], nfr[idx]
    # "reference has flanking N[...] sequence" (call_locus.py:786-787): only loci with an N next to the tract are looked at
    n_code = (ord("N"), ord("n"))
    tr0 = seq_off[:-1] + nfl_i
    sus = np.flatnonzero(np.isin(seqs[tr0 - 1], n_code) | np.isin(seqs[np.minimum(tr0 + ntr_i, len(seqs) - 1)], n_code))
    drop = set()
    for k in sus.tolist():
        i = int(idx[k])
        m_ = int(mlen[i])
        a0 = int(tr0[k])
        fl_s = seqs[int(seq_off[k]):a0].tobytes().decode()
        fr_s = seqs[a0 + int(ntr_i[k]):int(seq_off[k + 1])].tobytes().decode()
        if fl_s.endswith("N" * m_) or fr_s.startswith("N" * m_):
            drop.add(k)
    if drop:
        keep = np.array([k not in drop for k in range(len(idx))])
        return _merge_kept(block, ref, respect_ref, context, idx[keep], out)
    est = np.rint(ntr_i / mlen[idx]).astype(np.int64)           # round(len(ref_seq) / motif_size): half to even
    # get_reference_rc_params (repeat_count_params.py:17-42)
    max_iters = np.where(est >= 2000, 50, np.where(est >= 1000, 150, np.where(est >= 200, 200, DEFAULT_REF_MAX_ITERS)))
    step = np.where(est >= 2000, 15, np.where(est >= 1000, 5, np.where(est >= 200, 3, 1)))
    lsr = np.where(est >= 2000, 1, 3)
    motifs = b"".join(block[i].motif.encode() for i in idx.tolist())
    motif_off = np.concatenate(([0], np.cumsum(mlen[idx])))
    o9 = get_ref_repeat_counts_packed(est, seqs, seq_off, nfl_i, ntr_i, nfr_i, np.frombuffer(motifs, np.uint8), motif_off,
                                      ntr_i, max_iters, lsr, step, VCF_ANCHOR_SIZE, respect_ref, context)
    text = seqs.tobytes().decode("ascii")
    so = seq_off.tolist()
    o9l = o9.tolist()
    last_chr = [chr(x) for x in last_base.tolist()]
    for k, i in enumerate(idx.tolist()):
        locus = block[i]
        cn, _sc, l_off, r_off, _n1, _n2, a, b, _c = o9l[k]
        base = so[k]
        total_end = so[k + 1]
        # the reference's ref_total_seq carries one more base (call_locus.py:770-772)
        out[i] = {"ref_cn": cn, "ref_total_seq": text[base:total_end] + last_chr[k],
                  "ref_seq": text[base + a:base + a + b], "ref_left_flank_seq": text[base:base + a],
                  "ref_right_flank_seq": text[base + a + b:total_end],
                  "left_coord_adj": locus.left_coord if respect_ref else locus.left_coord - max(0, l_off),
                  "right_coord_adj": locus.right_coord if respect_ref else locus.right_coord + max(0, r_off)}
    return out


def _merge_kept(block, ref, respect_ref, context, keep_idx, out):
    """Rare path of the packed reference side: some loci were dropped after the gather (flanking N runs); the kept ones
    are run again as their own block."""
    sub = [block[int(i)] for i in keep_idx]
    res = _loci_with_ref_data_packed(sub, ref, respect_ref, context) if len(sub) > 1 else get_loci_with_ref_data(sub, ref, respect_ref, context)
    for i, r in zip(keep_idx.tolist(), res):
        out[i] = r
    return out


def get_locus_with_ref_data(locus: Locus, ref: Fasta, respect_ref: bool = False, context=None) -> dict | None:
    return get_loci_with_ref_data([locus], ref, respect_ref, context)[0]


def _distributed() -> bool:
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "torch" not in sys.modules:
        return False                      # a plain run never pays for importing torch
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:  # noqa: BLE001
        return False


def _locus_dict(locus: Locus) -> dict:
    """STRkitLocus.to_dict() + the always-present call keys (call_locus.py:1013-1017, json_report.py:69-74)."""
    return {"locus_index": locus.t_idx, "locus_id": locus.locus_id, "contig": locus.contig, "start": locus.left_coord,
            "end": locus.right_coord, "motif": locus.motif, "annotations": [], "assign_method": None, "call": None,
            "call_95_cis": None, "call_99_cis": None}


def call_locus(locus: Locus, bam: BamFile, ref: Fasta, flank_size: int = 70, realign: bool = False,
               min_avg_phred: int = MIN_AVG_PHRED, max_reads: int = MAX_READS, respect_ref: bool = False,
               rc_params: RepeatCountParams | None = None, min_read_align_score: float = MIN_READ_ALIGN_SCORE,
               ctx: _lib.Context | None = None, tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL,
               narrowing: int = _lib.STRK_NARROW_NONE) -> dict:
    """The per-locus entry point (strkit/call/call_locus.py:974-995) over this backend: one locus, its LocusResult
    record up to the read records (a block of one through the same path as call_sample)."""
    opts = CallOptions(flank_size, realign, min_avg_phred, max_reads, respect_ref, rc_params, min_read_align_score, tie_rule, end_flags, narrowing)
    return call_blocks([[locus]], bam, ref, opts, ctx)[0][0]


def call_sample(bam: BamFile | str, ref: Fasta | str, loci_file: str, flank_size: int = 70, realign: bool = False,
                min_avg_phred: int = MIN_AVG_PHRED, max_reads: int = MAX_READS, respect_ref: bool = False,
                sample_id: str | None = None, ctx: _lib.Context | None = None, processes: int = 1,
                rc_params: RepeatCountParams | None = None, min_read_align_score: float = MIN_READ_ALIGN_SCORE,
                tie_rule: int = _lib.STRK_TIE_FIRST, end_flags: int = _lib.STRK_SG_ALL, front_end: str = "auto",
                span_bytes: int = 4 << 30, narrowing: int = _lib.STRK_NARROW_NONE) -> dict:
    """`front_end` (for a `bam` given as a path): "device" = the file is inflated, scanned and cut on the GPU (DeviceBam) —
    whole when the compressed bytes plus their decompressed form (taken as `RESIDENT_FACTOR` times the file) fit into 90 % of
    the device memory that is free right now, else, with a .bai, in spans of at most `span_bytes` compressed bytes that follow
    the catalog; "host" = block-wise through the .bai on the host cores (IndexedBam) or, without an index, the whole stream
    (NativeBam); "auto" = "device" where that is possible, else "host".  A device reader that runs out of memory all the same
    (a file that inflates more than expected) is retried in spans when the file has an index and replaced by the host reader
    when it has none.  Under torch.distributed every rank opens the file on its own GPU and calls its share of the blocks."""
    t_open = time.perf_counter()
    own_reader = isinstance(bam, str)
    if isinstance(bam, str):
        import os
        if front_end not in ("auto", "device", "host"):
            raise ValueError("front_end must be auto, device or host")
        has_index = os.path.exists(bam + ".bai") or os.path.exists(os.path.splitext(bam)[0] + ".bai")
        # (the device of the rank, as _lib.default_context picks it — or the caller's context's: one process per GPU)
        dev = ctx.device if ctx is not None else int(os.environ.get("STRKIT_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        try:
            free_mem = _lib.device_mem(dev)[0]
        except Exception:  # noqa: BLE001  (no device: the reader below reports it)
            free_mem = 0
        small = os.path.getsize(bam) * RESIDENT_FACTOR < 0.9 * free_mem    # compressed + decompressed bytes stay in HBM
        use_device = front_end == "device" or (front_end == "auto" and (small or has_index))
        if use_device:                                            # a larger file goes through HBM span by span (needs the index)
            # The file is opened (read, uploaded, inflated, scanned: reader threads and the GPU, no Python) while this thread
            # loads the catalog and computes the reference side of every locus, which needs neither.  What both need — the
            # contig names — comes from the file's first blocks, inflated here.
            path = bam
            references = [c for c, _ in host_header(path)[1]]
            opened: list = []

            def open_reader():
                t_ = time.perf_counter()
                try:
                    # under torch.distributed a file with an index is read in spans whatever its size: a rank's blocks are one
                    # run of the catalog, so the spans it loads cover its share of the file only
                    whole = (small and not (_distributed() and has_index)) or not has_index
                    try:
                        opened.append(DeviceBam(path, device=dev, span_bytes=None if whole else span_bytes))
                    except _lib.StrkError as e:
                        if e.code != STRK_E_NOMEM:
                            raise
                        # it did not fit after all (a file that inflates more than RESIDENT_FACTOR says): the whole file goes
                        # through HBM in spans when it has an index; without one, or when even a span fails, the host reader
                        if whole and has_index:
                            opened.append(DeviceBam(path, device=dev, span_bytes=span_bytes))
                        else:
                            opened.append(IndexedBam(path) if has_index else NativeBam(path))
                except BaseException as e:  # noqa: BLE001  (handed to the caller's thread below)
                    opened.append(e)
                opened.append(time.perf_counter() - t_)

            opener = threading.Thread(target=open_reader, name="strkit_amd-open")
            opener.start()
            bam = None
        else:
            bam = IndexedBam(bam) if has_index else NativeBam(bam)
    t_open = time.perf_counter() - t_open       # (device reader: replaced below by the time its thread took)
    opts = CallOptions(flank_size, realign, min_avg_phred, max_reads, respect_ref, rc_params, min_read_align_score, tie_rule, end_flags, narrowing)
    try:
        ref = Fasta(ref) if isinstance(ref, str) else ref
        t0 = time.perf_counter()
        # catalog, alignment file and reference may or may not carry the "chr" prefix (call_locus.py:758 normalize_contig):
        # a locus is called when its contig exists, under either spelling, in both files
        both = {c for c in (bam.references if bam is not None else references) if resolve_contig(ref.references, c) is not None}
        blocks = load_loci(loci_file, flank_size, contigs=both, processes=processes)
        tm_pre: dict = {}
        ref_cache = None
        if bam is None and not _distributed():
            ref_cache = ref_side_of_blocks(blocks, ref, opts, ctx or _lib.default_context(), tm_pre)
    except BaseException:
        if bam is None:                          # do not leave a reader (gigabytes of device memory) behind
            opener.join()
            if isinstance(opened[0], DeviceBam):
                opened[0].close()
        raise
    t_wait = 0.0
    if bam is None:                              # the reader, or what kept it from opening
        t_w = time.perf_counter()
        opener.join()
        t_wait = time.perf_counter() - t_w
        if isinstance(opened[0], BaseException):
            raise opened[0]
        bam, t_open = opened[0], opened[1]
    with open(loci_file) as fh:                  # the catalog's size (the lines parse_loci_bed yields), without parsing them again
        n_catalog = sum(1 for raw in fh if raw.strip() and not raw.lstrip().startswith("#"))
    n_loaded = sum(len(b) for b in blocks)
    if n_loaded < n_catalog:
        print(f"strkit_amd: {n_catalog - n_loaded} of {n_catalog} catalog loci lie on contigs that the alignment file or "
              f"the reference does not have; they are not called", file=sys.stderr)

    def run(bl):
        return call_blocks(bl, bam, ref, opts, ctx, ref_cache=ref_cache)

    try:
        if _distributed():          # launched under torch.distributed (one rank per GPU): shard the blocks
            # a reader that loads what its blocks need (spans of the file / blocks through the index) gets ONE run of
            # consecutive blocks: every rank then reads and inflates its own byte range of the file, not the whole of it
            ranged = (isinstance(bam, DeviceBam) and bam.streamed) or isinstance(bam, IndexedBam)
            results, n_depth, tm = call_blocks_sharded(blocks, run, ref, respect_ref, contiguous=ranged)
        else:
            results, n_depth, tm = run(blocks)
    finally:
        fe_kernel_s = bam.kernel_s() if isinstance(bam, DeviceBam) else None
        if own_reader and isinstance(bam, DeviceBam):
            bam.close()             # gigabytes of device memory: not left to the garbage collector
    errors = tm.pop("errors", [])
    tm["ref_side_s"] = tm.get("ref_side_s", 0.0) + tm_pre.get("ref_side_s", 0.0)
    tm["open_s"] = t_open
    if own_reader and isinstance(bam, DeviceBam):
        tm["open_wait_s"] = t_wait              # what this thread still waited for the reader after catalog + reference side
    tm["front_end"] = "device" if isinstance(bam, DeviceBam) else "host"
    if isinstance(bam, DeviceBam) and bam.streamed:      # this rank's own share of the file (compressed bytes read, uploaded, inflated)
        tm["front_end_compressed_mb"] = bam.open_stage_s.get("compressed_mb", 0.0)
        tm["front_end_spans"] = bam.open_stage_s.get("spans", 0)
    if fe_kernel_s is not None:
        tm["front_end_device_s"] = fe_kernel_s      # inflation + record scan + extraction kernels (HIP events)
        tm["open_stage_s"] = dict(getattr(bam, "open_stage_s", {}))
    # same top-level layout as the reference's report (strkit/call/output/json_report.py:37-60,127-154)
    return {"sample_id": sample_id,
            "caller": {"name": "strkit_amd", "version": _lib.load().strk_version().decode()},
            "parameters": {"flank_size": flank_size, "realign": realign, "min_avg_phred": min_avg_phred,
                           "max_reads": max_reads, "respect_ref": respect_ref, "rc_method": "repalign",
                           "min_read_align_score": min_read_align_score, "processes": processes,
                           **({"tie_rule": tie_rule} if tie_rule != _lib.STRK_TIE_FIRST else {}),
                           **({"end_flags": end_flags} if end_flags != _lib.STRK_SG_ALL else {}),
                           **({"narrowing": narrowing} if narrowing != _lib.STRK_NARROW_NONE else {})},
            "contigs": sorted({r["contig"] for r in results}),
            "catalog": {"num_loci": len(results), "num_loci_unknown_contig": n_catalog - n_loaded},
            "results": results,
            "errors": errors,
            "avg_read_depth": n_depth / max(1, sum(1 for r in results if "reads" in r)),
            "runtime": time.perf_counter() - t0, "stage_times": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()}}


def deal_locus_blocks(blocks: list[list[Locus]], world: int, contiguous: bool = False) -> list[list[int]]:
    """Deterministic dealing of locus blocks to `world` ranks (indices into `blocks`), balanced by an estimate of the DP
    work: sum over loci of (tract + flanks) squared.  contiguous = False: longest-processing-time scatter (the best balance; the
    counting path, where a rank holds every read anyway).  contiguous = True: every rank gets ONE run of consecutive blocks
    whose cost is as close to an equal share as a prefix split allows — the file path: a rank then reads, uploads and inflates
    only the byte range of the alignment file its own blocks lie in (the reference's workers take consecutive blocks of a
    contig off one queue, call_sample.py:103-138,414-420)."""
    cost = [sum((l.right_coord - l.left_coord + 2 * l.flank_size) ** 2 for l in blk) for blk in blocks]
    if contiguous:
        total = float(sum(cost)) or 1.0
        owner_c: list[list[int]] = [[] for _ in range(world)]
        acc = 0.0
        for k, c in enumerate(cost):
            # the rank whose share the block's midpoint falls into
            r = min(world - 1, int((acc + c / 2.0) / total * world))
            owner_c[r].append(k)
            acc += c
        return owner_c
    load = [0] * world
    owner: list[list[int]] = [[] for _ in range(world)]
    for k in sorted(range(len(blocks)), key=lambda i: (-cost[i], i)):
        r = load.index(min(load))
        owner[r].append(k)
        load[r] += cost[k]
    return [sorted(o) for o in owner]


_STAGE_KEYS = ("ref_side_s", "realign_s", "extract_s", "count_s", "count_device_s", "report_s", "load_s", "load_wait_s")
_NAME_BYTES = 64          # fixed width of the read-name field of a gathered per-read record (longer names: the field grows)


def _encode_rows(rows: list[dict], errors: list[dict]):
    """Per-locus rows -> fixed-size records: loci int64[n, 6] = (locus_index, status, ref_cn, start_adj, end_adj, n reads)
    with status 0 called / 1 skipped (no reference data) / 2 failed; reads int64[m, 6] = (locus_index, cn, sl, flags,
    sc as float64 bits, w as float64 bits) with flags bit 0 reverse strand, bit 1 realigned, bit 2 sc is None;
    names uint8[m, W]."""
    n_reads = sum(len(r.get("reads") or {}) for r in rows)
    width = max([_NAME_BYTES] + [len(nm.encode()) for r in rows for nm in (r.get("reads") or {})])
    loci = np.zeros((len(rows) + len(errors), 6), np.int64)
    reads = np.zeros((n_reads, 6), np.int64)
    names = np.zeros((n_reads, width), np.uint8)
    k = 0
    for i, r in enumerate(rows):
        called = "ref_cn" in r
        rd = r.get("reads") or {}
        loci[i] = (r["locus_index"], 0 if called else 1, r.get("ref_cn", 0), r.get("start_adj", r["start"]), r.get("end_adj", r["end"]), len(rd))
        for nm, x in rd.items():
            b = nm.encode()
            names[k, :len(b)] = np.frombuffer(b, np.uint8)
            sc = x.get("sc")
            reads[k] = (r["locus_index"], x["cn"], x.get("sl", 0), (x["s"] == "-") | (2 if x.get("realn") else 0) | (4 if sc is None else 0),
                        np.float64(0.0 if sc is None else sc).view(np.int64), np.float64(x["w"]).view(np.int64))
            k += 1
    for i, e in enumerate(errors):
        loci[len(rows) + i] = (e["locus_index"], 2, 0, 0, 0, 0)
    return loci, reads, names


def _decode_rows(loci_by_index: dict, ref: Fasta, respect_ref: bool, loci_t: np.ndarray, reads_t: np.ndarray, names_t: np.ndarray):
    """The inverse of _encode_rows on the gathered tables: rows in catalog order and the failed loci.  The strings of a row
    (reference tract, anchor) are cut from the reference again: get_ref_repeat_count only ever MOVES flank bases into the
    tract (repeats.py:171-176), so the adjusted tract is reference[start_adj:end_adj]."""
    order = np.argsort(reads_t[:, 0], kind="stable") if len(reads_t) else np.zeros(0, np.int64)
    reads_t, names_t = reads_t[order], names_t[order]
    first = np.searchsorted(reads_t[:, 0], loci_t[:, 0], side="left") if len(reads_t) else np.zeros(len(loci_t), np.int64)
    rows, errors = [], []
    for (idx, status, ref_cn, s_adj, e_adj, n_reads), a in sorted(zip(loci_t.tolist(), first.tolist())):
        locus = loci_by_index[idx]
        if status == 2:
            errors.append({"locus_index": idx, "error": "failed on the rank that owned it (see that rank's log)"})
            continue
        if status == 1:
            rows.append(_locus_dict(locus))
            continue
        reads = {}
        for k in range(a, a + n_reads):
            _li, cn, sl, flags, sc_bits, w_bits = reads_t[k].tolist()
            nm = names_t[k].tobytes().rstrip(b"\0").decode()
            reads[nm] = {"s": "-" if flags & 1 else "+", "cn": cn, "w": float(np.int64(w_bits).view(np.float64)),
                         "sc": None if flags & 4 else float(np.int64(sc_bits).view(np.float64)), "sl": sl,
                         **({"realn": True} if flags & 2 else {})}
        rd = {"ref_cn": ref_cn, "left_coord_adj": s_adj, "right_coord_adj": e_adj,
              "ref_seq": ref.fetch(locus.contig, s_adj, e_adj),
              "ref_left_flank_seq": ref.fetch(locus.contig, max(0, s_adj - VCF_ANCHOR_SIZE), s_adj)}
        rows.append(_locus_row(locus, rd, reads, CallOptions(respect_ref=respect_ref)))
    return rows, errors


def _gather_padded(t, dist, device):
    """all_gather of a 2-D table whose first dimension differs between ranks: counts first, then ONE all_gather_into_tensor
    of the tables padded to the largest (fixed-size records: strkit_amd/sharding.py)."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0], t.shape[1]], dtype=torch.int64, device=device)
    ns = torch.zeros(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(ns, n)
    ns = ns.cpu().numpy().reshape(world, 2)
    rows, cols = int(ns[:, 0].max()), int(ns[:, 1].max())
    pad = torch.zeros((max(rows, 1), max(cols, 1)), dtype=t.dtype, device=device)
    pad[:t.shape[0], :t.shape[1]] = t.to(device)
    out = torch.zeros((world * pad.shape[0], pad.shape[1]), dtype=t.dtype, device=device)
    dist.all_gather_into_tensor(out, pad)
    out = out.cpu().numpy().reshape(world, pad.shape[0], pad.shape[1])
    return [out[w, :int(ns[w, 0])] for w in range(world)]


def call_blocks_sharded(blocks, call_fn, ref: Fasta | None = None, respect_ref: bool = False,
                        contiguous: bool = False) -> tuple[list[dict], int, dict]:
    """One process per GPU (`--processes N` of the reference <-> N ranks of a torch.distributed job): every rank calls
    its share of the locus blocks with `call_fn(blocks) -> (results, reads kept, stage times)` and all ranks get the
    merged results ordered by locus index, as the reference's ordered merge does (call_sample.py:195-197,420).
    The only communication is the collection of the results as FIXED-SIZE records — one per locus, one per read, the
    read names as a fixed-width byte field — with all_gather_into_tensor (RCCL over xGMI on the GPU box, gloo in the CPU
    tests); no pickled Python objects cross ranks."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = [blocks[k] for k in deal_locus_blocks(blocks, world, contiguous)[rank]]
    results, _n_depth, tm = call_fn(mine) if mine else ([], 0, {})
    loci_t, reads_t, names_t = _encode_rows(results, tm.get("errors", []))
    g_loci = _gather_padded(torch.from_numpy(loci_t), dist, device)
    g_reads = _gather_padded(torch.from_numpy(reads_t), dist, device)
    g_names = _gather_padded(torch.from_numpy(names_t), dist, device)
    width = max(x.shape[1] for x in g_names)
    g_names = [np.pad(x, ((0, 0), (0, width - x.shape[1]))) for x in g_names]
    stage_t = torch.tensor([[float(tm.get(k, 0.0)) for k in _STAGE_KEYS]], dtype=torch.float64)
    g_stage = np.concatenate(_gather_padded(stage_t, dist, device))
    by_index = {l.t_idx: l for blk in blocks for l in blk}
    merged, errors = _decode_rows(by_index, ref, respect_ref, np.concatenate(g_loci), np.concatenate(g_reads), np.concatenate(g_names))
    stage = {"errors": errors}
    for k, v in zip(_STAGE_KEYS, g_stage.max(axis=0).tolist()):      # ranks run side by side: the slowest one counts
        if v > 0 or k in tm:
            stage[k] = v
    return merged, sum(len(r.get("reads") or {}) for r in merged), stage


def ref_side_of_blocks(blocks, ref: Fasta, opts: CallOptions, ctx, tm: dict) -> dict:
    """Reference side of ALL loci of `blocks`, a few thousand per library call (each of its lock-step rounds is one device launch
    however many loci take part): {id(locus): reference data or None}.  A chunk that fails is left out — call_blocks' per-block
    path computes it again and isolates the locus."""
    ref_cache: dict[int, dict | None] = {}
    flat = [l for blk in blocks for l in blk]
    t_a = time.perf_counter()
    for c0 in range(0, len(flat), 4096):
        chunk = flat[c0:c0 + 4096]
        try:
            for locus, rd in zip(chunk, get_loci_with_ref_data(chunk, ref, opts.respect_ref, ctx)):
                ref_cache[id(locus)] = rd
        except (_lib.StrkError, ValueError):
            pass
    tm["ref_side_s"] = tm.get("ref_side_s", 0.0) + time.perf_counter() - t_a
    return ref_cache


def call_blocks(blocks, bam: BamFile, ref: Fasta, opts: CallOptions | None = None, ctx=None, ref_cache: dict | None = None):
    """Worker loop over blocks of loci (strkit/call/call_sample.py:103-197): (results in locus order, reads kept,
    stage times).  An error of the library inside a block is handled the way the reference's worker handles any
    exception of call_locus (call_sample.py:159-166: logged, the locus is dropped, the run goes on): the block is
    re-run locus by locus so that only the locus that fails is lost; `stage times["errors"]` lists them."""
    opts = opts or CallOptions()
    ctx = ctx or _lib.default_context()
    native = isinstance(bam, (NativeBam, IndexedBam, DeviceBam))
    run_block = _call_block_native if native else _call_block_python
    results: list[dict] = []
    n_depth = 0
    tm = {"ref_side_s": 0.0, "realign_s": 0.0, "extract_s": 0.0, "count_s": 0.0, "errors": []}

    # block-wise access: the first block's records are being inflated while the reference side is computed
    pool = fut = None
    if isinstance(bam, IndexedBam):
        from concurrent.futures import ThreadPoolExecutor
        blocks = [b for blk in blocks for b in _one_contig_blocks(blk)]
        tm["load_s"] = tm["load_wait_s"] = 0.0

        def load(block, slot):
            t0 = time.perf_counter()
            reg = bam.region(block[0].contig, min(l.left_flank_coord for l in block), max(l.right_flank_coord for l in block) + 1,
                             slot=slot)
            return reg, time.perf_counter() - t0

        pool = ThreadPoolExecutor(1)
        fut = pool.submit(load, blocks[0], 0) if blocks else None

    # reference side of ALL loci first (unless the caller has it already)
    if ref_cache is None:
        ref_cache = ref_side_of_blocks(blocks, ref, opts, ctx, tm)

    def safe(block, records):
        nonlocal n_depth
        try:
            known = [ref_cache[id(l)] for l in block] if all(id(l) in ref_cache for l in block) else None
            rows, n = run_block(block, records, ref, opts, ctx, tm, known)
        except _lib.StrkError as e:
            if len(block) > 1:
                for locus in block:
                    safe([locus], records)
                return
            print(f"strkit_amd: {block[0].log_str()} - skipping locus: {e}", file=sys.stderr)
            tm["errors"].append({"locus_index": block[0].t_idx, "error": str(e)})
            return
        results.extend(rows)
        n_depth += n

    # The report is hundreds of thousands of small dicts that reference nothing but strings and numbers: the cyclic collector
    # finds nothing in them and costs a third of the time it takes to build them (it runs every 700 new containers, and its
    # older generations grow with the report).  It is paused while the blocks run.
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    if pool is not None:
        # block-wise access: the records of block k + 1 are inflated (all host cores, outside the GIL) while block k is
        # being called; memory holds three blocks' worth of the alignment file, never the file
        try:
            for k, block in enumerate(blocks):
                t0 = time.perf_counter()
                records, dt = fut.result()
                tm["load_wait_s"] += time.perf_counter() - t0
                tm["load_s"] += dt
                # (three buffers in rotation: the block being called, the one being loaded, and one of slack)
                fut = pool.submit(load, blocks[k + 1], (k + 1) % 3) if k + 1 < len(blocks) else None
                safe(block, records)
        finally:
            pool.shutdown(wait=True)
            if gc_was_on:
                gc.enable()
    elif isinstance(bam, DeviceBam) and bam.streamed:
        # a file larger than device memory: span by span through HBM (DeviceBam.plan / load_span)
        try:
            tm["load_s"] = 0.0
            for contig, beg, end, group in bam.plan([b for blk in blocks for b in _one_contig_blocks(blk)]):
                t0 = time.perf_counter()
                bam.load_span(contig, beg, end)
                tm["load_s"] += time.perf_counter() - t0
                for block in group:
                    safe(block, bam)
        finally:
            if gc_was_on:
                gc.enable()
    else:
        try:
            for block in blocks:
                safe(block, bam)
        finally:
            if gc_was_on:
                gc.enable()
    results.sort(key=lambda r: r["locus_index"])
    return results, n_depth, tm


def _one_contig_blocks(block):
    """A block as the loader builds it stays on one contig (loci.py:277-280); a hand-made one is split."""
    out: list[list[Locus]] = []
    for locus in block:
        if out and out[-1][0].contig == locus.contig:
            out[-1].append(locus)
        else:
            out.append([locus])
    return out


def _empty_counts(n_loci):
    return ({k: np.zeros(0, np.int32) for k in ("cn", "score", "n_iters", "start")},
            {"sc": np.zeros(0), "keep": np.zeros(0, bool), "locus_ok": np.ones(n_loci, bool)})


def _count(batch: LocusBatch, opts: CallOptions, ctx, tm=None):
    if not batch.n_reads:
        return _empty_counts(batch.n_loci)
    out = count_loci(batch, opts.rc_params, ctx=ctx, tie_rule=opts.tie_rule, end_flags=opts.end_flags, narrowing=opts.narrowing, with_stats=tm is not None)
    res = out
    if tm is not None and isinstance(out, tuple):
        res, st = out
        tm["count_device_s"] = tm.get("count_device_s", 0.0) + st["kernel_ms"] / 1e3      # HIP-event time of the device work
    return res, filter_reads(batch, res, opts.min_read_align_score)


def _locus_row(locus: Locus, rd: dict, reads: dict, opts: CallOptions) -> dict:
    row = _locus_dict(locus)
    row["ref_cn"] = int(rd["ref_cn"])
    if not opts.respect_ref:
        row["start_adj"], row["end_adj"] = rd["left_coord_adj"], rd["right_coord_adj"]
    row["ref_start_anchor"] = rd["ref_left_flank_seq"][-VCF_ANCHOR_SIZE:].upper()      # call_locus.py:1350
    row["ref_seq"] = rd["ref_seq"]                                                      # call_locus.py:1351 (case kept)
    # allele calling is not part of this backend: the record stops where call_locus.py:1300 starts
    row["peaks"], row["read_peaks_called"] = None, False
    row["reads"] = reads
    return row


def _call_block_python(block, bam: BamFile, ref: Fasta, opts: CallOptions, ctx, tm, ref_data=None):
    """One block through bam.py / extract.py (the readable statement of the front end): (rows, reads kept)."""
    flank_size = opts.flank_size
    results: list[dict] = []
    n_depth = 0
    prepared = []                     # (locus, ref data, [(segment, query_coords, ref_coords) ...])
    realign_jobs = []                 # (index into prepared, index of the segment)
    t_a = time.perf_counter()
    if ref_data is None:
        ref_data = get_loci_with_ref_data(block, ref, opts.respect_ref, ctx)
    tm["ref_side_s"] += time.perf_counter() - t_a
    for locus, rd in zip(block, ref_data):
        if rd is None:
            results.append(_locus_dict(locus))    # SkipLocus: locus fields + empty call (call_locus.py:1032-1036)
            continue
        segs = bam.fetch(locus.contig, locus.left_flank_coord, locus.right_flank_coord)[:opts.max_reads]
        entries = []
        for seg in segs:
            if opts.realign and seg.soft_clip_overlaps_locus(locus):
                realign_jobs.append((len(prepared), len(entries)))
            entries.append([seg, None, False])
        prepared.append((locus, rd, entries))
    t_a = time.perf_counter()
    if realign_jobs:                  # every soft-clipped read of the block in one device call (realign.py:75-154)
        refs_, reads_, lfcs = [], [], []
        for pi, ei in realign_jobs:
            locus, rd, entries = prepared[pi]
            seg = entries[ei][0]
            refs_.append(rd["ref_total_seq"])
            reads_.append(calculate_seq_with_wildcards(seg.query_sequence, seg.query_qualities, 3))
            lfcs.append(locus.left_flank_coord)
        for (pi, ei), ac in zip(realign_jobs, realign_reads(refs_, reads_, lfcs, flank_size, context=ctx)):
            if ac is not None:
                prepared[pi][2][ei][1] = (ac.query_coords, ac.ref_coords)
                prepared[pi][2][ei][2] = True
    tm["realign_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    # triples of every read of the block -> one batched device call
    loci_reads, meta = [], []
    for locus, rd, entries in prepared:
        triples, names = [], []
        for seg, pairs, realigned in entries:
            if pairs is not None:     # realigned: the pairs of the new alignment
                coords = get_read_coords_from_matched_pairs(locus.left_flank_coord, rd["left_coord_adj"],
                                                            rd["right_coord_adj"], locus.right_flank_coord, *pairs)
            else:
                coords = get_read_coords_from_cigar(locus.left_flank_coord, rd["left_coord_adj"],
                                                    rd["right_coord_adj"], locus.right_flank_coord, seg)
            if coords.is_incomplete():
                continue
            try:
                sd = get_sequence_data_for_locus(seg, coords, flank_size, opts.min_avg_phred)
            except LowMeanBaseQual:
                continue
            triples.append((sd.flank_left_seq_wc[-flank_size:], sd.tr_seq_wc, sd.flank_right_seq_wc[:flank_size]))
            names.append((seg.name, seg.strand, realigned, len(sd.tr_seq)))
        loci_reads.append((locus.motif, triples))
        meta.append(names)
    if not prepared:
        return results, 0
    batch = LocusBatch.from_reads(loci_reads)
    tm["extract_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    res, flt = _count(batch, opts, ctx)
    tm["count_s"] += time.perf_counter() - t_a
    for li, (locus, rd, _) in enumerate(prepared):
        r0, r1 = int(batch.read_off[li]), int(batch.read_off[li + 1])
        kept = [r for r in range(r0, r1) if flt["keep"][r]]
        reads = {}
        # read weights (call_locus.py:1254-1259): from the lengths of ALL segments fetched for the locus
        lens_sorted = np.sort(np.array([e[0].length for e in prepared[li][2]], np.int64))
        tlwf = (batch.nfl[r0:r1] + batch.ntr[r0:r1] + batch.nfr[r0:r1]).astype(np.int64)
        ws = read_weights(lens_sorted, tlwf)
        for r in kept:
            name, strand, realigned, sl = meta[li][r - r0]
            sc = float(flt["sc"][r])
            reads[name] = {"s": strand, "cn": int(res["cn"][r]), "w": float(ws[r - r0]),
                           "sc": None if np.isnan(sc) else sc, "sl": sl, **({"realn": True} if realigned else {})}
        row = _locus_row(locus, rd, reads if flt["locus_ok"][li] else {}, opts)
        n_depth += len(row["reads"])
        results.append(row)
    return results, n_depth


def _call_block_native(block, bam, ref: Fasta, opts: CallOptions, ctx, tm, ref_data=None):
    """One block over the records of a NativeBam / an IndexedBam region / a DeviceBam: no Python per read before the report.
    The overlapping records of all loci come from one vectorised interval query (`fetch_many`), ONE extraction call cuts every
    read of every locus, one device call counts them, numpy filters them (_block_device_stage); only the rows of the report
    are built read by read (_block_report_stage, timed apart as report_s).  (rows, reads kept)"""
    return _block_report_stage(_block_device_stage(block, bam, ref, opts, ctx, tm, ref_data), opts, tm)


def _block_device_stage(block, bam, ref: Fasta, opts: CallOptions, ctx, tm, ref_data=None):
    """Everything of a block that touches the reader and the device: interval query, extraction, counting, filters, the names
    of the reads that are kept.  Returns what the report stage needs (or the finished rows when no locus is live)."""
    flank_size = opts.flank_size
    results: list[dict] = []
    t_a = time.perf_counter()
    if ref_data is None:
        ref_data = get_loci_with_ref_data(block, ref, opts.respect_ref, ctx)
    tm["ref_side_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    live = [(locus, rd) for locus, rd in zip(block, ref_data) if rd is not None]
    results.extend(_locus_dict(locus) for locus, rd in zip(block, ref_data) if rd is None)
    if not live:
        return {"results": results, "live": live}
    lfc = np.array([l.left_flank_coord for l, _ in live], np.int64)
    rfc = np.array([l.right_flank_coord for l, _ in live], np.int64)
    lca = np.array([rd["left_coord_adj"] for _, rd in live], np.int64)
    rca = np.array([rd["right_coord_adj"] for _, rd in live], np.int64)
    contigs = {l.contig for l, _ in live}
    if len(contigs) == 1:
        rec, counts = bam.fetch_many(live[0][0].contig, lfc, rfc, opts.max_reads)
    else:                                    # a hand-made block that mixes contigs
        parts = [bam.fetch_indices(l.contig, int(a), int(b))[:opts.max_reads] for (l, _), a, b in zip(live, lfc, rfc)]
        rec = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        counts = np.array([len(x) for x in parts], np.int64)
    item_locus = np.repeat(np.arange(len(live)), counts)
    coords = np.stack((lfc, lca, rca, rfc), axis=1)[item_locus]
    tm["extract_s"] += time.perf_counter() - t_a
    alt = None
    if opts.realign and rec.size:     # soft-clipped reads of the whole block in one device call (realign.py:75-154)
        t_a = time.perf_counter()
        lf, rf = lfc[item_locus], rfc[item_locus]
        left = (bam.clip_l[rec] > 0) & (bam.pos[rec] >= lf) & (bam.pos[rec] <= rf)
        right = (bam.clip_r[rec] > 0) & (bam.end[rec] >= lf) & (bam.end[rec] <= rf)
        cand = np.nonzero(left | right)[0]
        if cand.size:
            refs_, reads_ = [], []
            for it in cand:
                seg = bam.segment(int(rec[it]))
                refs_.append(live[int(item_locus[it])][1]["ref_total_seq"])
                reads_.append(calculate_seq_with_wildcards(seg.query_sequence, seg.query_qualities, 3))
            gate = realign_gate(flank_size)
            alt = {}
            for it, (sc, _e, cg) in zip(cand, realign_pairs(refs_, reads_, context=ctx)):
                if sc >= gate:
                    alt[int(it)] = (realign_cigar_to_read_alignment(cg), int(lf[it]))
        tm["realign_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    ex = extract_reads(bam, rec, coords, flank_size, opts.min_avg_phred, 3, alt)
    ok = ex["status"] == 0
    n_ok_per_locus = np.bincount(item_locus[ok], minlength=len(live))
    motifs = [l.motif.encode() for l, _ in live]
    mlen = np.array([len(m) for m in motifs], np.int64)
    ntr_ok = ex["ntr"][ok]
    batch = LocusBatch(
        seqs=ex["seqs"], seq_off=np.concatenate(([0], ex["seq_off"][1:][ok])).astype(np.int64),
        nfl=ex["nfl"][ok], ntr=ntr_ok, nfr=ex["nfr"][ok],
        est_cn=np.rint(ntr_ok / mlen[item_locus[ok]]).astype(np.int32),      # round(len(tr) / motif_size), half to even
        read_off=np.concatenate(([0], np.cumsum(n_ok_per_locus))).astype(np.int32),
        motifs=np.frombuffer(b"".join(motifs), np.uint8).copy(),
        motif_off=np.concatenate(([0], np.cumsum(mlen))).astype(np.int32))
    if "d_seqs" in ex:
        batch.d_seqs = ex["d_seqs"]          # extracted on the device: counted where they are
    tm["extract_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    res, flt = _count(batch, opts, ctx, tm)
    tm["count_s"] += time.perf_counter() - t_a
    t_a = time.perf_counter()
    ok_items = np.nonzero(ok)[0]
    read_locus = item_locus[ok]
    keep = flt["keep"] & flt["locus_ok"][read_locus]
    kept = np.nonzero(keep)[0]
    kept_rec = rec[ok_items[kept]]
    st = {"results": results, "live": live, "rec": rec, "counts": counts, "item_locus": item_locus, "ok_items": ok_items,
          "read_locus": read_locus, "kept": kept, "kept_rec": kept_rec, "names": bam.names(kept_rec),
          "minus": (bam.flag[kept_rec] & 16) != 0, "lens_all": bam.l_seq[rec].astype(np.int64), "cn": res["cn"], "sc": flt["sc"],
          "nfl": batch.nfl, "ntr": batch.ntr, "nfr": batch.nfr, "alt": alt}
    tm["names_s"] = tm.get("names_s", 0.0) + time.perf_counter() - t_a
    return st


def _block_report_stage(st, opts: CallOptions, tm):
    """Report rows of a block (call_locus.py:1279-1288,1340-1352) from what _block_device_stage left: Python and numpy only."""
    results, live = st["results"], st["live"]
    if not live:
        return results, 0
    t_a = time.perf_counter()
    rec, counts, item_locus, ok_items, read_locus, kept = (st[k] for k in ("rec", "counts", "item_locus", "ok_items", "read_locus", "kept"))
    names, alt = st["names"], st["alt"]
    n_kept = np.bincount(read_locus[kept], minlength=len(live))
    strands = np.where(st["minus"], "-", "+").tolist()
    cns = st["cn"][kept].tolist()
    scs = [None if x != x else x for x in st["sc"][kept].tolist()]
    sls = st["ntr"][kept].tolist()
    # read weights (call_locus.py:1254-1259, output.read_weights) for all loci at once: the lengths of ALL records fetched
    # for a locus, sorted inside the locus; L = mean length of those that could contain flank + tract + flank
    big = np.int64(1) << 40
    lens_all = st["lens_all"]
    order = np.lexsort((lens_all, item_locus))
    key = item_locus[order] * big + lens_all[order]
    csum = np.concatenate(([0], np.cumsum(lens_all[order])))
    loc_end = np.cumsum(counts)                                   # end of each locus' run in `order`
    tlwf = (st["nfl"][kept].astype(np.int64) + st["ntr"][kept] + st["nfr"][kept])
    part = np.searchsorted(key, read_locus[kept] * big + tlwf, side="left")
    e_ = loc_end[read_locus[kept]]
    L = (csum[e_] - csum[part]) / np.maximum(e_ - part, 1)
    ws = ((L + tlwf - 2.0) / (L - tlwf + 1.0)).tolist()
    realn = [bool(alt) and int(it) in alt for it in ok_items[kept]] if alt else None
    first = np.concatenate(([0], np.cumsum(n_kept))).tolist()
    # the read records of the whole block in one comprehension (values as locals: no indexing), then a dict per locus
    recs = [{"s": s_, "cn": c_, "w": w_, "sc": q_, "sl": l_} for s_, c_, w_, q_, l_ in zip(strands, cns, ws, scs, sls)]
    if realn is not None:
        for k, r_ in enumerate(realn):
            if r_:
                recs[k]["realn"] = True
    for li, (locus, rd) in enumerate(live):
        a, b = first[li], first[li + 1]
        results.append(_locus_row(locus, rd, dict(zip(names[a:b], recs[a:b])), opts))
    tm["report_s"] = tm.get("report_s", 0.0) + time.perf_counter() - t_a
    return results, int(len(kept))


def write_json(report: dict, path: str) -> None:
    with open(path, "w") as fh:
        json.dump(report, fh, indent=1)
