"""Synthetic alignment data at BASELINE.json's stated scale (10 000 loci x 30x HiFi reads of ~15 kb): reference FASTA,
catalog BED, coordinate-sorted BAM and its BAI index, written by all host cores.  Test / benchmark input only.

The small generator (`synth_dataset.make_dataset`) builds reads base by base in Python; at 4.5 G bases that takes hours.
Here a read is a numpy slice of the reference with its allele's tract spliced in, sequencing errors are applied with
vectorised numpy operations (the scheme of `strkit_amd.synth._mutate`), every worker process encodes and BGZF-compresses
the records of its own run of loci, and the parent only concatenates the parts (BGZF blocks are independent gzip
members) and writes the index from the offsets the workers report.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

__all__ = ["make_dataset_large", "write_bai"]

_A = np.frombuffer(b"ACGT", np.uint8)
_NIB = np.zeros(256, np.uint8)
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _NIB[_c] = _i
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_BLOCK = 0xFF00
OP_M, OP_I, OP_D, OP_EQ, OP_X = 0, 1, 2, 7, 8


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _bgzf(data: bytes) -> tuple[bytes, np.ndarray]:
    """BGZF blocks of `data`; returns (compressed bytes, compressed offset of every block)."""
    out = bytearray()
    offs = []
    for i in range(0, len(data), _BLOCK):
        chunk = data[i:i + _BLOCK]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(chunk) + comp.flush()
        offs.append(len(out))
        out += struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(body) + 25)
        out += body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out), np.array(offs, np.int64)


def _errors(rng, hap: np.ndarray, in_ref: np.ndarray, sub: float, indel: float):
    """Sequencing errors on a haplotype.  `in_ref[i]`: hap base i is matched to a reference base (else it is inserted
    sequence).  Returns (read bases, CIGAR op per emitted column, with D columns that emit no base)."""
    n = len(hap)
    u = rng.random(n)
    is_del = u < indel / 2
    is_ins = (u >= indel / 2) & (u < indel)
    is_sub = (u >= indel) & (u < indel + sub)
    base = hap.copy()
    k = int(is_sub.sum())
    if k:
        base[is_sub] = _A[(np.searchsorted(_A, hap[is_sub]) + 1 + rng.integers(3, size=k)) % 4]
    op = np.where(in_ref, np.where(is_sub, OP_X, OP_EQ), OP_I).astype(np.uint8)
    # columns per hap base: normal 1, insertion after the base 2, deletion of a reference-matched base 1 (a D column),
    # deletion of an inserted base 0
    cols = np.where(is_ins, 2, np.where(is_del & ~in_ref, 0, 1))
    ops = np.repeat(op, cols)
    starts = np.cumsum(cols) - cols
    ops[starts[is_del & in_ref]] = OP_D
    ops[starts[is_ins] + 1] = OP_I
    emit = np.repeat(base, cols)
    if is_ins.any():
        emit[starts[is_ins] + 1] = _A[rng.integers(4, size=int(is_ins.sum()))]
    return emit[ops != OP_D], ops


def _rle(ops: np.ndarray) -> np.ndarray:
    if ops.size == 0:
        return np.zeros(0, np.uint32)
    cut = np.flatnonzero(np.diff(ops)) + 1
    starts = np.concatenate(([0], cut))
    lens = np.diff(np.concatenate((starts, [ops.size])))
    return ((lens.astype(np.uint32) << 4) | ops[starts].astype(np.uint32)).astype(np.uint32)


def _worker(args):
    """Records of loci [l0, l1): BGZF part file + what the index needs."""
    (path, genome_path, glen, loci, l0, l1, depth, read_len, seed, sub, indel, low_qual, flank, part) = args
    genome = np.memmap(genome_path, np.uint8, "r", shape=(glen,))
    rng = np.random.default_rng([seed, l0])
    buf = bytearray()
    rec_u, rec_pos, rec_end, truth = [], [], [], []
    for li in range(l0, l1):
        start, end, motif, ref_cn = loci[li]
        m = np.frombuffer(motif.encode(), np.uint8)
        k = len(m)
        a1 = max(1, ref_cn + int(rng.integers(-3, 4)))
        a2 = max(1, ref_cn + int(rng.integers(-3, 4)))
        reads = []
        for ri in range(depth):
            cn = (a1, a2)[ri & 1]
            room = read_len - cn * k - 2 * (flank + 200)
            left_len = flank + 200 + (int(rng.integers(0, room)) if room > 0 else 0)
            right_len = max(flank + 200, read_len - left_len - cn * k)
            reads.append((start - left_len, ri, cn, left_len, right_len))
        reads.sort()
        for pos, ri, cn, left_len, right_len in reads:
            d = cn - ref_cn
            hap = np.concatenate((genome[pos:start], np.tile(m, cn), genome[end:end + right_len]))
            in_ref = np.ones(len(hap), bool)
            if d > 0:
                in_ref[left_len + ref_cn * k:left_len + cn * k] = False
            emit, ops = _errors(rng, hap, in_ref, sub, indel)
            if d < 0:       # contraction: the missing copies are one D run right after the last tract base of the read
                target = left_len + cn * k                       # reference bases consumed before the run
                ref_used = np.cumsum(ops != OP_I)
                at = int(np.searchsorted(ref_used, target, side="left")) + 1 if target > 0 else 0
                ops = np.concatenate((ops[:at], np.full(-d * k, OP_D, np.uint8), ops[at:]))
            cig = _rle(ops)
            l_seq = len(emit)
            qual = np.full(l_seq, 40, np.uint8)
            if low_qual > 0:
                qual[rng.random(l_seq) < low_qual] = 2
            nib = _NIB[emit]
            if l_seq & 1:
                nib = np.concatenate((nib, np.zeros(1, np.uint8)))
            packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)
            ref_len = int(((cig & 15) != OP_I).astype(np.int64) @ (cig >> 4).astype(np.int64))
            name = f"l{li}_r{ri}".encode() + b"\0"
            flag = 16 if rng.random() < 0.5 else 0
            body = struct.pack("<iiBBHHHIiii", 0, pos, len(name), 60, _reg2bin(pos, pos + max(ref_len, 1)), len(cig), flag,
                               l_seq, -1, -1, 0) + name + cig.tobytes() + packed.tobytes() + qual.tobytes()
            rec_u.append(len(buf))
            rec_pos.append(pos)
            rec_end.append(pos + ref_len)
            buf += struct.pack("<i", len(body)) + body
            truth.append((li, ri, cn))
    comp, block_off = _bgzf(bytes(buf))
    with open(part, "wb") as fh:
        fh.write(comp)
    return (len(comp), len(buf), block_off, np.array(rec_u, np.int64), np.array(rec_pos, np.int64), np.array(rec_end, np.int64),
            np.array(truth, np.int32).reshape(-1, 3))


def write_bai(path: str, n_ref: int, tid: np.ndarray, pos: np.ndarray, end: np.ndarray, voff: np.ndarray, voff_end: np.ndarray) -> None:
    """BAI index (SAM specification §5.2) of a coordinate-sorted BAM: per reference the bins with their chunks
    (consecutive records of one bin are merged) and the 16 kb linear index."""
    out = bytearray(b"BAI\x01" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        sel = np.flatnonzero(tid == t)
        bins: dict[int, list] = {}
        n_win = int(end[sel].max() >> 14) + 1 if sel.size else 0
        lin = np.zeros(n_win, np.uint64)
        for i in sel:
            b = _reg2bin(int(pos[i]), int(max(end[i], pos[i] + 1)))
            ch = bins.setdefault(b, [])
            if ch and ch[-1][1] == int(voff[i]):
                ch[-1][1] = int(voff_end[i])
            else:
                ch.append([int(voff[i]), int(voff_end[i])])
            w0, w1 = int(pos[i]) >> 14, (int(max(end[i], pos[i] + 1)) - 1) >> 14
            for w in range(w0, w1 + 1):
                if lin[w] == 0 or voff[i] < lin[w]:
                    lin[w] = voff[i]
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            out += struct.pack("<Ii", b, len(bins[b]))
            for c0, c1 in bins[b]:
                out += struct.pack("<QQ", c0, c1)
        # windows no record overlaps point at the next record (htslib fills them the same way)
        for w in range(n_win - 2, -1, -1):
            if lin[w] == 0:
                lin[w] = lin[w + 1]
        out += struct.pack("<i", n_win) + lin.astype("<u8").tobytes()
    with open(path, "wb") as fh:
        fh.write(out)


def make_dataset_large(out_dir: str, n_loci: int = 10000, depth: int = 30, read_len: int = 15000, seed: int = 1,
                       motif_len: tuple[int, int] = (3, 6), cn_range: tuple[int, int] = (8, 40), sub: float = 0.001,
                       indel: float = 0.002, low_qual: float = 0.0005, flank_size: int = 70, spacing: int | None = None,
                       procs: int | None = None) -> dict:
    """Writes ref.fa (+ .fai), loci.bed, reads.bam and reads.bam.bai under out_dir.  Returns the paths and the truth as arrays:
    `truth[:, 0]` locus, `[:, 1]` read number (name l<locus>_r<read>), `[:, 2]` the copy number of the read's allele."""
    import multiprocessing as mp
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    spacing = spacing or read_len + 5000
    assert spacing > read_len, "reads of neighbouring loci must not interleave"
    # reference: spacing random bases, tract, ... ; the base before and after a tract never continues the repeat
    motifs, ref_cns = [], rng.integers(cn_range[0], cn_range[1] + 1, size=n_loci)
    for _ in range(n_loci):
        while True:
            m = _A[rng.integers(4, size=int(rng.integers(motif_len[0], motif_len[1] + 1)))].tobytes().decode()
            if len(set(m)) > 1 and not any(len(m) % p == 0 and m == m[:p] * (len(m) // p) for p in range(1, len(m))):
                break
        motifs.append(m)
    tract_len = np.array([len(m) for m in motifs]) * ref_cns
    starts = spacing * np.arange(1, n_loci + 1) + np.concatenate(([0], np.cumsum(tract_len)[:-1]))
    glen = int(starts[-1] + tract_len[-1] + spacing)
    gpath = os.path.join(out_dir, "genome.u8")
    genome = np.memmap(gpath, np.uint8, "w+", shape=(glen,))
    genome[:] = _A[rng.integers(4, size=glen, dtype=np.uint8)]
    loci = []
    for i in range(n_loci):
        s, e = int(starts[i]), int(starts[i] + tract_len[i])
        m = np.frombuffer(motifs[i].encode(), np.uint8)
        genome[s:e] = np.tile(m, int(ref_cns[i]))
        if genome[s - 1] == m[-1]:
            genome[s - 1] = _A[(np.searchsorted(_A, m[-1]) + 1) % 4]
        if genome[e] == m[0]:
            genome[e] = _A[(np.searchsorted(_A, m[0]) + 1) % 4]
        loci.append((s, e, motifs[i], int(ref_cns[i])))
    genome.flush()
    # FASTA, 60 columns
    full, rem = divmod(glen, 60)
    with open(os.path.join(out_dir, "ref.fa"), "wb") as fh:
        fh.write(b">chr1\n")
        lines = np.empty((full, 61), np.uint8)
        lines[:, :60] = np.asarray(genome[:full * 60]).reshape(full, 60)
        lines[:, 60] = ord("\n")
        fh.write(lines.tobytes())
        if rem:
            fh.write(np.asarray(genome[full * 60:]).tobytes() + b"\n")
    with open(os.path.join(out_dir, "ref.fa.fai"), "w") as fh:      # what `samtools faidx` writes (the reference needs it too)
        fh.write(f"chr1\t{glen}\t6\t60\t61\n")
    with open(os.path.join(out_dir, "loci.bed"), "w") as fh:
        for i, (s, e, m, _) in enumerate(loci):
            fh.write(f"chr1\t{s}\t{e}\t" + (m if i % 2 else f"ID=syn{i};MOTIF={m}") + "\n")
    # records, in parallel over runs of loci
    procs = procs or max(1, min(16, len(os.sched_getaffinity(0))))
    per = max(1, min(64, (n_loci + procs * 4 - 1) // (procs * 4)))
    jobs = []
    for j, l0 in enumerate(range(0, n_loci, per)):
        jobs.append((None, gpath, glen, loci, l0, min(n_loci, l0 + per), depth, read_len, seed, sub, indel, low_qual, flank_size,
                     os.path.join(out_dir, f"part{j:05d}.bgzf")))
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_worker, jobs)
    text = "@HD\tVN:1.6\tSO:coordinate\n" + f"@SQ\tSN:chr1\tLN:{glen}\n"
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", glen)
    head_comp, _ = _bgzf(head)
    bam_path = os.path.join(out_dir, "reads.bam")
    coff = len(head_comp)
    voff, voff_end, pos_all, end_all, truth = [], [], [], [], []
    with open(bam_path, "wb") as out:
        out.write(head_comp)
        for job, (n_comp, n_raw, block_off, rec_u, rec_pos, rec_end, tr) in zip(jobs, parts):
            with open(job[-1], "rb") as fh:
                out.write(fh.read())
            os.remove(job[-1])
            blk = rec_u // _BLOCK
            v = ((block_off[blk] + coff) << 16) | (rec_u % _BLOCK)
            nxt = np.concatenate((rec_u[1:], [n_raw]))
            # end of a record = start of the next one (the last one of a part ends where the next part begins)
            nb = np.minimum(nxt // _BLOCK, len(block_off) - 1)
            ve = np.where(nxt < n_raw, ((block_off[nb] + coff) << 16) | (nxt % _BLOCK), (coff + n_comp) << 16)
            voff.append(v); voff_end.append(ve); pos_all.append(rec_pos); end_all.append(rec_end); truth.append(tr)
            coff += n_comp
        out.write(_BGZF_EOF)
    voff, voff_end = np.concatenate(voff), np.concatenate(voff_end)
    pos_all, end_all = np.concatenate(pos_all), np.concatenate(end_all)
    write_bai(bam_path + ".bai", 1, np.zeros(len(pos_all), np.int32), pos_all, end_all, voff, voff_end)
    del genome
    os.remove(gpath)
    return {"paths": {"ref": os.path.join(out_dir, "ref.fa"), "loci": os.path.join(out_dir, "loci.bed"), "bam": bam_path,
                      "bai": bam_path + ".bai"},
            "truth": np.concatenate(truth), "ref_cn": np.asarray(ref_cns, np.int32), "motifs": motifs, "n_reads": int(len(pos_all)),
            "genome_len": glen}
