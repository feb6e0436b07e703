"""Synthetic alignment data set (reference FASTA + catalog BED + coordinate-sorted BAM) with known truth, in the
shape BASELINE.json's configs describe: diploid loci, reads drawn from either allele, sequencing errors recorded in the
CIGAR the way an aligner would report them, optional soft-clipped reads over large expansions (the --realign case,
call_locus.py:860-867).  Test/benchmark input only."""
from __future__ import annotations

import os

import numpy as np

from .bam import write_bam
from .fasta import write_fasta

__all__ = ["make_dataset"]

_A = np.frombuffer(b"ACGT", np.uint8)


def _rand(rng, n) -> str:
    return _A[rng.integers(4, size=n)].tobytes().decode()


def _motif(rng, lo, hi) -> str:
    while True:
        m = _rand(rng, int(rng.integers(lo, hi + 1)))
        if len(set(m)) > 1 and not any(len(m) % p == 0 and m == m[:p] * (len(m) // p) for p in range(1, len(m))):
            return m


def _sequence_read(rng, hap: str, ops_ref: list[tuple[int, str]], sub: float, indel: float):
    """Apply per-base errors to `hap` whose alignment to the reference is `ops_ref` (run-length list over hap/ref
    columns: '=' match run, 'I' bases only in hap, 'D' bases only in ref).  Returns (read, cigar runs).
    Error positions are drawn first (a handful per read at HiFi rates), the read is assembled from the slices
    between them."""
    n = len(hap)
    ev = {}
    if sub > 0 or indel > 0:
        x = rng.random(n)
        for p in np.nonzero(x < sub + indel)[0]:
            ev[int(p)] = 0 if x[p] < sub else (1 if x[p] < sub + indel / 2 else 2)   # 0 sub, 1 deletion, 2 insertion
    out, cig = [], []

    def push(op, k=1):
        if k <= 0:
            return
        if cig and cig[-1][1] == op:
            cig[-1][0] += k
        else:
            cig.append([k, op])

    keys = sorted(ev)
    ki = 0
    pos = 0
    for ln, op in ops_ref:
        if op == "D":
            push("D", ln)
            continue
        end = pos + ln
        while pos < end:
            nxt = keys[ki] if ki < len(keys) and keys[ki] < end else end
            if nxt > pos:                       # clean stretch
                out.append(hap[pos:nxt])
                push(op, nxt - pos)
                pos = nxt
                continue
            ch, kind = hap[pos], ev[pos]
            ki += 1
            pos += 1
            if kind == 0:
                out.append("ACGT".replace(ch, "")[int(rng.integers(3))])
                push("X" if op == "=" else "I")
            elif kind == 1:
                if op == "=":
                    push("D")                   # base missing from the read
            else:
                out.append(ch + "ACGT"[int(rng.integers(4))])
                push(op)
                push("I")
    return "".join(out), [(k, o) for k, o in cig]


def make_dataset(out_dir: str, n_loci: int = 20, reads_per_locus: int = 12, read_len: int = 3000, seed: int = 7,
                 motif_len: tuple[int, int] = (3, 6), cn_range: tuple[int, int] = (8, 40), sub: float = 0.0,
                 indel: float = 0.0, low_qual: float = 0.0, soft_clip_frac: float = 0.0, expansion: int = 0,
                 flank_size: int = 70, spacing: int = 5000) -> dict:
    """Writes ref.fa, loci.bed, reads.bam under out_dir; returns the truth: per locus motif, ref_cn, alleles and
    per read (name -> allele copy number)."""
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    pieces, loci, pos = [], [], 0
    for li in range(n_loci):
        gap = _rand(rng, spacing)
        motif = _motif(rng, *motif_len)
        # flanks must not continue the repeat
        while gap.endswith(motif[-1]):
            gap = gap[:-1] + "ACGT".replace(motif[-1], "")[int(rng.integers(3))]
        ref_cn = int(rng.integers(cn_range[0], cn_range[1] + 1))
        pieces.append(gap)
        pos += len(gap)
        loci.append({"contig": "chr1", "start": pos, "end": pos + ref_cn * len(motif), "motif": motif, "ref_cn": ref_cn})
        pieces.append(motif * ref_cn)
        pos += ref_cn * len(motif)
    tail = _rand(rng, spacing)
    while tail.startswith(loci[-1]["motif"][0]):
        tail = "ACGT".replace(loci[-1]["motif"][0], "")[int(rng.integers(3))] + tail[1:]
    pieces.append(tail)
    # right-flank starts must not continue the repeat either
    genome = "".join(pieces)
    g = list(genome)
    for L in loci:
        if g[L["end"]] == L["motif"][0]:
            g[L["end"]] = "ACGT".replace(L["motif"][0], "")[int(rng.integers(3))]
    genome = "".join(g)
    write_fasta(os.path.join(out_dir, "ref.fa"), {"chr1": genome})
    with open(os.path.join(out_dir, "loci.bed"), "w") as fh:
        fh.write("# synthetic catalog\n")
        for i, L in enumerate(loci):
            last = L["motif"] if i % 2 else f"ID=syn{i};MOTIF={L['motif']}"
            fh.write(f"{L['contig']}\t{L['start']}\t{L['end']}\t{last}\n")
    records, truth = [], []
    for li, L in enumerate(loci):
        m, k = L["motif"], len(L["motif"])
        a1 = max(1, L["ref_cn"] + int(rng.integers(-3, 4)))
        a2 = max(1, L["ref_cn"] + int(rng.integers(-3, 4)) + (expansion if expansion else 0))
        L["alleles"] = (a1, a2)
        reads = {}
        for ri in range(reads_per_locus):
            cn = (a1, a2)[ri % 2]
            left_len = int(rng.integers(flank_size + 200, read_len - flank_size - 200 - cn * k)) if read_len - 2 * flank_size - 400 - cn * k > 0 else flank_size + 200
            start = L["start"] - left_len
            right_len = max(flank_size + 200, read_len - left_len - cn * k)
            left = genome[start:L["start"]]
            right = genome[L["end"]:L["end"] + right_len]
            hap = left + m * cn + right
            d = cn - L["ref_cn"]
            ops = [(len(left), "=")]
            if d >= 0:
                ops += [(L["ref_cn"] * k, "=")] + ([(d * k, "I")] if d else [])
            else:
                ops += [(cn * k, "="), (-d * k, "D")]
            ops += [(len(right), "=")]
            read, cig = _sequence_read(rng, hap, ops, sub, indel)
            qual = rng.integers(20, 41, size=len(read)).astype(np.uint8)
            if low_qual > 0:
                qual[rng.random(len(read)) < low_qual] = 2
            name = f"l{li}_r{ri}"
            flag = 16 if rng.random() < 0.5 else 0
            if soft_clip_frac > 0 and d > 0 and rng.random() < soft_clip_frac:
                # the aligner gave up inside the expansion: keep the left part, soft-clip the rest
                keep_ref = len(left) + L["ref_cn"] * k // 2
                q = r = 0
                new = []
                for n, o in cig:
                    if r >= keep_ref:
                        break
                    take = n
                    if o in "=XD" and r + n > keep_ref:
                        take = keep_ref - r
                    new.append((take, o))
                    if o in "=XI":
                        q += take
                    if o in "=XD":
                        r += take
                cig = new + [(len(read) - q, "S")]
            records.append(dict(name=name, flag=flag, contig="chr1", pos=start, mapq=60, cigar=cig, seq=read, qual=qual))
            reads[name] = cn
        truth.append({"motif": m, "ref_cn": L["ref_cn"], "alleles": (a1, a2), "reads": reads,
                      "start": L["start"], "end": L["end"]})
    records.sort(key=lambda r: r["pos"])
    write_bam(os.path.join(out_dir, "reads.bam"), [("chr1", len(genome))], records)
    return {"loci": truth, "paths": {k: os.path.join(out_dir, v) for k, v in
                                     (("ref", "ref.fa"), ("loci", "loci.bed"), ("bam", "reads.bam"))}}
