"""Seeded synthetic inputs for the repeat-count path.

The kernel only ever sees the per-read window ``(flank_left, tr, flank_right)`` that the caller
extracts (reference: strkit/call/call_locus.py:1101-1146), so the generator emits those triples
directly, CSR-packed, together with the per-locus motif and the caller's integer start estimate
``round(len(tr) / len(motif))`` (call_locus.py:1129, cf. :796).

Configs follow BASELINE.json / SURVEY.md §8(d): seed = 0xC0FFEE + config_id.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

__all__ = ["LocusBatch", "make_batch", "make_catalog_batch", "CONFIGS", "make_config"]

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_X = ord("X")


@dataclass
class LocusBatch:
    """CSR-packed (locus, read) inputs.  Read r of the batch owns ``seqs[seq_off[r]:seq_off[r+1]]``
    laid out ``fl | tr | fr``; locus l owns reads ``read_off[l]:read_off[l+1]`` (in caller order) and
    motif ``motifs[motif_off[l]:motif_off[l+1]]``."""

    seqs: np.ndarray  # uint8
    seq_off: np.ndarray  # int64 [n_reads + 1]
    nfl: np.ndarray  # int32 [n_reads]
    ntr: np.ndarray
    nfr: np.ndarray
    est_cn: np.ndarray  # int32 [n_reads]
    read_off: np.ndarray  # int32 [n_loci + 1]
    motifs: np.ndarray  # uint8
    motif_off: np.ndarray  # int32 [n_loci + 1]
    true_cn: np.ndarray = field(default=None)  # int32 [n_reads], generator's allele (not an expected output)

    @property
    def n_reads(self) -> int:
        return len(self.nfl)

    @property
    def n_loci(self) -> int:
        return len(self.read_off) - 1

    def motif(self, l: int) -> str:
        return self.motifs[self.motif_off[l]:self.motif_off[l + 1]].tobytes().decode()

    def read(self, r: int) -> tuple[str, str, str]:
        b = self.seqs[self.seq_off[r]:self.seq_off[r + 1]].tobytes().decode()
        a, t = int(self.nfl[r]), int(self.ntr[r])
        return b[:a], b[a:a + t], b[a + t:]

    def locus_slice(self, lo: int, hi: int) -> "LocusBatch":
        """Loci [lo, hi) as an independent batch (used for sharding)."""
        r0, r1 = int(self.read_off[lo]), int(self.read_off[hi])
        s0, s1 = int(self.seq_off[r0]), int(self.seq_off[r1])
        m0, m1 = int(self.motif_off[lo]), int(self.motif_off[hi])
        return LocusBatch(
            seqs=self.seqs[s0:s1], seq_off=self.seq_off[r0:r1 + 1] - s0, nfl=self.nfl[r0:r1], ntr=self.ntr[r0:r1],
            nfr=self.nfr[r0:r1], est_cn=self.est_cn[r0:r1], read_off=self.read_off[lo:hi + 1] - r0,
            motifs=self.motifs[m0:m1], motif_off=self.motif_off[lo:hi + 1] - m0,
            true_cn=None if self.true_cn is None else self.true_cn[r0:r1])

    @staticmethod
    def concat(parts: "list[LocusBatch]") -> "LocusBatch":
        """The loci of `parts`, in order, as one batch."""
        def offs(arrs, dtype):
            out, base = [np.zeros(1, dtype)], 0
            for a in arrs:
                out.append(a[1:].astype(dtype) + base)
                base += int(a[-1])
            return np.concatenate(out)
        tc = None if any(p.true_cn is None for p in parts) else np.concatenate([p.true_cn for p in parts])
        return LocusBatch(
            seqs=np.concatenate([p.seqs for p in parts]), seq_off=offs([p.seq_off for p in parts], np.int64),
            nfl=np.concatenate([p.nfl for p in parts]), ntr=np.concatenate([p.ntr for p in parts]),
            nfr=np.concatenate([p.nfr for p in parts]), est_cn=np.concatenate([p.est_cn for p in parts]),
            read_off=offs([p.read_off for p in parts], np.int32), motifs=np.concatenate([p.motifs for p in parts]),
            motif_off=offs([p.motif_off for p in parts], np.int32), true_cn=tc)

    def algorithmic_bytes(self) -> int:
        """per read |fl|+|tr|+|fr| + 4 in, 12 out; per locus |motif| + 8."""
        return int(self.seq_off[-1]) + 16 * self.n_reads + int(self.motif_off[-1]) + 8 * self.n_loci

    @staticmethod
    def from_reads(loci: list[tuple[str, list[tuple[str, str, str]]]], est_cns: list[list[int]] | None = None):
        """Build from [(motif, [(fl, tr, fr), ...]), ...]."""
        seqs, seq_off, nfl, ntr, nfr, est, read_off, motifs, motif_off = [], [0], [], [], [], [], [0], [], [0]
        for li, (motif, reads) in enumerate(loci):
            for ri, (fl, tr, fr) in enumerate(reads):
                seqs.append((fl + tr + fr).encode())
                seq_off.append(seq_off[-1] + len(fl) + len(tr) + len(fr))
                nfl.append(len(fl)); ntr.append(len(tr)); nfr.append(len(fr))
                est.append(est_cns[li][ri] if est_cns is not None else round(len(tr) / len(motif)))
            read_off.append(read_off[-1] + len(reads))
            motifs.append(motif.encode())
            motif_off.append(motif_off[-1] + len(motif))
        return LocusBatch(
            seqs=np.frombuffer(b"".join(seqs), np.uint8).copy(), seq_off=np.array(seq_off, np.int64),
            nfl=np.array(nfl, np.int32), ntr=np.array(ntr, np.int32), nfr=np.array(nfr, np.int32),
            est_cn=np.array(est, np.int32), read_off=np.array(read_off, np.int32),
            motifs=np.frombuffer(b"".join(motifs), np.uint8).copy(), motif_off=np.array(motif_off, np.int32))


def _reducible(m: np.ndarray) -> bool:
    n = len(m)
    for p in range(1, n):
        if n % p == 0 and np.array_equal(np.tile(m[:p], n // p), m):
            return True
    return False


def _mutate(rng: np.random.Generator, seq: np.ndarray, sub: float, indel: float, xrate: float) -> np.ndarray:
    """Per-base substitution / single-base insertion / deletion / low-quality 'X' (call_locus.py:79)."""
    n = len(seq)
    if n == 0:
        return seq
    u = rng.random(n)
    p_del, p_ins = indel / 2, indel
    is_del = u < p_del
    is_ins = (u >= p_del) & (u < p_ins)
    is_sub = (u >= p_ins) & (u < p_ins + sub)
    is_x = (u >= p_ins + sub) & (u < p_ins + sub + xrate)
    base = seq.copy()
    if is_sub.any():
        k = int(is_sub.sum())
        base[is_sub] = _BASES[(np.searchsorted(_BASES, seq[is_sub]) + 1 + rng.integers(3, size=k)) % 4]
    base[is_x] = _X
    counts = np.where(is_del, 0, np.where(is_ins, 2, 1))
    out = np.repeat(base, counts)
    if is_ins.any():
        starts = np.cumsum(counts) - counts  # output index of the first copy of each input base
        at = starts[is_ins]
        out[at] = _BASES[rng.integers(4, size=len(at))]  # the inserted base precedes the original one
    return out


def make_batch(seed: int, n_loci: int, reads_per_locus: int, motif_len: tuple[int, int], cn_range: tuple[int, int],
               sub: float, indel: float, xrate: float, flank: int = 70, slip: float = 0.0,
               motif_mix: tuple[tuple[tuple[int, int], float], ...] | None = None) -> LocusBatch:
    """`motif_mix` (optional) = ((length range, probability), ...): the motif length is drawn from a mixture of ranges
    (1-6 bp with probability 0.7, 7-20 bp with 0.3) instead of uniformly from `motif_len`."""
    rng = np.random.default_rng(seed)
    loci = []
    true_cn = []
    for _ in range(n_loci):
        ml_range = motif_len
        if motif_mix is not None:
            u, acc = rng.random(), 0.0
            for rg, pr in motif_mix:
                acc += pr
                ml_range = rg
                if u < acc:
                    break
        while True:
            m = _BASES[rng.integers(4, size=int(rng.integers(ml_range[0], ml_range[1] + 1)))]
            if len(m) == 1 or not _reducible(m):
                break
        ml = len(m)
        while True:  # flanks must not continue the repeat
            fl = _BASES[rng.integers(4, size=flank)]
            if not np.array_equal(fl[-ml:], m):
                break
        while True:
            fr = _BASES[rng.integers(4, size=flank)]
            if not np.array_equal(fr[:ml], m):
                break
        ref_cn = int(rng.integers(cn_range[0], cn_range[1] + 1))
        alleles = [max(1, ref_cn + int(s) * int(rng.geometric(0.5) - 1)) for s in rng.choice((-1, 1), size=2)]
        reads = []
        for r in range(reads_per_locus):
            cn = alleles[r & 1]
            if slip and rng.random() < slip:  # in-tract motif-unit slippage
                cn = max(1, cn + int(rng.choice((-1, 1))))
            tr = np.tile(m, cn)
            reads.append(tuple(_mutate(rng, s, sub, indel, xrate).tobytes().decode() for s in (fl, tr, fr)))
            true_cn.append(cn)
        loci.append((m.tobytes().decode(), reads))
    b = LocusBatch.from_reads(loci)
    b.true_cn = np.array(true_cn, np.int32)
    return b


# code -> member bases, as the reference has it (strkit/iupac.py:9-21, including "D" = A, C, T)
_IUPAC = {"R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC", "B": "CGT", "D": "ACT", "H": "ACT",
          "V": "ACG", "N": "ACGT"}


def make_catalog_batch(loci: list[tuple[str, int]], seed: int, reads_per_locus: int = 30, sub: float = 0.001,
                       indel: float = 0.002, xrate: float = 0.0005, flank: int = 70, slip: float = 0.02) -> LocusBatch:
    """BASELINE.json config 1's shape: loci of a real catalog (`loci` = [(motif, reference tract length), ...], e.g. the
    44 loci of catalogs/pathogenic_assoc.hg38.tsv, whose motifs include IUPAC codes such as AARRG or GCN) over a
    synthetic genome.  Every copy of an IUPAC motif draws its own member bases, the way such repeats vary between
    copies; reads carry the HiFi error model of config 2."""
    rng = np.random.default_rng(seed)
    out, true_cn = [], []
    for motif, ref_len in loci:
        ml = len(motif)
        choices = [np.frombuffer(_IUPAC.get(ch, ch).encode(), np.uint8) for ch in motif]

        def copies(n):
            if n <= 0:
                return np.zeros(0, np.uint8)
            return np.stack([c[rng.integers(len(c), size=n)] for c in choices], axis=1).reshape(-1)

        fl = _BASES[rng.integers(4, size=flank)]
        fr = _BASES[rng.integers(4, size=flank)]
        ref_cn = max(1, round(ref_len / ml))
        alleles = [max(1, ref_cn + int(s) * int(rng.geometric(0.5) - 1)) for s in rng.choice((-1, 1), size=2)]
        tracts = [copies(a) for a in alleles]               # one haplotype sequence per allele
        reads = []
        for r in range(reads_per_locus):
            tr, cn = tracts[r & 1], alleles[r & 1]
            if slip and rng.random() < slip:                 # in-tract motif-unit slippage
                d = int(rng.choice((-1, 1)))
                tr = np.concatenate([tr, copies(1)]) if d > 0 else tr[:max(ml, len(tr) - ml)]
                cn = max(1, cn + d)
            reads.append(tuple(_mutate(rng, x, sub, indel, xrate).tobytes().decode() for x in (fl, tr, fr)))
            true_cn.append(cn)
        out.append((motif, reads))
    b = LocusBatch.from_reads(out)
    b.true_cn = np.array(true_cn, np.int32)
    return b


# BASELINE.json configs.  cfg1 is CPU plumbing only; cfg4/5 are the 8-GPU shapes.
CONFIGS = {
    1: dict(n_loci=44, reads_per_locus=30, motif_len=(3, 6), cn_range=(5, 60), sub=0.001, indel=0.002, xrate=0.0005,
            slip=0.02),
    2: dict(n_loci=1000, reads_per_locus=30, motif_len=(3, 6), cn_range=(5, 60), sub=0.001, indel=0.002,
            xrate=0.0005, slip=0.02),
    3: dict(n_loci=10000, reads_per_locus=20, motif_len=(2, 20), cn_range=(5, 60), sub=0.03, indel=0.04, xrate=0.01),
    4: dict(n_loci=170000, reads_per_locus=30, motif_len=(1, 20), motif_mix=(((1, 6), 0.7), ((7, 20), 0.3)),
            cn_range=(5, 60), sub=0.001, indel=0.002, xrate=0.0005, slip=0.02),
    5: dict(n_loci=2000, reads_per_locus=40, motif_len=(1, 6), cn_range=(50, 2000), sub=0.001, indel=0.002,
            xrate=0.0005, slip=0.02),
}


def make_config(config_id: int, n_loci: int | None = None, seed_shift: int = 0, **over) -> LocusBatch:
    """BASELINE.json config `config_id`; `seed_shift` gives every rank of a multi-GPU run its own loci."""
    kw = dict(CONFIGS[config_id])
    if n_loci is not None:
        kw["n_loci"] = n_loci
    kw.update(over)
    return make_batch(0xC0FFEE + config_id + 7919 * seed_shift, **kw)
