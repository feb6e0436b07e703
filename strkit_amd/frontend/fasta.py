"""FASTA access for the reference genome — `ref.fetch(contig, start, end)` of call_locus.py:772 (pysam FastaFile in
the reference).  Whole file in memory, plain or gzip/BGZF-compressed; sized for synthetic test genomes."""
from __future__ import annotations

import gzip

from .loci import resolve_contig

__all__ = ["Fasta", "write_fasta"]


class Fasta:
    def __init__(self, path: str):
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rb") as fh:
            raw = fh.read()
        self.seqs: dict[str, str] = {}
        pos = 0
        n = len(raw)
        while pos < n:                                   # one record per iteration: header line, then the sequence lines
            if raw[pos:pos + 1] != b">":
                nl = raw.find(b"\n", pos)
                pos = n if nl < 0 else nl + 1
                continue
            nl = raw.find(b"\n", pos)
            nl = n if nl < 0 else nl
            name = raw[pos + 1:nl].split()[0].decode() if nl > pos + 1 else ""
            nxt = raw.find(b"\n>", nl)
            end = n if nxt < 0 else nxt + 1
            self.seqs[name] = raw[nl + 1:end].translate(None, b"\r\n \t").decode("ascii")
            pos = end

    @property
    def references(self) -> list[str]:
        return list(self.seqs)

    def array(self, contig: str):
        """The contig as a uint8 array (one byte per base, case kept), for vectorised window gathering; cached."""
        import numpy as np
        name = resolve_contig(self.seqs, contig) or contig
        cache = self.__dict__.setdefault("_arrays", {})
        if name not in cache:
            cache[name] = np.frombuffer(self.seqs[name].encode("ascii"), np.uint8)
        return cache[name]

    def get_reference_length(self, contig: str) -> int:
        return len(self.seqs[resolve_contig(self.seqs, contig) or contig])

    def fetch(self, contig: str, start: int, end: int) -> str:
        seq = self.seqs[resolve_contig(self.seqs, contig) or contig]    # KeyError for an unknown contig (InvalidLocus)
        if start < 0 or start > len(seq):
            raise IndexError(f"{contig}:{start}-{end} out of range")
        return seq[start:end]   # case kept, as pysam does: soft-masked reference stays lower case (docs/output_formats.md:96)


def write_fasta(path: str, seqs: dict[str, str], width: int = 60) -> None:
    with open(path, "w") as fh:
        for name, seq in seqs.items():
            fh.write(f">{name}\n")
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + "\n")
