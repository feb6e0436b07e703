"""FASTA access for the reference genome — `ref.fetch(contig, start, end)` of call_locus.py:772 (pysam FastaFile in
the reference).  Whole file in memory, plain or gzip/BGZF-compressed; sized for synthetic test genomes."""
from __future__ import annotations

import gzip

from .loci import resolve_contig

__all__ = ["Fasta", "write_fasta"]


class Fasta:
    def __init__(self, path: str):
        opener = gzip.open if path.endswith(".gz") else open
        self.seqs: dict[str, str] = {}
        name, chunks = None, []
        with opener(path, "rt") as fh:
            for line in fh:
                if line.startswith(">"):
                    if name is not None:
                        self.seqs[name] = "".join(chunks)
                    name, chunks = line[1:].split()[0], []
                else:
                    chunks.append(line.strip())
        if name is not None:
            self.seqs[name] = "".join(chunks)

    @property
    def references(self) -> list[str]:
        return list(self.seqs)

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
