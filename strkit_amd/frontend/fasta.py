"""FASTA access for the reference genome — `ref.fetch(contig, start, end)` of call_locus.py:772 (pysam FastaFile in
the reference).  Whole file in memory, plain or gzip/BGZF-compressed."""
from __future__ import annotations

import gzip

from .loci import resolve_contig

__all__ = ["Fasta", "write_fasta"]


class Fasta:
    """Contigs as uint8 arrays (one byte per base, case kept): line ends are removed with one vectorised pass per record."""

    def __init__(self, path: str):
        import numpy as np
        if path.endswith(".gz"):
            with gzip.open(path, "rb") as fh:
                raw = np.frombuffer(fh.read(), np.uint8)
        else:
            raw = np.fromfile(path, np.uint8)
        self._arr: dict[str, "np.ndarray"] = {}
        n = raw.size
        # headers: '>' at the start of the file or right after a newline
        gt = np.flatnonzero(raw == 62)
        starts = gt[(gt == 0) | (raw[np.maximum(gt, 1) - 1] == 10)]
        for k, pos in enumerate(starts.tolist()):
            end = int(starts[k + 1]) if k + 1 < len(starts) else n
            rec = raw[pos:end]
            nl = np.flatnonzero(rec[:4096] == 10)
            if nl.size == 0:
                nl = np.flatnonzero(rec == 10)
            h_end = int(nl[0]) if nl.size else rec.size
            words = rec[1:h_end].tobytes().split()
            name = words[0].decode() if words else ""
            self._arr[name] = self._strip_line_ends(rec[h_end + 1:])

    @staticmethod
    def _strip_line_ends(body):
        """Sequence lines of one record without their line ends.  Lines of one width (what every FASTA writer produces) are
        one strided copy; anything else (ragged lines, CR LF, blanks) goes through a mask."""
        import numpy as np
        nl = np.flatnonzero(body[:1 << 16] == 10)
        w = int(nl[0]) if nl.size else body.size
        if 0 < w < body.size:
            n_full = body.size // (w + 1)
            grid = body[:n_full * (w + 1)].reshape(n_full, w + 1)
            tail = body[n_full * (w + 1):]
            tail = tail[:-1] if tail.size and tail[-1] == 10 else tail
            if (grid[:, w] == 10).all() and not (tail == 10).any():
                out = np.concatenate((grid[:, :w].reshape(-1), tail))
                lut = np.zeros(256, bool)
                lut[[9, 10, 13, 32]] = True
                if not lut[out].any():                  # (one pass: no line end or blank is left inside)
                    return out
        return body[(body != 10) & (body != 13) & (body != 32) & (body != 9)]

    @property
    def references(self) -> list[str]:
        return list(self._arr)

    def array(self, contig: str):
        """The contig as a uint8 array (one byte per base, case kept), for vectorised window gathering."""
        return self._arr[resolve_contig(self._arr, contig) or contig]

    def get_reference_length(self, contig: str) -> int:
        return int(self._arr[resolve_contig(self._arr, contig) or contig].size)

    def fetch(self, contig: str, start: int, end: int) -> str:
        seq = self._arr[resolve_contig(self._arr, contig) or contig]    # KeyError for an unknown contig (InvalidLocus)
        if start < 0 or start > seq.size:
            raise IndexError(f"{contig}:{start}-{end} out of range")
        # case kept, as pysam does: soft-masked reference stays lower case (docs/output_formats.md:96)
        return seq[start:max(start, end)].tobytes().decode("ascii")


def write_fasta(path: str, seqs: dict[str, str], width: int = 60) -> None:
    with open(path, "w") as fh:
        for name, seq in seqs.items():
            fh.write(f">{name}\n")
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + "\n")
