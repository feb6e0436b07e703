"""FASTA access for the reference genome — `ref.fetch(contig, start, end)` of call_locus.py:772 (pysam FastaFile in
the reference, i.e. htslib's faidx: offsets by arithmetic on the line width).  Same idea here: a contig whose lines have one
width is never copied, base p lives at byte p + (p // bases_per_line) * line_end_bytes of its record and windows are gathered
straight from the file's bytes.  With a `.fai` next to a plain file (what `samtools faidx` / pysam leave there) the file is
memory-mapped and only the pages of the windows that are asked for are ever read; without one the file is read once, the
records are found and their layout is verified.  Records with ragged lines are stripped into an array of their own;
gzip/BGZF-compressed files are decompressed whole."""
from __future__ import annotations

import gzip
import os

import numpy as np

from .loci import resolve_contig

__all__ = ["Fasta", "write_fasta"]


class _Contig:
    """Bases of one record with lines of one width, addressed in place: `c[idx]` (integer array) and `c[a:b]` give uint8
    arrays of bases, `len(c)` / `c.size` the number of bases."""
    __slots__ = ("raw", "size", "w", "extra")

    def __init__(self, raw, size: int, w: int, stride: int):
        self.raw, self.size, self.w, self.extra = raw, int(size), int(w), int(stride - w)

    def __len__(self) -> int:
        return self.size

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b, step = key.indices(self.size)
            if step != 1:
                raise IndexError("contigs are sliced with step 1")
            if b <= a:
                return np.zeros(0, np.uint8)
            if a // self.w == (b - 1) // self.w:                # inside one line
                o = a + (a // self.w) * self.extra
                return np.array(self.raw[o:o + (b - a)])
            key = np.arange(a, b, dtype=np.int64)
        idx = np.asarray(key, np.int64)
        if idx.size and (int(idx.min()) < 0 or int(idx.max()) >= self.size):
            raise IndexError("base index outside the contig")
        return self.raw[idx + (idx // self.w) * self.extra]

    def tobytes(self) -> bytes:
        return self[0:self.size].tobytes()


def _count_blanks(body, piece: int = 1 << 20) -> int:
    """Bytes <= ' ' (line ends, blanks, tabs), counted piece-wise so that the comparison's temporary stays in cache."""
    return sum(int(np.count_nonzero(body[i:i + piece] <= 32)) for i in range(0, body.size, piece))


class Fasta:
    def __init__(self, path: str):
        self._arr: dict[str, "np.ndarray | _Contig"] = {}
        fai = path + ".fai"
        if not path.endswith(".gz") and os.path.exists(fai) and os.path.getmtime(fai) >= os.path.getmtime(path) and self._open_indexed(path, fai):
            return
        if path.endswith(".gz"):
            with gzip.open(path, "rb") as fh:
                raw = np.frombuffer(fh.read(), np.uint8)
        else:
            raw = np.fromfile(path, np.uint8)
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
            self._arr[name] = self._record_bases(rec[h_end + 1:])

    def _open_indexed(self, path: str, fai: str) -> bool:
        """The `.fai` of samtools faidx: name, bases, offset of the first base, bases per line, bytes per line."""
        size = os.path.getsize(path)
        entries = []
        with open(fai) as fh:
            for line in fh:
                f = line.rstrip("\n").split("\t")
                if len(f) < 5:
                    return False
                name, ln, off, lb, lw = f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])
                if ln < 0 or off < 0 or lb <= 0 and ln > 0 or lw < lb:
                    return False
                last = off + ln + ((ln - 1) // lb) * (lw - lb) if ln > 0 else off
                if last > size:
                    return False                                # an index of another file
                entries.append((name, ln, off, max(lb, 1), max(lw, 1), last))
        if not entries:
            return False
        mm = np.memmap(path, np.uint8, "r")
        for name, ln, off, lb, lw, last in entries:
            self._arr[name] = _Contig(mm[off:last], ln, lb, lw)
        return True

    @staticmethod
    def _record_bases(body):
        """The bases of one record.  Lines of one width (what every FASTA writer produces; LF or CR LF) stay where they are
        once that layout is verified: every line end where the width says, and no other blank anywhere in the record.
        Anything else (ragged lines, blank lines, blanks inside lines) is stripped through a mask."""
        nl = np.flatnonzero(body[:1 << 16] == 10)
        first = int(nl[0]) if nl.size else body.size
        if 0 < first < body.size:
            cr = 1 if first >= 2 and body[first - 1] == 13 else 0
            w, stride = first - cr, first + 1
            n_full = body.size // stride
            tail = body[n_full * stride:]
            tail_ends = 0
            if tail.size and tail[-1] == 10:
                tail_ends = 1 + (1 if cr and tail.size >= 2 and tail[-2] == 13 else 0)
            ok = bool((body[stride - 1::stride][:n_full] == 10).all())
            if ok and cr:
                ok = bool((body[stride - 2::stride][:n_full] == 13).all())
            if ok and tail.size - tail_ends <= w and _count_blanks(body) == n_full * (1 + cr) + tail_ends:
                return _Contig(body, n_full * w + tail.size - tail_ends, w, stride)
        return body[(body != 10) & (body != 13) & (body != 32) & (body != 9)]

    @property
    def references(self) -> list[str]:
        return list(self._arr)

    def array(self, contig: str):
        """The contig's bases (one byte per base, case kept) for vectorised window gathering: `a[index_array]`, `a[i:j]`,
        `len(a)` — a uint8 array, or the in-place view of a record with lines of one width."""
        return self._arr[resolve_contig(self._arr, contig) or contig]

    def get_reference_length(self, contig: str) -> int:
        return int(self._arr[resolve_contig(self._arr, contig) or contig].size)

    def fetch(self, contig: str, start: int, end: int) -> str:
        seq = self._arr[resolve_contig(self._arr, contig) or contig]    # KeyError for an unknown contig (InvalidLocus)
        if start < 0 or start > seq.size:
            raise IndexError(f"{contig}:{start}-{end} out of range")
        # case kept, as pysam does: soft-masked reference stays lower case (docs/output_formats.md:96)
        return seq[start:max(start, end)].tobytes().decode("ascii")


def write_fasta(path: str, seqs: dict[str, str], width: int = 60, index: bool = False) -> None:
    """index=True also writes the `.fai` that `samtools faidx` would."""
    fai = []
    off = 0
    with open(path, "w") as fh:
        for name, seq in seqs.items():
            head = f">{name}\n"
            fh.write(head)
            off += len(head)
            fai.append(f"{name}\t{len(seq)}\t{off}\t{width}\t{width + 1}\n")
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + "\n")
            off += len(seq) + (len(seq) + width - 1) // width
    if index:
        with open(path + ".fai", "w") as fh:
            fh.writelines(fai)
    elif os.path.exists(path + ".fai"):
        os.remove(path + ".fai")                                # never leave an index of an older file behind
