"""Alignment file access and read extraction through the library's host-side entry points (strk_bam_scan,
strk_extract_reads): the per-read work of the front end in C++, as the reference has it in Rust.  `bam.py` /
`extract.py` remain the readable statement of the same rules; tests/test_frontend.py compares the two."""
from __future__ import annotations

import gzip
import struct

import numpy as np

from .. import _lib
from .bam import AlignedSegment, real_cigar
from .loci import resolve_contig

__all__ = ["NativeBam", "extract_reads", "realign_cigar_to_read_alignment", "bgzf_read"]

_SEQ_LUT = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)


def bgzf_read(path: str, threads: int = 0) -> np.ndarray:
    """The decompressed content of a BGZF file (all cores: strk_bgzf_inflate); a plain gzip file goes through Python."""
    comp = np.fromfile(path, np.uint8)
    L = _lib.load()
    n = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, None, 0, 0)
    if n < 0:
        with gzip.open(path, "rb") as fh:
            return np.frombuffer(fh.read(), np.uint8)
    out = np.empty(int(n), np.uint8)
    got = L.strk_bgzf_inflate(comp.ctypes.data, comp.size, out.ctypes.data, out.size, int(threads))
    if got < 0:
        _lib.check(int(got))
    return out


class NativeBam:
    """Decompressed BAM stream + per-record arrays (one C pass); records are addressed by index."""

    def __init__(self, path: str):
        self.data = bgzf_read(path)
        raw = self.data
        if raw[:4].tobytes() != b"BAM\x01":
            raise ValueError(f"{path}: not a BAM file")
        l_text, = struct.unpack_from("<i", raw, 4)
        self.header_text = raw[8:8 + l_text].tobytes().rstrip(b"\0").decode("utf-8", "replace")
        off = 8 + l_text
        n_ref, = struct.unpack_from("<i", raw, off)
        off += 4
        self.contigs: list[tuple[str, int]] = []
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", raw, off)
            name = raw[off + 4:off + 4 + l_name - 1].tobytes().decode()
            l_ref, = struct.unpack_from("<i", raw, off + 4 + l_name)
            self.contigs.append((name, l_ref))
            off += 8 + l_name
        L = _lib.load()
        n = L.strk_bam_scan(raw.ctypes.data, raw.size, off, 0, *([None] * 8))
        if n < 0:
            _lib.check(int(n))
        self.n_records = int(n)
        self.rec_off = np.zeros(n, np.int64)
        a32 = lambda: np.zeros(n, np.int32)  # noqa: E731
        self.tid, self.pos, self.end, self.flag, self.l_seq, self.clip_l, self.clip_r = a32(), a32(), a32(), a32(), a32(), a32(), a32()
        got = L.strk_bam_scan(raw.ctypes.data, raw.size, off, n, self.rec_off.ctypes.data, self.tid.ctypes.data,
                              self.pos.ctypes.data, self.end.ctypes.data, self.flag.ctypes.data, self.l_seq.ctypes.data,
                              self.clip_l.ctypes.data, self.clip_r.ctypes.data)
        if got < 0:
            _lib.check(int(got))
        # coordinate-sorted record indices per contig (mapped records only)
        self._by_contig: dict[str, tuple[np.ndarray, np.ndarray, np.ndarray]] = {}
        mapped = (self.flag & 4) == 0
        for t, (name, _) in enumerate(self.contigs):
            idx = np.nonzero(mapped & (self.tid == t))[0]
            idx = idx[np.argsort(self.pos[idx], kind="stable")]
            self._by_contig[name] = (self.pos[idx], self.end[idx], idx)

    @property
    def references(self) -> list[str]:
        return [c for c, _ in self.contigs]

    def fetch_indices(self, contig: str, start: int, end: int) -> np.ndarray:
        """Indices of the mapped records that overlap [start, end), in coordinate order."""
        contig = resolve_contig(self._by_contig, contig)
        if contig is None:
            return np.zeros(0, np.int64)
        pos, rend, idx = self._by_contig[contig]
        hi = int(np.searchsorted(pos, end, side="left"))
        return idx[:hi][rend[:hi] > start]

    def name(self, i: int) -> str:
        o = int(self.rec_off[i]) + 4
        l_name = int(self.data[o + 8])
        return self.data[o + 32:o + 32 + l_name - 1].tobytes().decode()

    def strand(self, i: int) -> str:
        return "-" if self.flag[i] & 16 else "+"

    def segment(self, i: int) -> AlignedSegment:
        """The record as an AlignedSegment (realignment and tests; the hot loop never builds these)."""
        o = int(self.rec_off[i]) + 4
        raw = self.data
        l_name = int(raw[o + 8])
        n_cig = int(raw[o + 12]) | (int(raw[o + 13]) << 8)
        l_seq = int(self.l_seq[i])
        p = o + 32 + l_name
        cigar = raw[p:p + 4 * n_cig].view(np.uint32).copy() if n_cig else np.zeros(0, np.uint32)
        p += 4 * n_cig
        packed = raw[p:p + (l_seq + 1) // 2]
        nib = np.empty(packed.size * 2, np.uint8)
        nib[0::2] = packed >> 4
        nib[1::2] = packed & 15
        seq = _SEQ_LUT[nib[:l_seq]].tobytes().decode("ascii")
        p += (l_seq + 1) // 2
        qual = raw[p:p + l_seq].copy()
        tid = int(self.tid[i])
        block, = struct.unpack_from("<i", raw, o - 4)
        tags = raw[p + l_seq:o + block].tobytes()
        return AlignedSegment(self.name(i), int(self.flag[i]), self.contigs[tid][0] if tid >= 0 else "*", int(self.pos[i]),
                              int(raw[o + 9]), real_cigar(cigar, l_seq, tags), seq, None if l_seq and qual[0] == 0xFF else qual, tags)

    def soft_clip_overlaps(self, idx: np.ndarray, left_flank_coord: int, right_flank_coord: int) -> np.ndarray:
        """AlignedSegment.soft_clip_overlaps_locus for many records."""
        left = (self.clip_l[idx] > 0) & (self.pos[idx] >= left_flank_coord) & (self.pos[idx] <= right_flank_coord)
        right = (self.clip_r[idx] > 0) & (self.end[idx] >= left_flank_coord) & (self.end[idx] <= right_flank_coord)
        return left | right


def realign_cigar_to_read_alignment(cigar: np.ndarray) -> np.ndarray:
    """CIGAR of strk_realign (reference window as "query", read as "ref", leading free read bases as one D run) turned
    into the read's alignment to the reference: I and D swap, the leading run becomes a soft clip."""
    out = np.asarray(cigar, np.uint32).copy()
    ops = out & 15
    swapped = np.where(ops == 1, 2, np.where(ops == 2, 1, ops)).astype(np.uint32)
    out = (out & ~np.uint32(15)) | swapped
    if out.size and (int(cigar[0]) & 15) == 2:
        out[0] = (out[0] & ~np.uint32(15)) | np.uint32(4)
    return out


def extract_reads(bam: NativeBam, rec_idx: np.ndarray, coords: np.ndarray, flank_size: int, min_avg_phred: int,
                  wildcard_threshold: int = 3, alt: dict[int, tuple[np.ndarray, int]] | None = None) -> dict:
    """strk_extract_reads for items (record index, four locus boundaries).  `alt` maps item number to
    (read-alignment CIGAR, reference start) for realigned reads."""
    n = int(len(rec_idx))
    rec_off = np.ascontiguousarray(bam.rec_off[rec_idx], np.int64)
    coords = np.ascontiguousarray(coords, np.int64).reshape(n, 4)
    status, nfl, ntr, nfr = (np.zeros(n, np.int32) for _ in range(4))
    seq_off = np.zeros(n + 1, np.int64)
    a_cig = a_off = a_start = None
    if alt:
        a_off = np.zeros(n + 1, np.int64)
        a_start = np.zeros(n, np.int64)
        parts = []
        for i in range(n):
            if i in alt:
                parts.append(np.asarray(alt[i][0], np.uint32))
                a_start[i] = alt[i][1]
                a_off[i + 1] = a_off[i] + parts[-1].size
            else:
                a_off[i + 1] = a_off[i]
        a_cig = np.concatenate(parts) if parts else np.zeros(1, np.uint32)
    L = _lib.load()

    def call(seqs_ptr, cap):
        return L.strk_extract_reads(bam.data.ctypes.data, bam.data.size, n, rec_off.ctypes.data, coords.ctypes.data,
                                    a_cig.ctypes.data if a_cig is not None else None,
                                    a_off.ctypes.data if a_off is not None else None,
                                    a_start.ctypes.data if a_start is not None else None,
                                    int(flank_size), int(min_avg_phred), int(wildcard_threshold), status.ctypes.data,
                                    nfl.ctypes.data, ntr.ctypes.data, nfr.ctypes.data, seqs_ptr, cap, seq_off.ctypes.data)

    # size query first: a read may carry an expansion many times the reference window (the flagship case), so the
    # buffer is sized by what the reads actually hold, not by the locus
    _lib.check(call(None, 0))
    cap = int(seq_off[-1]) + 16
    seqs = np.empty(cap, np.uint8)
    _lib.check(call(seqs.ctypes.data, cap))
    return {"status": status, "nfl": nfl, "ntr": ntr, "nfr": nfr, "seqs": seqs[:int(seq_off[-1])], "seq_off": seq_off}
