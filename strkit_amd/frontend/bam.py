"""BGZF / BAM reading (and writing, for synthetic test data) without pysam or htslib.

Stands in for the reference's `STRkitBAMReader` / `STRkitAlignedSegment` (strkit_rust_ext, not in its tree): the
attributes below are the ones the per-locus path reads (call_locus.py:837-958,1082-1146; call_sample.py:81-131).
A BAM file is a series of gzip members (BGZF blocks), so Python's `gzip` module reads it; the whole file is parsed
into memory and `fetch` filters by interval — sized for test data, not for a 100 GB alignment file.
"""
from __future__ import annotations

import gzip
import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

from .loci import resolve_contig

__all__ = ["AlignedSegment", "BamFile", "read_bam", "write_bam", "real_cigar", "CIGAR_OPS"]

CIGAR_OPS = "MIDNSHP=X"
_SEQ_CODES = "=ACMGRSVTWYHKDBN"
_SEQ_LUT = np.frombuffer(_SEQ_CODES.encode(), np.uint8)
_CONSUMES_QUERY = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1], bool)
_CONSUMES_REF = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1], bool)
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


@dataclass
class AlignedSegment:
    name: str
    flag: int
    contig: str
    start: int                    # 0-based leftmost reference coordinate
    mapq: int
    cigar: np.ndarray             # uint32, BAM encoding (len << 4 | op)
    query_sequence: str
    query_qualities: np.ndarray | None
    tags: bytes = b""
    end: int = field(init=False)  # reference end, exclusive

    def __post_init__(self):
        ops, lens = self.cigar & 15, self.cigar >> 4
        self.end = self.start + int(lens[_CONSUMES_REF[ops]].sum())

    @property
    def length(self) -> int:
        return len(self.query_sequence)

    @property
    def is_reverse(self) -> bool:
        return bool(self.flag & 16)

    @property
    def is_unmapped(self) -> bool:
        return bool(self.flag & 4)

    @property
    def strand(self) -> str:
        return "-" if self.is_reverse else "+"

    def soft_clips(self) -> tuple[int, int]:
        """(left, right) soft-clip lengths."""
        if self.cigar.size == 0:
            return 0, 0
        first, last = int(self.cigar[0]), int(self.cigar[-1])
        return (first >> 4 if first & 15 == 4 else 0), (last >> 4 if last & 15 == 4 else 0)

    def soft_clip_overlaps_locus(self, locus) -> bool:
        """A soft clip that starts inside the locus + flank window (call_locus.py:860-867): the aligned part of the
        read begins or ends between the flank coordinates while clipped bases hang over."""
        left, right = self.soft_clips()
        return bool((left and locus.left_flank_coord <= self.start <= locus.right_flank_coord)
                    or (right and locus.left_flank_coord <= self.end <= locus.right_flank_coord))


class BamFile:
    def __init__(self, contigs: list[tuple[str, int]], segments: list[AlignedSegment], header_text: str = ""):
        self.contigs = contigs
        self.header_text = header_text
        self.segments = segments
        self._by_contig: dict[str, tuple[np.ndarray, np.ndarray, list[AlignedSegment]]] = {}
        for name, _ in contigs:
            segs = sorted((s for s in segments if s.contig == name and not s.is_unmapped), key=lambda s: s.start)
            self._by_contig[name] = (np.array([s.start for s in segs], np.int64), np.array([s.end for s in segs], np.int64), segs)

    @property
    def references(self) -> list[str]:
        return [c for c, _ in self.contigs]

    def fetch(self, contig: str, start: int, end: int) -> list[AlignedSegment]:
        """Mapped segments that overlap [start, end), in coordinate order."""
        contig = resolve_contig(self._by_contig, contig)
        if contig is None:
            return []
        starts, ends, segs = self._by_contig[contig]
        hi = int(np.searchsorted(starts, end, side="left"))
        return [segs[i] for i in np.nonzero(ends[:hi] > start)[0]]


def real_cigar(cigar: np.ndarray, l_seq: int, tags: bytes) -> np.ndarray:
    """Alignments with more than 65 535 CIGAR operations store the placeholder ``<l_seq>S<ref_len>N`` in the fixed
    field and the real CIGAR in the tag ``CG:B,I`` (SAM specification §4.2.2); everything downstream sees the real one."""
    if not (cigar.size == 2 and l_seq > 0 and int(cigar[0]) & 15 == 4 and int(cigar[0]) >> 4 == l_seq and int(cigar[1]) & 15 == 3):
        return cigar
    t, n = 0, len(tags)
    fixed = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while t + 3 <= n:
        tag, ty = tags[t:t + 2], chr(tags[t + 2])
        v = t + 3
        if ty in fixed:
            size = fixed[ty]
        elif ty in "ZH":
            size = tags.index(b"\0", v) - v + 1
        elif ty == "B":
            sub = chr(tags[v])
            cnt, = struct.unpack_from("<I", tags, v + 1)
            size = 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2}.get(sub, 4)
            if tag == b"CG" and sub == "I":
                return np.frombuffer(tags, np.uint32, cnt, v + 5).copy()
        else:
            break
        t = v + size
    return cigar


def _decode_seq(packed: np.ndarray, l_seq: int) -> str:
    nib = np.empty(packed.size * 2, np.uint8)
    nib[0::2] = packed >> 4
    nib[1::2] = packed & 15
    return _SEQ_LUT[nib[:l_seq]].tobytes().decode("ascii")


def read_bam(path: str) -> BamFile:
    with gzip.open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"BAM\x01":
        raise ValueError(f"{path}: not a BAM file")
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].rstrip(b"\0").decode("utf-8", "replace")
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, off)
    off += 4
    contigs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, off)
        name = data[off + 4:off + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", data, off + 4 + l_name)
        contigs.append((name, l_ref))
        off += 8 + l_name
    segs = []
    while off + 4 <= len(data):
        block_size, = struct.unpack_from("<i", data, off)
        rec = off + 4
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen = struct.unpack_from("<iiBBHHHIiii", data, rec)
        p = rec + 32
        name = data[p:p + l_name - 1].decode()
        p += l_name
        cigar = np.frombuffer(data, np.uint32, n_cig, p).copy()
        p += 4 * n_cig
        seq = _decode_seq(np.frombuffer(data, np.uint8, (l_seq + 1) // 2, p), l_seq)
        p += (l_seq + 1) // 2
        qual = np.frombuffer(data, np.uint8, l_seq, p).copy()
        p += l_seq
        tags = data[p:rec + block_size]
        segs.append(AlignedSegment(name, flag, contigs[ref_id][0] if ref_id >= 0 else "*", pos, mapq,
                                   real_cigar(cigar, l_seq, tags), seq, None if l_seq and qual[0] == 0xFF else qual, tags))
        off = rec + block_size
    return BamFile(contigs, segs, text)


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bgzf_block(chunk: bytes, body: bytes) -> bytes:
    """One BGZF block (SAM specification 4.1) around `body`, the raw deflate stream of `chunk` (at most 64 KiB in all)."""
    bsize = len(body) + 25
    return struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize) + body + struct.pack("<II", zlib.crc32(chunk), len(chunk))


def _bgzf_blocks(data: bytes) -> bytes:
    out = bytearray()
    for i in range(0, len(data), 0xFF00):
        chunk = data[i:i + 0xFF00]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        out += bgzf_block(chunk, comp.compress(chunk) + comp.flush())
    return bytes(out) + _BGZF_EOF


def write_bam(path: str, contigs: list[tuple[str, int]], records: list[dict]) -> None:
    """records: dicts with name, flag, contig, pos, mapq, cigar [(len, op letter)], seq, qual (array or None);
    written in the given order (sort by (contig, pos) for a coordinate-sorted file)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{ln}\n" for n, ln in contigs)
    tid = {n: i for i, (n, _) in enumerate(contigs)}
    buf = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs)))
    for n, ln in contigs:
        buf += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    lut = np.full(256, 15, np.uint8)
    for i, ch in enumerate(_SEQ_CODES):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    for r in records:
        cig = np.array([(ln << 4) | CIGAR_OPS.index(op) for ln, op in r["cigar"]], np.uint32)
        seq = r["seq"]
        if len(cig) > 65535 or r.get("long_cigar"):   # SAM §4.2.2: placeholder in the fixed field, real CIGAR in CG:B,I
            ref_len0 = sum(ln for ln, op in r["cigar"] if op in "MDN=X")
            r = dict(r, tags=r.get("tags", b"") + b"CGBI" + struct.pack("<I", len(cig)) + cig.tobytes())
            cig = np.array([(len(seq) << 4) | 4, (ref_len0 << 4) | 3], np.uint32)
        nib = lut[np.frombuffer(seq.encode("ascii"), np.uint8)]
        if len(seq) & 1:
            nib = np.concatenate((nib, np.zeros(1, np.uint8)))
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)
        qual = r.get("qual")
        qb = bytes([0xFF]) * len(seq) if qual is None else np.asarray(qual, np.uint8).tobytes()
        ref_len = sum(ln for ln, op in r["cigar"] if op in "MDN=X")
        name = r["name"].encode() + b"\0"
        body = struct.pack("<iiBBHHHIiii", tid[r["contig"]], r["pos"], len(name), r.get("mapq", 60),
                           _reg2bin(r["pos"], r["pos"] + max(ref_len, 1)), len(cig), r.get("flag", 0), len(seq), -1, -1, 0)
        body += name + cig.tobytes() + packed.tobytes() + qb + r.get("tags", b"")
        buf += struct.pack("<i", len(body)) + body
    with open(path, "wb") as fh:
        fh.write(_bgzf_blocks(bytes(buf)))
