"""Alignment file access and read extraction through the library's host-side entry points (strk_bam_scan,
strk_extract_reads): the per-read work of the front end in C++, as the reference has it in Rust.  `bam.py` /
`extract.py` remain the readable statement of the same rules; tests/test_frontend.py compares the two."""
from __future__ import annotations

import ctypes as C
import gzip
import os
import struct
import time

import numpy as np

from .. import _lib
from .bam import AlignedSegment, real_cigar
from .loci import resolve_contig

__all__ = ["NativeBam", "IndexedBam", "DeviceBam", "extract_reads", "realign_cigar_to_read_alignment", "bgzf_read"]

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


class _Records:
    """Per-record arrays over a decompressed stretch of a BAM stream (one C pass, strk_bam_scan[_piece]); records are
    addressed by index.  Base of NativeBam (whole file) and of the regions an IndexedBam hands out."""

    data: np.ndarray
    contigs: list

    def _set_arrays(self, n: int):
        self.n_records = int(n)
        self.rec_off = np.zeros(n, np.int64)
        a32 = lambda: np.zeros(n, np.int32)  # noqa: E731
        self.tid, self.pos, self.end, self.flag, self.l_seq, self.clip_l, self.clip_r = a32(), a32(), a32(), a32(), a32(), a32(), a32()

    def _scan_ptrs(self):
        return [a.ctypes.data for a in (self.rec_off, self.tid, self.pos, self.end, self.flag, self.l_seq, self.clip_l, self.clip_r)]

    def _trim(self, n: int):
        for k in ("rec_off", "tid", "pos", "end", "flag", "l_seq", "clip_l", "clip_r"):
            setattr(self, k, getattr(self, k)[:n])
        self.n_records = int(n)

    def _build_index(self):
        # coordinate-sorted record indices per contig (mapped records only)
        self._by_contig: dict[str, tuple[np.ndarray, np.ndarray, np.ndarray]] = {}
        mapped = (self.flag & 4) == 0
        tids = np.unique(self.tid[mapped]) if self.n_records else []
        for t in tids:
            if t < 0 or t >= len(self.contigs):
                continue
            idx = np.nonzero(mapped & (self.tid == t))[0]
            idx = idx[np.argsort(self.pos[idx], kind="stable")]
            self._by_contig[self.contigs[int(t)][0]] = (self.pos[idx], self.end[idx], idx)

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

    def fetch_many(self, contig: str, starts: np.ndarray, ends: np.ndarray, max_reads: int) -> tuple[np.ndarray, np.ndarray]:
        """fetch_indices for many intervals of one contig at once (no Python per interval): (record indices of all
        intervals concatenated, number of records per interval), at most `max_reads` per interval, coordinate order."""
        starts, ends = np.asarray(starts, np.int64), np.asarray(ends, np.int64)
        contig = resolve_contig(self._by_contig, contig)
        if contig is None or len(starts) == 0:
            return np.zeros(0, np.int64), np.zeros(len(starts), np.int64)
        pos, rend, idx = self._by_contig[contig]
        hi = np.searchsorted(pos, ends, side="left")
        span = int((rend - pos).max()) if len(pos) else 0            # no record is longer than this on the reference
        lo = np.searchsorted(pos, starts - span, side="left")
        cnt = np.maximum(hi - lo, 0)
        tot = int(cnt.sum())
        owner = np.repeat(np.arange(len(starts)), cnt)
        base = np.cumsum(cnt) - cnt
        k = lo[owner] + (np.arange(tot) - base[owner])
        keep = rend[k] > starts[owner]
        owner, k = owner[keep], k[keep]
        n_per = np.bincount(owner, minlength=len(starts))
        if n_per.size and n_per.max() > max_reads:                   # "using the first max_reads" (call_locus.py:1056-1058)
            first = np.cumsum(n_per) - n_per
            rank = np.arange(len(owner)) - first[owner]
            sel = rank < max_reads
            owner, k = owner[sel], k[sel]
            n_per = np.minimum(n_per, max_reads)
        return idx[k], n_per

    def name(self, i: int) -> str:
        o = int(self.rec_off[i]) + 4
        l_name = int(self.data[o + 8])
        return self.data[o + 32:o + 32 + l_name - 1].tobytes().decode()

    def names(self, idx: np.ndarray) -> list[str]:
        """Read names of many records (one library call)."""
        n = int(len(idx))
        if n == 0:
            return []
        L = _lib.load()
        rec_off = np.ascontiguousarray(self.rec_off[idx], np.int64)
        off = np.zeros(n + 1, np.int64)
        tot = L.strk_bam_names(self.data.ctypes.data, self.data.size, n, rec_off.ctypes.data, None, 0, off.ctypes.data)
        if tot < 0:
            _lib.check(int(tot))
        buf = np.empty(max(int(tot), 1), np.uint8)
        tot = L.strk_bam_names(self.data.ctypes.data, self.data.size, n, rec_off.ctypes.data, buf.ctypes.data, buf.size, off.ctypes.data)
        if tot < 0:
            _lib.check(int(tot))
        text = buf[:int(tot)].tobytes().decode()
        o = off.tolist()
        return [text[o[i]:o[i + 1]] for i in range(n)]

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


def _parse_header(raw: np.ndarray) -> tuple[str, list[tuple[str, int]], int]:
    """(header text, contigs, offset of the first alignment record) of a decompressed BAM stream (or its beginning)."""
    if raw[:4].tobytes() != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack_from("<i", raw, 4)
    text = raw[8:8 + l_text].tobytes().rstrip(b"\0").decode("utf-8", "replace")
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, off)
    off += 4
    contigs: list[tuple[str, int]] = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, off)
        name = raw[off + 4:off + 4 + l_name - 1].tobytes().decode()
        l_ref, = struct.unpack_from("<i", raw, off + 4 + l_name)
        contigs.append((name, l_ref))
        off += 8 + l_name
    return text, contigs, off


class NativeBam(_Records):
    """The whole decompressed BAM stream in memory (all cores inflate it) + per-record arrays."""

    def __init__(self, path: str):
        self.data = bgzf_read(path)
        raw = self.data
        try:
            self.header_text, self.contigs, off = _parse_header(raw)
        except ValueError:
            raise ValueError(f"{path}: not a BAM file") from None
        L = _lib.load()
        n = L.strk_bam_scan(raw.ctypes.data, raw.size, off, 0, *([None] * 8))
        if n < 0:
            _lib.check(int(n))
        self._set_arrays(int(n))
        got = L.strk_bam_scan(raw.ctypes.data, raw.size, off, n, *self._scan_ptrs())
        if got < 0:
            _lib.check(int(got))
        self._build_index()


class _Region(_Records):
    pass


def host_header(path: str, comp: np.ndarray | None = None) -> tuple[str, list[tuple[str, int]], int]:
    """(header text, contigs, offset of the first record in the decompressed stream) of a BAM file, from its first BGZF blocks
    inflated on the host: what a caller needs (the contig names) before any reader has opened the file."""
    comp = np.memmap(path, np.uint8, "r") if comp is None else comp
    L = _lib.load()
    head = np.empty(1 << 20, np.uint8)
    nxt = C.c_int64(0)
    while True:                                  # the header may span several blocks
        got = L.strk_bgzf_inflate_range(comp.ctypes.data, comp.size, 0, head.ctypes.data, head.size, C.byref(nxt), 1)
        if got < 0:
            _lib.check(int(got))
        try:
            return _parse_header(head[:int(got)])
        except (struct.error, IndexError, ValueError):
            if nxt.value >= comp.size or head.size > (1 << 28):
                raise ValueError(f"{path}: not a BAM file") from None
            head = np.empty(head.size * 4, np.uint8)


class IndexedBam:
    """Block-wise access to a coordinate-sorted BAM through its .bai index (SAM specification §5.2), the way the reference's
    reader fetches one block of loci at a time (call_sample.py:121): the compressed file is memory-mapped, `region()`
    starts at the virtual offset the 16 kb linear index gives for the region's first window, inflates consecutive BGZF
    blocks on all cores (strk_bgzf_inflate_range) and stops at the first record that starts past the region.  Memory
    use is one region, not the file."""

    def __init__(self, path: str, index: str | None = None):
        self.path = path
        self.comp = np.memmap(path, np.uint8, "r")
        self.header_text, self.contigs, self._first = host_header(path, self.comp)
        index = index or (path + ".bai" if os.path.exists(path + ".bai") else os.path.splitext(path)[0] + ".bai")
        self._lin = self._read_bai(index, len(self.contigs), path)

    @staticmethod
    def _read_bai(path: str, n_contigs: int, bam_path: str | None = None) -> list[np.ndarray]:
        # An index older than its alignment file MAY describe another file's record chain, but a plain `cp`, an rsync without
        # -t or a download in the other order look the same: warn, as htslib does, and go on.  A stale or foreign index is
        # caught where it matters: a record chain that does not end on the next entry point (k_dbam_scan's chain-end status,
        # the host scan's bounds checks).
        if bam_path is not None and os.path.getmtime(path) + 1.0 < os.path.getmtime(bam_path):
            import warnings
            warnings.warn(f"{path} is older than {bam_path}: the index may be stale (re-index the alignment file if records go missing)",
                          RuntimeWarning, stacklevel=3)
        raw = np.fromfile(path, np.uint8)
        if raw[:4].tobytes() != b"BAI\x01":
            raise ValueError(f"{path}: not a BAI index")
        n_ref, = struct.unpack_from("<i", raw, 4)
        off = 8
        lin = []
        for _ in range(n_ref):
            n_bin, = struct.unpack_from("<i", raw, off)
            off += 4
            for _b in range(n_bin):
                _bin, n_chunk = struct.unpack_from("<Ii", raw, off)
                off += 8 + 16 * n_chunk
            n_intv, = struct.unpack_from("<i", raw, off)
            off += 4
            lin.append(raw[off:off + 8 * n_intv].view("<u8").astype(np.uint64))
            off += 8 * n_intv
        return lin

    @property
    def references(self) -> list[str]:
        return [c for c, _ in self.contigs]

    def region(self, contig: str, beg: int, end: int, threads: int = 0, slot: int | None = None) -> _Region:
        """The records that can overlap [beg, end) on `contig` (plus, possibly, a few in front of it).
        `slot`: the decompressed bytes go into the object's buffer of that number, which is kept and used again by the next
        region of the same slot — a caller that walks a file block by block (call_blocks: three slots in rotation) then touches
        fresh memory only while the buffers grow to the size of its largest block; first-touch page faults of a new buffer per
        block cost more than the inflation itself.  A region of a slot is valid until the next region of that slot."""
        L = _lib.load()
        out = _Region()
        out.contigs = self.contigs
        names = self.references
        name = resolve_contig(names, contig)
        tid = names.index(name) if name is not None else -1
        lin = self._lin[tid] if 0 <= tid < len(self._lin) else np.zeros(0, np.uint64)
        w = max(0, int(beg)) >> 14
        nz = np.flatnonzero(lin[w:]) if w < len(lin) else np.zeros(0, np.int64)
        if tid < 0 or nz.size == 0:
            out.data = np.zeros(16, np.uint8)
            out._set_arrays(0)
            out._build_index()
            return out
        voff = int(lin[w + int(nz[0])])
        coff, scan_from = voff >> 16, voff & 0xFFFF
        # a first guess of the bytes the region needs: its share of the compressed file (x 4 for compression), doubled
        # whenever it turns out too small; every piece is scanned once (the scan resumes where the last one stopped)
        glen = max(1, self.contigs[tid][1])
        cap = int(min(1 << 31, max(16 << 20, 4.0 * self.comp.size * (end - beg + 40000) / glen)))
        pool = getattr(self, "_pool", None)
        if pool is None:
            pool = self._pool = {}
        buf = pool.get(slot) if slot is not None else None
        if buf is None or buf.size < cap:
            buf = np.empty(cap, np.uint8)
        n = 0                                     # bytes of `buf` that are filled
        parts = []
        keys = ("rec_off", "tid", "pos", "end", "flag", "l_seq", "clip_l", "clip_r")
        while True:
            nxt = C.c_int64(0)
            got = L.strk_bgzf_inflate_range(self.comp.ctypes.data, self.comp.size, coff, buf[n:].ctypes.data, buf.size - n, C.byref(nxt), int(threads))
            if got < 0:
                _lib.check(int(got))
            n += int(got)
            coff = nxt.value
            guess = (n - scan_from) // 2048 + 1024      # long reads: records of kilobytes; short ones: scanned twice
            while True:
                piece = _Region()
                piece._set_arrays(guess)
                end_off = C.c_int64(0)
                k = L.strk_bam_scan_piece(buf.ctypes.data, n, scan_from, piece.n_records, *piece._scan_ptrs(), C.byref(end_off))
                if k < 0:
                    _lib.check(int(k))
                if k <= guess:
                    break
                guess = int(k)
            piece._trim(int(k))
            scan_from = int(end_off.value)
            past = (piece.tid != tid) | (piece.pos >= end)
            if past.any():
                piece._trim(int(np.argmax(past)))
            parts.append(piece)
            if past.any() or coff >= self.comp.size:
                break
            if buf.size - n < (1 << 17):          # no further block fits: more room, then on
                buf = np.concatenate((buf[:n], np.empty(buf.size, np.uint8)))
        for k_ in keys:
            setattr(out, k_, np.concatenate([getattr(p_, k_) for p_ in parts]))
        out.n_records = int(len(out.rec_off))
        last = int(out.rec_off[-1]) + 4 + int(struct.unpack_from("<i", buf, int(out.rec_off[-1]))[0]) if out.n_records else 0
        out.data = buf[:max(last, 16)]
        if slot is not None:
            pool[slot] = buf
        out._build_index()
        return out


class DeviceBam(_Records):
    """The alignment file resident in HBM: the compressed bytes are uploaded once, inflated by the device (strk_dbam_inflate:
    one GPU lane per BGZF block), the records are found by the device (strk_dbam_scan: one lane per entry of the .bai linear
    index walks the record chain to the next entry) and only their per-record fields come back to the host, where the
    interval queries of _Records run on them as for any other reader.  Read extraction (extract_reads) and read names go
    through the device as well, and the extracted bases never leave it: strk_count_loci_dseqs counts them where they are.
    A record comes to the host only when somebody asks for it as a segment (soft-clipped reads on their way to realignment).

    `span_bytes` (needs the .bai): STREAMED mode for a file whose decompressed form does not fit in device memory — a 30x
    whole-genome file is 60-100 GB compressed, six times that inflated.  Nothing is loaded at first; `plan(blocks)` groups the
    catalog blocks of a contig into spans of at most that many compressed bytes, `load_span` reads one span of the file
    (strk_dbam_inflate_file_range: the BGZF blocks between two offsets the linear index gives), inflates and scans it on the
    device, and the object then serves that span exactly as it serves a whole file.  call_blocks walks the spans in turn."""

    def __init__(self, path: str, index: str | None = None, device: int = 0, span_bytes: int | None = None):
        self.path = path
        L = _lib.load()
        h = C.c_void_p()
        _lib.check(L.strk_dbam_open(int(device), C.byref(h)))
        self._h = h
        self.streamed = span_bytes is not None
        if self.streamed:
            try:
                self._init_streamed(path, index, int(span_bytes))
            except Exception:
                self.close()
                raise
            return
        tm = self.open_stage_s = {}
        t0 = time.perf_counter()
        tot = L.strk_dbam_inflate_file(h, os.fsencode(path), 0, None)
        tm["upload_inflate_s"] = round(time.perf_counter() - t0, 4)
        ms = (C.c_double * 3)()
        L.strk_dbam_file_ms(h, ms)
        tm["buffers_s"], tm["read_upload_s"], tm["inflate_s"] = (round(x / 1e3, 4) for x in ms)
        if tot < 0:
            self.close()
            _lib.check(int(tot))
        self.n_bytes = int(tot)
        self.data = None
        # header: text, contigs, offset of the first record
        n_head = min(self.n_bytes, 1 << 20)
        while True:
            head = np.empty(n_head, np.uint8)
            _lib.check(L.strk_dbam_download(h, 0, n_head, head.ctypes.data))
            try:
                self.header_text, self.contigs, first = _parse_header(head)
                break
            except (struct.error, IndexError, ValueError):
                if n_head >= self.n_bytes or n_head > (1 << 28):
                    self.close()
                    raise ValueError(f"{path}: not a BAM file") from None
                n_head = min(self.n_bytes, n_head * 4)
        # entry points of the record scan: the first record and what the 16 kb linear index points at
        index = index or (path + ".bai" if os.path.exists(path + ".bai") else os.path.splitext(path)[0] + ".bai")
        starts = [np.array([first], np.int64)]
        if os.path.exists(index):
            lin = IndexedBam._read_bai(index, len(self.contigs), path)
            voff = np.unique(np.concatenate([x[x != 0] for x in lin] or [np.zeros(0, np.uint64)]).astype(np.uint64))
            if voff.size:
                off = np.empty(voff.size, np.int64)
                _lib.check(L.strk_dbam_voffsets(h, voff.ctypes.data, voff.size, off.ctypes.data))
                starts.append(off[(off > first) & (off < self.n_bytes)])
        starts = np.unique(np.concatenate(starts)).astype(np.int64)
        tm["header_index_s"] = round(time.perf_counter() - t0 - tm["upload_inflate_s"], 4)
        t1 = time.perf_counter()
        self._scan(starts)
        tm["scan_s"] = round(time.perf_counter() - t1, 4)
        t1 = time.perf_counter()
        self._build_index()
        tm["record_index_s"] = round(time.perf_counter() - t1, 4)

    def _scan(self, starts: np.ndarray) -> None:
        """Record scan of the resident stretch from the entry points `starts`: the per-record arrays, in file order."""
        L = _lib.load()
        cap = 0                                                 # (capacity 0: the count alone, the first of the scan's two passes)
        while True:
            self._set_arrays(cap)
            self.l_name = np.zeros(cap, np.int32)
            n = L.strk_dbam_scan(self._h, starts.ctypes.data, starts.size, cap, *self._scan_ptrs(), self.l_name.ctypes.data)
            if n < 0:
                self.close()
                _lib.check(int(n))
            if n <= cap:
                break
            cap = int(n)
        self._trim(int(n))
        self.l_name = self.l_name[:int(n)]

    # ---- streamed mode -------------------------------------------------------------------------------------------------
    def _init_streamed(self, path: str, index: str | None, span_bytes: int) -> None:
        ib = IndexedBam(path, index)             # header through the host (the first blocks), the 16 kb linear index
        self.header_text, self.contigs, self._lin = ib.header_text, ib.contigs, ib._lin
        del ib
        self.span_bytes = max(int(span_bytes), 1 << 20)
        self._file_size = os.path.getsize(path)
        voff = np.unique(np.concatenate([x[x != 0] for x in self._lin] or [np.zeros(0, np.uint64)]).astype(np.uint64))
        self._voffs = voff                                        # every record start the index knows, in file order
        self._block_starts = np.unique((voff >> np.uint64(16)).astype(np.int64))     # block boundaries among them
        self._span = None                                         # (tid, beg, end) of what is resident
        self.n_bytes = 0
        self.data = None
        self._set_arrays(0)
        self.l_name = np.zeros(0, np.int32)
        self._build_index()
        self.open_stage_s = {"spans": 0, "load_s": 0.0, "compressed_mb": 0.0}

    def _tid(self, contig: str) -> int:
        names = self.references
        name = resolve_contig(names, contig)
        return names.index(name) if name is not None else -1

    def _coff_range(self, tid: int, beg: int, end: int, margin: int) -> tuple[int, int, int] | None:
        """(compressed offset of the first block, of the end, virtual offset of the first record) for the records of contig
        `tid` that start before `end` and can overlap [beg, end).  The upper end is a guess — the linear index says where the
        records that OVERLAP a window begin, not where those that start in it do: load_span checks it and widens `margin`."""
        lin = self._lin[tid] if 0 <= tid < len(self._lin) else np.zeros(0, np.uint64)
        w = max(0, int(beg)) >> 14
        nz = np.flatnonzero(lin[w:]) if w < len(lin) else np.zeros(0, np.int64)
        if nz.size == 0:
            return None
        v_lo = int(lin[w + int(nz[0])])
        w2 = ((int(end) + margin) >> 14) + 1
        later = lin[w2:][lin[w2:] != 0] if w2 < len(lin) else np.zeros(0, np.uint64)
        if later.size:
            cand = int(later[0]) >> 16
        else:                                                    # the contig ends: up to where the next one begins
            cand = -1
            for t in range(tid + 1, len(self._lin)):
                nzt = self._lin[t][self._lin[t] != 0]
                if nzt.size:
                    cand = int(nzt[0]) >> 16
                    break
        if cand < 0:
            return v_lo >> 16, self._file_size, v_lo
        k = int(np.searchsorted(self._block_starts, max(cand, v_lo >> 16), side="right"))    # the block after it
        hi = int(self._block_starts[k]) if k < self._block_starts.size else self._file_size
        return v_lo >> 16, hi, v_lo

    def plan(self, blocks: list) -> list[tuple[str, int, int, list]]:
        """Catalog blocks (one contig each) -> spans (contig, beg, end, blocks) of at most span_bytes compressed bytes, in
        catalog order; a block that needs more by itself is a span of its own."""
        spans: list[tuple[str, int, int, list]] = []
        cur = None
        for block in blocks:
            contig = block[0].contig
            beg = min(l.left_flank_coord for l in block)
            end = max(l.right_flank_coord for l in block) + 1
            if cur is not None and cur[0] == contig:
                nb, ne = min(cur[1], beg), max(cur[2], end)
                r = self._coff_range(self._tid(contig), nb, ne, 1 << 20)
                if r is not None and r[1] - r[0] <= self.span_bytes:
                    cur = (contig, nb, ne, cur[3] + [block])
                    continue
            if cur is not None:
                spans.append(cur)
            cur = (contig, beg, end, [block])
        if cur is not None:
            spans.append(cur)
        return spans

    def load_span(self, contig: str, beg: int, end: int) -> None:
        """Makes the records of `contig` that can overlap [beg, end) resident (replacing what was)."""
        L = _lib.load()
        t0 = time.perf_counter()
        tid = self._tid(contig)
        margin = 1 << 20
        while True:
            r = self._coff_range(tid, beg, end, margin) if tid >= 0 else None
            if r is None:
                self.n_bytes = 0
                self._set_arrays(0)
                self.l_name = np.zeros(0, np.int32)
                break
            lo, hi, v_lo = r
            tot = L.strk_dbam_inflate_file_range(self._h, os.fsencode(self.path), lo, hi, 0, None)
            if tot < 0:
                _lib.check(int(tot))
            self.n_bytes = int(tot)
            a = int(np.searchsorted(self._voffs, np.uint64(v_lo), side="left"))
            b = int(np.searchsorted(self._voffs, np.uint64(hi) << np.uint64(16), side="left"))
            voff = np.ascontiguousarray(self._voffs[a:b])
            off = np.empty(voff.size, np.int64)
            _lib.check(L.strk_dbam_voffsets(self._h, voff.ctypes.data, voff.size, off.ctypes.data))
            starts = np.unique(off[(off >= 0) & (off < self.n_bytes)]).astype(np.int64)
            self._scan(starts)
            # complete when the file order has gone past `end` (sorted by position) or the file ended
            n = self.n_records
            done = hi >= self._file_size or n == 0
            if n:
                lt, lp = int(self.tid[n - 1]), int(self.pos[n - 1])
                done = done or lt != tid or lp >= end
            if done or margin > (1 << 34):
                break
            margin *= 8                                           # reads longer than the margin lie across the end: further
        self._build_index()
        self._span = (tid, int(beg), int(end))
        st = self.open_stage_s
        st["spans"] += 1
        st["load_s"] = round(st["load_s"] + time.perf_counter() - t0, 4)
        if r is not None:
            st["compressed_mb"] = round(st["compressed_mb"] + (hi - lo) / 1e6, 1)

    def kernel_s(self) -> float:
        """HIP-event time of all the kernels this reader has launched (inflation, record scan, extraction), in seconds."""
        return float(_lib.load().strk_dbam_kernel_ms(self._h)) / 1e3 if getattr(self, "_h", None) is not None else 0.0

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().strk_dbam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def region(self, contig: str, beg: int, end: int, threads: int = 0, slot: int | None = None):
        """Every record is resident: a region is the reader itself.  (Streamed mode: the span is loaded unless the resident
        one covers it.)"""
        if self.streamed:
            tid = self._tid(contig)
            if self._span is None or self._span[0] != tid or beg < self._span[1] or end > self._span[2]:
                self.load_span(contig, beg, end)
        return self

    def _download(self, off: int, n: int) -> np.ndarray:
        buf = np.empty(int(n), np.uint8)
        _lib.check(_lib.load().strk_dbam_download(self._h, int(off), int(n), buf.ctypes.data))
        return buf

    def name(self, i: int) -> str:
        return self.names(np.array([i]))[0]

    def names(self, idx: np.ndarray) -> list[str]:
        idx = np.asarray(idx, np.int64)
        n = int(idx.size)
        if n == 0:
            return []
        ln = self.l_name[idx]
        with_nul = bool((ln >= 1).all())         # l_name counts the terminating NUL: copied along, the text splits at it
        off = np.concatenate(([0], np.cumsum(ln if with_nul else np.maximum(ln - 1, 0)))).astype(np.int64)
        rec_off = np.ascontiguousarray(self.rec_off[idx], np.int64)
        buf = np.empty(max(int(off[-1]), 1), np.uint8)
        _lib.check(_lib.load().strk_dbam_names(self._h, n, rec_off.ctypes.data, off.ctypes.data, buf.ctypes.data))
        text = buf[:int(off[-1])].tobytes().decode()
        if with_nul:
            names = text.split("\0")
            if len(names) == n + 1 and not names[-1]:
                return names[:n]
            o = off.tolist()                     # (a NUL inside a name: cut by the offsets)
            return [text[o[i]:o[i + 1] - 1] for i in range(n)]
        o = off.tolist()
        return [text[o[i]:o[i + 1]] for i in range(n)]

    def segment(self, i: int) -> AlignedSegment:
        """One record brought to the host (tests, a look at a read; the hot path never does this)."""
        o = int(self.rec_off[i])
        block, = struct.unpack("<i", self._download(o, 4).tobytes())
        one = _Region()
        one.contigs = self.contigs
        one.data = self._download(o, 4 + block)
        one._set_arrays(1)
        for k in ("tid", "pos", "end", "flag", "l_seq", "clip_l", "clip_r"):
            getattr(one, k)[0] = getattr(self, k)[i]
        return one.segment(0)


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
    if isinstance(bam, DeviceBam):
        name_len = np.zeros(max(n, 1), np.int32)
        d_seqs = C.c_void_p()
        _lib.check(_lib.load().strk_dbam_extract(bam._h, n, rec_off.ctypes.data, coords.ctypes.data,
                                                 a_cig.ctypes.data if a_cig is not None else None,
                                                 a_off.ctypes.data if a_off is not None else None,
                                                 a_start.ctypes.data if a_start is not None else None,
                                                 int(flank_size), int(min_avg_phred), int(wildcard_threshold), status.ctypes.data,
                                                 nfl.ctypes.data, ntr.ctypes.data, nfr.ctypes.data, seq_off.ctypes.data,
                                                 name_len.ctypes.data, C.byref(d_seqs)))
        # the bases stay on the device: `seqs` is a stand-in for code that only asks for sizes
        return {"status": status, "nfl": nfl, "ntr": ntr, "nfr": nfr, "seqs": np.zeros(1, np.uint8), "seq_off": seq_off,
                "d_seqs": d_seqs.value}
    L = _lib.load()

    def call(seqs_ptr, cap):
        return L.strk_extract_reads(bam.data.ctypes.data, bam.data.size, n, rec_off.ctypes.data, coords.ctypes.data,
                                    a_cig.ctypes.data if a_cig is not None else None,
                                    a_off.ctypes.data if a_off is not None else None,
                                    a_start.ctypes.data if a_start is not None else None,
                                    int(flank_size), int(min_avg_phred), int(wildcard_threshold), status.ctypes.data,
                                    nfl.ctypes.data, ntr.ctypes.data, nfr.ctypes.data, seqs_ptr, cap, seq_off.ctypes.data)

    # A first attempt with room for tracts up to four times the reference window (one pass over the records); a read
    # may carry an expansion many times that (the flagship case): the library then says how much is needed, nothing is
    # truncated, and the second call is sized by what the reads actually hold.
    span = coords[:, 3] - coords[:, 0] if n else np.zeros(0, np.int64)
    cap = int(np.minimum(bam.l_seq[rec_idx] if hasattr(bam, "l_seq") else span * 4, 4 * span + 64).sum()) + 16
    seqs = np.empty(cap, np.uint8)
    rc = call(seqs.ctypes.data, cap)
    if rc == -12:                                      # STRK_E_NOMEM: seq_off[n] holds the bytes needed
        cap = int(seq_off[-1]) + 16
        seqs = np.empty(cap, np.uint8)
        rc = call(seqs.ctypes.data, cap)
    _lib.check(rc)
    return {"status": status, "nfl": nfl, "ntr": ntr, "nfr": nfr, "seqs": seqs[:int(seq_off[-1])], "seq_off": seq_off}
