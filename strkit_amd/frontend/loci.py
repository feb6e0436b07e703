"""Catalog loader — behaviour of strkit/call/loci.py:60-152 (valid_motif, validate_locus, parse_loci_bed,
parse_last_column) and the block builder of load_loci (loci.py:170-215, 249-300): loci are handed to workers in
blocks of at most 200 that stay on one contig and whose neighbours are at most 20 000 bases apart."""
from __future__ import annotations

import functools
import re
from dataclasses import dataclass
from typing import Iterable, Iterator

__all__ = ["Locus", "LocusValidationError", "valid_motif", "validate_locus", "parse_loci_bed", "parse_last_column",
           "load_loci", "normalize_contig", "resolve_contig", "MAX_BLOCK_SIZE", "MAX_BLOCK_INTER_READ_DIST"]

_ID_MOTIF = re.compile(r"[Ii][Dd]=([^;=\s]+);[Mm][Oo][Tt][Ii][Ff]=([^;=\s]+)$")
_IUPAC_MOTIF_LETTERS = frozenset("ACGTRYSWKMBDHVN")   # loci.py:30: nucleotide codes, no 'X', no lower case
MAX_BLOCK_SIZE = 200                                   # loci.py:193
MAX_BLOCK_INTER_READ_DIST = 20000                      # loci.py:194


def normalize_contig(contig: str, has_chr: bool) -> str:
    """The name of `contig` in a file whose contigs do (has_chr) or do not carry the "chr" prefix — what the reference's
    `normalize_contig(locus.contig, ref_file_has_chr)` does between catalog, alignment file and reference genome
    (call_locus.py:758; `ref_file_has_chr` = any contig starts with "chr", call_sample.py:312)."""
    if has_chr:
        return contig if contig.startswith("chr") else "chr" + contig
    return contig[3:] if contig.startswith("chr") else contig


def resolve_contig(names, contig: str) -> str | None:
    """`contig` as one of `names` (a container of a file's contig names), with or without the "chr" prefix; None when
    the file does not have it either way."""
    if contig in names:
        return contig
    alt = normalize_contig(contig, not contig.startswith("chr"))
    return alt if alt in names else None


class LocusValidationError(ValueError):
    def __init__(self, error_str: str, hint_msg: str = ""):
        super().__init__(error_str)
        self.error_str = error_str
        self.hint_msg = hint_msg

    def log_error(self, logger) -> None:
        logger.critical(self.error_str)
        logger.critical(self.hint_msg)


@dataclass(frozen=True)
class Locus:
    """The fields of STRkitLocus the per-locus path reads (call_locus.py:765-772)."""
    t_idx: int
    locus_id: str
    contig: str
    left_coord: int
    right_coord: int
    motif: str
    flank_size: int = 70

    @property
    def left_flank_coord(self) -> int:
        return max(0, self.left_coord - self.flank_size)

    @property
    def right_flank_coord(self) -> int:
        return self.right_coord + self.flank_size

    @property
    def motif_size(self) -> int:
        return len(self.motif)

    def log_str(self) -> str:
        return f"locus {self.t_idx} (id={self.locus_id}): {self.contig}:{self.left_coord}-{self.right_coord} [{self.motif}]"


@functools.lru_cache(maxsize=4096)
def valid_motif(motif: str) -> bool:
    return len(motif) > 0 and all(ch in _IUPAC_MOTIF_LETTERS for ch in motif)


def validate_locus(locus: Locus) -> None:
    if locus.left_coord >= locus.right_coord:
        raise LocusValidationError(
            f"BED catalog format error: invalid coordinates on line {locus.t_idx}: start ({locus.left_coord}) >= end "
            f"({locus.right_coord})", "BED catalog: coordinates must be 0-based, half-open - [start, end)")
    if not valid_motif(locus.motif):
        raise LocusValidationError(f"BED catalog format error: invalid motif on line {locus.t_idx}: {locus.motif}",
                                   "BED catalog: motifs must contain only valid IUPAC nucleotide codes.")


def parse_loci_bed(loci_file: str) -> Iterator[tuple[str, ...]]:
    with open(loci_file) as fh:
        for raw in fh:
            line = raw.strip()
            if line and not line.startswith("#"):
                yield tuple(line.split("\t"))


def parse_last_column(t_idx: int, val: str) -> dict:
    """Last BED column: a bare motif, or `;`-separated `key=value` pairs with keys ID and MOTIF (any case, blanks
    around `=` and after `;` ignored); the motif is upper-cased, the default id is ``locus<t_idx>``."""
    if ";" not in val and "=" not in val:
        return {"id": f"locus{t_idx}", "motif": val}
    m = _ID_MOTIF.match(val)                             # the common spelling, without blanks: nothing to strip
    if m:
        return {"id": m.group(1), "motif": m.group(2).upper()}
    hint = "BED catalog: last column must either be motif or ID=locusID;MOTIF=motif"
    bad = LocusValidationError(f"BED catalog format error: could not parse last column on line {t_idx}: {val}", hint)
    out = {"id": f"locus{t_idx}"}
    for part in val.split(";"):
        part = part.lstrip(" ")
        if part.count("=") != 1:
            raise bad
        key, _, value = part.partition("=")
        key, value = key.rstrip(" ").lower(), value.strip()
        if key not in ("id", "motif"):
            raise bad
        if not value:
            raise LocusValidationError(
                f"BED catalog format error: cannot have empty value in last column on line {t_idx} for key {key}", hint)
        out[key] = value.upper() if key == "motif" else value
    if not out.get("motif"):
        raise bad
    return out


def load_loci(loci_file: str, flank_size: int = 70, contigs: Iterable[str] | None = None, processes: int = 1,
              max_block_size: int | None = None) -> list[list[Locus]]:
    """Blocks of validated loci in file order.  A block ends at a contig change, when the next locus starts more than
    20 000 bases past the block's right edge, or at min(200, max(n_loci // processes, 1)) loci (loci.py:193-194).
    Loci on contigs the alignment file does not have are dropped, as the reference does (loci.py:262-266)."""
    rows = list(parse_loci_bed(loci_file))
    cap = max_block_size or min(MAX_BLOCK_SIZE, max(len(rows) // max(processes, 1), 1))
    known = set(contigs) if contigs is not None else None
    if known is not None:       # "chr1" in the catalog and "1" in the files (or the reverse) are the same contig
        known |= {normalize_contig(c, True) for c in known} | {normalize_contig(c, False) for c in known}
    blocks: list[list[Locus]] = []
    cur: list[Locus] = []
    cur_right = -1
    for t_idx, row in enumerate(rows, 1):
        if len(row) < 4:
            raise LocusValidationError(f"BED catalog format error: line {t_idx} has fewer than 4 columns",
                                       "BED catalog: contig, start, end, motif")
        contig, start, end = row[0], int(row[1]), int(row[2])
        if known is not None and contig not in known:
            continue
        last = parse_last_column(t_idx, row[-1])
        locus = Locus(t_idx, last["id"], contig, start, end, last["motif"], flank_size)
        validate_locus(locus)
        if cur and (cur[-1].contig != contig or len(cur) >= cap
                    or locus.left_flank_coord - cur_right > MAX_BLOCK_INTER_READ_DIST):
            blocks.append(cur)
            cur, cur_right = [], -1
        cur.append(locus)
        cur_right = max(cur_right, locus.right_flank_coord)
    if cur:
        blocks.append(cur)
    return blocks
