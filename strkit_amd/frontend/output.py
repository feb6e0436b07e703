"""What happens to the per-read copy numbers after the hot path: the records the reference's
allele calling starts from and the read-level fields of its VCF.

* `read_weights` — `locus_segments.get_read_weight(targeted, segment.length, tr_len_w_flank)` (call_locus.py:1259).  The
  Rust original is not in the reference's tree; the formula is the one STRkit's earlier pure-Python releases carried at
  this very spot, and it reproduces the documented example (docs/output_formats.md:92-104: `sl` 31, flank 70 -> `w`
  1.0217 for HiFi reads of ~15.8 kb).  UNPINNED like the rest of the un-vendored arithmetic.
* `allele_calling_inputs` — the two arrays `call_alleles_with_gmm` builds from the read records and hands to
  `allele.call_alleles` (call_locus.py:188-204; allele.py:176-189): int32 copy numbers, float64 weights normalised to 1.
* `mcrl_field` / `slr_field` — the `MCRL` / `SLR` sample fields (output/vcf.py:322-342): per allele ("peak") a histogram
  `CNxCOUNT|CNxCOUNT...` of the read-level copy numbers / tract lengths of the reads assigned to it.
* `write_vcf` — a VCF 4.2 text writer with the reference's header lines and the record layout of
  create_result_vcf_records (output/vcf.py:67-156,173-342) for what this backend knows.  Alleles, genotypes and the
  per-allele fields come from allele calling, which stays with the reference; a row that carries peak labels (`p`) per
  read and a `call` gets `MC` / `MCRL` / `SLR` per peak exactly as there.  For rows without a call the reference writes
  none of the read-level fields; this writer adds, under its own header note, the one-group histograms of all kept reads
  so that the read-level answers of the hot path are visible in the VCF too.
"""
from __future__ import annotations

from collections import Counter
from datetime import datetime

import numpy as np

__all__ = ["read_weights", "allele_calling_inputs", "format_count_pair", "mcrl_field", "slr_field", "write_vcf", "VCF_ANCHOR_SIZE"]

VCF_ANCHOR_SIZE = 5


def read_weights(read_lengths_sorted: np.ndarray, tr_len_with_flank: np.ndarray, read_length: np.ndarray | None = None,
                 targeted: bool = False) -> np.ndarray:
    """Weight of each read of ONE locus.  `read_lengths_sorted`: lengths of all segments fetched for the locus, ascending
    (`locus_segments.sorted_read_lengths`, call_locus.py:1057); `tr_len_with_flank` per read = |flank| + |tract| + |flank| as
    extracted (call_locus.py:1254).  A read large enough to contain the tract is the rarer the longer the tract is:
    w = (L + t - 2) / (L - t + 1) with L the mean length of the reads that could contain it (targeted: the read's own
    length).  NaN where no fetched read is long enough (the reference voids the locus there, call_locus.py:1261-1271)."""
    t = np.asarray(tr_len_with_flank, np.float64)
    lens = np.asarray(read_lengths_sorted, np.float64)
    if targeted:
        L = np.asarray(read_length, np.float64)
    else:
        part = np.searchsorted(lens, t, side="left")
        suffix = np.concatenate((np.cumsum(lens[::-1])[::-1], [0.0]))
        cnt = len(lens) - part
        L = np.where(cnt > 0, suffix[np.minimum(part, len(lens))] / np.maximum(cnt, 1), np.nan)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (L + t - 2.0) / (L - t + 1.0)


def allele_calling_inputs(row: dict) -> tuple[np.ndarray, np.ndarray]:
    """(read_cns int32, read_weights float64 summing to 1) of a result row, in read order — exactly what
    call_alleles_with_gmm derives from `read_dict` (call_locus.py:188-192) and passes as `repeats_fwd` /
    `read_weights_fwd` to allele.call_alleles (call_locus.py:201-214)."""
    rdvs = tuple((row.get("reads") or {}).values())
    cns = np.fromiter((r["cn"] for r in rdvs), dtype=np.int32, count=len(rdvs))
    w = np.fromiter((r["w"] for r in rdvs), dtype=np.float64, count=len(rdvs))
    if len(w):
        w /= w.sum()
    return cns, w


def format_count_pair(pair) -> str:
    return "x".join(map(str, pair))      # CNxCOUNT or SLxCOUNT (output/vcf.py:57-58)


def _hist_field(reads: dict, key: str, n_peaks: int | None) -> tuple[str, ...]:
    vals = list(reads.values())
    if n_peaks is None:                  # no peak labels: one group with every read
        groups = [vals]
    else:
        groups = [[r for r in vals if r.get("p") == pi] for pi in range(n_peaks)]
    return tuple("|".join(map(format_count_pair, sorted(Counter(r[key] for r in g).items()))) for g in groups)


def mcrl_field(reads: dict, n_peaks: int | None = None) -> tuple[str, ...]:
    """`MCRL` (output/vcf.py:322-330): e.g. ("7x1|8x10|9x1", "8x2|9x12") for two alleles of 8 and 9 copies."""
    return _hist_field(reads, "cn", n_peaks)


def slr_field(reads: dict, n_peaks: int | None = None) -> tuple[str, ...]:
    """`SLR` (output/vcf.py:332-342): the same for the read-level tract lengths."""
    return _hist_field(reads, "sl", n_peaks)


_FORMATS = (("AD", ".", "Integer", "Read depth for each allele"),
            ("DP", "1", "Integer", "Read depth"),
            ("DPS", "1", "Integer", "Read depth (supporting reads only)"),
            ("GT", "1", "String", "Genotype"),
            ("MC", ".", "Integer", "Motif copy number for each allele"),
            ("MCCI", ".", "String", "Motif copy number 95% confidence interval for each allele"),
            ("MCRL", ".", "String", "Read-level motif copy numbers for each allele"),
            ("MMAS", "1", "Float", "Mean model (candidate TR sequence) alignment score across reads."),
            ("PM", "1", "String", "Peak-calling method (dist/snv+dist/snv/hp)"),
            ("SLR", ".", "String", "Read-level sequence lengths for each allele"))
_INFOS = (("VT", "1", "String", "Variant record type (str/snv)"),
          ("MOTIF", "1", "String", "Motif string"),
          ("REFMC", "1", "Integer", "Motif copy number in the reference genome"),
          ("BED_START", "1", "Integer", "Original start position of the locus as defined in the catalog (0-based inclusive)"),
          ("BED_END", "1", "Integer", "Original end position of the locus as defined in the catalog (0-based exclusive, i.e., 1-based)"),
          ("ANCH", "1", "Integer", "Five-prime anchor size"))


def write_vcf(report: dict, path: str, ref=None, sample_id: str | None = None, n_alleles: int = 2, date: str | None = None) -> int:
    """Writes the loci of a report as VCF 4.2 text; returns the number of records.  `ref` (a Fasta) supplies the contig
    lengths of the header.  Loci without reference data have no anchor and are skipped, as the reference does
    (output/vcf.py:184-186)."""
    sample = sample_id or report.get("sample_id") or "sample"
    now = datetime.now()  # noqa: DTZ005
    lines = ["##fileformat=VCFv4.2", "##fileDate=" + (date or f"{now.year}{now.month:02d}{now.day:02d}"), "##source=strkit_amd",
             "##strkitCommand=call", f"##strkitCatalogNumLoci={report.get('catalog', {}).get('num_loci', len(report['results']))}",
             "##strkitAmdNote=rows without an allele call carry MCRL and SLR as ONE group over all kept reads (STRkit writes them per called allele only)"]
    if ref is not None:
        lines += [f"##contig=<ID={c},length={ref.get_reference_length(c)}>" for c in ref.references]
    lines += [f'##FORMAT=<ID={i},Number={n},Type={t},Description="{d}">' for i, n, t, d in _FORMATS]
    lines += [f'##INFO=<ID={i},Number={n},Type={t},Description="{d}">' for i, n, t, d in _INFOS]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample)
    n_rec = 0
    for row in sorted(report["results"], key=lambda r: (r["contig"], r.get("start_adj", r["start"]))):
        if "ref_start_anchor" not in row or "ref_seq" not in row:
            continue
        anchor, ref_seq = row["ref_start_anchor"].upper(), row["ref_seq"].upper()
        # without consensus sequences of the alleles nothing of the anchor is shared and can be cut: one base stays for VCF
        # compliance only when alleles are written; here the whole anchor is kept (anchor_offset 0, output/vcf.py:213-218)
        start0 = row.get("start_adj", row["start"]) - len(anchor)
        reads = row.get("reads") or {}
        call = row.get("call")
        n_peaks = len(call) if call else None
        info = f"VT=str;MOTIF={row['motif']};REFMC={row['ref_cn']};BED_START={row['start']};BED_END={row['end']};ANCH={len(anchor)}"
        keys, vals = ["GT", "DP"], ["/".join(["."] * n_alleles), str(len(reads))]
        if row.get("assign_method"):
            keys.append("PM"); vals.append(str(row["assign_method"]))
        if call:
            keys.append("MC"); vals.append(",".join(str(int(c)) for c in call))
        if reads:
            keys += ["MCRL", "SLR"]
            vals += [",".join(mcrl_field(reads, n_peaks)), ",".join(slr_field(reads, n_peaks))]
        lines.append("\t".join((row["contig"], str(start0 + 1), row["locus_id"], anchor + ref_seq, ".", ".", ".", info, ":".join(keys), ":".join(vals))))
        n_rec += 1
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return n_rec
