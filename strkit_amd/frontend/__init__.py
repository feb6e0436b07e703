"""Minimal front end around the repeat-count backend: catalog loader, BGZF/BAM and FASTA
readers (+ writers for synthetic test data), CIGAR walk to the per-read (left flank, tract, right flank) triple,
and a `call` driver that feeds blocks of loci to the device.

The reference does this through pysam and the un-vendored `strkit_rust_ext` (STRkitBAMReader, STRkitAlignedSegment,
get_read_coords_from_matched_pairs, ...); only the call sites are in its tree, so every function here cites the
call site whose behaviour it reproduces.  Pure Python/numpy host code: nothing here is on the device hot path.
"""
from .loci import Locus, LocusValidationError, load_loci, parse_last_column, parse_loci_bed, valid_motif, validate_locus
from .fasta import Fasta, write_fasta
from .bam import AlignedSegment, BamFile, read_bam, write_bam
from .extract import (CigarIndex, LocusReadCoords, LowMeanBaseQual, find_pair_by_ref_pos, get_aligned_pairs,
                      get_read_coords_from_cigar, get_read_coords_from_matched_pairs, get_sequence_data_for_locus)
from .native import DeviceBam, IndexedBam, NativeBam, extract_reads
from .call import call_blocks, call_locus, call_sample, write_json

__all__ = ["Locus", "LocusValidationError", "load_loci", "parse_last_column", "parse_loci_bed", "valid_motif",
           "validate_locus", "Fasta", "write_fasta", "AlignedSegment", "BamFile", "read_bam", "write_bam",
           "CigarIndex", "LocusReadCoords", "LowMeanBaseQual", "find_pair_by_ref_pos", "get_aligned_pairs", "get_read_coords_from_cigar",
           "get_read_coords_from_matched_pairs", "get_sequence_data_for_locus", "call_sample", "call_locus", "call_blocks", "write_json", "NativeBam", "IndexedBam", "DeviceBam", "extract_reads"]
