/*
 * strkit_amd.h — C ABI of the MI355X (gfx950) repeat-count backend for STRkit's `strkit call`.
 *
 * This is the drop-in boundary for the per-read repeat-count hot path.  Reference call sites
 * (paths relative to the STRkit tree) each entry point replaces:
 *
 *   strk_repeat_count        strkit_rust_ext.get_repeat_count(start, tr, fl, fr, motif, max_iters,
 *                            local_search_range, step_size, use_shortcuts=False)
 *                            — strkit/call/repeats.py:58-68 (import at repeats.py:7)
 *   strk_count_loci[_device] the per-read loop of call_locus() that calls get_repeat_count once per
 *                            read with the running start-count feedback
 *                            — strkit/call/call_locus.py:1082,1125-1161; sharded over workers at
 *                            strkit/call/call_sample.py:103-138,414
 *   strk_submit_loci_device / strk_finish   the same loop, split so that successive blocks of loci
 *                            overlap on the device (the reference overlaps them across its worker
 *                            processes, strkit/call/call_sample.py:414)
 *   strk_ref_repeat_count    get_ref_repeat_count() incl. score_ref_boundaries(), once per locus
 *                            — strkit/call/repeats.py:23-43,73-192 (parasail sg_qe_scan_profile_sat);
 *                            strk_score_ref_table is its scoring primitive
 *   strk_ref_repeat_count_batch  the same for every locus of a block (call_locus.py:796-810 once per locus)
 *   strk_realign             the parasail sg_dx_trace_scan_16 call + CIGAR of realign_read()
 *                            — strkit/call/realign.py:56-72 (gate at realign.py:65, caller call_locus.py:867-901)
 *   strk_score_table         one parasail semi-global alignment score per candidate copy number
 *                            (the innermost operation; shape of strkit/call/repeats.py:33,40,124)
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, no allocation crosses the ABI
 * except the opaque context.  Every function returns 0 on success or a negative STRK_E_* code;
 * strk_last_error() returns a thread-local message.  A context is bound to one HIP device and
 * must be used from one host thread at a time; it is created lazily inside a worker process
 * (fork-safe: nothing touches HIP before strk_init).
 *
 * Sequences are 1 byte per base, ASCII over ACGT + IUPAC codes + 'X' (low-quality wildcard),
 * case-insensitive (strkit/call/align_matrix.py:25-39).  Scoring is the reference's: match +2,
 * mismatch -7, gap 5 per base (align_matrix.py:15-17).
 */
#ifndef STRKIT_AMD_H
#define STRKIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRK_OK 0
#define STRK_E_INVALID (-22) /* bad argument (EINVAL) */
#define STRK_E_NOMEM (-12)   /* device or host allocation failed (ENOMEM); also: an input beyond what the scratch parts may grow
                                to (a window of more than 2^20 candidate rows, 32 GiB of generic-kernel rows) — strk_last_error()
                                says which.  Scratch that is merely too small for a call grows and the call runs again. */
#define STRK_E_DEVICE (-5)   /* HIP runtime / kernel failure (EIO) */
#define STRK_E_NODEV (-19)   /* no usable gfx950 device (ENODEV) */
#define STRK_E_EMPTY (-61)   /* nothing could be scored (Python's max() of an empty dict) */

/* end-gap flags; s1 = the read window fl+tr+fr, s2 = the candidate fl+motif*i+fr */
#define STRK_DB_BEG_FREE 1
#define STRK_DB_END_FREE 2
#define STRK_CAND_BEG_FREE 4
#define STRK_CAND_END_FREE 8
#define STRK_SG_ALL 15

#define STRK_TIE_FIRST 0 /* Python max(): first maximal element (repeats.py:135,154) */
#define STRK_TIE_LAST 1
#define STRK_NARROW_NONE 0       /* local_search_range fixed for the whole search (repeats.py:100-151) */
#define STRK_NARROW_DECREMENT 1  /* one less after every explored stack entry, never below 1 */
#define STRK_NARROW_HALVE 2      /* halved after every explored stack entry, never below 1 */
#define STRK_NARROW_AFTER_SEED 3 /* the three seed entries use it as given, every chased entry 1 */

typedef struct strk_ctx strk_ctx;

/* Search parameters = RepeatCountParams (strkit/call/repeat_count_params.py:9-14) + the two
 * semantic switches the un-vendored Rust crate leaves open (see DESIGN.md "parity unpinned"). */
typedef struct strk_params {
    int32_t max_iters;          /* rc_params.max_iters                (params.py:45  -> 50) */
    int32_t local_search_range; /* rc_params.initial_local_search_range (params.py:26 -> 3) */
    int32_t step_size;          /* rc_params.initial_step_size          (params.py:27 -> 1)
                                   The reference calls both INITIAL values that "can be narrowed within the
                                   get_repeat_count fn" (repeat_count_params.py:14); that function is in the un-vendored
                                   Rust crate, so here they stay fixed for the whole search, as in the in-tree sibling
                                   get_ref_repeat_count (repeats.py:100-151): a third unpinned choice next to tie_rule
                                   and end_flags; a different schedule would change
                                   search_replay() only, the score table is schedule-independent. */
    int32_t tie_rule;           /* STRK_TIE_FIRST */
    int32_t end_flags;          /* STRK_SG_ALL */
    int32_t feedback;           /* 1: start-count feedback across the reads of a locus
                                   (call_locus.py:1129-1136,1161); 0: start = est_cn as given */
    int32_t window;             /* half-width of the speculative score table per read; 0 = default */
    int32_t no_dedupe;          /* 0 (default): reads of a locus with identical bytes, split and estimate
                                   share one score table (the reference's lru_cache, repeats.py:47);
                                   1: score every read separately */
    int32_t no_band;            /* 0 (default): reads whose window is much wider than the band the search can
                                   reach are scored by the banded kernel first and re-scored exactly only when
                                   the exactness certificate fails; 1: exact kernels only */
    int32_t narrowing;          /* schedule by which local_search_range / step_size shrink inside one search
                                   (repeat_count_params.py:14: "can be narrowed within the get_repeat_count fn").
                                   STRK_NARROW_NONE (0): fixed for the whole search, as get_ref_repeat_count does
                                   (repeats.py:100-151) — the only schedule the tree states, and the default.  The Rust
                                   function's own schedule is not in the tree; STRK_NARROW_DECREMENT / _HALVE / _AFTER_SEED
                                   are the plausible forms (strk_search.h: LsrSchedule; step_size stays fixed in all of
                                   them), there so that reference vectors (tests/test_reference_vectors.py) or a report
                                   diff (tools/compare_strkit_json.py) can name the one that fits.  Other values:
                                   STRK_E_INVALID.  The scalar entry point strk_repeat_count always runs STRK_NARROW_NONE. */
} strk_params;

/* CSR-packed batch of loci.  Read r owns seqs[seq_off[r] .. seq_off[r+1]) laid out fl|tr|fr;
 * locus l owns reads read_off[l] .. read_off[l+1] (in caller order) and motif
 * motifs[motif_off[l] .. motif_off[l+1]).  est_cn[r] is the caller's integer start estimate
 * (get_est_copy_num(), call_locus.py:1129). */
typedef struct strk_batch {
    int32_t n_reads;
    int32_t n_loci;
    const uint8_t* seqs;
    const int64_t* seq_off; /* [n_reads + 1] */
    const int32_t* nfl;     /* [n_reads] */
    const int32_t* ntr;
    const int32_t* nfr;
    const int32_t* est_cn;
    const int32_t* read_off;  /* [n_loci + 1] */
    const uint8_t* motifs;
    const int32_t* motif_off; /* [n_loci + 1] */
} strk_batch;

/* Per-call statistics (all optional output). */
typedef struct strk_stats {
    int64_t dp_cells;      /* DP cell updates executed by the device kernels */
    int32_t n_fallback;    /* reads scored by the generic (non-systolic) kernel */
    int32_t n_miss_reads;  /* reads whose search left the speculative window (re-scored) */
    int32_t n_miss_rounds; /* extra launch rounds needed to resolve them */
    float kernel_ms;       /* HIP-event time of the device work of this call */
    float dp_kernel_ms;    /* ... of the exact DP kernel (k_dp_all) alone */
    int32_t n_dp_launches;
    int32_t n_dedup_reads; /* reads served by the score table of an identical earlier read */
    int32_t n_band_reads;  /* reads scored by the banded kernel ... */
    int32_t n_band_fallback; /* ... of which the certificate failed (re-scored by the exact kernels) */
    float band_kernel_ms;  /* HIP-event time of the banded kernel (k_dp_band) of this call */
    int32_t window_used;   /* half-width of the speculative candidate window this call ran with (params.window, or the
                              level the library's default has adapted to) */
    int64_t band_bytes;    /* algorithmic bytes (|window| + 16 per read) of the reads k_plan routed to k_dp_band / k_dp_band_wide */
    int64_t exact_bytes;   /* ... and to the exact kernels (k_dp_all / k_dp_long / generic) */
    float band_wide_kernel_ms; /* HIP-event time of k_dp_band_wide (band classes 2, 3: long windows) */
    float long_kernel_ms;      /* ... of k_dp_long (column-tiled exact kernel) */
    float generic_kernel_ms;   /* ... of k_dp_generic */
    float head_ms;             /* ... of what precedes the DP kernels (k_hash, k_plan) */
    float replay_ms;           /* ... of k_replay */
    int32_t n_long_reads;      /* reads scored by k_dp_long */
    int64_t wide_bytes;        /* algorithmic bytes of the reads routed to k_dp_band_wide (part of band_bytes) */
    int64_t long_bytes;        /* ... to k_dp_long (part of exact_bytes) */
    int64_t band_cells;        /* dp_cells by the kernel that executed them: k_dp_band (band classes of 8 and 16 lanes per read), */
    int64_t wide_cells;        /* ... k_dp_band_wide (32 and 64 lanes), */
    int64_t exact_cells;       /* ... k_dp_all / k_dp_ref, */
    int64_t long_cells;        /* ... k_dp_long; the remainder of dp_cells is the generic kernel's */
    int32_t window_bucket[5];  /* candidate-window half-width per motif-length bucket (1-2, 3-4, 5-6, 7-10, 11+ bases) this call
                                  ran with; 0: the call held no locus of that bucket */
    int32_t n_sub_batches;     /* strk_count_loci on host buffers: sub-batches the call was cut into (0: one piece) */
} strk_stats;
/* strk_count_loci on large HOST batches runs as a pipeline of sub-batches on two contexts (csrc/strk_host_pipe.inc): the *_ms
 * fields, *_bytes, *_cells and the n_* counts are then SUMS over the sub-batches — sub-batches overlap, so the sum of their
 * device times can exceed the call's wall time (it is not a throughput denominator); window_used / window_bucket are the last
 * sub-batch's.  Device-resident entry points (strk_count_loci_device, strk_submit_loci_device + strk_finish) report one call. */

int strk_init(int device, strk_ctx** out);
void strk_destroy(strk_ctx* ctx);
const char* strk_last_error(void);
const char* strk_version(void);
/* Forget what the library has learnt about the current sample (the default candidate-window level per motif-length bucket,
 * process-wide): call it when a process moves on to ANOTHER sample (the reference runs one sample per process,
 * call_sample.py:265).  Per-context state (band probation, grid history) goes with strk_destroy. */
void strk_adaptive_reset(void);
/* free / total memory of a device in bytes (hipMemGetInfo): how the file front end decides whether an alignment file's
 * decompressed form stays resident or is streamed in spans. */
int strk_device_mem(int device, int64_t* free_bytes, int64_t* total_bytes);
/* Page-locks (hipHostRegister) / releases a host buffer the caller owns.  strk_count_loci copies the bases (batch.seqs) of a
 * batch to the device by DMA straight from the caller's array when that lies in registered (or hipHostMalloc'ed) memory — no
 * staging copy through the library's own pinned blocks, which is what bounds the pageable path (the reference's worker would
 * register the per-block array it reuses, call_sample.py:103-138).  Registering costs a few milliseconds per 100 MB: worth
 * it for buffers that are reused. */
int strk_host_register(void* ptr, int64_t bytes);
int strk_host_unregister(void* ptr);
/* 1 if [ptr, ptr + bytes) is page-locked host memory the library would read in place, 0 if not. */
int strk_host_is_pinned(const void* ptr, int64_t bytes);

/* Scalar drop-in for strkit_rust_ext.get_repeat_count (repeats.py:58-68).  Return contract
 * (repeats.py:55-56): ((out_cn, out_score), out_n_explored, out_cn - start_count). */
int strk_repeat_count(strk_ctx* ctx, int32_t start_count, const uint8_t* tr, int32_t tr_len,
                      const uint8_t* fl, int32_t fl_len, const uint8_t* fr, int32_t fr_len,
                      const uint8_t* motif, int32_t motif_len, int32_t max_iters,
                      int32_t local_search_range, int32_t step_size, int32_t* out_cn,
                      int32_t* out_score, int32_t* out_n_explored);

/* Batched per-locus path, HOST buffers in and out (all out_* are [n_reads] int32). */
int strk_count_loci(strk_ctx* ctx, const strk_batch* batch, const strk_params* params,
                    int32_t* out_cn, int32_t* out_score, int32_t* out_n_iters, int32_t* out_start,
                    strk_stats* stats);

/* Same, every pointer inside `batch` and every out_* pointer is DEVICE memory on ctx's device;
 * work is enqueued on `stream` (a hipStream_t, NULL = default stream) and the call returns after
 * the results are complete in out_* (it synchronises the stream once to check for window misses). */
int strk_count_loci_device(strk_ctx* ctx, const strk_batch* batch, const strk_params* params,
                           int32_t* out_cn, int32_t* out_score, int32_t* out_n_iters,
                           int32_t* out_start, void* stream, strk_stats* stats);

/* Pipelined form of strk_count_loci_device: strk_submit_loci_device enqueues the whole call on
 * `stream` and returns at once; strk_finish waits for it, resolves window misses and fills `stats`.
 * One call may be in flight per context (use one context per stream to overlap batches); every
 * pointer inside `batch` and every out_* pointer must stay valid until strk_finish returns. */
int strk_submit_loci_device(strk_ctx* ctx, const strk_batch* batch, const strk_params* params,
                            int32_t* out_cn, int32_t* out_score, int32_t* out_n_iters,
                            int32_t* out_start, void* stream);
int strk_finish(strk_ctx* ctx, strk_stats* stats);

/* Parity primitive: scores[table_off[r] + k] = semi-global score of (fl + motif*(lo[r]+k) + fr)
 * against (fl+tr+fr) for k < n[r].  HOST buffers.  table_off is [n_reads + 1]. */
int strk_score_table(strk_ctx* ctx, const strk_batch* batch, const int32_t* lo, const int32_t* n,
                     const int64_t* table_off, int32_t end_flags, int32_t force_generic,
                     int32_t* scores, strk_stats* stats);

/* Reference-side primitive (score_ref_boundaries, repeats.py:23-43): for k < n[r],
 * scores[table_off[r] + k] / end_query[...] = parasail sg_qe score and query end of the candidate
 * fl + motif*(lo[r]+k) (NO right flank) against the window fl+tr+fr whose end is free.  For the
 * reversed alignment pass the reversed window with the flanks swapped and the reversed motif.  HOST buffers. */
int strk_score_ref_table(strk_ctx* ctx, const strk_batch* batch, const int32_t* lo, const int32_t* n,
                         const int64_t* table_off, int32_t force_generic, int32_t* scores,
                         int32_t* end_query, strk_stats* stats);

/* Drop-in for get_ref_repeat_count (repeats.py:73-192), once per locus: boundary-extension search
 * + final count.  out9 = {cn, score, l_offset, r_offset, n_offset_scores, n_iters_final,
 * new fl_len, new tr_len, new fr_len} (the bases never change, only where the flank/tract
 * boundaries fall inside fl|tr|fr). */
int strk_ref_repeat_count(strk_ctx* ctx, int32_t start_count, const uint8_t* tr, int32_t tr_len,
                          const uint8_t* fl, int32_t fl_len, const uint8_t* fr, int32_t fr_len,
                          const uint8_t* motif, int32_t motif_len, int32_t ref_size,
                          int32_t vcf_anchor_size, int32_t max_iters, int32_t local_search_range,
                          int32_t step_size, int32_t respect_coords, int32_t* out9);

/* strk_ref_repeat_count for a block of loci in lock-step: every round of boundary-extension scoring is one device
 * call for all loci that still need scores, the final counts one call per distinct search schedule.  Locus i owns
 * seqs[seq_off[i] .. seq_off[i+1]) laid out fl|tr|fr and motifs[motif_off[i] .. motif_off[i+1]); max_iters /
 * local_search_range / step_size are per locus (get_reference_rc_params, repeat_count_params.py:17-42, depends on
 * the estimate).  out9 is [n_loci * 9], fields as strk_ref_repeat_count.  HOST buffers. */
int strk_ref_repeat_count_batch(strk_ctx* ctx, int32_t n_loci, const int32_t* start_count, const uint8_t* seqs,
                                const int64_t* seq_off, const int32_t* nfl, const int32_t* ntr, const int32_t* nfr,
                                const uint8_t* motifs, const int32_t* motif_off, const int32_t* ref_size,
                                int32_t vcf_anchor_size, const int32_t* max_iters, const int32_t* local_search_range,
                                const int32_t* step_size, int32_t respect_coords, int32_t* out9);

/* Batched drop-in for the parasail call of realign_read (strkit/call/realign.py:56-63,71):
 * sg_dx_trace_scan_16(s1 = reference window, s2 = wildcarded read, open, extend, dna_matrix) — s1 aligned end
 * to end, both ends of s2 free, a gap of length k costs open + (k-1)*extend (the reference passes 7 and 0).
 * Pair p owns s1[s1_off[p] .. s1_off[p+1]) and s2[s2_off[p] .. s2_off[p+1]); HOST buffers.
 * Outputs per pair: out_score = pr.score; out_end_ref = 0-based s2 position of the last aligned base
 * (pr.end_ref); the CIGAR of pr.cigar.seq in BAM encoding (len << 4 | op, ops "MIDNSHP=X": I consumes s1,
 * D consumes s2), written to cigar[cigar_off[p] ..] with out_n_cigar[p] runs; it starts at s2 position 0
 * (free leading s2 bases are one D run) and ends at out_end_ref.  2*|s1| + 4 runs always suffice.
 * gap_pref: which gap kind wins an exact score tie after the diagonal (0: I before D, the default; 1: D before I).
 * stats (optional): kernel_ms, dp_cells, exact_bytes = trace bytes written. */
int strk_realign(strk_ctx* ctx, int32_t n_pairs, const uint8_t* s1, const int64_t* s1_off, const uint8_t* s2,
                 const int64_t* s2_off, int32_t open, int32_t extend, int32_t gap_pref, int32_t* out_score,
                 int32_t* out_end_ref, int32_t* out_n_cigar, uint32_t* cigar, const int64_t* cigar_off,
                 strk_stats* stats);

/* What parasail's fixed 16-bit kernel (sg_dx_trace_scan_16, strkit/call/realign.py:56) would have done with these pairs:
 * strk_realign computes in 32 bits and never saturates, the reference's call can.  Per pair out_flags[p] =
 *   STRK_I16_SCORE_SATURATES (2)  the final score itself exceeds what the 16-bit kernel can hold (> 32767 - 2, the matrix
 *                                 maximum is kept as head-room): parasail would return a saturated result and
 *                                 realign_read would compare garbage with its threshold (realign.py:65);
 *   STRK_I16_CELL_MAY_SATURATE (1) the score fits, but an intermediate cell can reach the limit (2 * min(|s1|, |s2|) > 32765):
 *                                 whether the reference's result is affected cannot be told from the score alone;
 *   0                              no cell can reach the limit (every window of at most 16 382 bases).
 * Pure host arithmetic on the lengths and on strk_realign's scores; no context. */
#define STRK_I16_CELL_MAY_SATURATE 1
#define STRK_I16_SCORE_SATURATES 2
int strk_realign_i16_flags(int32_t n_pairs, const int64_t* s1_off, const int64_t* s2_off, const int32_t* scores,
                           int32_t* out_flags);

/* ---- host-side front end (CPU only; no context, thread-safe) ---------------------------------------------------
 * What the reference's Rust extension does before the counter runs: walk the alignment records
 * (STRkitBAMReader / STRkitAlignedSegment, call sites strkit/call/call_sample.py:81-131) and cut each read into
 * (left flank, tract, right flank) for a locus (get_read_coords_from_matched_pairs + get_sequence_data_for_locus,
 * strkit/call/call_locus.py:875-877,1101-1146).  The rules are spelled out in strkit_amd/frontend/extract.py.
 *
 * strk_bam_scan: `buf` = the decompressed BAM stream, `first_rec` = byte offset of the first alignment record (after
 * header and reference list).  Writes, for up to `cap` records, the record's byte offset, refID, 0-based start, exclusive
 * reference end, flag, l_seq and the soft-clip lengths at its two ends; returns the number of records in the stream
 * (call again with larger arrays if it exceeds `cap`) or a negative STRK_E_* code. */
int64_t strk_bam_scan(const uint8_t* buf, int64_t n_bytes, int64_t first_rec, int64_t cap, int64_t* rec_off, int32_t* tid,
                      int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r);

/* strk_bam_scan_piece: strk_bam_scan over a PIECE of the decompressed stream (block-wise access): a record cut off at the
 * end of the piece is not an error; *end_off = offset just past the last complete record (where the next piece resumes). */
int64_t strk_bam_scan_piece(const uint8_t* buf, int64_t n_bytes, int64_t first_rec, int64_t cap, int64_t* rec_off, int32_t* tid,
                            int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r,
                            int64_t* end_off);

/* strk_bam_names: the read names of n records (without their terminating NUL), concatenated; out_off is [n + 1].
 * out == NULL: size query.  Returns the total length or a negative STRK_E_* code. */
int64_t strk_bam_names(const uint8_t* buf, int64_t n_bytes, int64_t n, const int64_t* rec_off, uint8_t* out, int64_t out_cap,
                       int64_t* out_off);

/* strk_bgzf_inflate: decompresses a whole BGZF stream (BAM, bgzipped FASTA) with n_threads host threads (0 = all
 * cores); blocks are independent deflate streams, every block's CRC is checked.  out == NULL: returns the decompressed
 * size.  Otherwise returns the number of bytes written, or a negative STRK_E_* code. */
int64_t strk_bgzf_inflate(const uint8_t* comp, int64_t n_comp, uint8_t* out, int64_t out_cap, int32_t n_threads);

/* strk_bgzf_inflate_range: block-wise access (what an index such as .bai points into): inflates the consecutive BGZF blocks
 * that start at compressed offset `coff` for as long as whole blocks fit into out_cap bytes; *next_coff = compressed offset
 * of the first block NOT inflated (n_comp at the end of the file).  Returns the bytes written or a negative STRK_E_* code.
 * A BAM virtual offset is (coff << 16 | offset inside the inflated block). */
int64_t strk_bgzf_inflate_range(const uint8_t* comp, int64_t n_comp, int64_t coff, uint8_t* out, int64_t out_cap,
                                int64_t* next_coff, int32_t n_threads);

/* strk_extract_reads: item i = (record at rec_off[i], locus boundaries coords[4i..4i+3] = left_flank_coord, left_coord,
 * right_coord, right_flank_coord).  An item with alt_cigar_off[i+1] > alt_cigar_off[i] uses that CIGAR (BAM encoding)
 * starting at alt_start[i] instead of the record's own alignment (a realigned read); alt_* may be NULL.
 * status[i]: 0 = extracted, 1 = the read does not span both flanks (skipped), 2 = mean base quality of the tract below
 * min_avg_phred (skipped).  Extracted items append fl|tr|fr (flanks cut to flank_size, bases with PHRED <=
 * wildcard_threshold replaced by 'X') to seqs; seq_off is [n_items + 1]; nfl/ntr/nfr the three lengths.
 * seqs == NULL is a size query: status, the lengths and seq_off are filled (seq_off[n_items] = bytes needed), no bases are
 * written.  A record that holds the long-CIGAR placeholder (<l_seq>S<ref_len>N) is read through its CG:B,I tag.  Items are
 * processed by all host cores when there are more than a few hundred. */
int strk_extract_reads(const uint8_t* buf, int64_t n_bytes, int32_t n_items, const int64_t* rec_off, const int64_t* coords,
                       const uint32_t* alt_cigar, const int64_t* alt_cigar_off, const int64_t* alt_start, int32_t flank_size,
                       int32_t min_avg_phred, int32_t wildcard_threshold, int32_t* status, int32_t* nfl, int32_t* ntr,
                       int32_t* nfr, uint8_t* seqs, int64_t seq_cap, int64_t* seq_off);

/* ---- alignment file on the device (strk_dbam): BGZF inflation, record scan and read extraction as HIP kernels ----------
 * Replaces, for a whole file or a stretch of it, the host functions above (what strkit_rust_ext's STRkitBAMReader and
 * STRkitAlignedSegment do in the reference, call sites strkit/call/call_sample.py:103-121, call_locus.py:837-958): the
 * decompressed stream lives in HBM only.  One object = one device buffer set; not thread-safe. */
typedef struct strk_dbam strk_dbam;
int strk_dbam_open(int device, strk_dbam** out);
void strk_dbam_close(strk_dbam* d);
/* A closed reader leaves its two largest device buffers (the decompressed stream, the compressed bytes) to the next reader of
 * the process instead of freeing them (allocating gigabytes right after freeing them can take longer than inflating the file);
 * this frees them. */
void strk_dbam_release_cache(void);
/* Inflates the consecutive BGZF blocks of `comp` (HOST memory) from byte `coff` on, as many as decompress into at most
 * max_out bytes, into the object's device buffer (one GPU lane per block, CRC checked); *next_coff = offset of the first block
 * not taken.  Returns the decompressed bytes or a negative STRK_E_* code. */
int64_t strk_dbam_inflate(strk_dbam* d, const uint8_t* comp, int64_t n_comp, int64_t coff, int64_t max_out, int64_t* next_coff);
/* The same for a whole file read by the library: `threads` readers (0: default) pread pieces of it into a ring of pinned
 * buffers, each piece is copied to the device as it comes in, the block headers are walked meanwhile and all blocks are
 * inflated by one launch at the end.  *n_comp (may be NULL) = size of the file.  Returns the decompressed bytes or a negative
 * STRK_E_* code. */
int64_t strk_dbam_inflate_file(strk_dbam* d, const char* path, int threads, int64_t* n_comp);
/* The same for the BGZF blocks in the file's bytes [coff_lo, coff_hi): both must be block boundaries (what the .bai's virtual
 * offsets >> 16 are); coff_hi < 0 or past the end = up to the end of the file.  A file whose decompressed form does not fit in
 * device memory is walked span by span with it (frontend.DeviceBam, streamed mode). */
int64_t strk_dbam_inflate_file_range(strk_dbam* d, const char* path, int64_t coff_lo, int64_t coff_hi, int threads, int64_t* n_comp);
/* bytes [off, off + n) of the decompressed stream -> host (headers, a single record for realignment, tests) */
int strk_dbam_download(strk_dbam* d, int64_t off, int64_t n, uint8_t* out);
/* the first n bytes of the bases of the last strk_dbam_extract -> host (tests) */
int strk_dbam_download_seqs(strk_dbam* d, int64_t n, uint8_t* out);
/* HIP-event time (ms) of all kernels the object has launched so far (inflation, scan, extraction) */
double strk_dbam_kernel_ms(strk_dbam* d);
/* wall-clock stages (ms) of the last strk_dbam_inflate_file: buffers (device + pinned ring), read + upload, inflation */
void strk_dbam_file_ms(strk_dbam* d, double* out3);
/* device address and size of the decompressed stream (valid until the next strk_dbam_inflate / strk_dbam_close) */
void* strk_dbam_data(strk_dbam* d, int64_t* n_bytes);
/* BAM virtual offsets (block offset << 16 | offset in the block) -> offsets in the decompressed stream (-1: not in it) */
int strk_dbam_voffsets(strk_dbam* d, const uint64_t* voff, int64_t n, int64_t* out);
/* strk_bam_scan on the device.  `starts`: ascending, distinct offsets of record starts (the first record and what the .bai
 * linear index points at); one GPU lane walks the record chain from each to the next.  Returns the number of records; the
 * host arrays (capacity cap) are filled, in stream order, when it fits. */
int64_t strk_dbam_scan(strk_dbam* d, const int64_t* starts, int64_t n_starts, int64_t cap, int64_t* rec_off, int32_t* tid,
                       int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r, int32_t* l_name);
/* strk_extract_reads on the device (same arguments, alt_* may be NULL): the bases stay in HBM (*d_seqs, valid until the next
 * call on the object), status / lengths / seq_off and the name length of every item come back. */
int strk_dbam_extract(strk_dbam* d, int32_t n_items, const int64_t* rec_off, const int64_t* coords, const uint32_t* alt_cigar,
                      const int64_t* alt_cigar_off, const int64_t* alt_start, int32_t flank_size, int32_t min_avg_phred,
                      int32_t wildcard_threshold, int32_t* status, int32_t* nfl, int32_t* ntr, int32_t* nfr, int64_t* seq_off,
                      int32_t* name_len, void** d_seqs);
/* read names of n records into out[0 .. out_off[n]); out_off = running sum of the name lengths (input) */
int strk_dbam_names(strk_dbam* d, int64_t n, const int64_t* rec_off, const int64_t* out_off, uint8_t* out);
/* strk_count_loci with the bases already on the device (d_seqs; batch->seqs is ignored), everything else on the host */
int strk_count_loci_dseqs(strk_ctx* ctx, const strk_batch* batch, const void* d_seqs, const strk_params* params, int32_t* out_cn,
                          int32_t* out_score, int32_t* out_n_iters, int32_t* out_start, strk_stats* stats);
/* test aid: the two statements of the CIGAR -> locus boundaries walk on one alignment (bit 0: run index, bit 1: one pass) */
int strk_read_coords_both(const uint32_t* cigar, int32_t n_cigar, int64_t start, const int64_t* coords, int64_t* out_runs,
                          int64_t* out_linear);
/* strk_bgzf_inflate's contract, served by the DEVICE inflater's code compiled for the host, one thread (test aid) */
int64_t strk_bgzf_inflate_sw(const uint8_t* comp, int64_t n_comp, uint8_t* out, int64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif /* STRKIT_AMD_H */
