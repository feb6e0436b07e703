// strk_kernels.h — gfx950 device code of the repeat-count path (included by strk_api.hip only): shared
// definitions, k_hash and k_plan here; the DP kernels and k_replay in the headers included at the end.
//
// What is computed (reference: strkit/call/repeats.py:58-68 -> strkit_rust_ext.get_repeat_count
// -> parasail semi-global DP, once per candidate copy number i):
//     S[i] = sg_score( db = fl+tr+fr ,  cand_i = fl + motif*i + fr ),  gap 5/base, dna_matrix.
//
// How (MI355X-first):
//   * open == extend (align_matrix.py:17) collapses Gotoh to H = max(diag+W, up-g, left-g); with
//     G(r,j) = H(r,j) + g*(r+j) this is G = max3(up, left, diag + W + 2g): one v_add + one
//     v_max3_i32 per cell, all values >= 0.
//   * All candidates share the row prefix fl+motif*i.  One forward DP over fl+motif*i_hi rows and
//     one backward DP over the fr rows give every S[i] as max_j(Gf(R_i,j) + Gb(j)) - const at the
//     "fork rows" R_i = |fl| + i*|motif| (exact because gaps are linear, so DP nodes carry no
//     affine state).  ~8x fewer cell updates than one DP per candidate.
//   * A read is owned by a group of G lanes (8, 16, 32 or 64); each lane keeps CL consecutive db columns
//     of the DP row in VGPRs and the group runs a skewed (anti-diagonal) systolic wavefront: lane
//     l works on row t-l at step t and hands its last column to lane l+1 with one DPP row/wave
//     shift.  Substitution scores come from one v_perm_b32 per 4 cells on an 8-byte per-row word
//     staged in LDS (the db's distinct symbols are remapped to <= 8 classes per read).
//   * No MFMA: integer max-plus DP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "strk_scoring.h"
#include "strk_search.h"

namespace strk {

constexpr int kTableMax = 32;       // max candidates per read in one DP item
constexpr int kNegInf = -(1 << 29);
constexpr int kFastFlankMax = 127;  // right-flank rows the fast classes stage in LDS (the default flank is 70)
constexpr int kMotifMax = 256;     // the fast kernel stages the encoded motif in LDS (<= 2 * smallest capacity bytes)
constexpr int kRowSlack = 160;      // a class of capacity CAP accepts up to CAP + kRowSlack prefix rows

// Fast-kernel classes: (G lanes per read, CL columns per lane); capacity = G*CL slots >= |db| + 1.
//   classes 0-6  : G = 8,  CL = 16..40  (capacity 128..320,   8 reads per wave)
//   classes 7-8  : G = 16, CL = 24, 28  (capacity 384, 448,   4 reads per wave)
//   classes 9-12 : G = 32, CL = 16..28  (capacity 512..896,   2 reads per wave)
//   classes 13-16: G = 64, CL = 16..28  (capacity 1024..1792, 1 read per wave)
// Fewer lanes per read = fewer skew steps and less per-step overhead per cell; the VGPR budget
// (two CL-sized row arrays) and the per-read LDS footprint cap CL at 40.
constexpr int kNumClasses = 17;
__host__ __device__ constexpr int class_G(int c) { return c < 7 ? 8 : (c < 9 ? 16 : (c < 13 ? 32 : 64)); }
__host__ __device__ constexpr int class_CL(int c) {
    return c < 7 ? 16 + 4 * c : (c < 9 ? 24 + 4 * (c - 7) : (c < 13 ? 16 + 4 * (c - 9) : 16 + 4 * (c - 13)));
}
__host__ __device__ constexpr int class_cap(int c) { return class_G(c) * class_CL(c); }
constexpr int kLongClass = kNumClasses;         // list of the column-tiled long-read kernel (k_dp_long)
constexpr int kGenericClass = kNumClasses + 1;  // list of the generic kernel
constexpr int kBandClass0 = kNumClasses + 2;    // band kernel lists: + class (strk_search.h: band_class_G lanes x band_class_D diagonals)
constexpr int kNumLists = kNumClasses + 2 + kNumBandClasses;
constexpr int kLongTile = 64 * 28;              // slots per column tile of k_dp_long (G = 64, CL = 28)
constexpr int kLongMaxTiles = 64;               // |db| + 1 <= 114 688
constexpr int kLongFlankMax = 255;              // k_dp_long keeps both flanks' row symbols in LDS

__constant__ int8_t c_mat[kNSym][kNSym];
__constant__ uint8_t c_enc[256];

struct KArgs {
    // inputs (device)
    const uint8_t* seqs;
    const int64_t* seq_off;
    const int32_t* nfl;
    const int32_t* ntr;
    const int32_t* nfr;
    const int32_t* est_cn;
    const int32_t* read_off;
    const uint8_t* motifs;
    const int32_t* motif_off;
    int32_t n_reads, n_loci;
    // plan / workspace (device)
    int32_t* read_locus;  // [n_reads]
    int32_t* win_lo;      // [n_reads]
    int32_t* win_n;       // [n_reads]
    int64_t* tab_off;     // [n_reads]
    int32_t* table;       // score table
    int32_t* cls_list;    // [kNumLists * list_stride * 2]  (read, chunk start) pairs
    int4* band_recs;      // [kNumBandClasses * list_stride * 3]  everything a band item needs, written by k_plan:
                          //   (read, locus, nfl, ntr) (nfr, m, lo, n) (seq_off lo, seq_off hi, motif_off, est_cn)
    const int32_t* long_sorted;   // k_dp_long's item list in k_sort_long's order (longest first), or NULL: the class list as it was filled
    int4* band_recs_w;    // the records k_dp_band_wide reads: band_recs, or k_sort_wide's copy of the four wide classes, longest first
    int32_t* wide_hist;   // [2][kNumWideLists][256] items of the wide band classes by prefix rows / 64 (k_sort_wide_hist), then the cursors
                          //    k_sort_wide hands positions out with (zeroed with the counters)
    int32_t* counters;    // see Counter enum
    unsigned long long* cells;  // DP cells executed; cells[2] / cells[3]: algorithmic bytes (|window| + 16 per read) of the
                                // items routed to the band kernels / to the exact kernels by k_plan; cells[4] / cells[5]: the part
                                // of those that k_dp_band_wide / k_dp_long take; cells[6..9]: the cells of cells[0] by the kernel
                                // that executes them: k_dp_band, k_dp_band_wide, k_dp_all / k_dp_ref, k_dp_long (kCell*)
    int32_t* scratch;     // generic kernel rows
    long long scratch_cap;      // in int32 units; [0, long_waves * long_slot) belongs to k_dp_long (one slot
                                //   per resident wave), the rest is handed out by the generic kernel's bump allocator
    unsigned long long* scratch_used;
    long long long_slot;        // int32 units per k_dp_long wave
    int32_t long_waves;
    int4* spec;           // [n_reads] speculative search result for start == est_cn (or NULL)
    unsigned long long* rhash;  // [n_reads] content hash of each read window (dedupe) or NULL
    int32_t* rep;         // [n_reads] earliest identical read of the same locus (itself if none)
    int32_t list_stride;
    int32_t end_flags;
    int32_t window;       // half width (plan kernel): the largest of window_b
    int32_t window_b[5];  // ... per motif-length bucket (win_bucket): the estimate round(|tr| / |motif|) is off by the read's
                          //    indel drift DIVIDED by the motif length, so long motifs get by with narrow windows
    int32_t table_stride; // entries per read (plan kernel)
    int32_t max_iters, lsr, step, tie_last;  // search parameters (speculative search in k_dp_all)
    int32_t narrow;                          // schedule of local_search_range inside one search (strk_search.h: kNarrow*)
    int32_t band_mode;    // 1: eligible reads go through k_dp_band first (strk_search.h, "Banded scoring")
    int32_t band_limit;   // only reads with index < band_limit are eligible (a context on probation tries the band on a sample)
    BandTune band_tune;   // where the forward band lies (strk_search.h: band_geometry)
    uint8_t* exact;       // [n_reads] 1: the read's table holds exact scores, 0: band lower bounds
    int32_t dbg;          // profiling aid (env STRKIT_AMD_DBG, results are wrong when set): 1 no forward pass, 2 no backward
                          //    pass, 4 no in-kernel search, 8 no fork rows, 16 band items in arrival order, 32 no wave priorities
    int32_t ref_mode;     // 1: reference-side scoring (repeats.py:23-43): candidate = fl + motif*i only, the
                          //    table holds (score, end_query) pairs, end_flags must be STRK_DB_END_FREE
};

enum Counter {
    kCntClass0 = 0,                       // [0..kNumLists) list lengths (fast classes, long, generic)
    kCntMiss = kNumLists,                 // loci whose search left the table window
    kCntError = kNumLists + 1,            // sticky error bits
    kCntNextChunk = kNumLists + 2,        // work queue head of k_dp_all
    kCntDup = kNumLists + 3,              // reads that share the score table of an identical earlier read
    kCntNextLong = kNumLists + 4,         // work queue head of k_dp_long
    kCntNextBand = kNumLists + 5,         // work queue heads of k_dp_band (+0) and k_dp_band_wide (+1)
    kCntBandFallback = kNumLists + 7,     // band reads whose search could not be certified (re-scored exactly)
    kCntMissB = kNumLists + 8,            // [kWinBuckets] loci whose search left the table window, per motif-length bucket
    kCntLociB = kNumLists + 8 + 5,        // [kWinBuckets] loci per motif-length bucket
    kCntTotal = kNumLists + 8 + 10,
    kCntLongNeed = 56                     // the largest scratch slot (int32 units) an item of k_dp_long asked for and did not get
};
static_assert(kCntTotal <= 48 && kCntLongNeed < 64, "counters: 64 ints; 45..55 belong to the -DSTRK_PHASE_TIMING aid");
// motif-length buckets of the adaptive candidate window: 1-2, 3-4, 5-6, 7-10, 11 and more bases (round 4: the two long-motif
// buckets may go below +-6 — tools/window_need.py: no locus of BASELINE config 4 with a motif of 11+ bases needs more than
// +-4, 0.7 % of those with 7-10 bases do — and a narrow window is what puts a long motif's band into 256 instead of 384 diagonals)
constexpr int kWinBuckets = 5;
__host__ __device__ constexpr int win_bucket(int m) { return m <= 2 ? 0 : (m <= 4 ? 1 : (m <= 6 ? 2 : (m <= 10 ? 3 : 4))); }
static_assert(kCntTotal <= 48, "counters 48..55 belong to the phase-timing aid, the 64-bit counters start at int 64");
// wide band classes (k_dp_band_wide: 2, 3, 6, 7) -> 0..3, and rows -> bucket of the longest-first order
constexpr int kNumWideLists = 4;
__host__ __device__ constexpr int wide_slot(int band_cls) { return (band_cls & 1) | ((band_cls >> 2) << 1); }
__host__ __device__ constexpr int wide_slot_class(int slot) { return 2 + (slot & 1) + ((slot >> 1) << 2); }
__host__ __device__ constexpr int wide_bucket(int rows) { return rows >= (255 << 6) ? 255 : (rows < 0 ? 0 : rows >> 6); }
// slots of KArgs::cells that split cells[0] by kernel; cell_slot_of_list: the slot of a class list's kernel (-1: generic)
constexpr int kCellBand = 6, kCellWide = 7, kCellExact = 8, kCellLong = 9;
__host__ __device__ constexpr int cell_slot_of_list(int c) {
    return c >= kBandClass0 ? (band_class_wide_kernel(c - kBandClass0) ? kCellWide : kCellBand)
                            : (c == kLongClass ? kCellLong : (c == kGenericClass ? -1 : kCellExact));
}
constexpr int kErrBadInput = 1;   // empty motif / negative length
constexpr int kErrScratch = 2;    // generic scratch exhausted
constexpr int kErrEmpty = 4;      // nothing scored for some read
constexpr int kErrList = 8;       // a class list is full (more items than the call sized its lists for)
constexpr int kErrLongSlot = 16;  // a window too long for a k_dp_long scratch slot
constexpr int kSpecMiss = 1, kSpecEmpty = 2;

// ---------------------------------------------------------------------------------------------
// Plan: per read -> locus id, candidate window, table slot, kernel class.
// ---------------------------------------------------------------------------------------------
__device__ inline int classify(int nfl, int ntr, int nfr, int m, int lo, int n, int force_generic, int ref_mode) {
    const long long ndb = (long long)nfl + ntr + nfr;
    const long long rows = (long long)nfl + (long long)(lo + n - 1) * m;
    if (force_generic || nfl < 1 || (nfr < 1 && !ref_mode) || n > kTableMax || m > kMotifMax) return kGenericClass;
    for (int c = ref_mode ? kNumClasses - 1 : 0; c < kNumClasses; ++c) {  // ref mode: one specialisation (the widest class)
        const int cap = class_cap(c);
        if (ndb + 1 <= cap && rows <= cap + kRowSlack && (nfr <= kFastFlankMax || ref_mode)) return c;
    }
    if (!ref_mode && nfl <= kLongFlankMax && nfr <= kLongFlankMax && ndb + 1 <= (long long)kLongTile * kLongMaxTiles &&
        rows <= (1 << 20))
        return kLongClass;
    return kGenericClass;
}

// ---------------------------------------------------------------------------------------------
// Dedupe (the device-side counterpart of the reference's lru_cache, strkit/call/repeats.py:47):
// reads of one locus with byte-identical fl|tr|fr, equal split lengths and equal start estimate
// share ONE score table.  k_hash gives every read a 64-bit content hash; k_plan compares a read
// with the earlier reads of its locus (hash first, then every byte) and lists only first occurrences.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hash_mix(unsigned long long h, unsigned long long v) {
    h ^= v;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// Eight lanes per read: lane s hashes the 8-byte words s, s + 8, ... of the window (coalesced 64-byte rows), the
// partial hashes are folded in lane order with three DPP-free shuffles.  Equal reads give equal hashes; unequal
// reads with equal hashes are told apart byte by byte in k_plan.
__global__ void __launch_bounds__(256) k_hash(KArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 3, sl = gid & 7;
    const bool act = r < a.n_reads;
    unsigned long long h = 0xCBF29CE484222325ull + (unsigned long long)sl;
    if (act) {
        const uint8_t* p = a.seqs + a.seq_off[r];
        const int len = (int)(a.seq_off[r + 1] - a.seq_off[r]);
        const int nw = (len + 7) >> 3;
        for (int w = sl; w < nw; w += 8) {
            unsigned long long v = 0;
            const int i = w << 3;
            if (i + 8 <= len) __builtin_memcpy(&v, p + i, 8);
            else for (int k = 0; i + k < len; ++k) v |= (unsigned long long)p[i + k] << (8 * k);
            h = hash_mix(h, v + 0x9E3779B97F4A7C15ull * (unsigned long long)(w + 1));
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {   // lanes 0..7 of a read are adjacent: fold (0,1) (2,3).. then pairs of pairs
        const unsigned long long other = (unsigned long long)__shfl_down((long long)h, o, 8);
        h = hash_mix(h, other);
    }
    if (act && sl == 0) {
        h = hash_mix(h, ((unsigned long long)(unsigned)a.nfl[r] << 32) | (unsigned)a.ntr[r]);
        h = hash_mix(h, ((unsigned long long)(unsigned)a.nfr[r] << 32) | (unsigned)a.est_cn[r]);
        a.rhash[r] = h;
    }
}

__device__ inline bool same_read(const KArgs& a, int r, int q) {
    if (a.nfl[r] != a.nfl[q] || a.ntr[r] != a.ntr[q] || a.nfr[r] != a.nfr[q] || a.est_cn[r] != a.est_cn[q]) return false;
    const uint8_t* x = a.seqs + a.seq_off[r];
    const uint8_t* y = a.seqs + a.seq_off[q];
    const int len = a.nfl[r] + a.ntr[r] + a.nfr[r];
    int i = 0;
    for (; i + 8 <= len; i += 8) {
        unsigned long long u, v;
        __builtin_memcpy(&u, x + i, 8);
        __builtin_memcpy(&v, y + i, 8);
        if (u != v) return false;
    }
    for (; i < len; ++i)
        if (x[i] != y[i]) return false;
    return true;
}

// mode 0: windows from est_cn +/- window, table slot r*table_stride;
// mode 1: windows and table offsets are already in win_lo / win_n / tab_off (strk_score_table,
//         window-miss rounds); `items` (optional) restricts the launch to a list of reads.
// Class lists are filled with one global atomic per class per block (LDS histogram first).
__global__ void __launch_bounds__(256) k_plan(KArgs a, int mode, const int32_t* items, int n_items, int force_generic) {
    __shared__ int s_cnt[kNumLists];
    __shared__ int s_base[kNumLists];
    __shared__ unsigned long long s_cells, s_bytes_band, s_bytes_exact, s_bytes_wide, s_bytes_long, s_cells_k[4];
    __shared__ __attribute__((aligned(16))) unsigned s_key[256];
    __shared__ int s_below[kNumBandClasses];   // band items of this block in band classes below c
    __shared__ int s_lb[kWinBuckets];   // loci per motif-length bucket (counted at a locus's first read)
    if (threadIdx.x < kWinBuckets) s_lb[threadIdx.x] = 0;
    if (threadIdx.x < kNumLists) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_cells = 0; s_bytes_band = 0; s_bytes_exact = 0; s_bytes_wide = 0; s_bytes_long = 0; }
    if (threadIdx.x < 4) s_cells_k[threadIdx.x] = 0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int r = 0, l = 0, nfl = 0, ntr = 0, nfr = 0, m = 1, lo = 0, n = 0;
    if (gid < n_items) {
        r = items ? items[gid] : gid;
        // locus of read r: last l with read_off[l] <= r
        int lo_l = 0, hi_l = a.n_loci;
        while (hi_l - lo_l > 1) {
            const int mid = (lo_l + hi_l) >> 1;
            if (a.read_off[mid] <= r) lo_l = mid; else hi_l = mid;
        }
        l = lo_l;
        a.read_locus[r] = l;
        m = a.motif_off[l + 1] - a.motif_off[l];
        if (mode == 0 && r == a.read_off[l] && m >= 1) atomicAdd(&s_lb[win_bucket(m)], 1);
        nfl = a.nfl[r]; ntr = a.ntr[r]; nfr = a.nfr[r];
        if (m < 1 || nfl < 0 || ntr < 0 || nfr < 0) {
            atomicOr(&a.counters[kCntError], kErrBadInput);
            a.win_n[r] = 0;
            n = 0;
        } else if (mode == 0) {
            const long long est = a.est_cn[r];
            // the estimate round(|tr| / |motif|) drifts with the copy number (indels accumulate): widen the
            // window by one size per 128 copies, as far as the table stride allows
            const long long w = min((long long)a.window_b[win_bucket(m)] + min(max(est, 0ll) >> 7, 7ll), (long long)(a.table_stride - 1) / 2);
            long long w_lo = est - w, w_hi = est + w;
            if (w_lo < 0) w_lo = 0;
            if (w_hi < w_lo) w_hi = w_lo;  // negative estimates: keep a one-entry window at 0
            if (w_hi - w_lo + 1 > a.table_stride) w_hi = w_lo + a.table_stride - 1;
            lo = (int)w_lo;
            n = (int)(w_hi - w_lo + 1);
            int rep = r;
            if (a.rhash) {
                const unsigned long long h = a.rhash[r];
                for (int q = a.read_off[l]; q < r; ++q)
                    if (a.rhash[q] == h && same_read(a, r, q)) { rep = q; break; }  // first occurrence
            }
            a.rep[r] = rep;
            // no speculative result yet: the kernels that score a read AND search it overwrite this, a read that ends in
            // the generic kernel (scores only) keeps it and is searched by k_replay on its table
            if (a.spec) a.spec[r] = make_int4(0, 0, 0, kSpecMiss);
            a.win_lo[r] = lo;
            a.win_n[r] = n;
            a.tab_off[r] = (int64_t)rep * a.table_stride;
            if (rep != r) {
                n = 0;  // no DP item: the table (and the speculative search) of `rep` serve this read too
                atomicAdd(&a.counters[kCntDup], 1);
            }
        } else {
            lo = a.win_lo[r];
            n = a.win_n[r];
        }
    }
    // Long windows (window-miss rounds, explicit tables) are cut into items of <= kTableMax sizes.
    unsigned long long cells = 0;
    const unsigned long long ndb = (unsigned long long)nfl + ntr + nfr;
    int band_list = -1;   // >= 0: the read goes to the band kernel first
    if (a.band_mode && mode == 0 && n > 0 && n <= kTableMax && !force_generic && r < a.band_limit) {
        const BandGeo geo = band_geometry(nfl, ntr, nfr, m, lo, n, a.band_tune);
        if (geo.ok) band_list = kBandClass0 + geo.cls;
    }
    if (a.exact && gid < n_items && n > 0) a.exact[r] = band_list < 0;
    int first_c = -1;
    for (int k0 = 0; k0 < n; k0 += kTableMax) {
        const int nn = min(kTableMax, n - k0);
        const int c = band_list >= 0 ? band_list : classify(nfl, ntr, nfr, m, lo + k0, nn, force_generic, a.ref_mode);
        if (k0 == 0) first_c = c;
        atomicAdd(&s_cnt[c], 1);
        unsigned long long cc = 0;
        if (c == kGenericClass) {
            for (int k = 0; k < nn; ++k) cc += ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + k) * m + nfr);
        } else if (band_list >= 0) {
            cc = (unsigned long long)band_class_wd(band_list - kBandClass0) * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + nn - 1) * m + nfr);
        } else {
            cc = ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + nn - 1) * m + nfr);
        }
        cells += cc;
        if (cc && cell_slot_of_list(c) >= 0) atomicAdd(&s_cells_k[cell_slot_of_list(c) - kCellBand], cc);
    }
    if (cells) {
        atomicAdd(&s_cells, cells);
        atomicAdd(band_list >= 0 ? &s_bytes_band : &s_bytes_exact, ndb + 16);
        if (band_list >= 0 && band_class_wide_kernel(band_list - kBandClass0)) atomicAdd(&s_bytes_wide, ndb + 16);   // k_dp_band_wide's classes
        if (first_c == kLongClass) atomicAdd(&s_bytes_long, ndb + 16);
    }
    __syncthreads();
    if (threadIdx.x < kNumLists) {
        const int c = threadIdx.x;
        s_base[c] = s_cnt[c] ? atomicAdd(&a.counters[kCntClass0 + c], s_cnt[c]) : 0;
    }
    __syncthreads();
    if (threadIdx.x < kNumBandClasses) {
        int below = 0;
        for (int k = 0; k < (int)threadIdx.x; ++k) below += s_cnt[kBandClass0 + k];
        s_below[threadIdx.x] = below;
    }
    __syncthreads();
    if (threadIdx.x < kNumLists) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < kWinBuckets && s_lb[threadIdx.x]) atomicAdd(&a.counters[kCntLociB + threadIdx.x], s_lb[threadIdx.x]);
    if (threadIdx.x == 0 && s_cells) {
        atomicAdd(a.cells, s_cells);
        if (s_bytes_band) atomicAdd(a.cells + 2, s_bytes_band);
        if (s_bytes_exact) atomicAdd(a.cells + 3, s_bytes_exact);
        if (s_bytes_wide) atomicAdd(a.cells + 4, s_bytes_wide);
        if (s_bytes_long) atomicAdd(a.cells + 5, s_bytes_long);
        for (int k = 0; k < 4; ++k)
            if (s_cells_k[k]) atomicAdd(a.cells + kCellBand + k, s_cells_k[k]);
    }
    __syncthreads();
    // Band items of a block are listed in the order of (band class, prefix rows): the band kernel runs the 8 (4, 2, 1) items of
    // a chunk in lock-step for as many steps as the longest one needs (chunks of a 256-read block in arrival order: 8.5 % more
    // steps than the items need; sorted by rows: 4 %).  rank = items of this block that sort before mine.
    int band_rank = 0;
    {
        // 32-bit keys (class | prefix rows | thread): items that sort before mine = keys below mine, four per LDS read; those of
        // the lower classes are the same number for every item of a class (s_below)
        const unsigned mine = band_list >= 0 ? ((unsigned)(band_list - kBandClass0) << 28) |
                                                   ((unsigned)min(nfl + (lo + n - 1) * m, (1 << 20) - 1) << 8) | threadIdx.x
                                             : ~0u;
        s_key[threadIdx.x] = mine;
        __syncthreads();
        if (band_list >= 0) {
            const uint4* k4 = reinterpret_cast<const uint4*>(s_key);
            for (int q = 0; q < 64; ++q) {
                const uint4 o = k4[q];
                band_rank += (o.x < mine) + (o.y < mine) + (o.z < mine) + (o.w < mine);
            }
            band_rank -= s_below[band_list - kBandClass0];
        }
    }
    for (int k0 = 0; k0 < n; k0 += kTableMax) {
        const int nn = min(kTableMax, n - k0);
        const int c = band_list >= 0 ? band_list : classify(nfl, ntr, nfr, m, lo + k0, nn, force_generic, a.ref_mode);
        const int idx = s_base[c] + ((band_list >= 0 && !(a.dbg & 16)) ? band_rank : atomicAdd(&s_cnt[c], 1));
        if (idx < a.list_stride) {
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx] = r;
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx + 1] = band_list >= 0 ? l : k0;   // band items: the locus (k0 is 0)
            if (band_list >= 0) {   // the band kernel fetches an item with one level of loads (and one chunk ahead)
                int4* rec = a.band_recs + ((size_t)(band_list - kBandClass0) * a.list_stride + idx) * 3;
                const long long so = a.seq_off[r];
                rec[0] = make_int4(r, l, nfl, ntr);
                rec[1] = make_int4(nfr, m, lo, n);
                rec[2] = make_int4((int)(so & 0xffffffffll), (int)(so >> 32), a.motif_off[l], a.est_cn[r]);
            }
        } else {
            atomicOr(&a.counters[kCntError], kErrList);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One read (strk_repeat_count, the scalar drop-in of repeats.py:47-70): the plan of its short launch chain.  Zeroes the counters,
// gives read 0 its window by k_plan's rule, its class and its list entry; k_dp_all (one block) then scores the window and
// searches it from the start count (the "speculative" search for start == est_cn IS the call's search: no feedback here).
// A read no fast class takes, or whose search leaves the window, keeps spec[0] = miss and goes the general way.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_scalar_plan(KArgs a, int n_counter_ints) {
    for (int i = threadIdx.x; i < n_counter_ints; i += 64) a.counters[i] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int nfl = a.nfl[0], ntr = a.ntr[0], nfr = a.nfr[0], m = a.motif_off[1] - a.motif_off[0];
    a.read_locus[0] = 0;
    a.tab_off[0] = 0;
    a.spec[0] = make_int4(0, 0, 0, kSpecMiss);
    a.win_lo[0] = 0;
    a.win_n[0] = 0;
    if (m < 1 || nfl < 0 || ntr < 0 || nfr < 0) { a.counters[kCntError] = kErrBadInput; return; }
    const long long est = a.est_cn[0];
    const long long w = min((long long)a.window + min(max(est, 0ll) >> 7, 7ll), (long long)(a.table_stride - 1) / 2);
    long long w_lo = est - w, w_hi = est + w;
    if (w_lo < 0) w_lo = 0;
    if (w_hi < w_lo) w_hi = w_lo;
    if (w_hi - w_lo + 1 > a.table_stride) w_hi = w_lo + a.table_stride - 1;
    const int lo = (int)w_lo, n = (int)(w_hi - w_lo + 1);
    a.win_lo[0] = lo;
    a.win_n[0] = n;
    int c = classify(nfl, ntr, nfr, m, lo, n, 0, 0);
    if (c < kNumClasses) {
        // one read alone: latency, not throughput — the class whose wave finishes first (more lanes, fewer columns per lane: a step
        // costs about 9/4 CL + 12 instructions and a pass has rows + G - 1 of them), not the one with the fewest lanes
        const long long ndb = (long long)nfl + ntr + nfr, rows = (long long)nfl + (long long)(lo + n - 1) * m;
        long long best = -1;
        for (int k = c; k < kNumClasses; ++k) {
            const int cap = class_cap(k);
            if (ndb + 1 > cap || rows > cap + kRowSlack) continue;
            const long long cost = (rows + nfr + 2 * class_G(k)) * (9 * class_CL(k) / 4 + 12);
            if (best < 0 || cost < best) { best = cost; c = k; }
        }
        a.cls_list[(size_t)c * a.list_stride * 2] = 0;
        a.cls_list[(size_t)c * a.list_stride * 2 + 1] = 0;
        a.counters[kCntClass0 + c] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Longest first: the items of the four wide band classes, re-listed by descending prefix rows (counting sort on rows / 64 from
// k_plan's census; order inside a bucket is whatever the cursors hand out — results do not depend on it).  A wave of
// k_dp_band_wide works for milliseconds on one 12 000-row read: the queue must not hold such a chunk back until other waves
// have nothing left (BASELINE config 5 at one GPU's share: 6 400 chunks for 1 900 waves, class by class but in arrival order
// inside a class), and two reads that share a wave in lock step should be of one length.
// ---------------------------------------------------------------------------------------------
// Two kernels of kSortWideBlocks blocks per class (round 4, last hours).  The first form took the histogram from k_plan — one
// global atomic per wide item — which cost config 4's shard (16 000 wide chunks) what the order gave back, so only small queues
// were sorted; a single block per class with the histogram in LDS took a millisecond for 32 000 records.  Now every block
// counts its slice in LDS and adds its 256 counts to the class's histogram (k_sort_wide_hist); then (k_sort_wide) it counts
// again, turns the class histogram into bucket starts, reserves its range per bucket with one atomic on the bucket's cursor
// and scatters its records by LDS-local offsets.  Global atomics: two per block and occupied bucket.
constexpr int kSortWideBlocks = 16;
__device__ __forceinline__ int wide_rec_bucket(const int4* recs, int idx) {
    const int4 q0 = recs[(size_t)idx * 3], q1 = recs[(size_t)idx * 3 + 1];
    return 255 - wide_bucket(q0.z + (q1.z + q1.w - 1) * q1.y);   // nfl + (lo + n - 1) * m; bucket 0 = the longest
}
__global__ void __launch_bounds__(256) k_sort_wide_hist(KArgs a) {
    __shared__ int s_hist[256];
    const int s = blockIdx.y, cls = wide_slot_class(s);
    const int count = min(a.counters[kCntClass0 + kBandClass0 + cls], a.list_stride);
    if (count <= 0) return;
    const int4* recs = a.band_recs + (size_t)cls * a.list_stride * 3;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += kSortWideBlocks * 256) atomicAdd(&s_hist[wide_rec_bucket(recs, i)], 1);
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&a.wide_hist[s * 256 + threadIdx.x], s_hist[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_sort_wide(KArgs a, int4* out) {
    __shared__ int s_cnt[256], s_base[256], s_wave[4];
    const int s = blockIdx.y, cls = wide_slot_class(s);
    const int count = min(a.counters[kCntClass0 + kBandClass0 + cls], a.list_stride);
    if (count <= 0) return;
    const int4* recs = a.band_recs + (size_t)cls * a.list_stride * 3;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += kSortWideBlocks * 256) atomicAdd(&s_cnt[wide_rec_bucket(recs, i)], 1);
    __syncthreads();
    {   // where bucket t of the class begins (exclusive prefix sum of the class histogram), plus this block's reserved offset in it
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int v = a.wide_hist[s * 256 + threadIdx.x];
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        int base = incl - v;
        for (int k = 0; k < w; ++k) base += s_wave[k];
        const int mine = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = mine ? base + atomicAdd(&a.wide_hist[(kNumWideLists + s) * 256 + threadIdx.x], mine) : 0;
        s_cnt[threadIdx.x] = 0;   // re-used as this block's cursor inside its range
    }
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += kSortWideBlocks * 256) {
        const int4 q0 = recs[(size_t)i * 3], q1 = recs[(size_t)i * 3 + 1], q2 = recs[(size_t)i * 3 + 2];
        const int b = 255 - wide_bucket(q0.z + (q1.z + q1.w - 1) * q1.y);
        const int pos = s_base[b] + atomicAdd(&s_cnt[b], 1);
        int4* o = out + ((size_t)cls * a.list_stride + pos) * 3;
        o[0] = q0; o[1] = q1; o[2] = q2;
    }
}

// ---------------------------------------------------------------------------------------------
// Longest first for k_dp_long too.  One read per wave and 2 048 resident waves: a call with more long items than that (config
// 5's shape when no certificate holds — noisy reads) is as long as whatever its last waves pick up, and the items differ by a
// factor of a thousand in cells (|db| x rows, 50-2 000 copies).  Measured by itself (tools/prof_long.sh, 13 560 items): the
// waves were resident for 65 % of the launch.  The list is filled by k_plan, by the band kernels' fall-backs and by k_replay,
// so the order is made here, right in front of the kernel: one block, a histogram of cost (column tiles x steps, in 1 024
// buckets), positions from a suffix sum, a scatter.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_sort_long(KArgs a, int32_t* out) {
    __shared__ int s_hist[1024], s_start[1024];
    const int count = min(a.counters[kCntClass0 + kLongClass], a.list_stride);
    if (count <= 0) return;
    const int32_t* list = a.cls_list + (size_t)kLongClass * a.list_stride * 2;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    auto bucket_of = [&](int r, int k0) {
        const int l = a.read_locus[r];
        const int m = a.motif_off[l + 1] - a.motif_off[l];
        const long long ndb = (long long)a.nfl[r] + a.ntr[r] + a.nfr[r];
        const long long rows = (long long)a.nfl[r] + (long long)(a.win_lo[r] + k0 + min(kTableMax, a.win_n[r] - k0) - 1) * m;
        const long long cost = ((ndb + kLongTile) / kLongTile) * (rows + 64);
        return 1023 - (int)min(1023ll, cost >> 9);        // bucket 0 = the longest
    };
    for (int i = threadIdx.x; i < count; i += 1024) atomicAdd(&s_hist[bucket_of(list[2 * i], list[2 * i + 1])], 1);
    __syncthreads();
    {   // exclusive prefix sum over the 1 024 buckets: per-wave scan, then the 16 wave totals
        __shared__ int s_wave[16];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int v = s_hist[threadIdx.x];
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w; ++k) base += s_wave[k];
        s_start[threadIdx.x] = base + incl - v;
        s_hist[threadIdx.x] = 0;                            // re-used as the buckets' cursors
    }
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 1024) {
        const int r = list[2 * i], k0 = list[2 * i + 1];
        const int b = bucket_of(r, k0);
        const int pos = s_start[b] + atomicAdd(&s_hist[b], 1);
        out[2 * pos] = r;
        out[2 * pos + 1] = k0;
    }
}

}  // namespace strk

#include "strk_dp_exact.h"
#include "strk_dp_band.h"
#include "strk_dp_long.h"
#include "strk_replay.h"
