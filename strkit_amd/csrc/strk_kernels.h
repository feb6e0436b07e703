// strk_kernels.h — gfx950 device code of the repeat-count path (included by strk_api.hip only).
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
constexpr int kBandClass0 = kNumClasses + 2;    // band kernel lists: + class (G = 8 << class lanes, 16 G diagonals)
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
    int32_t* counters;    // see Counter enum
    unsigned long long* cells;  // DP cells executed; cells[2] / cells[3]: algorithmic bytes (|window| + 16 per read) of the
                                // items routed to the band kernel / to the exact kernels by k_plan
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
    int32_t window;       // half width (plan kernel)
    int32_t table_stride; // entries per read (plan kernel)
    int32_t max_iters, lsr, step, tie_last;  // search parameters (speculative search in k_dp_all)
    int32_t band_mode;    // 1: eligible reads go through k_dp_band first (strk_search.h, "Banded scoring")
    uint8_t* exact;       // [n_reads] 1: the read's table holds exact scores, 0: band lower bounds
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
    kCntTotal = kNumLists + 8
};
constexpr int kErrBadInput = 1;   // empty motif / negative length
constexpr int kErrScratch = 2;    // generic scratch exhausted
constexpr int kErrEmpty = 4;      // nothing scored for some read
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

__global__ void __launch_bounds__(256) k_hash(KArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    const uint8_t* p = a.seqs + a.seq_off[r];
    const int len = (int)(a.seq_off[r + 1] - a.seq_off[r]);
    unsigned long long h = 0xCBF29CE484222325ull;
    int i = 0;
    for (; i + 8 <= len; i += 8) {
        unsigned long long v;
        __builtin_memcpy(&v, p + i, 8);
        h = hash_mix(h, v);
    }
    unsigned long long tail = 0;
    for (int k = 0; i + k < len; ++k) tail |= (unsigned long long)p[i + k] << (8 * k);
    h = hash_mix(h, tail);
    h = hash_mix(h, ((unsigned long long)(unsigned)a.nfl[r] << 32) | (unsigned)a.ntr[r]);
    h = hash_mix(h, ((unsigned long long)(unsigned)a.nfr[r] << 32) | (unsigned)a.est_cn[r]);
    a.rhash[r] = h;
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
    __shared__ unsigned long long s_cells, s_bytes_band, s_bytes_exact;
    if (threadIdx.x < kNumLists) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_cells = 0; s_bytes_band = 0; s_bytes_exact = 0; }
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
        nfl = a.nfl[r]; ntr = a.ntr[r]; nfr = a.nfr[r];
        if (m < 1 || nfl < 0 || ntr < 0 || nfr < 0) {
            atomicOr(&a.counters[kCntError], kErrBadInput);
            a.win_n[r] = 0;
            n = 0;
        } else if (mode == 0) {
            const long long est = a.est_cn[r];
            // the estimate round(|tr| / |motif|) drifts with the copy number (indels accumulate): widen the
            // window by one size per 128 copies, as far as the table stride allows
            const long long w = min((long long)a.window + min(max(est, 0ll) >> 7, 7ll), (long long)(a.table_stride - 1) / 2);
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
    if (a.band_mode && mode == 0 && n > 0 && n <= kTableMax && !force_generic) {
        const BandGeo geo = band_geometry(nfl, ntr, nfr, m, lo, n);
        if (geo.ok) band_list = kBandClass0 + geo.cls;
    }
    if (a.exact && gid < n_items && n > 0) a.exact[r] = band_list < 0;
    for (int k0 = 0; k0 < n; k0 += kTableMax) {
        const int nn = min(kTableMax, n - k0);
        const int c = band_list >= 0 ? band_list : classify(nfl, ntr, nfr, m, lo + k0, nn, force_generic, a.ref_mode);
        atomicAdd(&s_cnt[c], 1);
        if (c == kGenericClass) {
            for (int k = 0; k < nn; ++k) cells += ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + k) * m + nfr);
        } else if (band_list >= 0) {
            cells += (unsigned long long)(128 << (band_list - kBandClass0)) * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + nn - 1) * m + nfr);
        } else {
            cells += ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + nn - 1) * m + nfr);
        }
    }
    if (cells) {
        atomicAdd(&s_cells, cells);
        atomicAdd(band_list >= 0 ? &s_bytes_band : &s_bytes_exact, ndb + 16);
    }
    __syncthreads();
    if (threadIdx.x < kNumLists) {
        const int c = threadIdx.x;
        s_base[c] = s_cnt[c] ? atomicAdd(&a.counters[kCntClass0 + c], s_cnt[c]) : 0;
        s_cnt[c] = 0;
    }
    if (threadIdx.x == 0 && s_cells) {
        atomicAdd(a.cells, s_cells);
        if (s_bytes_band) atomicAdd(a.cells + 2, s_bytes_band);
        if (s_bytes_exact) atomicAdd(a.cells + 3, s_bytes_exact);
    }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += kTableMax) {
        const int nn = min(kTableMax, n - k0);
        const int c = band_list >= 0 ? band_list : classify(nfl, ntr, nfr, m, lo + k0, nn, force_generic, a.ref_mode);
        const int idx = s_base[c] + atomicAdd(&s_cnt[c], 1);
        if (idx < a.list_stride) {
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx] = r;
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx + 1] = band_list >= 0 ? l : k0;   // band items: the locus (k0 is 0)
        } else {
            atomicOr(&a.counters[kCntError], kErrScratch);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast DP kernel
// ---------------------------------------------------------------------------------------------
constexpr int kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;  // gfx9 DPP controls, present on gfx950
constexpr int kCLMax = 40;   // columns per lane of the largest class
constexpr int kNQMax = kCLMax / 4;

constexpr int kDppRowShr1 = 0x111, kDppRowShl1 = 0x101;
// from_left<G>(keep, v): lane l gets v of lane l-1; the first lane of every group gets `keep`.
//   G = 16: a DPP row is one group (row_shr:1 leaves `keep` in its first lane);
//   G = 64: wave_shr:1;  G = 32 / 8: wave_shr:1 / row_shr:1, then a select patches the seam lanes.
// `keep` must be wave-uniform (it is the boundary value of the group's edge lane at this step).
template <int G>
__device__ __forceinline__ int from_left(int keep, int v, bool edge_lane) {
    if (G == 16) return __builtin_amdgcn_update_dpp(keep, v, kDppRowShr1, 0xf, 0xf, false);
    const int x = __builtin_amdgcn_update_dpp(keep, v, G == 8 ? kDppRowShr1 : kDppWaveShr1, 0xf, 0xf, false);
    return ((G == 32 || G == 8) && edge_lane) ? keep : x;
}
template <int G>
__device__ __forceinline__ int from_right(int keep, int v, bool edge_lane) {
    if (G == 16) return __builtin_amdgcn_update_dpp(keep, v, kDppRowShl1, 0xf, 0xf, false);
    const int x = __builtin_amdgcn_update_dpp(keep, v, G == 8 ? kDppRowShl1 : kDppWaveShl1, 0xf, 0xf, false);
    return ((G == 32 || G == 8) && edge_lane) ? keep : x;
}

// LDS operations of one group never leave its wave: order them with a wave-level fence.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// max over the (at most eight) groups of a wave of a value that is uniform inside each group
__device__ __forceinline__ int wave_max_over_groups(int v) {
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int l = 8; l < 64; l += 8) m = max(m, __builtin_amdgcn_readlane(v, l));
    return m;
}

// Per-group LDS layout of a class (G lanes x CL columns); all offsets are multiples of 16.
struct DpLayout {
    int cap, off_db, off_cp, off_ct, off_b0, group_bytes;
    static constexpr int OFF_TBL = 0;                          // 18 x 8 B row words
    static constexpr int OFF_COMB = OFF_TBL + 18 * 8;          // kTableMax ints
    static constexpr int OFF_LMAX = OFF_COMB + kTableMax * 4;  // kTableMax ints
    static constexpr int OFF_MISC = OFF_LMAX + kTableMax * 4;  // 4 ints: symmask, zfree
    __host__ __device__ constexpr DpLayout(int G, int CL)
        : cap(G * CL),
          off_db(OFF_MISC + 16),                                             // 4 pad + CAP + 4 pad bytes
          off_cp(off_db + ((G * CL + 8 + 15) & ~15)),                        // prefix rows
          off_ct(off_cp + ((G * CL + kRowSlack + 2 * G + 4 + 15) & ~15)),    // tail rows (reversed fr)
          off_b0(off_ct + ((kFastFlankMax + 2 * G + 4 + 15) & ~15)),         // backward result, u16 per slot
          group_bytes(off_b0 + ((G * CL * 2 + 15) & ~15)) {}
};
__host__ __device__ constexpr int wave_lds_bytes(int c) { return (64 / class_G(c)) * DpLayout(class_G(c), class_CL(c)).group_bytes; }
__host__ __device__ constexpr int max_wave_lds_bytes(int c) {
    return c < 0 ? 0 : (wave_lds_bytes(c) > max_wave_lds_bytes(c - 1) ? wave_lds_bytes(c) : max_wave_lds_bytes(c - 1));
}
constexpr int kWaveLdsBytes = max_wave_lds_bytes(kNumClasses - 1);
constexpr int kLdsSlack = 2048 + 256;  // stale row symbols (any byte) may index up to 255*8 B past a row-word table

typedef const __attribute__((address_space(4))) KArgs* KArgsKernarg;

// Everything the two DP passes of one wave need; G and CL are wave-uniform run-time values.
struct PassCtx {
    int G, CL, lig;
    bool first, last, act;
    int ndb;
    bool dbBeg, dbEnd, cBeg, cEnd;
    const uint2* tbl;        // LDS: per-symbol row words
    const unsigned* selw;    // LDS: selector words of this lane, selw[q] <-> db[lig*CL + 4q - 4 .. -1]
    uint2* b0;               // LDS: backward result of this lane, b0[q * G] <-> slots 4q..4q+3 (u16 each)
};

// One DP row in G-space over the lane's 4*NQ columns: dst = max3(up, left, diag + w).  FWD walks
// the columns left to right, the backward pass right to left; src/dst alternate (no register copies).
template <int NQ, bool FWD>
__device__ __forceinline__ int dp_row(const int (&src)[4 * NQ], int (&dst)[4 * NQ], const unsigned (&sel)[NQ], uint2 word,
                                      int edge, int edge_prev) {
    int d = edge_prev, l = edge;
#pragma unroll
    for (int i = 0; i < 4 * NQ; ++i) {
        const int c = FWD ? i : 4 * NQ - 1 - i;
        const unsigned wb = __builtin_amdgcn_perm(word.y, word.x, sel[c / 4]);
        const int nh = max(max(src[c], l), d + (int)((wb >> (8 * (c % 4))) & 0xffu));
        d = src[c]; l = nh; dst[c] = nh;
    }
    return l;  // the lane's outgoing column
}

// Backward pass over the fr rows (k' = 1..rowsT consume fr[rowsT-k']).  Slot s holds node j = s
// (db chars s.. remain) for s < ndb; slots >= ndb are inert pads that carry the boundary value.
// Leaves Gb(rowsT, .) in LDS (b0) and returns max_{k'<rowsT} (Gb(k', 0) - g*k') for lane 0.
template <int NQ, int G>
__device__ __forceinline__ int bwd_pass(const PassCtx& x, int rowsT, const uint8_t* ct) {
    constexpr int g = kGap, CL = 4 * NQ;
    int Ha[CL], Hb[CL];
    unsigned sel[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sel[q] = x.act ? x.selw[q + 1] : 0x0c0c0c0cu;
#pragma unroll
    for (int c = 0; c < CL; ++c) {
        const int s = x.lig * CL + c;
        int v = 0;
        if (s < x.ndb && x.dbEnd) v = g * (x.ndb - s) - (s == 0 ? g : 0);
        Ha[c] = v;
    }
    const int Tb = (wave_max_over_groups(rowsT > 0 ? rowsT + G - 1 : 0) + 1) & ~1;
    const int bstep = x.cEnd ? g : 0;
    const int gkEvent = g * rowsT;
    int zsave = 0;
    int hout = Ha[0];
    int edgePrev = from_right<G>(0, hout, x.last);
    int gk = g * (x.lig - (G - 1));   // g * k' of the row this lane finished before step 0
    int zmax = Ha[0];
    const uint8_t* pa = ct + x.lig;   // row symbol of step t is pa[t]
    uint2 wordNext = x.tbl[pa[0]];
    unsigned symNext = pa[1];
#define STRK_BSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = x.tbl[symNext];                                                           \
        symNext = pa[(T) + 2];                                                               \
        const int edge = from_right<G>(bstep * ((T) + 1), hout, x.last);                        \
        hout = dp_row<NQ, false>(SRC, DST, sel, word, edge, edgePrev);                       \
        edgePrev = edge;                                                                     \
        gk += g;                                                                             \
        if (gk == gkEvent) {                                                                 \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                   \
                x.b0[q * G] = make_uint2((unsigned)DST[4 * q] | ((unsigned)DST[4 * q + 1] << 16), \
                                         (unsigned)DST[4 * q + 2] | ((unsigned)DST[4 * q + 3] << 16)); \
            zsave = zmax;                                                                    \
        }                                                                                    \
        zmax = max(zmax, hout - gk);                                                         \
        if ((T) == G - 2) zmax = hout;                                                       \
    }
    for (int t = 0; t < Tb; t += 2) {
        STRK_BSTEP(Ha, Hb, t)
        STRK_BSTEP(Hb, Ha, t + 1)
    }
#undef STRK_BSTEP
    return zsave;
}

// Forward pass over fl + motif*i_hi.  Slot 0 is an inert pad carrying the left boundary; slot
// s = 1..ndb holds node j = s (consumes db[s-1]); slots > ndb replicate the last column.  At the
// fork rows R_k = nfl + (lo+k)*m it folds max_s(Gf + Gb) into comb[k] and records the running
// last-column maximum in lmaxA[k].
template <int NQ, int G>
__device__ __forceinline__ void fwd_pass(const PassCtx& x, int rowsP, const uint8_t* cp, int nEff, int fork0, int m,
                                         int* comb, int* lmaxA) {
    constexpr int g = kGap, CL = 4 * NQ;
    int Ha[CL], Hb[CL];
    unsigned sel[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sel[q] = x.act ? __builtin_amdgcn_alignbyte(x.selw[q + 1], x.selw[q], 3) : 0x0c0c0c0cu;
#pragma unroll
    for (int c = 0; c < CL; ++c) Ha[c] = x.dbBeg ? g * min(x.lig * CL + c, x.ndb) : 0;
    const int Tf = (wave_max_over_groups(nEff > 0 ? rowsP + G - 1 : 0) + 1) & ~1;
    const int bstep = x.cBeg ? g : 0;
    const int gm = g * m;
    int hout = Ha[CL - 1];
    int edgePrev = from_left<G>(0, hout, x.first);
    int gr = -g * x.lig;           // g * row this lane finished before step 0
    int lastmax = kNegInf;
    int forkG = nEff > 0 ? g * fork0 : 0x7fffffff;
    int forkIdx = 0;
    const uint8_t* pa = cp + (G - 1) - x.lig;
    uint2 wordNext = x.tbl[pa[0]];
    unsigned symNext = pa[1];
#define STRK_FSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = x.tbl[symNext];                                                           \
        symNext = pa[(T) + 2];                                                               \
        const int edge = from_left<G>(bstep * ((T) + 1), hout, x.first);                        \
        hout = dp_row<NQ, true>(SRC, DST, sel, word, edge, edgePrev);                        \
        edgePrev = edge;                                                                     \
        gr += g;                                                                             \
        lastmax = max(lastmax, hout - gr);                                                   \
        if ((T) == G - 2) lastmax = kNegInf;                                                 \
        if (gr == forkG) {                                                                   \
            int acc = kNegInf;                                                               \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                 \
                const uint2 bw = x.b0[q * G];                                                \
                acc = max(max(acc, DST[4 * q] + (int)(bw.x & 0xffffu)), DST[4 * q + 1] + (int)(bw.x >> 16)); \
                acc = max(max(acc, DST[4 * q + 2] + (int)(bw.y & 0xffffu)), DST[4 * q + 3] + (int)(bw.y >> 16)); \
            }                                                                                \
            atomicMax(&comb[forkIdx], acc);                                                  \
            if (x.last) lmaxA[forkIdx] = lastmax;                                            \
            ++forkIdx;                                                                       \
            forkG = forkIdx < nEff ? forkG + gm : 0x7fffffff;                                \
        }                                                                                    \
    }
    for (int t = 0; t < Tf; t += 2) {
        STRK_FSTEP(Ha, Hb, t)
        STRK_FSTEP(Hb, Ha, t + 1)
    }
#undef STRK_FSTEP
}

// Reference-side forward pass (score_ref_boundaries, strkit/call/repeats.py:23-43): the candidate is
// fl + motif*i with NO right flank, the db end is free, and both the score and the db position where
// the alignment ends (parasail's end_query) are wanted.  At fork row R_k every slot j >= 1 offers
// H(R_k, j) = G - g*(R_k + j); the fold keeps (value, smallest j) as one 64-bit key
// ((G + g*(ndb - j)) << 20 | (2^20 - 1 - j)) with an LDS 64-bit atomic max.  One instance (G = 64,
// CL = 28) serves every shape: this path runs once per locus, not once per read.
template <int NQ, int G>
__device__ __forceinline__ void fwd_pass_ref(const PassCtx& x, int rowsP, const uint8_t* cp, int nEff, int fork0, int m,
                                             unsigned long long* comb64) {
    constexpr int g = kGap, CL = 4 * NQ;
    int Ha[CL], Hb[CL];
    unsigned sel[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sel[q] = x.act ? __builtin_amdgcn_alignbyte(x.selw[q + 1], x.selw[q], 3) : 0x0c0c0c0cu;
#pragma unroll
    for (int c = 0; c < CL; ++c) Ha[c] = x.dbBeg ? g * min(x.lig * CL + c, x.ndb) : 0;
    const int Tf = (wave_max_over_groups(nEff > 0 ? rowsP + G - 1 : 0) + 1) & ~1;
    const int bstep = x.cBeg ? g : 0;
    const int gm = g * m;
    int hout = Ha[CL - 1];
    int edgePrev = from_left<G>(0, hout, x.first);
    int gr = -g * x.lig;
    int forkG = nEff > 0 ? g * fork0 : 0x7fffffff;
    int forkIdx = 0;
    const uint8_t* pa = cp + (G - 1) - x.lig;
    uint2 wordNext = x.tbl[pa[0]];
    unsigned symNext = pa[1];
#define STRK_RSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = x.tbl[symNext];                                                           \
        symNext = pa[(T) + 2];                                                               \
        const int edge = from_left<G>(bstep * ((T) + 1), hout, x.first);                     \
        hout = dp_row<NQ, true>(SRC, DST, sel, word, edge, edgePrev);                        \
        edgePrev = edge;                                                                     \
        gr += g;                                                                             \
        if (gr == forkG) {                                                                   \
            unsigned long long acc = 0;                                                      \
            _Pragma("unroll") for (int c = 0; c < CL; ++c) {                                 \
                const int s = x.lig * CL + c;                                                \
                const int j = min(s, x.ndb);                                                 \
                const unsigned long long key = ((unsigned long long)(unsigned)(DST[c] + g * (x.ndb - j)) << 20) | \
                                               (unsigned long long)(0xFFFFF - j);            \
                if (s >= 1) acc = key > acc ? key : acc;                                     \
            }                                                                                \
            atomicMax(&comb64[forkIdx], acc);                                                \
            ++forkIdx;                                                                       \
            forkG = forkIdx < nEff ? forkG + gm : 0x7fffffff;                                \
        }                                                                                    \
    }
    for (int t = 0; t < Tf; t += 2) {
        STRK_RSTEP(Ha, Hb, t)
        STRK_RSTEP(Hb, Ha, t + 1)
    }
#undef STRK_RSTEP
}

// Processes the items [base, base + 64/G) of class list `cls`, one per group of G lanes of this
// wave.  G (16/32/64) and CL (columns per lane) are wave-uniform run-time values: set-up and
// epilogue are one body, only the two hot loops are specialised on CL/4 (six copies each).
// `ap` points at the kernel's KArgs in the kernarg segment: fields are scalar-loaded where they are
// used instead of living in SGPRs across the hot loops.
template <bool REF>
__device__ __forceinline__ void dp_wave(KArgsKernarg ap, int cls, int base, uint8_t* Lw, const uint8_t* s_enc,
                                        const int8_t* s_mat) {
    constexpr int g = kGap;
    const int G = class_G(cls), CL = class_CL(cls), nq = CL / 4;
    const DpLayout lay(G, CL);
    const int lane = threadIdx.x & 63;
    const int lig = lane & (G - 1);                  // lane in group
    const int grp = lane / G;
    const bool first = lig == 0, last = lig == G - 1;
    uint8_t* const Lg = Lw + grp * lay.group_bytes;
    uint2* const tbl = reinterpret_cast<uint2*>(Lg + DpLayout::OFF_TBL);
    int* const comb = reinterpret_cast<int*>(Lg + DpLayout::OFF_COMB);
    int* const lmaxA = reinterpret_cast<int*>(Lg + DpLayout::OFF_LMAX);
    int* const misc = reinterpret_cast<int*>(Lg + DpLayout::OFF_MISC);
    uint8_t* const dbs = Lg + lay.off_db;   // dbs[4 + j] <-> db[j]
    uint8_t* const cp = Lg + lay.off_cp;
    uint8_t* const ct = Lg + lay.off_ct;

    const int end_flags = ap->end_flags;
    const bool cBeg = end_flags & 4, cEnd = end_flags & 8;
    const int list_stride = ap->list_stride;
    const int count = min(ap->counters[kCntClass0 + cls], list_stride);

    const int it = base + grp;
    bool act = it < count;
    int r = 0, k0 = 0, nfl = 1, ntr = 0, nfr = 1, m = 1, lo = 0, n = 0;
    long long soff = 0;
    const uint8_t* motif = ap->motifs;
    if (act) {
        const int32_t* list = ap->cls_list + (size_t)cls * list_stride * 2;
        r = list[2 * it];
        k0 = list[2 * it + 1];
        nfl = ap->nfl[r]; ntr = ap->ntr[r]; nfr = ap->nfr[r];
        soff = ap->seq_off[r];
        const int l = ap->read_locus[r];
        const int mo = ap->motif_off[l];
        motif += mo;
        m = ap->motif_off[l + 1] - mo;
        lo = ap->win_lo[r] + k0;
        n = min(kTableMax, ap->win_n[r] - k0);
    }
    const int ndb = nfl + ntr + nfr;
    const int rowsP = act ? nfl + (lo + n - 1) * m : 0;
    constexpr int ref_mode = REF ? 1 : 0;   // k_dp_ref (reference side) / k_dp_all (reads)
    const int rowsT = (act && !ref_mode) ? nfr : 0;

    // ---- stage the encoded read window and collect its symbol set ------------------------------
    uint8_t* const motifL = Lg + lay.off_b0;   // encoded motif; the area is free until the backward pass ends
    if (first) misc[0] = 0;
    wave_lds_sync();
    {
        unsigned mask = 0;
        const uint8_t* seq = ap->seqs + soff;
        const int total = lay.cap + 8;
        for (int s0 = lig; s0 < total; s0 += 4 * G) {   // four independent loads in flight per lane
            int raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = s0 + u * G - 4;
                raw[u] = (act && j >= 0 && j < ndb) ? (int)seq[j] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * G;
                int sym = 0xff;
                if (raw[u] >= 0) {
                    sym = s_enc[raw[u]];
                    mask |= 1u << sym;
                }
                if (s < total) dbs[s] = (uint8_t)sym;
            }
        }
        if (mask) atomicOr(reinterpret_cast<unsigned*>(&misc[0]), mask);
        for (int k = lig; k < m; k += G) motifL[k] = act ? s_enc[motif[k]] : (uint8_t)kNullSym;
    }
    wave_lds_sync();
    const unsigned symmask = (unsigned)misc[0];
    if (act && __popc(symmask) > 8) {
        // more distinct symbols than one v_perm word can hold: hand the item to the generic kernel
        if (first) {
            int32_t* counters = ap->counters;
            const int idx = atomicAdd(&counters[kCntClass0 + kGenericClass], 1);
            if (idx < list_stride) {
                int32_t* gl = ap->cls_list + (size_t)kGenericClass * list_stride * 2;
                gl[2 * idx] = r;
                gl[2 * idx + 1] = k0;
            } else {
                atomicOr(&counters[kCntError], kErrScratch);
            }
        }
        act = false;
    }
    const int nEff = act ? n : 0;

    // ---- per-row substitution words: byte k = W(row symbol, k-th db symbol class) + 2g ----------
    for (int e = lig; e < 18; e += G) {
        unsigned wlo = 0, whi = 0;
        if (e < kNSym) {
            int k = 0;
            for (int s = 0; s < kNSym; ++s) {
                if (!((symmask >> s) & 1u)) continue;
                if (k < 8) {
                    const unsigned b = (unsigned)(s_mat[e * kNSym + s] + kWBias) & 0xffu;
                    if (k < 4) wlo |= b << (8 * k); else whi |= b << (8 * (k - 4));
                }
                ++k;
            }
        }
        tbl[e] = make_uint2(wlo, whi);
    }
    for (int e = lig; e < kTableMax; e += G) {   // ref mode reuses the two arrays as 32 x u64 keys (0 = empty)
        comb[e] = ref_mode ? 0 : kNegInf;
        lmaxA[e] = ref_mode ? 0 : kNegInf;
    }
    // ---- candidate row symbols: null padding | fl | motif*i_hi | null padding ------------------
    {
        const int lenP = rowsP + 2 * (G - 1) + 4;
        const int gstep = G % m;
        int ph = (lig - (G - 1) - nfl) % m;   // phase of this lane's first row inside the motif
        if (ph < 0) ph += m;
        for (int idx = lig; idx < lenP; idx += G) {
            const int row = idx - (G - 1);  // 0-based row
            int sym = kNullSym;
            if (row >= 0 && row < rowsP) sym = row < nfl ? dbs[4 + row] : motifL[ph];
            cp[idx] = (uint8_t)sym;
            ph += gstep;
            if (ph >= m) ph -= m;
        }
        const int lenT = rowsT + 2 * (G - 1) + 4;
        for (int idx = lig; idx < lenT; idx += G) {
            const int row = idx - (G - 1);  // backward row k' - 1
            int sym = kNullSym;
            if (row >= 0 && row < rowsT) sym = dbs[4 + ndb - 1 - row];
            ct[idx] = (uint8_t)sym;
        }
    }
    wave_lds_sync();
    // ---- db symbols -> v_perm selector bytes (class id = rank of the symbol's bit; pads -> 0x0c = constant 0)
    {
        unsigned* const dbw = reinterpret_cast<unsigned*>(dbs);
        for (int wi = lig; wi < (lay.cap + 8) / 4; wi += G) {
            const unsigned v = dbw[wi];
            unsigned o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned sym = (v >> (8 * b)) & 0xffu;
                o |= (sym < (unsigned)kNSym ? (unsigned)__popc(symmask & ((1u << sym) - 1u)) : 0x0cu) << (8 * b);
            }
            dbw[wi] = o;
        }
    }
    wave_lds_sync();

    PassCtx x;
    x.G = G; x.CL = CL; x.lig = lig; x.first = first; x.last = last; x.act = act; x.ndb = ndb;
    x.dbBeg = end_flags & 1; x.dbEnd = end_flags & 2; x.cBeg = cBeg; x.cEnd = cEnd;
    x.tbl = tbl;
    x.selw = reinterpret_cast<const unsigned*>(dbs) + lig * nq;
    x.b0 = reinterpret_cast<uint2*>(Lg + lay.off_b0) + lig;

    // the two hot loops are specialised on (CL/4, G): 14 instances each, everything else is one body
    int zsave = 0;
    const int fork0 = nfl + lo * m;
#define STRK_PASSES(NQ_, G_)                                             \
    {                                                                    \
        zsave = bwd_pass<NQ_, G_>(x, rowsT, ct);                          \
        if (first) misc[1] = zsave;                                      \
        fwd_pass<NQ_, G_>(x, rowsP, cp, nEff, fork0, m, comb, lmaxA);     \
    }
    if constexpr (REF) {
        fwd_pass_ref<7, 64>(x, rowsP, cp, nEff, fork0, m, reinterpret_cast<unsigned long long*>(comb));
    } else
    switch (cls) {
    case 0: STRK_PASSES(4, 8) break;
    case 1: STRK_PASSES(5, 8) break;
    case 2: STRK_PASSES(6, 8) break;
    case 3: STRK_PASSES(7, 8) break;
    case 4: STRK_PASSES(8, 8) break;
    case 5: STRK_PASSES(9, 8) break;
    case 6: STRK_PASSES(10, 8) break;
    case 7: STRK_PASSES(6, 16) break;
    case 8: STRK_PASSES(7, 16) break;
    case 9: STRK_PASSES(4, 32) break;
    case 10: STRK_PASSES(5, 32) break;
    case 11: STRK_PASSES(6, 32) break;
    case 12: STRK_PASSES(7, 32) break;
    case 13: STRK_PASSES(4, 64) break;
    case 14: STRK_PASSES(5, 64) break;
    case 15: STRK_PASSES(6, 64) break;
    default: STRK_PASSES(7, 64) break;
    }
#undef STRK_PASSES
    wave_lds_sync();
    // ---- assemble S[lo + k] (fields re-read from the kernarg segment: nothing was kept live) -----
    KArgsKernarg ap2 = ap;
    asm volatile("" : "+s"(ap2));
    if (act && ref_mode) {
        const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(comb);
        int32_t* const out = ap2->table + ap2->tab_off[r] + 2 * k0;
        for (int k = lig; k < n; k += G) {
            const unsigned long long key = keys[k];
            const int R = nfl + (lo + k) * m;
            out[2 * k] = (int)(key >> 20) - g * ndb - g * R;          // score
            out[2 * k + 1] = (0xFFFFF - (int)(key & 0xFFFFF)) - 1;    // end_query: last aligned db index
        }
    } else if (act) {
        const int zfree = misc[1] - g * ndb;
        int32_t* const out = ap2->table + ap2->tab_off[r] + k0;
        for (int k = lig; k < n; k += G) {
            const int R = nfl + (lo + k) * m;
            int sc = comb[k] - g * (R + nfr + ndb);
            if (cEnd) sc = max(sc, lmaxA[k] - g * ndb);
            if (cBeg) sc = max(sc, zfree);
            comb[k] = sc;
            out[k] = sc;
        }
    }
    wave_lds_sync();
    // ---- speculative search for start == est_cn (the no-feedback guess), replayed from LDS --------
    int4* const spec = ap2->spec;
    if (spec && !ref_mode && act && first && k0 == 0) {
        SeenMask64 seen;
        const SearchResult res = search_replay(ap2->est_cn[r], ap2->step, ap2->lsr, ap2->max_iters, ap2->tie_last, comb, lo, n, seen);
        spec[r] = make_int4(res.cn, res.score, res.n_explored, (res.miss ? kSpecMiss : 0) | (res.empty ? kSpecEmpty : 0));
    }
    wave_lds_sync();
}
static_assert(kNumClasses == 17 && class_CL(0) == 16 && class_CL(6) == kCLMax && class_CL(7) == 24 && class_CL(9) == 16 &&
                  class_CL(13) == 16 && class_G(16) == 64 && class_CL(16) == 28,
              "dp_wave dispatches the 17 (CL/4, G) classes by index");

template <bool REF>
__device__ __forceinline__ void dp_kernel_body() {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * kWaveLdsBytes + kLdsSlack];
    __shared__ uint8_t s_enc[256];
    __shared__ int8_t s_mat[kNSym * kNSym + 3];
    s_enc[threadIdx.x] = c_enc[threadIdx.x];
    for (int i = threadIdx.x; i < kNSym * kNSym; i += 256) s_mat[i] = c_mat[i / kNSym][i % kNSym];
    __syncthreads();
    uint8_t* const Lw = lds + (threadIdx.x >> 6) * kWaveLdsBytes;
    const KArgsKernarg kernarg = (KArgsKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        __builtin_amdgcn_wave_barrier();   // the wave enters every iteration whole (see k_realign_dp)
        KArgsKernarg ap = kernarg;
        asm volatile("" : "+s"(ap));
        int32_t* const counters = ap->counters;
        const int list_stride = ap->list_stride;
        int c = 0;
        if ((threadIdx.x & 63) == 0) c = atomicAdd(&counters[kCntNextChunk], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        int cls = -1, base = 0, acc = 0;
        for (int k = kNumClasses - 1; k >= 0; --k) {
            const int ngw = 64 / class_G(k);
            const int cnt = min(counters[kCntClass0 + k], list_stride);
            const int nch = (cnt + ngw - 1) / ngw;
            if (c < acc + nch) { cls = k; base = (c - acc) * ngw; break; }
            acc += nch;
        }
        if (cls < 0) break;
        dp_wave<REF>(ap, cls, base, Lw, s_enc, s_mat);
    }
}

// All fast classes in ONE launch: every wave pulls chunks (one item per group) from a device-side
// queue, most expensive classes first.  KArgs must be the kernel's only argument (dp_wave reads it
// through the kernarg segment pointer).
__global__ void __launch_bounds__(256) k_dp_all(KArgs a_by_value) {
    (void)a_by_value;
    dp_kernel_body<false>();
}

// Reference-side scoring (get_ref_repeat_count, once per locus): same set-up, forward pass only,
// (score, end_query) pairs.
__global__ void __launch_bounds__(256) k_dp_ref(KArgs a_by_value) {
    (void)a_by_value;
    dp_kernel_body<true>();
}
static_assert(class_G(kNumClasses - 1) == 64 && class_CL(kNumClasses - 1) == 28, "k_dp_ref / k_dp_long use the widest class");

// ---------------------------------------------------------------------------------------------
// Band kernel (see strk_search.h "Banded scoring with an exactness certificate").  Lanes own
// DIAGONALS instead of columns: lane l of a group keeps the 16 diagonals d = dlo + 16 l .. + 15 of the
// current row, so a group of 8 (16) lanes covers a band of 128 (256) diagonals that follows the
// alignment down the matrix.  Per row and slot k:
//     up   = (r-1, j)   = old[k+1]   (the next lane's old[0] for k = 15: a second DPP, mid-step)
//     left = (r, j-1)   = new[k-1]   (the previous lane's new[15] for k = 0: the systolic skew)
//     diag = (r-1, j-1) = old[k] + w
// and the selector bytes of the lane's 16 columns slide by one column per row (four v_alignbyte plus
// one LDS byte).  Cells outside the band are 0 in G-space (= -inf: every real value is >= 0), cells
// left of column 1 carry the left-boundary value, cells right of the last column replicate it.
// The backward pass is the same function on the reversed right flank and the reversed window.
// ---------------------------------------------------------------------------------------------
struct BandLayout {
    // class-byte array: selb[pad + x] <-> db[x]; `pad` selector-0x0c bytes in front and pad + kBandHiPad
    // behind, sized so that no column the two passes can ask for (virtual rows, rows beyond |db|, the
    // two-step prefetch) falls outside it: the hot loop indexes it without clamping.
    int wd, pad, maxdb, maxcol, off_sel, off_cp, off_ct, off_b0, group_bytes, sel_len;
    static constexpr int OFF_TBL = 0, OFF_COMB = 18 * 8, OFF_MISC = OFF_COMB + kTableMax * 4, OFF_LMAX = OFF_MISC + 16;
    static constexpr int kBandHiPad = kBandRowSlack + 32;
    __host__ __device__ constexpr BandLayout(int c)   // c = band class
        : wd(128 << c), pad((128 << c) + (8 << c) + 8), maxdb(band_max_db(c)), maxcol(band_max_col(c)),
          off_sel(OFF_LMAX + (band_class_fly(c) ? kTableMax * 4 : 0)),
          off_cp(off_sel + ((band_max_db(c) + 2 * ((128 << c) + (8 << c) + 8) + kBandHiPad + 15) & ~15)),
          // forward row symbols: the whole prefix (classes 0, 1) or 256 flank + 256 motif symbols (2, 3)
          off_ct(off_cp + (band_class_fly(c) ? 512 : ((band_max_db(c) + kBandRowSlack + 2 * (8 << c) + 4 + 15) & ~15))),
          off_b0(off_ct + ((kBandMaxFlank + 2 * (8 << c) + 4 + 15) & ~15)),
          group_bytes(off_b0 + ((band_max_col(c) * 2 + 15) & ~15)),
          sel_len((band_max_db(c) + 2 * ((128 << c) + (8 << c) + 8) + kBandHiPad) & ~3) {}
};
__host__ __device__ constexpr int band_wave_lds(int c) { return (64 / (8 << c)) * BandLayout(c).group_bytes; }
__host__ __device__ constexpr int max_band_wave_lds(int c) {
    return c < 0 ? 0 : (band_wave_lds(c) > max_band_wave_lds(c - 1) ? band_wave_lds(c) : max_band_wave_lds(c - 1));
}
constexpr int kBandWaveLds = max_band_wave_lds(kNumBandClasses - 1);
constexpr int kBandNeg16 = -20000;

struct BandCtx {
    int lig;
    bool first, last;
    const uint2* tbl;
    const uint8_t* selb;   // class-byte array: selb[pad + x] <-> db[x], 0x0c elsewhere
    int pad, maxidx, ndb;
    const uint8_t* flL;    // on-the-fly forward rows: 256 left-flank symbols, 256 motif symbols
    const uint8_t* motifL;
    int nfl, m;
};

// One banded pass over `nrows` rows.  BWD = false: forward pass (columns = db, left to right);
// BWD = true: backward pass in reversed coordinates (columns = reversed db).  dlo_ is the first
// diagonal of the band, topFree/leftFree the free-end flags of the top row / left column.
template <int G, bool BWD, bool FLY>
__device__ __forceinline__ void band_pass(const BandCtx& x, const uint8_t* rowsym, int nrows, int dlo_, bool topFree,
                                          bool leftFree, int nEff, int fork0, int m, int cmin, int ncol, int* comb,
                                          short* b0col, int* lmaxA) {
    constexpr int g = kGap;
    const int ncols = x.ndb;
    const int d0 = dlo_ + x.lig * 16;                 // diagonal of this lane's slot 0
    auto g0 = [&](int j) -> int { return topFree ? g * min(max(j, 0), ncols) : 0; };   // row-0 pattern
    auto col_addr = [&](int j) -> int {               // LDS index of the class byte of column j (1-based)
        return BWD ? x.pad + ncols - j : x.pad + j - 1;   // always inside the padded array (BandLayout)
    };
    int Ha[16], Hb[16];
    unsigned sel[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) Ha[k] = g0(-x.lig + d0 + k);
    {
        const int j0 = 1 - x.lig + d0;                // column of slot 0 at the row of step 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (unsigned)x.selb[col_addr(j0 + 4 * q + b)] << (8 * b);
            sel[q] = v;
        }
    }
    const int T = (wave_max_over_groups(nrows > 0 ? nrows + G - 1 : 0) + 1) & ~1;
    const int dhi_ = dlo_ + 16 * G - 1;
    int houtL = Ha[15];
    int gr = -g * x.lig;                               // g * (row finished before step 0)
    int jb = 1 - x.lig + d0;                           // column of slot 0 at the current step's row
    int forkG = (!BWD && nEff > 0) ? g * fork0 : 0x7fffffff;
    if (BWD) forkG = g * nrows;                        // the backward pass has one event: its last row
    int forkIdx = 0;
    const int gm = g * m;
    // wide classes (FLY): running maximum of the last column over the in-band rows.  Right of column |db|
    // the pads replicate G(r, |db|), so from row |db| - dhi on the last lane's slot 15 holds that value.
    int lastmax = kNegInf;
    const int grFirst = g * max(1, ncols - dhi_);
    // row symbols: staged in LDS with G-1 null rows in front (pa[t] is this lane's row at step t), or
    // generated two steps ahead from the left flank and the motif phase (long windows)
    const uint8_t* pa = rowsym + (G - 1) - x.lig;
    int rowi = -x.lig, ph = 0;
    auto next_sym = [&]() -> unsigned {
        unsigned sym = kNullSym;
        if (rowi >= 0) sym = rowi < x.nfl ? x.flL[rowi] : x.motifL[ph];
        if (rowi >= x.nfl) { ++ph; if (ph == x.m) ph = 0; }
        ++rowi;
        return sym;
    };
    uint2 wordNext;
    unsigned symNext;
    if (FLY) { wordNext = x.tbl[next_sym()]; symNext = next_sym(); }
    else { wordNext = x.tbl[pa[0]]; symNext = pa[1]; }
    unsigned nbNext = x.selb[col_addr(jb + 16)];       // class byte entering at the next row
#define STRK_BAND_STEP(SRC, DST, TT)                                                               \
    {                                                                                              \
        const uint2 word = wordNext;                                                               \
        wordNext = x.tbl[symNext];                                                                 \
        symNext = FLY ? next_sym() : (unsigned)pa[(TT) + 2];                                       \
        const unsigned nb = nbNext;                                                                \
        nbNext = x.selb[col_addr(jb + 17)];                                                        \
        /* lane 0 works on row TT+1; left of the band lies the left boundary while j-1 <= 0 */      \
        const int keepL = ((TT) + dlo_ <= 0 && leftFree) ? g * ((TT) + 1) : 0;                      \
        const int leftEdge = from_left<G>(keepL, houtL, x.first);                                  \
        /* the last lane works on row TT-G+2; above row 1 lies the row-0 pattern, else -inf */      \
        const int rl = (TT) - G + 2;                                                               \
        const int keepU = rl <= 1 ? g0(rl + dhi_) : 0;                                             \
        const unsigned w0 = __builtin_amdgcn_perm(word.y, word.x, sel[0]);                         \
        const unsigned w1 = __builtin_amdgcn_perm(word.y, word.x, sel[1]);                         \
        const unsigned w2 = __builtin_amdgcn_perm(word.y, word.x, sel[2]);                         \
        const unsigned w3 = __builtin_amdgcn_perm(word.y, word.x, sel[3]);                         \
        DST[0] = max(max(SRC[1], leftEdge), SRC[0] + (int)(w0 & 0xffu));                           \
        const int upEdge = from_right<G>(keepU, DST[0], x.last);                                   \
        _Pragma("unroll") for (int k = 1; k < 15; ++k) {                                           \
            const unsigned wq = k < 4 ? w0 : (k < 8 ? w1 : (k < 12 ? w2 : w3));                     \
            DST[k] = max(max(SRC[k + 1], DST[k - 1]), SRC[k] + (int)((wq >> (8 * (k % 4))) & 0xffu)); \
        }                                                                                          \
        DST[15] = max(max(upEdge, DST[14]), SRC[15] + (int)(w3 >> 24));                            \
        houtL = DST[15];                                                                           \
        if (FLY && !BWD) lastmax = (gr + g >= grFirst) ? max(lastmax, houtL - (gr + g)) : lastmax;  \
        sel[0] = __builtin_amdgcn_alignbyte(sel[1], sel[0], 1);                                    \
        sel[1] = __builtin_amdgcn_alignbyte(sel[2], sel[1], 1);                                    \
        sel[2] = __builtin_amdgcn_alignbyte(sel[3], sel[2], 1);                                    \
        sel[3] = __builtin_amdgcn_alignbyte(nb, sel[3], 1);                                        \
        gr += g;                                                                                   \
        if (gr == forkG) {                                                                         \
            if (BWD) {                                                                             \
                /* last row: slot k is reversed column jb + k, i.e. db node ndb - (jb + k) */       \
                _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                   \
                    const int jp = jb + k, idx = ncols - jp - cmin;                                \
                    if (jp >= 0 && jp <= ncols && idx >= 0 && idx < ncol) b0col[idx] = (short)DST[k]; \
                }                                                                                  \
                forkG = 0x7fffffff;                                                                \
            } else {                                                                               \
                const short* bc = b0col + (jb - cmin);                                             \
                int acc = kNegInf;                                                                 \
                _Pragma("unroll") for (int k = 0; k < 16; k += 2)                                  \
                    acc = max(max(acc, DST[k] + (int)bc[k]), DST[k + 1] + (int)bc[k + 1]);         \
                atomicMax(&comb[forkIdx], acc);                                                    \
                if (FLY && x.last) lmaxA[forkIdx] = lastmax;                                       \
                ++forkIdx;                                                                         \
                forkG = forkIdx < nEff ? forkG + gm : 0x7fffffff;                                  \
            }                                                                                      \
        }                                                                                          \
        ++jb;                                                                                      \
    }
    for (int t = 0; t < T; t += 2) {
        STRK_BAND_STEP(Ha, Hb, t)
        STRK_BAND_STEP(Hb, Ha, t + 1)
    }
#undef STRK_BAND_STEP
}

// Profiling aid (tools/phase_timing.sh builds a private copy of the library with -DSTRK_PHASE_TIMING): shader-clock
// ticks per phase of band_wave, summed over waves into the spare counter slots 40..47.
#ifdef STRK_PHASE_TIMING
#define STRK_PHASE(i)                                                                                  \
    do {                                                                                               \
        const unsigned long long t_ = __builtin_readcyclecounter();                                    \
        if (lane == 0) atomicAdd(&a.counters[40 + (i)], (int)((t_ - tphase) >> 6));                    \
        tphase = t_;                                                                                   \
    } while (0)
#else
#define STRK_PHASE(i) do { } while (0)
#endif

// Processes 64/G items of band class BC (G = 8 << BC lanes per read), one per group.
template <int BC>
__device__ __forceinline__ void band_wave(const KArgs& a, int base, uint8_t* Lw, const uint8_t* s_enc, const int8_t* s_mat) {
    constexpr int g = kGap, G = 8 << BC;
    constexpr bool FLY = band_class_fly(BC);
    constexpr BandLayout lay(BC);
    const int cls = kBandClass0 + BC;
    const int lane = threadIdx.x & 63;
#ifdef STRK_PHASE_TIMING
    unsigned long long tphase = __builtin_readcyclecounter();
#endif
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const bool first = lig == 0, last = lig == G - 1;
    uint8_t* const Lg = Lw + grp * lay.group_bytes;
    uint2* const tbl = reinterpret_cast<uint2*>(Lg + BandLayout::OFF_TBL);
    int* const comb = reinterpret_cast<int*>(Lg + BandLayout::OFF_COMB);
    int* const misc = reinterpret_cast<int*>(Lg + BandLayout::OFF_MISC);
    int* const lmaxA = reinterpret_cast<int*>(Lg + BandLayout::OFF_LMAX);   // wide classes only
    uint8_t* const selb = Lg + lay.off_sel;
    uint8_t* const cp = Lg + lay.off_cp;       // staged prefix rows, or (FLY) 256 flank + 256 motif symbols
    uint8_t* const ct = Lg + lay.off_ct;
    short* const b0col = reinterpret_cast<short*>(Lg + lay.off_b0);
    uint8_t* const motifL = FLY ? cp + 256 : Lg + lay.off_b0;   // non-FLY: the motif sits in b0col until that is initialised

    const int count = min(a.counters[kCntClass0 + cls], a.list_stride);
    const int32_t* list = a.cls_list + (size_t)cls * a.list_stride * 2;
    const int it = base + grp;
    bool act = it < count;
    int r = 0, nfl = 1, ntr = 0, nfr = 1, m = 1, lo = 0, n = 0;
    long long soff = 0;
    const uint8_t* motif = a.motifs;
    if (act) {
        const int2 item = reinterpret_cast<const int2*>(list)[it];   // (read, locus): two levels of dependent loads, not three
        r = item.x;
        const int l = item.y;
        const int mo0 = a.motif_off[l], mo1 = a.motif_off[l + 1];
        nfl = a.nfl[r]; ntr = a.ntr[r]; nfr = a.nfr[r];
        soff = a.seq_off[r];
        lo = a.win_lo[r];
        n = a.win_n[r];
        motif += mo0;
        m = mo1 - mo0;
    }
    const int ndb = nfl + ntr + nfr;
    const BandGeo geo = band_geometry(nfl, ntr, nfr, m, lo, max(n, 1));
    const int rowsP = act ? nfl + (lo + n - 1) * m : 0;
    const int rowsT = act ? nfr : 0;
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;

    STRK_PHASE(0);
    // ---- stage: class-byte array with pads, symbol set, row words, row symbols ------------------
    if (first) misc[0] = 0;
    wave_lds_sync();
    {
        // window bytes -> symbols, a dword per lane and eight dwords in flight (the loop is bound by load latency);
        // slots outside the window get 0xff
        unsigned mask = 0;
        const uint8_t* seq = a.seqs + soff;
        constexpr int ND = lay.sel_len / 4;
        unsigned* const selw = reinterpret_cast<unsigned*>(selb);
        for (int d0 = lig; d0 < ND; d0 += 8 * G) {
            unsigned w[8], ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * G, j0 = 4 * d - lay.pad;
                unsigned v = 0, o = 0;
                if (act && d < ND && j0 + 3 >= 0 && j0 < ndb) {
                    if (j0 >= 0 && j0 + 3 < ndb) {
                        __builtin_memcpy(&v, seq + j0, 4);   // unaligned dword load
                        o = 0xfu;
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if (j0 + b >= 0 && j0 + b < ndb) { v |= (unsigned)seq[j0 + b] << (8 * b); o |= 1u << b; }
                    }
                }
                w[u] = v;
                ok[u] = o;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * G;
                unsigned out = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    unsigned sym = 0xffu;
                    if ((ok[u] >> b) & 1u) { sym = s_enc[(w[u] >> (8 * b)) & 0xffu]; mask |= 1u << sym; }
                    out |= sym << (8 * b);
                }
                if (d < ND) selw[d] = out;
            }
        }
        if (mask) atomicOr(reinterpret_cast<unsigned*>(&misc[0]), mask);
        for (int k = lig; k < m; k += G) motifL[k] = act ? s_enc[motif[k]] : (uint8_t)kNullSym;
    }
    wave_lds_sync();
    STRK_PHASE(1);
    const unsigned symmask = (unsigned)misc[0];
    bool fallback = act && __popc(symmask) > 8;   // more symbol classes than a v_perm word holds: exact path decides
    for (int e = lig; e < 18; e += G) {
        unsigned wlo = 0, whi = 0;
        if (e < kNSym) {
            int k = 0;
            for (int s = 0; s < kNSym; ++s) {
                if (!((symmask >> s) & 1u)) continue;
                if (k < 8) {
                    const unsigned b = (unsigned)(s_mat[e * kNSym + s] + kWBias) & 0xffu;
                    if (k < 4) wlo |= b << (8 * k); else whi |= b << (8 * (k - 4));
                }
                ++k;
            }
        }
        tbl[e] = make_uint2(wlo, whi);
    }
    for (int e = lig; e < kTableMax; e += G) { comb[e] = kNegInf; if (FLY) lmaxA[e] = kNegInf; }
    {
        if (FLY) {
            for (int k = lig; k < 256; k += G) cp[k] = (uint8_t)(k < nfl ? selb[lay.pad + k] : kNullSym);
        } else {
            const int lenP = rowsP + 2 * (G - 1) + 4;
            const int gstep = G % m;
            int ph = (lig - (G - 1) - nfl) % m;
            if (ph < 0) ph += m;
            for (int idx = lig; idx < lenP; idx += G) {
                const int row = idx - (G - 1);
                int sym = kNullSym;
                if (row >= 0 && row < rowsP) sym = row < nfl ? selb[lay.pad + row] : motifL[ph];
                cp[idx] = (uint8_t)sym;
                ph += gstep;
                if (ph >= m) ph -= m;
            }
        }
        const int lenT = rowsT + 2 * (G - 1) + 4;
        for (int idx = lig; idx < lenT; idx += G) {
            const int row = idx - (G - 1);
            int sym = kNullSym;
            if (row >= 0 && row < rowsT) sym = selb[lay.pad + ndb - 1 - row];
            ct[idx] = (uint8_t)sym;
        }
    }
    wave_lds_sync();
    for (int wi = lig; wi < lay.sel_len / 4; wi += G) {   // symbols -> v_perm selector bytes
        unsigned* const w = reinterpret_cast<unsigned*>(selb) + wi;
        const unsigned v = *w;
        unsigned o = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned sym = (v >> (8 * b)) & 0xffu;
            o |= (sym < (unsigned)kNSym ? (unsigned)__popc(symmask & ((1u << sym) - 1u)) : 0x0cu) << (8 * b);
        }
        *w = o;
    }
    for (int k = lig; k < lay.maxcol; k += G) b0col[k] = (short)kBandNeg16;
    wave_lds_sync();
    STRK_PHASE(2);

    BandCtx x;
    x.lig = lig; x.first = first; x.last = last; x.tbl = tbl; x.selb = selb;
    x.pad = lay.pad; x.maxidx = lay.sel_len - 1; x.ndb = ndb;
    x.flL = cp; x.motifL = motifL; x.nfl = nfl; x.m = m;
    const bool run = act && !fallback && geo.ok;
    const int nEff = run ? n : 0;
    // backward pass (reversed right flank x reversed window), then forward pass with the fork rows
    band_pass<G, true, false>(x, ct, run ? rowsT : 0, geo.bdlo, dbEnd, cEnd, 0, 0, 1, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    STRK_PHASE(3);
    band_pass<G, false, FLY>(x, cp, run ? rowsP : 0, geo.dlo, dbBeg, cBeg, nEff, nfl + lo * m, m, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    STRK_PHASE(4);
    if (run) {
        for (int k = lig; k < n; k += G) {
            const int R = nfl + (lo + k) * m;
            int sc = max(comb[k], -(1 << 28)) - g * (R + nfr + ndb);
            if (FLY && cEnd) sc = max(sc, max(lmaxA[k], -(1 << 28)) - g * ndb);   // ends in the last column, in band
            comb[k] = sc;
            a.table[a.tab_off[r] + k] = sc;
        }
    }
    // the bounds of all candidates at once, one per lane (the search below runs on one lane and would otherwise
    // evaluate band_ub once per window entry); b0col is free after the forward pass
    int* const ubA = reinterpret_cast<int*>(b0col);
    if (run)
        for (int k = lig; k < n; k += G) ubA[k] = band_ub(geo, nfl, ntr, nfr, m, lo + k, a.end_flags);
    wave_lds_sync();
    if (act && first) {
        bool certified = false;
        if (run) {
            SeenMask64 seen;
            auto ub = [&](int k) { return ubA[k]; };
            const CertResult cr = search_replay_cert(a.est_cn[r], a.step, a.lsr, a.max_iters, a.tie_last, comb, lo, n, seen, ub);
            if (!cr.uncertain) {
                certified = true;
                a.spec[r] = make_int4(cr.res.cn, cr.res.score, cr.res.n_explored,
                                      (cr.res.miss ? kSpecMiss : 0) | (cr.res.empty ? kSpecEmpty : 0));
            }
        }
        if (!certified) {   // hand the read to the exact kernels (they run after this one)
            const int c = classify(nfl, ntr, nfr, m, lo, n, 0, 0);
            const int idx = atomicAdd(&a.counters[kCntClass0 + c], 1);
            if (idx < a.list_stride) {
                int32_t* gl = a.cls_list + (size_t)c * a.list_stride * 2;
                gl[2 * idx] = r;
                gl[2 * idx + 1] = 0;
            } else {
                atomicOr(&a.counters[kCntError], kErrScratch);
            }
            a.exact[r] = 1;
            atomicAdd(&a.counters[kCntBandFallback], 1);
            if (c != kGenericClass)
                atomicAdd(a.cells, (unsigned long long)ndb * ((unsigned long long)nfl + (unsigned long long)(lo + n - 1) * m + nfr));
        }
    }
    wave_lds_sync();
    STRK_PHASE(5);
}

// Two kernels so that the common short classes (0, 1) are not register-allocated together with the
// long-window classes (2, 3).  SET 0: classes 1 then 0;  SET 1: classes 3 then 2.  Each wave pulls chunks
// from the set's queue until it is empty.
template <int SET>
__device__ __forceinline__ void band_kernel_body(const KArgs& a) {
    constexpr int CA = SET ? 3 : 1, CB = SET ? 2 : 0;   // wider class first
    const int nA = min(a.counters[kCntClass0 + kBandClass0 + CA], a.list_stride);
    const int nB = min(a.counters[kCntClass0 + kBandClass0 + CB], a.list_stride);
    if (nA + nB <= 0) return;
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * kBandWaveLds + kLdsSlack];
    __shared__ uint8_t s_enc[256];
    __shared__ int8_t s_mat[kNSym * kNSym + 3];
    s_enc[threadIdx.x] = c_enc[threadIdx.x];
    for (int i = threadIdx.x; i < kNSym * kNSym; i += 256) s_mat[i] = c_mat[i / kNSym][i % kNSym];
    __syncthreads();
    uint8_t* const Lw = lds + (threadIdx.x >> 6) * kBandWaveLds;
    constexpr int perA = 64 / (8 << CA), perB = 64 / (8 << CB);   // reads per wave
    const int chA = (nA + perA - 1) / perA, chB = (nB + perB - 1) / perB;
    for (;;) {
        __builtin_amdgcn_wave_barrier();   // the wave enters every iteration whole (see k_realign_dp)
        int c = 0;
        if ((threadIdx.x & 63) == 0) c = atomicAdd(&a.counters[kCntNextBand + SET], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c < chA) band_wave<CA>(a, c * perA, Lw, s_enc, s_mat);
        else if (c - chA < chB) band_wave<CB>(a, (c - chA) * perB, Lw, s_enc, s_mat);
        else break;
    }
}
__global__ void __launch_bounds__(256) k_dp_band(KArgs a) { band_kernel_body<0>(a); }
__global__ void __launch_bounds__(256) k_dp_band_wide(KArgs a) { band_kernel_body<1>(a); }

// ---------------------------------------------------------------------------------------------
// Long-read kernel: the same shared-prefix systolic DP for windows wider than the largest fast class
// (BASELINE config 5: up to ~2 000 copies, |db| ~ 12 kb).  One read per wave (G = 64, CL = 28); the
// db columns are cut into tiles of kLongTile slots that are processed one after the other, the
// column between two tiles (one value per row) travels through a global scratch array that the wave
// reads/writes 64 rows at a time (coalesced) and feeds to the edge lane with v_readlane.  The
// backward result Gb(|fr|, .) of all tiles lives in global scratch (int32).  Row symbols are
// computed on the fly (fl from LDS, then the motif with a running phase) instead of being staged.
// ---------------------------------------------------------------------------------------------
struct LongLayout {   // per-wave LDS, bytes
    static constexpr int OFF_TBL = 0, OFF_COMB = 18 * 8, OFF_LMAX = OFF_COMB + kTableMax * 4, OFF_MISC = OFF_LMAX + kTableMax * 4;
    static constexpr int OFF_DB = OFF_MISC + 16;                                   // kLongTile + 8 selector bytes
    static constexpr int OFF_FL = OFF_DB + ((kLongTile + 8 + 15) & ~15);           // 256 left-flank row symbols
    static constexpr int OFF_MOTIF = OFF_FL + 256;                                 // 256 motif symbols
    static constexpr int OFF_CT = OFF_MOTIF + 256;                                 // reversed fr rows + null padding
    static constexpr int OFF_COLB = OFF_CT + ((kLongFlankMax + 2 * 64 + 4 + 15) & ~15);  // 2 x 324 ints
    static constexpr int COLB_INTS = 324;
    static constexpr int BYTES = OFF_COLB + 2 * COLB_INTS * 4;
};
static_assert(LongLayout::BYTES <= kWaveLdsBytes, "k_dp_long fits the per-wave LDS budget of k_dp_all");

__global__ void __launch_bounds__(256) k_dp_long(KArgs a) {
    constexpr int g = kGap, G = 64, NQ = 7, CL = 28, TW = kLongTile;
    if (a.counters[kCntClass0 + kLongClass] <= 0) return;   // nothing long in this batch (the usual case)
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * kWaveLdsBytes + kLdsSlack];
    __shared__ uint8_t s_enc[256];
    __shared__ int8_t s_mat[kNSym * kNSym + 3];
    s_enc[threadIdx.x] = c_enc[threadIdx.x];
    for (int i = threadIdx.x; i < kNSym * kNSym; i += 256) s_mat[i] = c_mat[i / kNSym][i % kNSym];
    __syncthreads();
    uint8_t* const Lw = lds + (threadIdx.x >> 6) * kWaveLdsBytes;
    const int lane = threadIdx.x & 63;
    const bool first = lane == 0, last = lane == 63;
    uint2* const tbl = reinterpret_cast<uint2*>(Lw + LongLayout::OFF_TBL);
    int* const comb = reinterpret_cast<int*>(Lw + LongLayout::OFF_COMB);
    int* const lmaxA = reinterpret_cast<int*>(Lw + LongLayout::OFF_LMAX);
    int* const misc = reinterpret_cast<int*>(Lw + LongLayout::OFF_MISC);
    uint8_t* const dbs = Lw + LongLayout::OFF_DB;
    uint8_t* const flL = Lw + LongLayout::OFF_FL;
    uint8_t* const motifL = Lw + LongLayout::OFF_MOTIF;
    uint8_t* const ct = Lw + LongLayout::OFF_CT;
    int* const colB = reinterpret_cast<int*>(Lw + LongLayout::OFF_COLB);
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;
    const int count = min(a.counters[kCntClass0 + kLongClass], a.list_stride);
    const int32_t* list = a.cls_list + (size_t)kLongClass * a.list_stride * 2;

    for (;;) {
        __builtin_amdgcn_wave_barrier();   // the wave enters every iteration whole (see k_realign_dp)
        int it = 0;
        if (first) it = atomicAdd(&a.counters[kCntNextLong], 1);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= count) break;
        const int r = list[2 * it], k0 = list[2 * it + 1];
        const int nfl = a.nfl[r], ntr = a.ntr[r], nfr = a.nfr[r];
        const int ndb = nfl + ntr + nfr;
        const uint8_t* seq = a.seqs + a.seq_off[r];
        const int l = a.read_locus[r];
        const uint8_t* motif = a.motifs + a.motif_off[l];
        const int m = a.motif_off[l + 1] - a.motif_off[l];
        const int lo = a.win_lo[r] + k0;
        const int n = min(kTableMax, a.win_n[r] - k0);
        const int rowsP = nfl + (lo + n - 1) * m;
        const int NT = (ndb + 1 + TW - 1) / TW;
        const int stepsF = (rowsP + G - 1 + 63) & ~63;              // forward steps per tile, whole 64-blocks
        const int colLen = stepsF + 64;                             // ints per boundary-column buffer
        // ---- global scratch (this wave's own slot): backward row of all tiles + two boundary columns
        const long long need = (long long)NT * TW + 2ll * colLen;
        if (need > a.long_slot) {
            if (first) atomicOr(&a.counters[kCntError], kErrScratch);
            continue;
        }
        const long long at = (long long)(blockIdx.x * 4 + (threadIdx.x >> 6)) * a.long_slot;
        int32_t* const b0g = a.scratch + at;
        int32_t* colF[2] = {b0g + (size_t)NT * TW, b0g + (size_t)NT * TW + colLen};

        // ---- symbol set of the whole window, row-word table, flank/motif row symbols -------------
        if (first) misc[0] = 0;
        wave_lds_sync();
        {
            unsigned mask = 0;
            for (int j = lane; j < ndb; j += 64) mask |= 1u << s_enc[seq[j]];
            if (mask) atomicOr(reinterpret_cast<unsigned*>(&misc[0]), mask);
            for (int k = lane; k < 256; k += 64) {
                flL[k] = (uint8_t)(k < nfl ? s_enc[seq[k]] : kNullSym);
                motifL[k] = (uint8_t)(k < m ? s_enc[motif[k]] : kNullSym);
            }
            const int lenT = nfr + 2 * (G - 1) + 4;
            for (int idx = lane; idx < lenT; idx += 64) {
                const int row = idx - (G - 1);
                ct[idx] = (uint8_t)((row >= 0 && row < nfr) ? s_enc[seq[ndb - 1 - row]] : kNullSym);
            }
            for (int e = lane; e < kTableMax; e += 64) { comb[e] = kNegInf; lmaxA[e] = kNegInf; }
        }
        wave_lds_sync();
        const unsigned symmask = (unsigned)misc[0];
        if (__popc(symmask) > 8) {   // hand the item to the generic kernel
            if (first) {
                const int idx = atomicAdd(&a.counters[kCntClass0 + kGenericClass], 1);
                if (idx < a.list_stride) {
                    int32_t* gl = a.cls_list + (size_t)kGenericClass * a.list_stride * 2;
                    gl[2 * idx] = r;
                    gl[2 * idx + 1] = k0;
                } else {
                    atomicOr(&a.counters[kCntError], kErrScratch);
                }
            }
            continue;
        }
        for (int e = lane; e < 18; e += 64) {
            unsigned wlo = 0, whi = 0;
            if (e < kNSym) {
                int k = 0;
                for (int sy = 0; sy < kNSym; ++sy) {
                    if (!((symmask >> sy) & 1u)) continue;
                    const unsigned b = (unsigned)(s_mat[e * kNSym + sy] + kWBias) & 0xffu;
                    if (k < 4) wlo |= b << (8 * k); else whi |= b << (8 * (k - 4));
                    ++k;
                }
            }
            tbl[e] = make_uint2(wlo, whi);
        }
        // selector bytes of one tile: dbs[4 + x] <-> db[tile*TW + x]  (0x0c outside the window)
        auto stage_tile = [&](int tile) {
            wave_lds_sync();
            for (int sidx = lane; sidx < TW + 8; sidx += 64) {
                const long long j = (long long)tile * TW + sidx - 4;
                unsigned v = 0x0c;
                if (j >= 0 && j < ndb) {
                    const unsigned sym = s_enc[seq[j]];
                    v = (unsigned)__popc(symmask & ((1u << sym) - 1u));
                }
                dbs[sidx] = (uint8_t)v;
            }
            wave_lds_sync();
        };
        const unsigned* const selw = reinterpret_cast<const unsigned*>(dbs) + lane * NQ;
        int Ha[CL], Hb[CL];
        unsigned sel[NQ];

        // =============================== backward pass, tiles right to left =======================
        int zsave = 0;
        {
            const int Tb = (nfr + G - 1 + 1) & ~1;
            const int bstep = cEnd ? g : 0;
            for (int tile = NT - 1; tile >= 0; --tile) {
                stage_tile(tile);
                int* const cIn = colB + ((tile + 1) & 1) * LongLayout::COLB_INTS;   // written by tile + 1
                int* const cOut = colB + (tile & 1) * LongLayout::COLB_INTS;
#pragma unroll
                for (int q = 0; q < NQ; ++q) sel[q] = selw[q + 1];
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    const long long sl = (long long)tile * TW + lane * CL + c;
                    int v = 0;
                    if (sl < ndb && dbEnd) v = g * (int)(ndb - sl) - (sl == 0 ? g : 0);
                    Ha[c] = v;
                }
                const bool inner = tile != NT - 1;   // right input comes from the tile to the right
                int hout = Ha[0];
                if (first) cOut[0] = hout;           // row 0 of this tile's left-most slot
                int edgePrev = from_right<G>(inner ? cIn[0] : 0, hout, last);
                int gk = g * (lane - (G - 1));
                int zmax = Ha[0];
                const uint8_t* pa = ct + lane;
                uint2 wordNext = tbl[pa[0]];
                unsigned symNext = pa[1];
#define STRK_LB(SRC, DST, T)                                                                        \
                {                                                                                   \
                    const uint2 word = wordNext;                                                    \
                    wordNext = tbl[symNext];                                                        \
                    symNext = pa[(T) + 2];                                                          \
                    const int keep = inner ? cIn[min((T) + 1, LongLayout::COLB_INTS - 1)] : bstep * ((T) + 1); \
                    const int edge = from_right<G>(keep, hout, last);                               \
                    hout = dp_row<NQ, false>(SRC, DST, sel, word, edge, edgePrev);                  \
                    edgePrev = edge;                                                                \
                    gk += g;                                                                        \
                    if (first && (T) - (G - 1) + 1 >= 1 && (T) - (G - 1) + 1 <= nfr) cOut[(T) - (G - 1) + 1] = hout; \
                    if (gk == g * nfr) {                                                            \
                        _Pragma("unroll") for (int c = 0; c < CL; ++c)                              \
                            b0g[(size_t)tile * TW + lane * CL + c] = DST[c];                        \
                        zsave = zmax;                                                               \
                    }                                                                               \
                    zmax = max(zmax, hout - gk);                                                    \
                    if ((T) == G - 2) zmax = hout;                                                  \
                }
                for (int t = 0; t < Tb; t += 2) {
                    STRK_LB(Ha, Hb, t)
                    STRK_LB(Hb, Ha, t + 1)
                }
#undef STRK_LB
                wave_lds_sync();
            }
        }
        if (first) misc[1] = zsave;   // only tile 0's lane 0 holds node 0: it ran last
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // b0g is re-read below through L2
        wave_lds_sync();

        // =============================== forward pass, tiles left to right =========================
        {
            const int bstep = cBeg ? g : 0;
            const int gm = g * m;
            for (int tile = 0; tile < NT; ++tile) {
                stage_tile(tile);
                const int32_t* const cIn = colF[(tile + 1) & 1];   // written by tile - 1
                int32_t* const cOut = colF[tile & 1];
                const bool inner = tile != 0;
                const bool lastTile = tile == NT - 1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) sel[q] = __builtin_amdgcn_alignbyte(selw[q + 1], selw[q], 3);
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    const long long sl = (long long)tile * TW + lane * CL + c;
                    Ha[c] = dbBeg ? g * (int)min(sl, (long long)ndb) : 0;
                }
                int hout = Ha[CL - 1];
                if (last) cOut[0] = hout;                      // row 0 of this tile's right-most slot
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                int edgePrev = from_left<G>(inner ? cIn[0] : 0, hout, first);
                int gr = -g * lane;
                int lastmax = kNegInf;
                int forkG = g * (nfl + lo * m);
                int forkIdx = 0;
                // row symbols are generated two steps ahead (fl from LDS, then the motif with a running
                // phase), the row word one step ahead, so both LDS latencies hide behind a DP row
                int rowi = -lane;                              // 0-based row of the symbol generated next
                int ph = 0;                                    // (rowi - nfl) mod m once rowi >= nfl
                auto next_sym = [&]() -> int {
                    int sym = kNullSym;
                    if (rowi >= 0) sym = rowi < nfl ? flL[rowi] : motifL[ph];
                    if (rowi >= nfl) { ++ph; if (ph == m) ph = 0; }
                    ++rowi;
                    return sym;
                };
                uint2 wordNext = tbl[next_sym()];
                int symNext = next_sym();
                for (int t0 = 0; t0 < stepsF; t0 += 64) {
                    int edgeIn = 0;
                    if (inner) edgeIn = cIn[t0 + 1 + lane];    // rows t0+1 .. t0+64 of the left neighbour slot
                    int outAcc = 0;
#define STRK_LF(SRC, DST, U)                                                                        \
                    {                                                                               \
                        const int t = t0 + (U);                                                     \
                        const uint2 word = wordNext;                                                \
                        wordNext = tbl[symNext];                                                    \
                        symNext = next_sym();                                                       \
                        const int keep = inner ? __builtin_amdgcn_readlane(edgeIn, (U)) : bstep * (t + 1); \
                        const int edge = from_left<G>(keep, hout, first);                           \
                        hout = dp_row<NQ, true>(SRC, DST, sel, word, edge, edgePrev);               \
                        edgePrev = edge;                                                            \
                        gr += g;                                                                    \
                        lastmax = max(lastmax, hout - gr);                                          \
                        if (t == G - 2) lastmax = kNegInf;                                          \
                        {   /* right-most slot of the tile, row t - 62, for the next tile */        \
                            const int v = __builtin_amdgcn_readlane(hout, 63);                      \
                            if (lane == (U)) outAcc = v;                                            \
                        }                                                                           \
                        if (gr == forkG) {                                                          \
                            int acc = kNegInf;                                                      \
                            const int32_t* bp = b0g + (size_t)tile * TW + lane * CL;                \
                            _Pragma("unroll") for (int c = 0; c < CL; ++c) acc = max(acc, DST[c] + bp[c]); \
                            atomicMax(&comb[forkIdx], acc);                                         \
                            if (last && lastTile) lmaxA[forkIdx] = lastmax;                         \
                            ++forkIdx;                                                              \
                            forkG = forkIdx < n ? forkG + gm : 0x7fffffff;                          \
                        }                                                                           \
                    }
                    for (int u = 0; u < 64; u += 2) {
                        STRK_LF(Ha, Hb, u)
                        STRK_LF(Hb, Ha, u + 1)
                    }
#undef STRK_LF
                    // lane u holds the value produced at step t0 + u = row t0 + u - 62 of the right-most slot
                    const int row = t0 + lane - (G - 2);
                    if (row >= 1 && row < colLen) cOut[row] = outAcc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                wave_lds_sync();
            }
        }
        wave_lds_sync();
        // ---- assemble S[lo + k], speculative search -----------------------------------------------
        {
            const int zfree = misc[1] - g * ndb;
            int32_t* const out = a.table + a.tab_off[r] + k0;
            for (int k = lane; k < n; k += 64) {
                const int R = nfl + (lo + k) * m;
                int sc = comb[k] - g * (R + nfr + ndb);
                if (cEnd) sc = max(sc, lmaxA[k] - g * ndb);
                if (cBeg) sc = max(sc, zfree);
                comb[k] = sc;
                out[k] = sc;
            }
        }
        wave_lds_sync();
        if (a.spec && first && k0 == 0) {
            SeenMask64 seen;
            const SearchResult res = search_replay(a.est_cn[r], a.step, a.lsr, a.max_iters, a.tie_last, comb, lo, n, seen);
            a.spec[r] = make_int4(res.cn, res.score, res.n_explored, (res.miss ? kSpecMiss : 0) | (res.empty ? kSpecEmpty : 0));
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: one thread per (item, candidate); plain row-by-row DP with the H row in global
// scratch.  Takes every shape the fast classes do not (empty flanks, > 8 distinct symbols in the
// read window, windows longer than the largest class).  Correctness path, not a fast path.
// ---------------------------------------------------------------------------------------------
__global__ void k_dp_generic(KArgs a) {
    const int count = min(a.counters[kCntClass0 + kGenericClass], a.list_stride);
    if (count <= 0) return;
    const int32_t* list = a.cls_list + (size_t)kGenericClass * a.list_stride * 2;
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;
    constexpr int g = kGap;
    const long long total = (long long)count * kTableMax;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const int it = (int)(w / kTableMax), k = (int)(w % kTableMax);
        const int r = list[2 * it], k0 = list[2 * it + 1];
        const int n = min(kTableMax, a.win_n[r] - k0);
        if (k >= n) continue;
        const int i = a.win_lo[r] + k0 + k;
        const int nfl = a.nfl[r], ntr = a.ntr[r], nfr = a.nfr[r];
        const int l = a.read_locus[r];
        const uint8_t* motif = a.motifs + a.motif_off[l];
        const int m = a.motif_off[l + 1] - a.motif_off[l];
        const uint8_t* db = a.seqs + a.seq_off[r];
        const int ndb = nfl + ntr + nfr;
        const int ncfr = a.ref_mode ? 0 : nfr;   // the reference-side candidate has no right flank
        const long long ncand = (long long)nfl + (long long)i * m + ncfr;
        int32_t* out = a.ref_mode ? a.table + a.tab_off[r] + 2 * (k0 + k) : a.table + a.tab_off[r] + k0 + k;
        if (ndb <= 0 || ncand <= 0) {
            out[0] = 0;
            if (a.ref_mode) out[1] = -1;
            continue;
        }
        const unsigned long long need = (unsigned long long)ndb + 1;
        const unsigned long long at = (unsigned long long)a.long_slot * a.long_waves + atomicAdd(a.scratch_used, need);
        if (at + need > (unsigned long long)a.scratch_cap) {
            atomicOr(&a.counters[kCntError], kErrScratch);
            out[0] = 0;
            if (a.ref_mode) out[1] = -1;
            continue;
        }
        int32_t* Hrow = a.scratch + at;
        Hrow[0] = 0;
        for (int j = 1; j <= ndb; ++j) Hrow[j] = dbBeg ? 0 : -g * j;
        int lastcol = kNegInf;
        for (long long rr = 1; rr <= ncand; ++rr) {
            const long long p = rr - 1;
            const uint8_t ch = p < nfl ? db[p] : (p < nfl + (long long)i * m ? motif[(p - nfl) % m] : db[nfl + ntr + (p - nfl - (long long)i * m)]);
            const int8_t* wrow = c_mat[c_enc[ch]];
            int diag = Hrow[0];
            int left = cBeg ? 0 : (int)(-g * rr);
            Hrow[0] = left;
            for (int j = 1; j <= ndb; ++j) {
                const int up = Hrow[j];
                int h = diag + wrow[c_enc[db[j - 1]]];
                h = max(h, max(up, left) - g);
                diag = up;
                left = h;
                Hrow[j] = h;
            }
            lastcol = max(lastcol, left);
        }
        int best = Hrow[ndb], bestj = ndb;
        if (cEnd) best = max(best, lastcol);
        if (dbEnd)
            for (int j = 1; j <= ndb; ++j)
                if (Hrow[j] > best || (Hrow[j] == best && j < bestj && !cEnd)) { best = Hrow[j]; bestj = j; }
        out[0] = best;
        if (a.ref_mode) out[1] = bestj - 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Search replay: one lane per locus walks its reads in caller order (call_locus.py:1082) with the
// start-count feedback (call_locus.py:1129-1136,1161) and replays the hill climb on the table.
// ---------------------------------------------------------------------------------------------
struct ReplayArgs {
    int32_t max_iters, lsr, step, tie_last, feedback;
    int32_t* out_cn;
    int32_t* out_score;
    int32_t* out_n;
    int32_t* out_start;
    // per-locus resume state
    int32_t* next_read;   // [n_loci] first read not yet finished (== read_off[l+1] when done)
    double* frac;         // [n_loci]
    int32_t* need_lo;     // [n_loci] window wanted by the read that missed
    int32_t* need_hi;
};

// One wave per locus: lane i holds the inputs of the locus's i-th read (coalesced loads), the
// in-order chain over the reads is wave-uniform ALU work on values fetched with v_readlane.
__global__ void __launch_bounds__(64) k_replay(KArgs a, ReplayArgs p) {
    const int l = blockIdx.x;
    const int lane = threadIdx.x;
    const int r_end = a.read_off[l + 1];
    double frac = 0.0;
    int r_next = a.read_off[l];   // first read not finished yet
    bool missed = false;
    for (int base = r_next; base < r_end && !missed; base += 64) {
        const int cnt = min(64, r_end - base);
        const int rl = base + lane;
        const int my_est = lane < cnt ? a.est_cn[rl] : 0;
        int4 my_spec = make_int4(0, 0, 0, kSpecMiss);
        if (a.spec && lane < cnt) my_spec = a.spec[a.rep[rl]];
        int o_cn = 0, o_score = 0, o_n = 0, o_start = 0;
        int done = 0;
        for (int i = 0; i < cnt; ++i) {
            const int est = __builtin_amdgcn_readlane(my_est, i);
            int start = est;
            double frac_try = frac;
            if (p.feedback) start = feedback_start(est, &frac_try);
            SearchResult res;
            const int spec_flags = __builtin_amdgcn_readlane(my_spec.w, i);
            if (start == est && !(spec_flags & kSpecMiss)) {
                // the DP kernel already replayed the search for the no-feedback guess
                res.cn = __builtin_amdgcn_readlane(my_spec.x, i);
                res.score = __builtin_amdgcn_readlane(my_spec.y, i);
                res.n_explored = __builtin_amdgcn_readlane(my_spec.z, i);
                res.miss = 0;
                res.empty = (spec_flags & kSpecEmpty) ? 1 : 0;
            } else {
                const int r = base + i;
                const int rp = a.rep[r];
                SeenMask64 seen;
                if (!a.band_mode || a.exact[rp]) {
                    res = search_replay(start, p.step, p.lsr, p.max_iters, p.tie_last, a.table + a.tab_off[r], a.win_lo[r],
                                        min(a.win_n[r], 64), seen);
                } else {
                    // banded table: lower bounds + certificate; an ambiguous comparison asks for exact scores
                    const int nfl = a.nfl[rp], ntr = a.ntr[rp], nfr = a.nfr[rp];
                    const int m = a.motif_off[l + 1] - a.motif_off[l];
                    const int wlo = a.win_lo[r], wn = min(a.win_n[r], 64);
                    const BandGeo geo = band_geometry(nfl, ntr, nfr, m, wlo, wn);
                    const int flags = a.end_flags;
                    auto ub = [&](int k) { return band_ub(geo, nfl, ntr, nfr, m, wlo + k, flags); };
                    const CertResult cr = search_replay_cert(start, p.step, p.lsr, p.max_iters, p.tie_last,
                                                             a.table + a.tab_off[r], wlo, wn, seen, ub);
                    res = cr.res;
                    if (cr.uncertain) { res.miss = 1; res.need_lo = wlo; res.need_hi = wlo + wn - 1; }
                }
            }
            if (res.miss) {
                if (lane == 0) {
                    p.need_lo[l] = res.need_lo;
                    p.need_hi[l] = res.need_hi;
                    atomicAdd(&a.counters[kCntMiss], 1);
                }
                missed = true;
                break;
            }
            frac = frac_try;
            if (res.empty) {
                if (lane == 0) atomicOr(&a.counters[kCntError], kErrEmpty);  // the reference would raise here
                res.cn = 0;
                res.score = 0;
            }
            if (lane == i) { o_cn = res.cn; o_score = res.score; o_n = res.n_explored; o_start = start; }
            if (p.feedback && !res.empty && res.cn != start) feedback_update(&frac, res.cn, start);  // += 0 otherwise
            done = i + 1;
        }
        if (lane < done) {
            p.out_cn[rl] = o_cn;
            p.out_score[rl] = o_score;
            p.out_n[rl] = o_n;
            p.out_start[rl] = o_start;
        }
        r_next = base + done;
    }
    if (lane == 0) {
        p.next_read[l] = r_next;
        p.frac[l] = frac;
    }
}

}  // namespace strk
