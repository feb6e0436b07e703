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
//   * A read is owned by a group of G lanes (16 or 64); each lane keeps CL consecutive db columns
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
constexpr int kRowSlack = 512;      // a class of capacity CAP accepts up to CAP + kRowSlack prefix rows

// Fast-kernel classes: (G lanes per read, CL columns per lane); capacity = G*CL slots >= |db| + 1.
constexpr int kNumClasses = 13;
constexpr int kGenericClass = kNumClasses;  // index of the generic list
__host__ __device__ constexpr int class_G(int c) { return c < 6 ? 16 : 64; }
__host__ __device__ constexpr int class_CL(int c) { return c < 6 ? 8 + 4 * c : 8 + 4 * (c - 6); }
__host__ __device__ constexpr int class_cap(int c) { return class_G(c) * class_CL(c); }

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
    int32_t* cls_list;    // [(kNumClasses + 1) * list_stride]
    int32_t* counters;    // see Counter enum
    unsigned long long* cells;  // DP cells executed
    int32_t* scratch;     // generic kernel rows
    long long scratch_cap;      // in int32 units
    unsigned long long* scratch_used;
    int32_t list_stride;
    int32_t end_flags;
    int32_t window;       // half width (plan kernel)
    int32_t table_stride; // entries per read (plan kernel)
};

enum Counter {
    kCntClass0 = 0,                       // [0..kNumClasses] list lengths (kNumClasses = generic)
    kCntMiss = kNumClasses + 1,           // loci whose search left the table window
    kCntError = kNumClasses + 2,          // sticky error bits
    kCntTotal = kNumClasses + 3
};
constexpr int kErrBadInput = 1;   // empty motif / negative length
constexpr int kErrScratch = 2;    // generic scratch exhausted
constexpr int kErrEmpty = 4;      // nothing scored for some read

// ---------------------------------------------------------------------------------------------
// Plan: per read -> locus id, candidate window, table slot, kernel class.
// ---------------------------------------------------------------------------------------------
__device__ inline int classify(int nfl, int ntr, int nfr, int m, int lo, int n, int force_generic) {
    const long long ndb = (long long)nfl + ntr + nfr;
    const long long rows = (long long)nfl + (long long)(lo + n - 1) * m;
    if (force_generic || nfl < 1 || nfr < 1 || n > kTableMax) return kGenericClass;
    for (int c = 0; c < kNumClasses; ++c) {
        const int cap = class_cap(c);
        if (ndb + 1 <= cap && rows <= cap + kRowSlack) return c;
    }
    return kGenericClass;
}

// mode 0: windows from est_cn +/- window, table slot r*table_stride;
// mode 1: windows and table offsets are already in win_lo / win_n / tab_off (strk_score_table,
//         window-miss rounds); `items` (optional) restricts the launch to a list of reads.
__global__ void k_plan(KArgs a, int mode, const int32_t* items, int n_items, int force_generic) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_items) return;
    const int r = items ? items[gid] : gid;
    // locus of read r: last l with read_off[l] <= r
    int lo_l = 0, hi_l = a.n_loci;
    while (hi_l - lo_l > 1) {
        const int mid = (lo_l + hi_l) >> 1;
        if (a.read_off[mid] <= r) lo_l = mid; else hi_l = mid;
    }
    const int l = lo_l;
    a.read_locus[r] = l;
    const int m = a.motif_off[l + 1] - a.motif_off[l];
    const int nfl = a.nfl[r], ntr = a.ntr[r], nfr = a.nfr[r];
    if (m < 1 || nfl < 0 || ntr < 0 || nfr < 0) {
        atomicOr(&a.counters[kCntError], kErrBadInput);
        a.win_n[r] = 0;
        return;
    }
    int lo, n;
    if (mode == 0) {
        const long long est = a.est_cn[r];
        long long w_lo = est - a.window, w_hi = est + a.window;
        if (w_lo < 0) w_lo = 0;
        if (w_hi < w_lo) w_hi = w_lo;  // negative estimates: keep a one-entry window at 0
        if (w_hi - w_lo + 1 > a.table_stride) w_hi = w_lo + a.table_stride - 1;
        lo = (int)w_lo;
        n = (int)(w_hi - w_lo + 1);
        a.win_lo[r] = lo;
        a.win_n[r] = n;
        a.tab_off[r] = (int64_t)r * a.table_stride;
    } else {
        lo = a.win_lo[r];
        n = a.win_n[r];
    }
    if (n <= 0) return;
    // Long windows (window-miss rounds, explicit tables) are cut into items of <= kTableMax sizes.
    for (int k0 = 0; k0 < n; k0 += kTableMax) {
        const int nn = min(kTableMax, n - k0);
        const int c = classify(nfl, ntr, nfr, m, lo + k0, nn, force_generic);
        const int idx = atomicAdd(&a.counters[kCntClass0 + c], 1);
        // item encoding: read index and chunk start are stored side by side
        if (idx < a.list_stride) {
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx] = r;
            a.cls_list[(size_t)c * a.list_stride * 2 + 2 * idx + 1] = k0;
        } else {
            atomicOr(&a.counters[kCntError], kErrScratch);
        }
        const unsigned long long ndb = (unsigned long long)nfl + ntr + nfr;
        if (c == kGenericClass) {
            unsigned long long cells = 0;
            for (int k = 0; k < nn; ++k) cells += ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + k) * m + nfr);
            atomicAdd(a.cells, cells);
        } else {
            atomicAdd(a.cells, ndb * ((unsigned long long)nfl + (unsigned long long)(lo + k0 + nn - 1) * m + nfr));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast DP kernel
// ---------------------------------------------------------------------------------------------
template <int G> struct Lanes;
template <> struct Lanes<16> {
    // row_shr:1 / row_shl:1 — a 16-lane DPP row is exactly one group
    static __device__ __forceinline__ int from_left(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, 0x111, 0xf, 0xf, false); }
    static __device__ __forceinline__ int from_right(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, 0x101, 0xf, 0xf, false); }
};
template <> struct Lanes<64> {
    // wave_shr:1 / wave_shl:1 (gfx9 DPP controls, present on gfx950)
    static __device__ __forceinline__ int from_left(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, 0x138, 0xf, 0xf, false); }
    static __device__ __forceinline__ int from_right(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, 0x130, 0xf, 0xf, false); }
};

// LDS operations of one group never leave its wave: order them with a wave-level fence.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int G>
__device__ __forceinline__ int wave_max_over_groups(int v) {
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int g = 1; g < 64 / G; ++g) m = max(m, __builtin_amdgcn_readlane(v, g * G));
    return m;
}

template <int G, int CL>
struct DpLayout {
    static constexpr int CAP = G * CL;
    static constexpr int NGW = 64 / G;           // groups per wave
    static constexpr int GPB = 4 * NGW;          // groups per 256-thread block
    static constexpr int ROWS_MAX = CAP + kRowSlack;
    static constexpr int OFF_TBL = 0;                                   // 18 x 8 B row words
    static constexpr int OFF_COMB = OFF_TBL + 18 * 8;                   // kTableMax ints
    static constexpr int OFF_LMAX = OFF_COMB + kTableMax * 4;           // kTableMax ints
    static constexpr int OFF_MISC = OFF_LMAX + kTableMax * 4;           // 4 ints: symmask, zfree
    static constexpr int OFF_DB = OFF_MISC + 16;                        // CAP bytes
    static constexpr int OFF_CP = OFF_DB + CAP;                         // prefix rows
    static constexpr int LEN_CP = (ROWS_MAX + 2 * G + 15) & ~15;
    static constexpr int OFF_CT = OFF_CP + LEN_CP;                      // tail rows (reversed fr)
    static constexpr int LEN_CT = (CAP + 2 * G + 15) & ~15;
    static constexpr int GROUP_BYTES = (OFF_CT + LEN_CT + 15) & ~15;
};

template <int G, int CL>
__global__ void __launch_bounds__(256) k_dp(KArgs a, int cls) {
    using L = DpLayout<G, CL>;
    constexpr int g = kGap;
    constexpr int NQ = CL / 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds[L::GPB * L::GROUP_BYTES];

    const int lane = threadIdx.x & 63;
    const int lig = lane % G;                       // lane in group
    const int gib = (threadIdx.x >> 6) * L::NGW + lane / G;  // group in block
    uint8_t* const Lg = lds + gib * L::GROUP_BYTES;
    uint2* const tbl = reinterpret_cast<uint2*>(Lg + L::OFF_TBL);
    int* const comb = reinterpret_cast<int*>(Lg + L::OFF_COMB);
    int* const lmaxA = reinterpret_cast<int*>(Lg + L::OFF_LMAX);
    int* const misc = reinterpret_cast<int*>(Lg + L::OFF_MISC);
    uint8_t* const dbs = Lg + L::OFF_DB;
    uint8_t* const cp = Lg + L::OFF_CP;
    uint8_t* const ct = Lg + L::OFF_CT;

    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;
    const int count = min(a.counters[kCntClass0 + cls], a.list_stride);
    const int32_t* list = a.cls_list + (size_t)cls * a.list_stride * 2;

    for (int base = blockIdx.x * L::GPB; base < count; base += gridDim.x * L::GPB) {
        const int it = base + gib;
        bool act = it < count;
        int r = 0, k0 = 0, nfl = 1, ntr = 0, nfr = 1, m = 1, lo = 0, n = 0;
        long long soff = 0;
        const uint8_t* motif = a.motifs;
        if (act) {
            r = list[2 * it];
            k0 = list[2 * it + 1];
            nfl = a.nfl[r]; ntr = a.ntr[r]; nfr = a.nfr[r];
            soff = a.seq_off[r];
            const int l = a.read_locus[r];
            motif = a.motifs + a.motif_off[l];
            m = a.motif_off[l + 1] - a.motif_off[l];
            lo = a.win_lo[r] + k0;
            n = min(kTableMax, a.win_n[r] - k0);
        }
        const int ndb = nfl + ntr + nfr;
        const int rowsP = act ? nfl + (lo + n - 1) * m : 0;
        const int rowsT = act ? nfr : 0;

        // ---- stage the encoded read window and collect its symbol set --------------------------
        if (lig == 0) misc[0] = 0;
        wave_lds_sync();
        {
            unsigned mask = 0;
            for (int s = lig; s < L::CAP; s += G) {
                int sym = 0xff;
                if (act && s < ndb) {
                    sym = c_enc[a.seqs[soff + s]];
                    mask |= 1u << sym;
                }
                dbs[s] = (uint8_t)sym;
            }
            if (mask) atomicOr(reinterpret_cast<unsigned*>(&misc[0]), mask);
        }
        wave_lds_sync();
        const unsigned symmask = (unsigned)misc[0];
        const int ncls = __popc(symmask);
        if (act && ncls > 8) {
            // more distinct symbols than one v_perm word can hold: hand the item to the generic kernel
            if (lig == 0) {
                const int idx = atomicAdd(&a.counters[kCntClass0 + kGenericClass], 1);
                if (idx < a.list_stride) {
                    int32_t* gl = a.cls_list + (size_t)kGenericClass * a.list_stride * 2;
                    gl[2 * idx] = r;
                    gl[2 * idx + 1] = k0;
                } else {
                    atomicOr(&a.counters[kCntError], kErrScratch);
                }
            }
            act = false;
        }
        const int nEff = act ? n : 0;

        // ---- per-row substitution words: byte k = W(row symbol, k-th db symbol class) + 2g ------
        for (int e = lig; e < 18; e += G) {
            unsigned wlo = 0, whi = 0;
            if (e < kNSym) {
                int k = 0;
                for (int s = 0; s < kNSym; ++s) {
                    if (!((symmask >> s) & 1u)) continue;
                    if (k < 8) {
                        const unsigned b = (unsigned)(c_mat[e][s] + kWBias) & 0xffu;
                        if (k < 4) wlo |= b << (8 * k); else whi |= b << (8 * (k - 4));
                    }
                    ++k;
                }
            }
            tbl[e] = make_uint2(wlo, whi);
        }
        for (int e = lig; e < kTableMax; e += G) { comb[e] = kNegInf; lmaxA[e] = kNegInf; }
        // ---- candidate row symbols: null padding | fl | motif*i_hi | null padding --------------
        {
            const int lenP = rowsP + 2 * (G - 1);
            for (int idx = lig; idx < lenP; idx += G) {
                const int row = idx - (G - 1);  // 0-based row
                int sym = kNullSym;
                if (row >= 0 && row < rowsP) sym = row < nfl ? dbs[row] : c_enc[motif[(row - nfl) % m]];
                cp[idx] = (uint8_t)sym;
            }
            const int lenT = rowsT + 2 * (G - 1);
            for (int idx = lig; idx < lenT; idx += G) {
                const int row = idx - (G - 1);  // backward row k' - 1
                int sym = kNullSym;
                if (row >= 0 && row < rowsT) sym = dbs[nfl + ntr + nfr - 1 - row];
                ct[idx] = (uint8_t)sym;
            }
        }
        wave_lds_sync();

        // class id of a db symbol = rank of its bit in symmask
        auto sel_of = [&](int slot_char) -> unsigned {  // slot_char: index into db, or -1 for a pad
            if (slot_char < 0 || slot_char >= ndb) return 0x0cu;  // v_perm: constant 0x00
            const unsigned sym = dbs[slot_char];
            return (unsigned)__popc(symmask & ((1u << sym) - 1u));
        };

        int H[CL], B0[CL];
        unsigned sel[NQ];

        // =============================== backward pass over fr ================================
        // slot s holds node j = s (db chars s.. remain), s < ndb; slots >= ndb are inert pads that
        // carry the boundary value.  Rows k' = 1..nfr consume fr[nfr-k'].
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= sel_of(act ? lig * CL + 4 * q + b : -1) << (8 * b);
            sel[q] = v;
        }
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            const int s = lig * CL + c;
            int v = 0;
            if (s < ndb && dbEnd) v = g * (ndb - s) - (s == 0 ? g : 0);
            H[c] = v;
        }
        int zsave = 0;
        {
            const int Tb = wave_max_over_groups<G>(rowsT > 0 ? rowsT + G - 1 : 0);
            const int idxMax = rowsT + 2 * (G - 1) - 1;
            int hout = H[0];
            int edgePrev = Lanes<G>::from_right(0, hout);
            int kq = lig - (G - 1) + 1;   // row k' this lane works on at step t = 0
            int gk = g * kq;
            int zmax = H[0];
            uint2 wordNext = tbl[ct[min(lig, idxMax < 0 ? 0 : idxMax)]];
            for (int t = 0; t < Tb; ++t) {
                const uint2 word = wordNext;
                {
                    int idx = t + 1 + lig;
                    idx = idx > idxMax ? idxMax : idx;
                    wordNext = tbl[ct[idx < 0 ? 0 : idx]];
                }
                const int bnd = cEnd ? gk : 0;
                const int edge = Lanes<G>::from_right(bnd, hout);
                int d = edgePrev, l = edge;
#pragma unroll
                for (int c = CL - 1; c >= 0; --c) {
                    const unsigned wb = __builtin_amdgcn_perm(word.y, word.x, sel[c / 4]);
                    const int w = (int)((wb >> (8 * (c % 4))) & 0xffu);
                    const int up = H[c];
                    const int nh = max(max(up, l), d + w);
                    d = up; l = nh; H[c] = nh;
                }
                edgePrev = edge;
                hout = H[0];
                if (kq == rowsT) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) B0[c] = H[c];
                    zsave = zmax;
                }
                zmax = kq <= 0 ? H[0] : max(zmax, H[0] - gk);
                ++kq; gk += g;
            }
            if (rowsT == 0) {  // unreachable for fast-class items (nfr >= 1); keeps B0 defined
#pragma unroll
                for (int c = 0; c < CL; ++c) B0[c] = H[c];
            }
        }
        if (lig == 0) misc[1] = zsave;

        // =============================== forward pass over fl + motif*i_hi =====================
        // slot 0 is an inert pad carrying the left boundary; slot s = 1..ndb holds node j = s
        // (consumes db[s-1]); slots > ndb replicate the last column.
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int s = lig * CL + 4 * q + b;
                v |= sel_of(act && s >= 1 ? s - 1 : -1) << (8 * b);
            }
            sel[q] = v;
        }
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            const int s = lig * CL + c;
            H[c] = dbBeg ? g * min(s, ndb) : 0;
        }
        {
            const int Tf = wave_max_over_groups<G>(nEff > 0 ? rowsP + G - 1 : 0);
            const int idxMax = rowsP + 2 * (G - 1) - 1;
            int hout = H[CL - 1];
            int edgePrev = Lanes<G>::from_left(0, hout);
            int rq = 1 - lig;          // row this lane works on at step t = 0
            int gr = g * rq;
            int lastmax = kNegInf;
            int nextFork = nEff > 0 ? nfl + lo * m : 0x7fffffff;
            int forkIdx = 0;
            uint2 wordNext = tbl[cp[min((G - 1) - lig, idxMax < 0 ? 0 : idxMax)]];
            for (int t = 0; t < Tf; ++t) {
                const uint2 word = wordNext;
                {
                    int idx = t + 1 + (G - 1) - lig;
                    idx = idx > idxMax ? idxMax : idx;
                    wordNext = tbl[cp[idx < 0 ? 0 : idx]];
                }
                const int bnd = cBeg ? gr : 0;
                const int edge = Lanes<G>::from_left(bnd, hout);
                int d = edgePrev, l = edge;
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    const unsigned wb = __builtin_amdgcn_perm(word.y, word.x, sel[c / 4]);
                    const int w = (int)((wb >> (8 * (c % 4))) & 0xffu);
                    const int up = H[c];
                    const int nh = max(max(up, l), d + w);
                    d = up; l = nh; H[c] = nh;
                }
                edgePrev = edge;
                hout = H[CL - 1];
                lastmax = rq <= 0 ? kNegInf : max(lastmax, hout - gr);
                if (rq == nextFork) {
                    int acc = kNegInf;
#pragma unroll
                    for (int c = 0; c < CL; ++c) acc = max(acc, H[c] + B0[c]);
                    atomicMax(&comb[forkIdx], acc);
                    if (lig == G - 1) lmaxA[forkIdx] = lastmax;
                    ++forkIdx;
                    nextFork = forkIdx < nEff ? nextFork + m : 0x7fffffff;
                }
                ++rq; gr += g;
            }
        }
        wave_lds_sync();
        // ---- assemble S[lo + k] ---------------------------------------------------------------
        if (act) {
            const int zfree = misc[1] - g * ndb;
            for (int k = lig; k < n; k += G) {
                const int R = nfl + (lo + k) * m;
                int sc = comb[k] - g * (R + nfr + ndb);
                if (cEnd) sc = max(sc, lmaxA[k] - g * ndb);
                if (cBeg) sc = max(sc, zfree);
                a.table[a.tab_off[r] + k0 + k] = sc;
            }
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
        const long long ncand = (long long)nfl + (long long)i * m + nfr;
        int32_t* out = a.table + a.tab_off[r] + k0 + k;
        if (ndb <= 0 || ncand <= 0) { *out = 0; continue; }
        const unsigned long long need = (unsigned long long)ndb + 1;
        const unsigned long long at = atomicAdd(a.scratch_used, need);
        if (at + need > (unsigned long long)a.scratch_cap) {
            atomicOr(&a.counters[kCntError], kErrScratch);
            *out = 0;
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
        int best = Hrow[ndb];
        if (cEnd) best = max(best, lastcol);
        if (dbEnd) for (int j = 1; j <= ndb; ++j) best = max(best, Hrow[j]);
        *out = best;
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

__global__ void k_replay(KArgs a, ReplayArgs p) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.n_loci) return;
    const int r_end = a.read_off[l + 1];
    double frac = 0.0;
    int r = a.read_off[l];
    for (; r < r_end; ++r) {
        const int est = a.est_cn[r];
        int start = est;
        double frac_try = frac;
        if (p.feedback) start = feedback_start(est, &frac_try);
        SeenMask64 seen;
        const SearchResult res = search_replay(start, p.step, p.lsr, p.max_iters, p.tie_last,
                                               a.table + a.tab_off[r], a.win_lo[r], min(a.win_n[r], 64), seen);
        if (res.miss) {
            p.need_lo[l] = res.need_lo;
            p.need_hi[l] = res.need_hi;
            atomicAdd(&a.counters[kCntMiss], 1);
            break;
        }
        frac = frac_try;
        if (res.empty) {
            atomicOr(&a.counters[kCntError], kErrEmpty);
            p.out_cn[r] = 0; p.out_score[r] = 0; p.out_n[r] = res.n_explored; p.out_start[r] = start;
            continue;  // the reference would raise here; the host turns the flag into an error
        }
        p.out_cn[r] = res.cn;
        p.out_score[r] = res.score;
        p.out_n[r] = res.n_explored;
        p.out_start[r] = start;
        if (p.feedback) feedback_update(&frac, res.cn, start);
    }
    p.next_read[l] = r;
    p.frac[l] = frac;
}

}  // namespace strk
