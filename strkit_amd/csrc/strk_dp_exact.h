// strk_dp_exact.h — exact column-owned systolic kernels k_dp_all / k_dp_ref (all fast classes in one launch)
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

// ---------------------------------------------------------------------------------------------
// Fast DP kernel
// ---------------------------------------------------------------------------------------------
constexpr int kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;  // gfx9 DPP controls, present on gfx950
constexpr int kCLMax = 40;   // columns per lane of the largest class
#ifndef STRK_STAIR_MIN_G
#define STRK_STAIR_MIN_G 16   // (config 3 on one MI355X: 7.97 ms per 200 000 reads at 32, 7.68 at 16, 7.76 at 8)
#endif
constexpr int kStairMinG = STRK_STAIR_MIN_G;   // classes with at least this many lanes per read fold fork rows along a staircase (bwd_pass)
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
    int cap, off_db, off_cp, off_ct, off_b0, off_bj, group_bytes;
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
          off_bj(off_b0 + ((G * CL * 2 + 15) & ~15)),                        // staircase fork rows: one int per lane
          group_bytes(off_bj + ((G * 4 + 15) & ~15)) {}
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
    int* bj;                 // LDS: staircase fork rows, this lane's first slot one backward row before its b0 row
};

// Row word of a row symbol that was staged PRE-MULTIPLIED by 8 (its byte offset in the row-word table): one add for the
// address instead of a mask and a shift-add per step.
__device__ __forceinline__ uint2 row_word8(const uint2* tbl, unsigned sym8) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(tbl) + sym8);
}

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
//
// STAIR (staircase fork rows, classes of 32 and 64 lanes per read): the forward pass then folds a fork row when its FIRST lane
// finishes it, lane l standing l rows above it — the cut through the matrix is a staircase, and lane l needs the backward
// values of a suffix that still holds the last l rows in front of the fork row.  Those rows are motif rows (the caller
// checks (smallest candidate) * |motif| >= G - 1) and, a fork row sitting at the end of a motif copy, the same for every
// candidate: the reversed motif, cyclically.  The backward pass simply runs on over them (row symbols staged behind the
// reversed right flank): lane l works G - 1 - l rows behind the last lane, so row rowsT + l of lane l is the pass's LAST
// step for every lane — the skew steps that used to compute nothing now compute the extension, and the result is stored
// once, by all lanes together, instead of at G different steps.  One value more per lane: its first slot one row earlier
// (bj), for the one transition the staircase cut does not see (fwd_pass).
template <int NQ, int G, bool STAIR>
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
    const int gkEvent = g * (rowsT + (STAIR ? x.lig : 0));   // STAIR: every lane of a group reaches its row at step rowsT + G - 2
    int zsave = 0;
    int hout = Ha[0];
    int edgePrev = from_right<G>(0, hout, x.last);
    int gk = g * (x.lig - (G - 1));   // g * k' of the row this lane finished before step 0
    int zmax = Ha[0];
    const uint8_t* pa = ct + x.lig;   // row symbol of step t is pa[t]
    uint2 wordNext = row_word8(x.tbl, pa[0]);
    unsigned symNext = pa[1];
#define STRK_BSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = row_word8(x.tbl, symNext);                                                \
        symNext = pa[(T) + 2];                                                               \
        asm volatile("" : "+v"(symNext));   /* (a plain 32-bit value from here on: no re-masking of the byte next step) */ \
        const int edge = from_right<G>(bstep * ((T) + 1), hout, x.last);                        \
        hout = dp_row<NQ, false>(SRC, DST, sel, word, edge, edgePrev);                       \
        edgePrev = edge;                                                                     \
        gk += g;                                                                             \
        if (gk == gkEvent) {                                                                 \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                   \
                x.b0[q * G] = make_uint2((unsigned)DST[4 * q] | ((unsigned)DST[4 * q + 1] << 16), \
                                         (unsigned)DST[4 * q + 2] | ((unsigned)DST[4 * q + 3] << 16)); \
            if (STAIR) *x.bj = SRC[0];                                                       \
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
//
// STAIR: a fork row is folded at the step at which the group's first lane finishes it (one execution of the fork block per
// fork row and group, every lane taking part, instead of one per lane and fork row); lane l then stands at row R_k - l and
// adds the backward values of the suffix "last l motif rows + right flank" (bwd_pass).  Every alignment passes through one
// cell of that staircase — a monotone path leaves the region above it exactly once, and its last cell above is a cut cell —
// with ONE exception: the diagonal step from the last column of lane l - 1, one row above that lane's cut row, into the
// first column of lane l, one row below lane l's cut row.  Lane l folds that transition by itself: the left neighbour's
// value is its own edge input of this step, the substitution score comes from its next row's word, and the backward value
// of the cell entered is its first slot one backward row earlier (bj).
template <int NQ, int G, bool STAIR>
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
    int forkG = nEff > 0 ? g * (fork0 - (STAIR ? x.lig : 0)) : 0x7fffffff;
    int forkIdx = 0;
    const int bjump = STAIR ? (x.first ? kNegInf : *x.bj) : 0;
    const uint8_t* pa = cp + (G - 1) - x.lig;
    uint2 wordNext = row_word8(x.tbl, pa[0]);
    unsigned symNext = pa[1];
#define STRK_FSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = row_word8(x.tbl, symNext);                                                \
        symNext = pa[(T) + 2];                                                               \
        asm volatile("" : "+v"(symNext));   /* (a plain 32-bit value from here on: no re-masking of the byte next step) */ \
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
            if (STAIR)                                                                       \
                acc = max(acc, edgePrev + (int)(__builtin_amdgcn_perm(wordNext.y, wordNext.x, sel[0]) & 0xffu) + bjump); \
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
    uint2 wordNext = row_word8(x.tbl, pa[0]);
    unsigned symNext = pa[1];
#define STRK_RSTEP(SRC, DST, T)                                                              \
    {                                                                                        \
        const uint2 word = wordNext;                                                         \
        wordNext = row_word8(x.tbl, symNext);                                                \
        symNext = pa[(T) + 2];                                                               \
        asm volatile("" : "+v"(symNext));   /* (a plain 32-bit value from here on: no re-masking of the byte next step) */ \
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
    // staircase fork rows (bwd_pass / fwd_pass): every item of the chunk must have G - 1 motif rows in its smallest candidate
    const bool stair = !ref_mode && G >= kStairMinG && !(ap->dbg & 32) &&
                       __builtin_amdgcn_ballot_w64(act && (long long)lo * m < G - 1) == 0;

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
                atomicOr(&counters[kCntError], kErrList);
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
            cp[idx] = (uint8_t)(8 * sym);   // (the byte offset of the symbol's row word: row_word8)
            ph += gstep;
            if (ph >= m) ph -= m;
        }
        const int lenT = rowsT + 2 * (G - 1) + 4;
        for (int idx = lig; idx < lenT; idx += G) {
            const int row = idx - (G - 1);  // backward row k' - 1
            int sym = kNullSym;
            if (row >= 0 && row < rowsT) sym = dbs[4 + ndb - 1 - row];
            else if (stair && row >= rowsT && row < rowsT + G - 1) sym = motifL[m - 1 - (row - rowsT) % m];   // the reversed motif, cyclically
            ct[idx] = (uint8_t)(8 * sym);
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
    x.bj = reinterpret_cast<int*>(Lg + lay.off_bj) + lig;

    // the two hot loops are specialised on (CL/4, G): 14 instances each, everything else is one body
    int zsave = 0;
    const int fork0 = nfl + lo * m;
#define STRK_PASSES(NQ_, G_)                                                     \
    {                                                                            \
        if (G_ >= kStairMinG && stair) {                                         \
            zsave = bwd_pass<NQ_, G_, (G_ >= kStairMinG)>(x, rowsT, ct);          \
            if (first) misc[1] = zsave;                                          \
            wave_lds_sync();                                                     \
            fwd_pass<NQ_, G_, (G_ >= kStairMinG)>(x, rowsP, cp, nEff, fork0, m, comb, lmaxA); \
        } else {                                                                 \
            zsave = bwd_pass<NQ_, G_, false>(x, rowsT, ct);                       \
            if (first) misc[1] = zsave;                                          \
            fwd_pass<NQ_, G_, false>(x, rowsP, cp, nEff, fork0, m, comb, lmaxA);  \
        }                                                                        \
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
        const SearchResult res = search_replay(ap2->est_cn[r], ap2->step, ap2->lsr, ap2->max_iters, ap2->tie_last, comb, lo, n, seen, ap2->narrow);
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

}  // namespace strk
