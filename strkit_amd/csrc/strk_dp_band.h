// strk_dp_band.h — diagonal-owned banded first pass k_dp_band / k_dp_band_wide with in-kernel certificate
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

// ---------------------------------------------------------------------------------------------
// Band kernel (see strk_search.h "Banded scoring with an exactness certificate").  Lanes own
// DIAGONALS instead of columns: lane l of a group keeps the D diagonals d = dlo + D l .. + D - 1 of the
// current row (D = 16; 12 in the narrow class), so a group of 8 (16) lanes covers a band
// of 128 (256) diagonals that follows the alignment down the matrix.  Per row and slot k:
//     up   = (r-1, j)   = old[k+1]   (the next lane's NEW [0] for k = D-1: it works one row behind; a DPP, mid-step)
//     left = (r, j-1)   = new[k-1]   (the previous lane's new[D-1] for k = 0: the systolic skew)
//     diag = (r-1, j-1) = old[k] + w
// and the selector bytes of the lane's D columns slide by one column per row (D/4 v_alignbyte plus
// one LDS byte).  Cells outside the band are 0 in G-space (= -inf: every real value is >= 0); the pad columns
// left of column 1 and the rows in front of row 1 carry the boundary values by themselves (row-word tables below),
// cells right of the last column replicate it.
// The backward pass is the same function on the reversed right flank and the reversed window.
// ---------------------------------------------------------------------------------------------
struct BandLayout {
    // class-byte array: selb[pad + x] <-> db[x]; `pad` pad selectors in front and up to pad + kBandHiPad
    // behind, sized so that no column the two passes can ask for (virtual rows, rows beyond |db|, the
    // prefetch) falls outside it: the hot loop indexes it without clamping.
    int wd, pad, maxdb, maxcol, off_sel, off_cp, off_ct, off_b0, group_bytes, sel_len;
    static constexpr int OFF_COMB = 0, OFF_MISC = OFF_COMB + kTableMax * 4, OFF_LMAX = OFF_MISC + 16;
    static constexpr int kBandHiPad = kBandRowSlack + 32;
    __host__ __device__ constexpr BandLayout(int c)   // c = layout class (band_class_layout: 0..3)
        : wd(128 << c), pad((128 << c) + (8 << c) + 8), maxdb(band_max_db(c)), maxcol(band_max_col(c)),
          off_sel(OFF_LMAX + (band_class_lmax(c) ? kTableMax * 4 : 0)),
          off_cp(off_sel + ((band_max_db(c) + 2 * ((128 << c) + (8 << c) + 8) + kBandHiPad + 15) & ~15)),
          // forward row symbols: the whole prefix (classes 0, 1) or 256 flank + 256 motif symbols (2, 3)
          off_ct(off_cp + (band_class_fly(c) ? 512 : ((band_max_db(c) + kBandRowSlack + 2 * (8 << c) + 4 + 15) & ~15))),
          off_b0(off_ct + ((kBandMaxFlank + 2 * (8 << c) + 4 + 15) & ~15)),
          group_bytes(off_b0 + ((band_max_col(c) * 2 + 15) & ~15)),
          sel_len((band_max_db(c) + 2 * ((128 << c) + (8 << c) + 8) + kBandHiPad) & ~3) {}
};
static_assert(band_class_layout(4) == 0 && band_class_layout(7) == 3 && band_class_wd(5) <= band_class_wd(1) && band_class_G(6) == band_class_G(2),
              "a 12-diagonal class uses the LDS layout of the 16-diagonal class with its lane count: pads sized for the wider band cover it");
__host__ __device__ constexpr int band_wave_lds(int c) { return (64 / (8 << c)) * BandLayout(c).group_bytes; }
__host__ __device__ constexpr int max_band_wave_lds(int c) {
    return c < 0 ? 0 : (band_wave_lds(c) > max_band_wave_lds(c - 1) ? band_wave_lds(c) : max_band_wave_lds(c - 1));
}
constexpr int kBandWaveLds = max_band_wave_lds(3);   // layout classes 0..3
constexpr int kBandNeg16 = -20000;
// Fixed symbol classes of the band kernels: what an alignment file's reads hold after wildcarding (call_locus.py:79):
// A C G T, N and the wildcard X, in either case.  v_perm selector = class index = bits 3:1 of the ASCII code (A 0, C 1, T 2,
// G 3, X 4, N 7: four window bytes become four selectors with one shift and one AND, a v_perm of the expected letters and
// an XOR tell whether all four really are one of the six); a read with another symbol (an IUPAC code, a byte outside the
// alphabet) takes the exact kernels, whose classes are per read.
// Row-word tables (8 bytes per row symbol, one table per pass): bytes 0..4 and 7 = W + 2g against the six classes, byte 5 =
// what a pad column in FRONT of the window adds on the diagonal, byte 6 = the same for a pad column BEHIND it.  A free left
// boundary is G(r, j <= 0) = g r in G-space, i.e. "+ g per row" along every diagonal left of column 1: the pass whose left
// side those pads are gets g there when that end is free (forward: front pads / candidate begin; backward, in reversed
// coordinates: back pads / candidate end) and 0 otherwise — the pads then carry the boundary value by themselves.  The same
// for the top boundary G(0, j) = g j: the rows in front of the matrix ("no row" symbol) add g per real column when the
// window's end on that side is free.  Only the cell right next to the boundary column / row still needs the boundary value
// handed in (one step per pass, band_pass).
// Entries: 0..17 by encoded symbol (16 = '*', 17 = no row), then one entry per class for flank rows, which are staged as
// selector bytes.  Row symbols are staged PRE-MULTIPLIED by 8 (the table offset of their word).
constexpr int kBandTblClass0 = 18;
constexpr int kBandSelFront = 5, kBandSelBack = 6;
constexpr int kBandTblBytes = 256;   // per pass: 26 entries used; a stale row symbol (any byte) stays inside it
// encoded symbol (strk_scoring.h alphabet "ACGTRYSWKMBDHVNX") of selector c, -1 for the two pad selectors
__host__ __device__ constexpr int band_class_symbol(int c) {
    return c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 3 : (c == 3 ? 2 : (c == 4 ? 15 : (c == 7 ? 14 : -1)))));
}
constexpr unsigned kBandLetterLo = 0x47544341u, kBandLetterHi = 0x4E000058u;   // "ACTG", "X\0\0N"
typedef const __attribute__((address_space(3))) uint8_t* lds_cu8;

struct BandCtx {
    int lig;
    bool first, last;
    int notFirst, notLast;   // 0 in the first / last lane of a group, else all ones
    const uint8_t* tbl;      // the pass's row-word table (set by band_wave before each pass)
    const uint8_t* selb;   // class-byte array: selb[pad + x] <-> db[x], 0x0c elsewhere
    int pad, maxidx, ndb;
    const uint8_t* flL;    // on-the-fly forward rows: 256 left-flank symbols, 256 motif symbols
    const uint8_t* motifL;
    int nfl, m;
    int flyP;              // on-the-fly rows: the whole number of motif copies (in bytes) the running address is taken back by
};

// min over the (at most eight) groups of a wave of a value that is uniform inside each group
__device__ __forceinline__ int wave_min_over_groups(int v) {
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int l = 8; l < 64; l += 8) m = min(m, __builtin_amdgcn_readlane(v, l));
    return m;
}
// lane l gets v of lane l - 1 (l + 1), the first (last) lane of every group 0 — the value of a cell outside the band in
// G-space.  One instruction: the DPP shift zero-fills at the ends of its row / of the wave and is an operand of the AND
// that clears the seam lanes where a group is shorter than that.
template <int G>
__device__ __forceinline__ int from_left0(int v, int notFirst) {
    const int x = __builtin_amdgcn_update_dpp(0, v, G <= 16 ? kDppRowShr1 : kDppWaveShr1, 0xf, 0xf, true);
    return (G == 16 || G == 64) ? x : (x & notFirst);
}
template <int G>
__device__ __forceinline__ int from_right0(int v, int notLast) {
    const int x = __builtin_amdgcn_update_dpp(0, v, G <= 16 ? kDppRowShl1 : kDppWaveShl1, 0xf, 0xf, true);
    return (G == 16 || G == 64) ? x : (x & notLast);
}

// One banded pass over `nrows` rows.  BWD = false: forward pass (columns = db, left to right);
// BWD = true: backward pass in reversed coordinates (columns = reversed db).  dlo_ is the first
// diagonal of the band, topFree/leftFree the free-end flags of the top row / left column.
// Steps come in two forms.  The plain one takes 0 (outside the band) from both sides.  The other one also hands in the
// boundary values: the row-0 pattern above the last lane's first row (step G - 1) and the left-boundary column while it lies
// right next to the band (step -dlo of each group; the pads further left carry the boundary by themselves, see the row-word
// tables above).  Only the pairs of steps that hold one of those moments run the second form.
template <int G, int D, bool BWD, bool FLY, bool LMAX>
__device__ __forceinline__ void band_pass(const BandCtx& x, const uint8_t* rowsym, int nrows, int dlo_, bool topFree,
                                          bool leftFree, int nEff, int fork0, int m, int cmin, int ncol, int* comb,
                                          short* b0col, int* lmaxA) {
    static_assert(D % 4 == 0 && D >= 8 && D <= 16, "a lane keeps 8, 12 or 16 diagonals");
    constexpr int g = kGap, NQ = D / 4;
    const int ncols = x.ndb;
    const int d0 = dlo_ + x.lig * D;                  // diagonal of this lane's slot 0
    auto g0 = [&](int j) -> int { return topFree ? g * min(max(j, 0), ncols) : 0; };   // row-0 pattern
    auto col_addr = [&](int j) -> int {               // LDS index of the class byte of column j (1-based)
        return BWD ? x.pad + ncols - j : x.pad + j - 1;   // always inside the padded array (BandLayout)
    };
    int Ha[D], Hb[D];
    int Hsave[BWD ? D : 1];   // backward pass: the lane's last row
    unsigned sel[NQ];
#pragma unroll
    for (int k = 0; k < (BWD ? D : 1); ++k) Hsave[k] = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) Ha[k] = g0(-x.lig + d0 + k);
    const int jb0 = 1 - x.lig + d0;                   // column of slot 0 at the row of step 0 (step t: jb0 + t)
    {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (unsigned)x.selb[col_addr(jb0 + 4 * q + b)] << (8 * b);
            sel[q] = v;
        }
    }
    const int T = (wave_max_over_groups(nrows > 0 ? nrows + G - 1 : 0) + 1) & ~1;
    const int dhi_ = dlo_ + D * G - 1;
    int houtL = Ha[D - 1];
    // the step at which this lane finishes its next fork row (forward), its last row (backward: the one event of that pass)
    int forkT = (!BWD && nEff > 0) ? fork0 - 1 + x.lig : 0x7fffffff;
    if (BWD && nrows > 0) forkT = nrows - 1 + x.lig;
    int forkIdx = 0;
    // classes of band_class_lmax: running maximum, over the rows from 1 on, of (last slot of the last lane) - g * row.  Right of
    // column |db| the pads replicate G(r, |db|), so from row |db| - dhi on that slot holds the last column's value: an alignment
    // that ends there.  Before that row it holds the band's rightmost cell (r, j < |db|), and G(r, j) - g r - g |db| =
    // H(r, j) - g (|db| - j) is the score of a real alignment too (gap the rest of the window, end in the last column), so it
    // may take part: S_band stays a lower bound, and the test that used to exclude those rows costs two instructions per step.
    // Kept as M = max_r (G(r, .) - g r) + g * (row just finished): M <- max(M + g, G(row, .)).
    int lastmax = kNegInf;
    // row symbols (8 * table index): staged in LDS with G-1 null rows in front (this lane's row at step t is psym[t]), or
    // generated two steps ahead from the left flank and the motif phase (long windows).  psym / pnb are running LDS
    // addresses, advanced once per pair of steps; the steps read at constant offsets from them.
    // FLY (long windows): the same running address over 256 flank + 256 motif symbols — the flank staged right-aligned in front of
    // the motif (null rows in front of it: lane l starts l rows before row 0), the motif repeated cyclically behind it — so the
    // address runs from the null rows through the flank into the motif without a test and is taken back by a whole number of
    // motif copies (x.flyP) once per PAIR of steps when it has passed them (add, compare, select per pair; a phase counter per
    // row with two compares and two selects cost a tenth of the wide classes' steps).
    lds_cu8 psym = FLY ? (lds_cu8)(x.flL + 256 - x.nfl - x.lig) : (lds_cu8)(rowsym + (G - 1) - x.lig);
    lds_cu8 const flyEnd = (lds_cu8)x.motifL + x.flyP;
    lds_cu8 const tbl = (lds_cu8)x.tbl;
    auto row_word = [&](unsigned sym8) -> uint2 {
        const unsigned long long v = *reinterpret_cast<const __attribute__((address_space(3))) unsigned long long*>(tbl + sym8);
        return make_uint2((unsigned)v, (unsigned)(v >> 32));
    };
    // row words are fetched two steps ahead (wordE / wordO: even / odd steps).  Every LDS byte is loaded at the head of a
    // step and used at its end, in the same basic block: the compiler then knows that ds_read_u8 zero-extends (a value that
    // crosses the fork-row branch would be masked again).
    uint2 wordE, wordO;
    wordE = row_word(psym[0]);
    wordO = row_word(psym[1]);
    // class byte entering the lane's window after step t: column jb0 + t + D
    lds_cu8 pnb = (lds_cu8)(x.selb + col_addr(jb0 + D)) - (BWD ? 1 : 0);
#define STRK_BAND_STEP(SRC, DST, TT, ODD, EDGE)                                                    \
    {                                                                                              \
        STRK_BAND_FORK(SRC, (TT) - 1)                                                              \
        const uint2 word = (ODD) ? wordO : wordE;                                                  \
        unsigned sym2 = (unsigned)psym[2 + (ODD)];                      /* row of step TT + 2 */   \
        unsigned nb = pnb[BWD ? 1 - (ODD) : (ODD)];                                                \
        int leftEdge, keepU = 0;                                                                   \
        if (EDGE) {                                                                                \
            /* lane 0 works on row TT+1; left of the band lies the left boundary while j-1 <= 0 */  \
            const int keepL = ((TT) + dlo_ <= 0 && leftFree) ? g * ((TT) + 1) : 0;                  \
            leftEdge = from_left<G>(keepL, houtL, x.first);                                        \
            /* the last lane works on row TT-G+2; above row 1 lies the row-0 pattern, else -inf */  \
            const int rl = (TT) - G + 2;                                                           \
            keepU = rl <= 1 ? g0(rl + dhi_) : 0;                                                   \
        } else {                                                                                   \
            leftEdge = from_left0<G>(houtL, x.notFirst);                                           \
        }                                                                                          \
        unsigned wq[NQ];                                                                           \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) wq[q] = __builtin_amdgcn_perm(word.y, word.x, sel[q]); \
        DST[0] = max(max(SRC[1], leftEdge), SRC[0] + (int)(wq[0] & 0xffu));                        \
        const int upEdge = (EDGE) ? from_right<G>(keepU, DST[0], x.last) : from_right0<G>(DST[0], x.notLast); \
        _Pragma("unroll") for (int k = 1; k < D - 1; ++k)                                          \
            DST[k] = max(max(SRC[k + 1], DST[k - 1]), SRC[k] + (int)((wq[k / 4] >> (8 * (k % 4))) & 0xffu)); \
        DST[D - 1] = max(max(upEdge, DST[D - 2]), SRC[D - 1] + (int)(wq[NQ - 1] >> 24));           \
        houtL = DST[D - 1];                                                                        \
        /* the two bytes as plain 32-bit values from here on (zero-extended by the load, which the compiler knows in   \
           this block only: their uses may be moved below the next fork-row branch) */                  \
        asm volatile("" : "+v"(nb), "+v"(sym2));                                                   \
        if (LMAX && !BWD) {                                                                        \
            lastmax = max(lastmax + g, houtL);      /* (kept shifted by g * row: add and max per step) */ \
            /* the last lane has just finished row 0 at step G - 2 = tTop, which only ever runs as a boundary step (a wave-uniform  \
               test in the plain steps became s_cselect vcc + v_cndmask: 23 cycles for a VOP2 select whose VCC no VALU compare \
               has just written, profiles/r04_valu_rate3.txt) */                                    \
            if ((EDGE) && (TT) == G - 2) lastmax = kNegInf;                                        \
        }                                                                                          \
        _Pragma("unroll") for (int q = 0; q + 1 < NQ; ++q) sel[q] = __builtin_amdgcn_alignbyte(sel[q + 1], sel[q], 1); \
        sel[NQ - 1] = __builtin_amdgcn_alignbyte(nb, sel[NQ - 1], 1);                              \
        if (ODD) wordO = row_word(sym2); else wordE = row_word(sym2);                              \
    }
    // the fork-row work of step TT on that step's output row V.  It runs at the head of the NEXT step, so that a step
    // itself is one basic block from its LDS byte loads to their uses.
#define STRK_BAND_FORK(V, TT)                                                                      \
    /* (backward: the body is a dozen register copies, which the compiler would predicate and execute at EVERY step; the     \
       wave-uniform test in front makes it a real branch) */                                                                 \
    if ((!BWD || __builtin_amdgcn_ballot_w64((TT) == forkT) != 0) && (TT) == forkT) {              \
        const int jb = jb0 + (TT);                      /* column of slot 0 at that step's row */  \
        if (BWD) {                                                                                 \
            /* last row of this lane: put aside (every lane reaches it at a step of its own; the store into the column   \
               array, with its bounds checks per slot, runs once behind the loop for all lanes together) */               \
            _Pragma("unroll") for (int k = 0; k < D; ++k)                                          \
                asm volatile("v_mov_b32 %0, %1" : "+v"(Hsave[k]) : "v"(V[k]));   /* (a plain copy makes the allocator merge the   \
                    save array with the row arrays and copy a whole row per pair of steps instead) */                      \
            forkT = 0x7fffffff;                                                                    \
        } else {                                                                                   \
            const short* bc = b0col + (jb - cmin);                                                 \
            int acc = kNegInf;                                                                     \
            _Pragma("unroll") for (int k = 0; k < D; k += 2)                                       \
                acc = max(max(acc, V[k] + (int)bc[k]), V[k + 1] + (int)bc[k + 1]);                 \
            atomicMax(&comb[forkIdx], acc);                                                        \
            if (LMAX && x.last) lmaxA[forkIdx] = lastmax - g * ((TT) + 1 - x.lig);   /* the step's row */ \
            ++forkIdx;                                                                             \
            forkT = forkIdx < nEff ? forkT + m : 0x7fffffff;                                       \
        }                                                                                          \
    }
    // pairs of steps that must hand in a boundary value: the one with step G - 1 (row-0 pattern above the last lane's row 1)
    // and those with the step -dlo of some group (left-boundary column right next to the band)
    const bool hasL = nrows > 0 && leftFree && dlo_ <= 0;
    const int tTop = (G - 1) & ~1;
    const int tL0 = wave_min_over_groups(hasL ? -dlo_ : 0x3fffffff) & ~1;
    const int tL1 = wave_max_over_groups(hasL ? -dlo_ : -1);
    // runs of plain pairs and runs of boundary pairs alternate, each run a loop of its own (one loop with both forms in
    // its body makes the compiler copy the two row arrays once per pair)
    auto advance = [&]() {
        psym += 2;
        if (FLY) {   // psym -= flyP when it has passed flyEnd — without a select (sub, ashr, and, sub: four fast-class instructions)
            int past = (int)(flyEnd - psym - 1) >> 31;   // all ones when psym >= flyEnd
            asm volatile("" : "+v"(past));                  // (or the compiler folds shift + and back into v_cmp + v_cndmask)
            psym -= x.flyP & past;
        }
        asm volatile("" : "+v"(psym));
        pnb += BWD ? -2 : 2;
        asm volatile("" : "+v"(pnb));
    };
    int t = 0;
    while (t < T) {
        int ts = T;                                    // the next boundary pair at or after t
        if (tTop >= t) ts = min(ts, tTop);
        if (tL1 >= t) ts = min(ts, max(tL0, t));
        for (; t < ts; t += 2) {
            STRK_BAND_STEP(Ha, Hb, t, 0, false)
            STRK_BAND_STEP(Hb, Ha, t + 1, 1, false)
            advance();
        }
        for (; t < T && (t == tTop || (t >= tL0 && t <= tL1)); t += 2) {
            STRK_BAND_STEP(Ha, Hb, t, 0, true)
            STRK_BAND_STEP(Hb, Ha, t + 1, 1, true)
            advance();
        }
    }
    STRK_BAND_FORK(Ha, T - 1)
    if (BWD && nrows > 0) {
        // last row: slot k is reversed column jb + k, i.e. db node ndb - (jb + k)
        const int jb = jb0 + nrows - 1 + x.lig;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int jp = jb + k, idx = ncols - jp - cmin;
            if (jp >= 0 && jp <= ncols && idx >= 0 && idx < ncol) b0col[idx] = (short)Hsave[k];
        }
    }
#undef STRK_BAND_STEP
#undef STRK_BAND_FORK
}

// Profiling aid (tools/phase_timing.sh builds a private copy of the library with -DSTRK_PHASE_TIMING): shader-clock
// ticks per phase of band_wave, summed over waves into the spare counter slots 48..55.
#ifdef STRK_PHASE_TIMING
#define STRK_PHASE(i)                                                                                  \
    do {                                                                                               \
        const unsigned long long t_ = __builtin_readcyclecounter();                                    \
        if (lane == 0) atomicAdd(&a.counters[48 + (i)], (int)((t_ - tphase) >> 6));                    \
        tphase = t_;                                                                                   \
    } while (0)
#else
#define STRK_PHASE(i) do { } while (0)
#endif

// Processes 64/G items of band class BC (G = band_class_G(BC) lanes per read), one per group.
// nextA / nextB / nextC: the caller's three steps towards the NEXT chunk (take it from the queue; fetch its records; touch its
// window bytes), called where each one's result has had time to arrive and where its own loads are not in the way of this
// chunk's (the memory counter retires loads in order: a load issued behind a slow one waits for it).
template <int BC, class NA, class NB, class NC>
__device__ __forceinline__ void band_wave(const KArgs& a, bool act, int4 q0, int4 q1, int4 q2, uint8_t* Lw, const uint8_t* s_enc,
                                          const uint8_t* s_tbl, NA& nextA, NB& nextB, NC& nextC) {
    constexpr int g = kGap, G = band_class_G(BC), D = band_class_D(BC);
    constexpr bool FLY = band_class_fly(BC);
    constexpr bool LMAX = band_class_lmax(BC);
    constexpr BandLayout lay(band_class_layout(BC));
    const int lane = threadIdx.x & 63;
#ifdef STRK_PHASE_TIMING
    unsigned long long tphase = __builtin_readcyclecounter();
    const unsigned long long tchunk = tphase;
#endif
    int lig_ = lane & (G - 1);
    // (opaque per chunk: what a pass derives from the lane index and the class's constants — sixteen clamped row-0 values of
    // the backward band, for one — would otherwise be hoisted out of the chunk loop and held, or spilled, across both passes)
    asm volatile("" : "+v"(lig_));
    const int lig = lig_;
    const int grp = lane / G;
    const bool first = lig == 0, last = lig == G - 1;
    uint8_t* const Lg = Lw + grp * lay.group_bytes;
    // row words: the same for every read (fixed symbol classes, band_kernel_body); forward table, then backward table
    int* const comb = reinterpret_cast<int*>(Lg + BandLayout::OFF_COMB);
    int* const misc = reinterpret_cast<int*>(Lg + BandLayout::OFF_MISC);
    int* const lmaxA = reinterpret_cast<int*>(Lg + BandLayout::OFF_LMAX);   // classes of band_class_lmax only
    uint8_t* const selb = Lg + lay.off_sel;
    uint8_t* const cp = Lg + lay.off_cp;       // staged prefix rows, or (FLY) 256 flank + 256 motif symbols
    uint8_t* const ct = Lg + lay.off_ct;
    short* const b0col = reinterpret_cast<short*>(Lg + lay.off_b0);
    uint8_t* const motifL = FLY ? cp + 256 : Lg + lay.off_b0;   // non-FLY: the motif sits in b0col until that is initialised

    // the item's record was fetched by the caller while the previous chunk was being processed
    int r = 0, nfl = 1, ntr = 0, nfr = 1, m = 1, lo = 0, n = 0;
    long long soff = 0;
    const uint8_t* motif = a.motifs;
    if (act) {
        r = q0.x; nfl = q0.z; ntr = q0.w;
        nfr = q1.x; m = q1.y; lo = q1.z; n = q1.w;
        soff = (long long)(((unsigned long long)(unsigned)q2.y << 32) | (unsigned)q2.x);
        motif += q2.z;
    }
    const int ndb = nfl + ntr + nfr;
    // (k_plan listed the item under this class: band_geometry's search over the classes need not be repeated)
    const BandGeo geo = band_geometry_of_class(BC, nfl, ntr, m, lo, max(n, 1), a.band_tune);
    const int rowsP = act ? nfl + (lo + n - 1) * m : 0;
    const int rowsT = act ? nfr : 0;
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;

    STRK_PHASE(0);
    // ---- stage: selector bytes with pads, row symbols ----
    if (first) misc[0] = 0;
    wave_lds_sync();
    {
        // pads in front of the window and as far behind it as a pass can reach; the window itself a dword per lane, eight
        // dwords in flight (the loop is bound by load latency)
        constexpr unsigned kFront = 0x01010101u * kBandSelFront, kBack = 0x01010101u * kBandSelBack;
        constexpr int ND = lay.sel_len / 4, PD = lay.pad / 4;
        constexpr int kBehind = (lay.wd + kBandRowSlack + G + 28 + 3) / 4;
        static_assert(lay.pad % 4 == 0, "the window starts on a dword of the selector array");
        unsigned other = 0;
        const uint8_t* seq = a.seqs + soff;
        unsigned* const selw = reinterpret_cast<unsigned*>(selb);
        const int nd = act ? (ndb + 3) >> 2 : 0;
        for (int d = lig; d < PD; d += G) selw[d] = kFront;
        for (int d = PD + nd + lig; d < min(ND, PD + nd + kBehind); d += G) selw[d] = kBack;
        STRK_PHASE(7);
        for (int d0 = lig; d0 < nd; d0 += 8 * G) {
            unsigned w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * G;
                unsigned v = 0;
                if (4 * d + 3 < ndb) {
                    __builtin_memcpy(&v, seq + 4 * d, 4);   // unaligned dword load
                } else {
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (4 * d + b < ndb) v |= (unsigned)seq[4 * d + b] << (8 * b);
                }
                w[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * G;
                unsigned out = (w[u] >> 1) & 0x07070707u;
                unsigned bad = (w[u] & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(kBandLetterHi, kBandLetterLo, out);
                if (4 * d + 3 >= ndb) {   // the window's last dword: the bytes behind its end are pads
                    const unsigned vm = 4 * d >= ndb ? 0u : (0xffffffffu >> (8 * (4 * d + 4 - ndb)));
                    out = (out & vm) | (kBack & ~vm);
                    bad &= vm;
                }
                other |= bad;
                if (d < nd) selw[PD + d] = out;
            }
        }
        STRK_PHASE(6);
        nextA();
        if (other) misc[0] = 1;
        for (int k = lig; k < m; k += G) motifL[k] = act ? s_enc[motif[k]] : (uint8_t)kNullSym;
    }
    wave_lds_sync();
    STRK_PHASE(1);
    const bool fallback = act && misc[0] != 0;   // a symbol outside the fixed classes (IUPAC code in a read): exact path decides
    for (int e = lig; e < kTableMax; e += G) { comb[e] = kNegInf; if (LMAX) lmaxA[e] = kNegInf; }
    {
        // row symbols index the row-word table: kBandTblClass0 + class for a flank base (its selector byte), the
        // encoded symbol for a motif base
        if (FLY) {
            // flank rows right-aligned in front of the motif symbols, null rows in front of them (band_pass)
            for (int k = lig; k < 256; k += G) {
                const int row = k - (256 - nfl);
                cp[k] = (uint8_t)(8 * (row >= 0 ? kBandTblClass0 + selb[lay.pad + row] : kNullSym));
            }
            wave_lds_sync();
            for (int k = lig; k < m; k += G) motifL[k] = (uint8_t)(8 * motifL[k]);   // (own entries only)
            wave_lds_sync();
            for (int k = m + lig; k < 256; k += G) motifL[k] = motifL[k % m];       // the motif, cyclically, up to the region's end
        } else {
            // (as far as the longest item of the wave reads: every byte a step can fetch is a valid table offset)
            // Three stretches, a loop each: G - 1 null rows in front, the flank rows (their selector bytes), the motif rows
            // (phase kept per lane), null rows behind.
            const int lenP = wave_max_over_groups(rowsP) + 2 * (G - 1) + 4;
            for (int idx = lig; idx < G - 1; idx += G) cp[idx] = (uint8_t)(8 * kNullSym);
            uint8_t* const cpr = cp + (G - 1);          // cpr[row]
            for (int row = lig; row < min(nfl, rowsP); row += G) cpr[row] = (uint8_t)(8 * (kBandTblClass0 + selb[lay.pad + row]));
            {
                const int gstep = G % m;
                int ph = lig % m;
                for (int row = nfl + lig; row < rowsP; row += G) {
                    cpr[row] = (uint8_t)(8 * motifL[ph]);
                    ph += gstep;
                    if (ph >= m) ph -= m;
                }
            }
            for (int row = rowsP + lig; row < lenP - (G - 1); row += G) cpr[row] = (uint8_t)(8 * kNullSym);
        }
        const int lenT = wave_max_over_groups(rowsT) + 2 * (G - 1) + 4;
        for (int idx = lig; idx < lenT; idx += G) {
            const int row = idx - (G - 1);
            int sym = kNullSym;
            if (row >= 0 && row < rowsT) sym = kBandTblClass0 + selb[lay.pad + ndb - 1 - row];
            ct[idx] = (uint8_t)(8 * sym);
        }
    }
    wave_lds_sync();
    {   // (two columns per store; as many as the widest item of the wave has: ncol = (n - 1) m + band width)
        constexpr unsigned kNeg2 = ((unsigned)(unsigned short)(short)kBandNeg16) * 0x00010001u;
        const int nc2 = min(lay.maxcol, wave_max_over_groups(act ? geo.ncol : 0) + 1) >> 1;
        unsigned* const b2 = reinterpret_cast<unsigned*>(b0col);
        for (int k = lig; k <= nc2; k += G) if (2 * k < lay.maxcol) b2[k] = kNeg2;
    }
    wave_lds_sync();
    nextB();
    STRK_PHASE(2);

    // Long-window classes: a chunk of 12 000 rows is the critical path of a launch that does not fill the chip many times over
    // (BASELINE config 5 at one GPU's share: the longest chunk alone ran 3.0 of the kernel's 3.95 ms, census of the phase-timing
    // build) — and it ran at HALF speed, sharing its SIMD's issue slots with the wave of a short chunk.  Priority goes by rows: the
    // long chunk takes the issue slots it can use (arbitration is by priority, then age), its partner gets the rest; the work is the
    // same, the longest chunk's latency is not.
    if (FLY && !(a.dbg & 32)) {   // (dbg 32: profiling aid, no priorities)
        const int rmax = wave_max_over_groups(rowsP);
        if (rmax >= 11264) __builtin_amdgcn_s_setprio(3);
        else if (rmax >= 8192) __builtin_amdgcn_s_setprio(2);
        else if (rmax >= 5120) __builtin_amdgcn_s_setprio(1);
    }
    BandCtx x;
    x.lig = lig; x.first = first; x.last = last; x.notFirst = first ? 0 : -1; x.notLast = last ? 0 : -1; x.selb = selb;
    asm volatile("" : "+v"(x.notFirst), "+v"(x.notLast));   // plain AND masks: operands of the DPP shifts, not selects
    x.pad = lay.pad; x.maxidx = lay.sel_len - 1; x.ndb = ndb;
    x.flL = cp; x.motifL = motifL; x.nfl = nfl; x.m = m;
    x.flyP = FLY ? ((256 - 4) / m) * m : 0;   // (a pair of steps reads up to three bytes behind the address: kBandFlyMaxMotif)
    const bool run = act && !fallback && geo.ok;
    const int nEff = run ? n : 0;
    // backward pass (reversed right flank x reversed window), then forward pass with the fork rows
    x.tbl = s_tbl + kBandTblBytes;
    band_pass<G, D, true, false, false>(x, ct, (run && !(a.dbg & 2)) ? rowsT : 0, geo.bdlo, dbEnd, cEnd, 0, 0, 1, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    nextC();
    STRK_PHASE(3);
    x.tbl = s_tbl;
    band_pass<G, D, false, FLY, LMAX>(x, cp, (run && !(a.dbg & 1)) ? rowsP : 0, geo.dlo, dbBeg, cBeg, (a.dbg & 8) ? 0 : nEff, nfl + lo * m, m, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    if (FLY) __builtin_amdgcn_s_setprio(0);
    STRK_PHASE(4);
    if (run) {
        for (int k = lig; k < n; k += G) {
            const int R = nfl + (lo + k) * m;
            int sc = max(comb[k], -(1 << 28)) - g * (R + nfr + ndb);
            if (LMAX && cEnd) sc = max(sc, max(lmaxA[k], -(1 << 28)) - g * ndb);   // ends in the last column, in band
            comb[k] = sc;
            a.table[(long long)r * a.table_stride + k] = sc;   // (a band item is the first of its copies: its own table slot)
        }
    }
    // the bounds of all candidates at once, one per lane (the search below runs on one lane and would otherwise
    // evaluate band_ub once per window entry); b0col is free after the forward pass
    int* const ubA = reinterpret_cast<int*>(b0col);
    if (run)
        for (int k = lig; k < n; k += G) ubA[k] = band_ub(geo, nfl, ntr, nfr, m, lo + k, a.end_flags);
    wave_lds_sync();
    // The search is replayed with the certificate from the ESTIMATE only.  (Replaying it from the starts the caller's feedback
    // can turn the estimate into — est +- 1, 2: five lanes, one start each — was tried so that k_replay would never meet an
    // uncertain table: half of the HiFi reads then fail, because a search window that does not hold the best size has a
    // maximum 5 |motif| to 7 |motif| per size below it, under the 2 * len(diagonal) bound of its inexact entries.  Reads that
    // turn out uncertain from their real start are re-scored on the device instead: k_replay's append lists, strk_replay.h.)
    if (act && first) {
        bool certified = (a.dbg & 4) != 0;
        if (run && !(a.dbg & 4)) {
            SeenMask64 seen;
            auto ub = [&](int k) { return ubA[k]; };
            const CertResult cr = search_replay_cert(q2.w, a.step, a.lsr, a.max_iters, a.tie_last, comb, lo, n, seen, ub, a.narrow);
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
                atomicOr(&a.counters[kCntError], kErrList);
            }
            a.exact[r] = 1;
            atomicAdd(&a.counters[kCntBandFallback], 1);
            if (c != kGenericClass) {
                const unsigned long long cc = (unsigned long long)ndb * ((unsigned long long)nfl + (unsigned long long)(lo + n - 1) * m + nfr);
                atomicAdd(a.cells, cc);
                atomicAdd(a.cells + cell_slot_of_list(c), cc);
            }
        }
    }
    wave_lds_sync();
    STRK_PHASE(5);
#ifdef STRK_PHASE_TIMING
    {   // the longest chunk of the launch (ticks / 64), the most rows a chunk had, chunks, and the sum of their longest items' rows
        const int rows_ = wave_max_over_groups(act ? nfl + (lo + n - 1) * m : 0);
        if (lane == 0) {
            atomicMax(&a.counters[45], (int)((tphase - tchunk) >> 6));
            atomicMax(&a.counters[46], rows_);
            atomicAdd(&a.counters[47], 1);
        }
    }
#endif
}

// Two kernels so that the common short classes (8 and 16 lanes per read) are not register-allocated together with the long-
// window classes (32 and 64 lanes).  (Round 2 held k_dp_band to 208 VGPRs so that k_replay ran beside two of its waves per
// SIMD with three calls in flight; with two calls in flight and the free eighth of the CU slots that no longer matters:
// tools/grid_sweep2.sh, a k_replay of 64 VGPRs gains 1 %.)  SET 0: classes 1, 5, 0, 4;  SET 1: classes 3, 7, 2, 6.  Each wave pulls chunks from the set's queue until
// it is empty (chunks of the most expensive class first).
template <int SET>
__device__ __forceinline__ void band_kernel_body(const KArgs& a) {
    constexpr int C0 = SET ? 3 : 1, C1 = SET ? 7 : 5, C2 = SET ? 2 : 0, C3 = SET ? 6 : 4;
    const int n0_ = min(a.counters[kCntClass0 + kBandClass0 + C0], a.list_stride);
    const int n1_ = min(a.counters[kCntClass0 + kBandClass0 + C1], a.list_stride);
    const int n2_ = min(a.counters[kCntClass0 + kBandClass0 + C2], a.list_stride);
    const int n3_ = min(a.counters[kCntClass0 + kBandClass0 + C3], a.list_stride);
    if (n0_ + n1_ + n2_ + n3_ <= 0) return;
    // the two row-word tables sit in front of the per-wave regions
    constexpr int kTblBytes = 2 * kBandTblBytes;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kTblBytes + 4 * kBandWaveLds + kLdsSlack];
    __shared__ uint8_t s_enc[256];
    uint8_t* const s_tbl = lds;
    {
        s_enc[threadIdx.x] = c_enc[threadIdx.x];
        if (threadIdx.x < kTblBytes / 8) {
            const int pass = threadIdx.x / (kBandTblBytes / 8), t = threadIdx.x % (kBandTblBytes / 8);   // 0 forward, 1 backward
            const bool topFree = a.end_flags & (pass ? 2 : 1), leftFree = a.end_flags & (pass ? 8 : 4);
            // row symbol of entry t: the encoded symbol itself, or (flank rows, staged as selectors) that of selector t - 18
            const int e = t < kBandTblClass0 ? t : (t < kBandTblClass0 + 8 ? band_class_symbol(t - kBandTblClass0) : -1);
            unsigned long long word = 0;
            for (int k = 0; k < 8; ++k) {
                const int cs = band_class_symbol(k);
                unsigned b = 0;
                if (t == kNullSym) {
                    // rows in front of the matrix: the row-0 pattern G(0, j) = g j moves one column per row
                    b = (cs >= 0 && topFree) ? kGap : 0;
                } else if (e >= 0 && e < kNSym) {
                    if (cs >= 0) b = (unsigned)(c_mat[e][cs] + kWBias) & 0xffu;
                    else b = (leftFree && k == (pass ? kBandSelBack : kBandSelFront)) ? kGap : 0;
                }
                word |= (unsigned long long)b << (8 * k);
            }
            reinterpret_cast<unsigned long long*>(s_tbl)[threadIdx.x] = word;
        }
    }
    __syncthreads();
    uint8_t* const Lw = lds + kTblBytes + (threadIdx.x >> 6) * kBandWaveLds;
    constexpr int G01 = band_class_G(C0), G23 = band_class_G(C2);   // (C1 / C3: the same lane counts, 12 diagonals)
    constexpr int per01 = 64 / G01, per23 = 64 / G23;   // reads per wave
    const int ch0 = (n0_ + per01 - 1) / per01, ch1 = (n1_ + per01 - 1) / per01, ch2 = (n2_ + per23 - 1) / per23, ch3 = (n3_ + per23 - 1) / per23;
    const int e0 = ch0, e1 = e0 + ch1, e2 = e1 + ch2, e3 = e2 + ch3;   // chunk index ranges of the four classes
    const int lane = threadIdx.x & 63;
    auto pop = [&]() {
        int c = 0;
        if (lane == 0) c = atomicAdd(&a.counters[kCntNextBand + SET], 1);
        return __builtin_amdgcn_readfirstlane(c);
    };
    // record of this lane's item in chunk c (chunks of the wider classes come first)
    auto fetch = [&](int c, bool& act, int4& q0, int4& q1, int4& q2) -> int {
        int cls = C0, it = 0, cnt = 0, G_ = G01;
        if (c < e0) { cls = C0; it = c * per01 + lane / G01; cnt = n0_; }
        else if (c < e1) { cls = C1; it = (c - e0) * per01 + lane / G01; cnt = n1_; }
        else if (c < e2) { cls = C2; it = (c - e1) * per23 + lane / G23; cnt = n2_; G_ = G23; }
        else if (c < e3) { cls = C3; it = (c - e2) * per23 + lane / G23; cnt = n3_; G_ = G23; }
        act = it < cnt;
        if (act) {
            const int4* rec = (SET ? a.band_recs_w : a.band_recs) + ((size_t)cls * a.list_stride + it) * 3;
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
        }
        return G_;   // lanes per item of that chunk
    };
    // Touch the window bytes of an item one chunk early (one byte per 64-byte line, a line per lane of the group), so
    // that the staging loop of its chunk finds them in cache.  The loaded byte itself is never used.
    auto touch = [&](bool act_, long long so, int ndb_, int G_) -> unsigned {
        unsigned sink = 0;
        if (act_) {
            const uint8_t* pp = a.seqs + so + min((lane & (G_ - 1)) * 64, max(ndb_ - 1, 0));
            asm volatile("global_load_ubyte %0, %1, off" : "=v"(sink) : "v"(pp) : "memory");
        }
        return sink;
    };
    int c = pop();
    bool act = false;
    int4 q0 = make_int4(0, 0, 0, 0), q1 = q0, q2 = q0;
    (void)fetch(c, act, q0, q1, q2);
    while (c < e3) {
        __builtin_amdgcn_wave_barrier();   // the wave enters every iteration whole (see k_realign_dp)
        // the next chunk is taken, its records are fetched and its window bytes touched while this one is being processed.
        // Of the records only what the touch needs stays in registers across the two passes (three values, not twelve: the
        // passes are where the kernel's register allocation peaks); the records themselves are fetched again, from cache,
        // when this chunk is done.
        int cn_raw = 0, cn = 0, g_n = 8, ndb_n = 0;
        long long so_n = 0;
        bool act_n = false;
        unsigned sink = 0;
        auto nextA = [&]() { if (lane == 0) cn_raw = atomicAdd(&a.counters[kCntNextBand + SET], 1); };
        auto nextB = [&]() {
            int4 n0 = make_int4(0, 0, 0, 0), n1 = n0, n2 = n0;
            cn = __builtin_amdgcn_readfirstlane(cn_raw);
            g_n = fetch(cn, act_n, n0, n1, n2);
            so_n = (long long)(((unsigned long long)(unsigned)n2.y << 32) | (unsigned)n2.x);
            ndb_n = n0.z + n0.w + n1.x;
        };
        auto nextC = [&]() { sink = touch(act_n, so_n, ndb_n, g_n); };
        if (c < e0) band_wave<C0>(a, act, q0, q1, q2, Lw, s_enc, s_tbl, nextA, nextB, nextC);
        else if (c < e1) band_wave<C1>(a, act, q0, q1, q2, Lw, s_enc, s_tbl, nextA, nextB, nextC);
        else if (c < e2) band_wave<C2>(a, act, q0, q1, q2, Lw, s_enc, s_tbl, nextA, nextB, nextC);
        else band_wave<C3>(a, act, q0, q1, q2, Lw, s_enc, s_tbl, nextA, nextB, nextC);
        c = cn;
        (void)fetch(c, act, q0, q1, q2);
        // the touch load's destination register stays reserved until the load has certainly landed
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(sink) : "memory");
    }
}
// two blocks (eight waves) per CU: the register allocation must stay within 256 VGPRs
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_dp_band(KArgs a) { band_kernel_body<0>(a); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_dp_band_wide(KArgs a) { band_kernel_body<1>(a); }

}  // namespace strk
