// strk_dp_band.h — diagonal-owned banded first pass k_dp_band / k_dp_band_wide with in-kernel certificate
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

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
    static constexpr int OFF_COMB = 0, OFF_MISC = OFF_COMB + kTableMax * 4, OFF_LMAX = OFF_MISC + 16;
    static constexpr int kBandHiPad = kBandRowSlack + 32;
    __host__ __device__ constexpr BandLayout(int c)   // c = band class
        : wd(128 << c), pad((128 << c) + (8 << c) + 8), maxdb(band_max_db(c)), maxcol(band_max_col(c)),
          off_sel(OFF_LMAX + (band_class_lmax(c) ? kTableMax * 4 : 0)),
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
// Fixed symbol classes of the band kernels: what an alignment file's reads hold after wildcarding (call_locus.py:79):
// A C G T, N, the wildcard X and '*' (any byte outside the alphabet).  v_perm selector = class index; a read with another
// symbol (an IUPAC code) takes the exact kernels, whose classes are per read.  Row-word table: entries 0..17 by encoded
// symbol (16 = '*', 17 = no row), then one entry per class for flank rows, which are staged as selector bytes.
constexpr int kBandNClass = 7;
constexpr int kBandTblClass0 = 18;
__host__ __device__ constexpr int band_class_symbol(int c) { return c < 4 ? c : (c == 4 ? 15 /* X */ : (c == 5 ? 14 /* N */ : kStar)); }

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
template <int G, bool BWD, bool FLY, bool LMAX>
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
    // classes of band_class_lmax: running maximum of the last column over the in-band rows.  Right of column |db|
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
#define STRK_BAND_STEP(SRC, DST, TT, EDGE)                                                         \
    {                                                                                              \
        const uint2 word = wordNext;                                                               \
        wordNext = x.tbl[symNext];                                                                 \
        symNext = FLY ? next_sym() : (unsigned)pa[(TT) + 2];                                       \
        const unsigned nb = nbNext;                                                                \
        nbNext = x.selb[col_addr(jb + 17)];                                                        \
        /* lane 0 works on row TT+1; left of the band lies the left boundary while j-1 <= 0 */      \
        const int keepL = ((EDGE) && (TT) + dlo_ <= 0 && leftFree) ? g * ((TT) + 1) : 0;            \
        const int leftEdge = from_left<G>(keepL, houtL, x.first);                                  \
        /* the last lane works on row TT-G+2; above row 1 lies the row-0 pattern, else -inf */      \
        const int rl = (TT) - G + 2;                                                               \
        const int keepU = ((EDGE) && rl <= 1) ? g0(rl + dhi_) : 0;                                 \
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
        if (LMAX && !BWD) lastmax = (gr + g >= grFirst) ? max(lastmax, houtL - (gr + g)) : lastmax; \
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
                if (LMAX && x.last) lmaxA[forkIdx] = lastmax;                                      \
                ++forkIdx;                                                                         \
                forkG = forkIdx < nEff ? forkG + gm : 0x7fffffff;                                  \
            }                                                                                      \
        }                                                                                          \
        ++jb;                                                                                      \
    }
    // The band touches the left boundary column while t <= -dlo and the row-0 pattern while t <= G - 1: after that
    // (for every group of the wave) both edge values are the constant 0 and the steps need not compute them.
    const int tEdge = min(T, (wave_max_over_groups(max(1 - dlo_, G)) + 1) & ~1);
    int t = 0;
    for (; t < tEdge; t += 2) {
        STRK_BAND_STEP(Ha, Hb, t, true)
        STRK_BAND_STEP(Hb, Ha, t + 1, true)
    }
    for (; t < T; t += 2) {
        STRK_BAND_STEP(Ha, Hb, t, false)
        STRK_BAND_STEP(Hb, Ha, t + 1, false)
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
__device__ __forceinline__ void band_wave(const KArgs& a, bool act, int4 q0, int4 q1, int4 q2, uint8_t* Lw, const uint8_t* s_enc,
                                          const uint8_t* s_sel, const uint2* s_tbl) {
    constexpr int g = kGap, G = 8 << BC;
    constexpr bool FLY = band_class_fly(BC);
    constexpr bool LMAX = band_class_lmax(BC);
    constexpr BandLayout lay(BC);
    const int lane = threadIdx.x & 63;
#ifdef STRK_PHASE_TIMING
    unsigned long long tphase = __builtin_readcyclecounter();
#endif
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const bool first = lig == 0, last = lig == G - 1;
    uint8_t* const Lg = Lw + grp * lay.group_bytes;
    const uint2* const tbl = s_tbl;   // row words: the same for every read (fixed symbol classes, band_kernel_body)
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
    const BandGeo geo = band_geometry(nfl, ntr, nfr, m, lo, max(n, 1));
    const int rowsP = act ? nfl + (lo + n - 1) * m : 0;
    const int rowsT = act ? nfr : 0;
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;

    STRK_PHASE(0);
    // ---- stage: selector bytes with pads (window byte -> fixed symbol class, one LDS look-up per byte), row symbols ----
    if (first) misc[0] = 0;
    wave_lds_sync();
    {
        // a dword per lane and eight dwords in flight (the loop is bound by load latency); slots outside the window
        // get selector 0x0c (constant 0: inert in G-space); a byte outside the fixed classes sets bit 7
        unsigned other = 0;
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
                unsigned out = 0x0c0c0c0cu;
                if (ok[u] == 0xfu) {
                    out = (unsigned)s_sel[w[u] & 0xffu] | ((unsigned)s_sel[(w[u] >> 8) & 0xffu] << 8) |
                          ((unsigned)s_sel[(w[u] >> 16) & 0xffu] << 16) | ((unsigned)s_sel[w[u] >> 24] << 24);
                } else if (ok[u]) {
                    out = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        out |= (((ok[u] >> b) & 1u) ? (unsigned)s_sel[(w[u] >> (8 * b)) & 0xffu] : 0x0cu) << (8 * b);
                }
                other |= out;
                if (d < ND) selw[d] = out;
            }
        }
        STRK_PHASE(6);
        if (other & 0x80808080u) misc[0] = 1;
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
            for (int k = lig; k < 256; k += G) cp[k] = (uint8_t)(k < nfl ? kBandTblClass0 + selb[lay.pad + k] : kNullSym);
        } else {
            const int lenP = rowsP + 2 * (G - 1) + 4;
            const int gstep = G % m;
            int ph = (lig - (G - 1) - nfl) % m;
            if (ph < 0) ph += m;
            for (int idx = lig; idx < lenP; idx += G) {
                const int row = idx - (G - 1);
                int sym = kNullSym;
                if (row >= 0 && row < rowsP) sym = row < nfl ? kBandTblClass0 + selb[lay.pad + row] : motifL[ph];
                cp[idx] = (uint8_t)sym;
                ph += gstep;
                if (ph >= m) ph -= m;
            }
        }
        const int lenT = rowsT + 2 * (G - 1) + 4;
        for (int idx = lig; idx < lenT; idx += G) {
            const int row = idx - (G - 1);
            int sym = kNullSym;
            if (row >= 0 && row < rowsT) sym = kBandTblClass0 + selb[lay.pad + ndb - 1 - row];
            ct[idx] = (uint8_t)sym;
        }
    }
    wave_lds_sync();
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
    band_pass<G, true, false, false>(x, ct, (run && !(a.dbg & 2)) ? rowsT : 0, geo.bdlo, dbEnd, cEnd, 0, 0, 1, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    STRK_PHASE(3);
    band_pass<G, false, FLY, LMAX>(x, cp, (run && !(a.dbg & 1)) ? rowsP : 0, geo.dlo, dbBeg, cBeg, (a.dbg & 8) ? 0 : nEff, nfl + lo * m, m, geo.cmin, geo.ncol, comb, b0col, lmaxA);
    wave_lds_sync();
    STRK_PHASE(4);
    if (run) {
        for (int k = lig; k < n; k += G) {
            const int R = nfl + (lo + k) * m;
            int sc = max(comb[k], -(1 << 28)) - g * (R + nfr + ndb);
            if (LMAX && cEnd) sc = max(sc, max(lmaxA[k], -(1 << 28)) - g * ndb);   // ends in the last column, in band
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
        bool certified = (a.dbg & 4) != 0;
        if (run && !(a.dbg & 4)) {
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
    // the row-word table sits in front of the per-wave regions: a stale row symbol (any byte) indexes at most 255 * 8 bytes
    // past its start, which is still inside this array
    constexpr int kTblBytes = 256;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kTblBytes + 4 * kBandWaveLds + kLdsSlack];
    __shared__ uint8_t s_enc[256];
    __shared__ uint8_t s_sel[256];
    uint2* const s_tbl = reinterpret_cast<uint2*>(lds);
    {
        const unsigned sym = c_enc[threadIdx.x];
        s_enc[threadIdx.x] = (uint8_t)sym;
        unsigned cls = 0x80u;   // not one of the fixed classes
        for (int c = 0; c < kBandNClass; ++c) if ((int)sym == band_class_symbol(c)) cls = (unsigned)c;
        s_sel[threadIdx.x] = (uint8_t)cls;
        if (threadIdx.x < kTblBytes / 8) {
            const int e = threadIdx.x < kBandTblClass0 ? (int)threadIdx.x : band_class_symbol(min((int)threadIdx.x - kBandTblClass0, kBandNClass - 1));
            unsigned wlo = 0, whi = 0;
            if (e < kNSym && (int)threadIdx.x < kBandTblClass0 + kBandNClass) {
                for (int k = 0; k < kBandNClass; ++k) {
                    const unsigned b = (unsigned)(c_mat[e][band_class_symbol(k)] + kWBias) & 0xffu;
                    if (k < 4) wlo |= b << (8 * k); else whi |= b << (8 * (k - 4));
                }
            }
            s_tbl[threadIdx.x] = make_uint2(wlo, whi);
        }
    }
    __syncthreads();
    uint8_t* const Lw = lds + kTblBytes + (threadIdx.x >> 6) * kBandWaveLds;
    constexpr int perA = 64 / (8 << CA), perB = 64 / (8 << CB);   // reads per wave
    const int chA = (nA + perA - 1) / perA, chB = (nB + perB - 1) / perB;
    const int lane = threadIdx.x & 63;
    auto pop = [&]() {
        int c = 0;
        if (lane == 0) c = atomicAdd(&a.counters[kCntNextBand + SET], 1);
        return __builtin_amdgcn_readfirstlane(c);
    };
    // record of this lane's item in chunk c (chunks of the wider class come first)
    auto fetch = [&](int c, bool& act, int4& q0, int4& q1, int4& q2) -> int {
        int cls = CA, it = 0, cnt = 0;
        if (c < chA) { cls = CA; it = c * perA + lane / (8 << CA); cnt = nA; }
        else if (c - chA < chB) { cls = CB; it = (c - chA) * perB + lane / (8 << CB); cnt = nB; }
        act = it < cnt;
        if (act) {
            const int4* rec = a.band_recs + ((size_t)cls * a.list_stride + it) * 3;
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
        }
        return 8 << cls;   // lanes per item of that chunk
    };
    // Touch the window bytes of an item one chunk early (one byte per 64-byte line, a line per lane of the group), so
    // that the staging loop of its chunk finds them in cache.  The loaded byte itself is never used.
    auto touch = [&](bool act_, const int4& q0_, const int4& q1_, const int4& q2_, int G_) -> unsigned {
        unsigned sink = 0;
        if (act_) {
            const long long so = (long long)(((unsigned long long)(unsigned)q2_.y << 32) | (unsigned)q2_.x);
            const int ndb_ = q0_.z + q0_.w + q1_.x;
            const uint8_t* pp = a.seqs + so + min((lane & (G_ - 1)) * 64, max(ndb_ - 1, 0));
            asm volatile("global_load_ubyte %0, %1, off" : "=v"(sink) : "v"(pp) : "memory");
        }
        return sink;
    };
    int c = pop();
    bool act = false;
    int4 q0 = make_int4(0, 0, 0, 0), q1 = q0, q2 = q0;
    (void)fetch(c, act, q0, q1, q2);
    while (c < chA + chB) {
        __builtin_amdgcn_wave_barrier();   // the wave enters every iteration whole (see k_realign_dp)
        // take the next chunk now and start loading its records: they arrive while this chunk is being processed
        const int cn = pop();
        bool act_n = false;
        int4 n0 = make_int4(0, 0, 0, 0), n1 = n0, n2 = n0;
        const int g_n = fetch(cn, act_n, n0, n1, n2);
        const unsigned sink = touch(act_n, n0, n1, n2, g_n);
        if (c < chA) band_wave<CA>(a, act, q0, q1, q2, Lw, s_enc, s_sel, s_tbl);
        else band_wave<CB>(a, act, q0, q1, q2, Lw, s_enc, s_sel, s_tbl);
        // the touch load's destination register stays reserved until the load has certainly landed
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(sink) : "memory");
        c = cn; act = act_n; q0 = n0; q1 = n1; q2 = n2;
    }
}
// two blocks (eight waves) per CU: the register allocation must stay within 256 VGPRs
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_dp_band(KArgs a) { band_kernel_body<0>(a); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_dp_band_wide(KArgs a) { band_kernel_body<1>(a); }

}  // namespace strk
