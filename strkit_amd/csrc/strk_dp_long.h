// strk_dp_long.h — column-tiled exact kernel for windows wider than the largest fast class, and the generic kernel
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

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
    const int32_t* list = a.long_sorted ? a.long_sorted : a.cls_list + (size_t)kLongClass * a.list_stride * 2;

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
            if (first) {   // the host makes the slots as large as the largest request and runs the call again (grow_scratch)
                atomicOr(&a.counters[kCntError], kErrLongSlot);
                atomicMax(&a.counters[kCntLongNeed], (int)min(need, 0x7fffffffll));
            }
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
                    atomicOr(&a.counters[kCntError], kErrList);
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
            const SearchResult res = search_replay(a.est_cn[r], a.step, a.lsr, a.max_iters, a.tie_last, comb, lo, n, seen, a.narrow);
            a.spec[r] = make_int4(res.cn, res.score, res.n_explored, (res.miss ? kSpecMiss : 0) | (res.empty ? kSpecEmpty : 0));
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: one WAVE per (item, candidate); plain row-by-row DP with the H row in global scratch, lane l of the
// wave owning the columns 1 + l, 65 + l, ... .  Takes every shape the fast classes do not (empty flanks, > 8 distinct
// symbols in the read window, motifs longer than kMotifMax).  Correctness path, not a fast path — but a batch of such reads
// with starts far from their tracts' sizes ran for minutes with one THREAD per candidate (tools/fuzz_parity.py), hence the
// wave: within a row, H(j) = max(h'(j), H(j-1) - g) with h'(j) = max(diag + w, up - g) is a running maximum of
// U(j) = H(j) + g j = max(h'(j) + g j, U(j-1)) — 64 columns at a time by an inclusive prefix maximum over the lanes, the carry
// U of the block before handed on.  Every scratch word is read and written by one lane only (diag comes by shuffle).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dp_generic(KArgs a) {
    const int count = min(a.counters[kCntClass0 + kGenericClass], a.list_stride);
    if (count <= 0) return;
    const int32_t* list = a.cls_list + (size_t)kGenericClass * a.list_stride * 2;
    const bool dbBeg = a.end_flags & 1, dbEnd = a.end_flags & 2, cBeg = a.end_flags & 4, cEnd = a.end_flags & 8;
    constexpr int g = kGap;
    const int lane = threadIdx.x & 63;
    const long long total = (long long)count * kTableMax;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    // a wave keeps the row it has and takes a new one only when the next window is longer: the pool then holds at most one
    // row (of the longest window) per wave however many items a call brings, and the host grows it when even that does not
    // fit (scratch_used counts what was ASKED for, served or not: strk_api.hip, grow_scratch)
    unsigned long long myAt = 0, myCap = 0;
    for (long long w = wave0; w < total; w += n_waves) {
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
            if (lane == 0) {
                out[0] = 0;
                if (a.ref_mode) out[1] = -1;
            }
            continue;
        }
        const unsigned long long need = (unsigned long long)ndb + 1;
        if (need > myCap) {
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(a.scratch_used, need);
            at = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at & 0xffffffffull)) |
                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32);
            myAt = (unsigned long long)a.long_slot * a.long_waves + at;
            myCap = myAt + need > (unsigned long long)a.scratch_cap ? 0 : need;
        }
        if (need > myCap) {
            if (lane == 0) {
                atomicOr(&a.counters[kCntError], kErrScratch);
                out[0] = 0;
                if (a.ref_mode) out[1] = -1;
            }
            continue;
        }
        int32_t* Hrow = a.scratch + myAt;          // Hrow[j], j = 1..ndb (column 0 lives in registers)
        const int n_blocks = (ndb + 63) >> 6;
        for (int j = 1 + lane; j <= ndb; j += 64) Hrow[j] = dbBeg ? 0 : -g * j;
        int lastcol = kNegInf;
        int h0_prev = 0;                           // H(rr - 1, 0)
        for (long long rr = 1; rr <= ncand; ++rr) {
            const long long p = rr - 1;
            const uint8_t ch = p < nfl ? db[p] : (p < nfl + (long long)i * m ? motif[(p - nfl) % m] : db[nfl + ntr + (p - nfl - (long long)i * m)]);
            const int8_t* wrow = c_mat[c_enc[ch]];
            const int h0 = cBeg ? 0 : (int)(-g * rr);
            int carryU = h0;                       // U(0) = H(rr, 0) + g * 0
            int diagCarry = h0_prev;
            for (int jb = 0; jb < n_blocks; ++jb) {
                const int j = 1 + (jb << 6) + lane;
                const bool valid = j <= ndb;
                const int up = valid ? Hrow[j] : kNegInf;
                int d = __shfl_up(up, 1);
                if (lane == 0) d = diagCarry;
                diagCarry = __builtin_amdgcn_readlane(up, 63);
                int t = kNegInf;
                if (valid) t = max(d + (int)wrow[c_enc[db[j - 1]]], up - g) + g * j;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(t, off);
                    if (lane >= off) t = max(t, v);
                }
                const int U = max(t, carryU);
                if (valid) Hrow[j] = U - g * j;
                carryU = __builtin_amdgcn_readlane(U, 63);   // (lanes behind the last column repeat its U)
            }
            lastcol = max(lastcol, carryU - g * ndb);         // H(rr, ndb)
            h0_prev = h0;
        }
        // what the sequential scan of the last row returns: the first column that holds the maximum (see the thread-per-
        // candidate form in the history: `H[j] > best || (H[j] == best && j < bestj && !cEnd)` for j = 1..ndb, from best = H[ndb])
        int hmax = kNegInf;
        for (int j = 1 + lane; j <= ndb; j += 64) hmax = max(hmax, Hrow[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hmax = max(hmax, __shfl_xor(hmax, off));
        int jmin = 0x7fffffff;
        for (int j = 1 + lane; j <= ndb; j += 64)
            if (Hrow[j] == hmax) { jmin = j; break; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) jmin = min(jmin, __shfl_xor(jmin, off));
        int hlast = ((ndb - 1) & 63) == lane ? Hrow[ndb] : kNegInf;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hlast = max(hlast, __shfl_xor(hlast, off));
        int best = hlast, bestj = ndb;
        if (cEnd) best = max(best, lastcol);
        if (dbEnd && (hmax > best || (hmax == best && !cEnd))) { best = hmax; bestj = jmin; }
        if (lane == 0) {
            out[0] = best;
            if (a.ref_mode) out[1] = bestj - 1;
        }
    }
}

}  // namespace strk
