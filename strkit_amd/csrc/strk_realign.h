// strk_realign.h — read realignment on gfx950: parasail sg_dx_trace (affine gaps, trace-back, CIGAR).
//
// Replaces the parasail call of strkit/call/realign.py:56-63 (sg_dx_trace_scan_16(ref window, read,
// open 7, extend 0, dna_matrix)) and the CIGAR walk behind pr.cigar.seq (realign.py:71).  s1 = the
// reference window (aligned end to end), s2 = the wildcarded read (both ends free).
//
// k_realign_dp: one wave per (s1, s2) pair.  Lane l owns CL consecutive s1 positions ("columns"); the
// wave streams over the s2 positions ("rows"), lane l working on row t - l at step t (anti-diagonal
// skew), so the only traffic between lanes is one DPP shift per step of the lane's last column.
// s1 is RIGHT-aligned in the 64*CL columns: the pad columns on the left score 0 against everything
// and therefore hold H = 0 on every row, which is exactly parasail's boundary column — the last real
// column is always the last column of lane 63, where the running last-row maximum is tracked.
// Windows longer than 64*32 are processed as column tiles; the tile's right edge (H and the s1-gap
// state per row) goes through a scratch array to the next tile.
//
// Values are 16*score + tag.  The tag bits order equal scores the way the trace-back needs
// (bits 3:2: 2 = diagonal, 1 / 0 = the two gap kinds; bit 0 / bit 1: the gap was extended), so a
// plain integer max makes both the decision and its record; one 4-bit trace entry per cell is
// written to HBM, [tile][step][lane] with CL/2 bytes per entry (coalesced per step).
//
// k_realign_trace: one thread per pair walks the trace from the end cell and writes the CIGAR
// (BAM encoding), leading free s2 bases as one D run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "strk_kernels.h"

namespace strk {

constexpr int kRealignMaxCL = 32;
constexpr int kRealignPadSym = 17;     // pad column: scores 0 against every row symbol
constexpr int kRealignRowStride = 32;  // bytes per row-symbol row of the LDS score table
constexpr int kRealignNeg = -(1 << 28);
constexpr int kRealignStrip = 1024;    // row symbols staged in LDS per wave
constexpr int kTagDiag = 8;            // bits 3:2 = 2
constexpr int kTagMask = 12;
constexpr int kFlagGiExt = 1, kFlagGdExt = 2;

struct RealignPair {
    int64_t s1_off, s2_off;      // into the raw base arrays
    int64_t trace_off;           // bytes into the trace workspace
    int64_t edge_off;            // ints into the edge scratch (2 * 2 * n2 ints, tiles ping-pong), -1: single tile
    int64_t cig_off;             // uint32 units into the CIGAR buffer
    int32_t n1, n2;
    int32_t cl;                  // columns per lane: 4, 8, 16, 32
    int32_t ntiles;
    int32_t pad;                 // pad columns on the left of tile 0
    int32_t cig_cap;
    int32_t orig;                // caller's pair index
    int32_t reserved;
};

struct RealignArgs {
    const RealignPair* pairs;    // sorted by decreasing work
    int32_t n_pairs;
    int32_t open, ext;           // gap of length k costs open + (k-1)*ext
    int32_t gap_pref;            // 0: on equal scores the s1-consuming gap ('I') beats the s2-consuming gap ('D'); 1: reverse
    const uint8_t* s1;
    const uint8_t* s2;
    uint8_t* trace;
    int32_t* edge;
    int32_t* score;              // [n_pairs], caller order
    int32_t* end2;
    int32_t* n_cigar;            // runs written, or -1 if cig_cap was too small
    uint32_t* cigar;
    int32_t* queue;              // work-queue counters, one per launch (zeroed by the host)
    unsigned long long* cells;   // DP cell updates (statistics)
    volatile int32_t* dbg;       // debug progress markers (host-pinned) or nullptr
};

template <int CL>
__device__ __forceinline__ void realign_store(uint8_t* p, const unsigned (&w)[(CL + 7) / 8]) {
    if (CL == 4) *reinterpret_cast<uint16_t*>(p) = (uint16_t)(w[0] >> 16);
    else if (CL == 8) *reinterpret_cast<unsigned*>(p) = w[0];
    else if (CL == 16) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
    else *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// One column tile of one pair.  Returns (in lane 63) the running maximum of the tile's last column.
template <int CL, bool EXT0>
__device__ __forceinline__ void realign_tile(const RealignArgs& a, const RealignPair& pr, int tile, const int8_t* tab,
                                             const uint8_t* s_enc, uint8_t* s2buf, int& best, int& bestj) {
    constexpr int NW = (CL + 7) / 8;
    const int lane = threadIdx.x & 63;
    const int n2 = pr.n2, pad = pr.pad;
    const int o16 = a.open * 16, e16 = a.ext * 16;
    const int tagGi = a.gap_pref ? 0 : 4, tagGd = a.gap_pref ? 4 : 0;
    const int kOpenGi = tagGi - o16, kOpenGd = tagGd - o16;
    const uint8_t* s1 = a.s1 + pr.s1_off;
    const uint8_t* s2 = a.s2 + pr.s2_off;
    const bool last_tile = tile == pr.ntiles - 1;
    const int ci0 = (tile * 64 + lane) * CL;   // first padded column of this lane

    // column symbols (packed 4 per dword); pad columns get the pad symbol
    unsigned apk[(CL + 3) / 4];
#pragma unroll
    for (int q = 0; q < (CL + 3) / 4; ++q) {
        unsigned v = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ci0 + q * 4 + u - pad;
            const unsigned real = (unsigned)s_enc[s1[max(i, 0)]];
            v |= (i >= 0 ? real : (unsigned)kRealignPadSym) << (8 * u);
        }
        apk[q] = v;
    }
    auto bnd = [&](int ci) { const int i = ci - pad; return i >= 0 ? -16 * (a.open + i * a.ext) : 0; };

    int Hrow[CL], Gd[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) { Hrow[c] = bnd(ci0 + c); Gd[c] = kRealignNeg + tagGd; }
    int Hl_prev = bnd(ci0 - 1);
    int lastH = 0, lastE = kRealignNeg + tagGi;   // this lane's last column of the previous step (garbage until the lane starts)
    int symrow = 16 * kRealignRowStride;

    // left edge of lane 0: boundary (tile 0) or the previous tile's right edge from scratch
    const int32_t* edge_in = nullptr;
    int32_t* edge_out = nullptr;
    if (pr.edge_off >= 0) {
        int32_t* base = a.edge + pr.edge_off;
        if (tile > 0) edge_in = base + (size_t)((tile - 1) & 1) * 2 * n2;
        if (!last_tile) edge_out = base + (size_t)(tile & 1) * 2 * n2;
    }
    uint8_t* trace = a.trace + pr.trace_off + (size_t)tile * (size_t)(n2 + 63) * 64 * (CL / 2);

    const int steps = n2 + 63;
    // Row symbols: 1 024 at a time are encoded into the wave's LDS strip (so that the step loop never waits on
    // the memory counter its trace stores also use); each block of 64 steps takes its 64 symbols from there,
    // one per lane.  Left-edge values of a later column tile come straight from the scratch array.
    auto refill = [&](int t0) {
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < kRealignStrip / 64; ++u) {
            const int jj = t0 + u * 64 + lane;
            s2buf[u * 64 + lane] = jj < n2 ? s_enc[s2[jj]] : (uint8_t)16;
        }
        wave_lds_sync();
    };
    auto load_chunk = [&](int t0, int& c_sym, int& c_h, int& c_e) {
        c_sym = (int)s2buf[(t0 & (kRealignStrip - 1)) + lane] * kRealignRowStride;
        c_h = 0;
        c_e = kRealignNeg + tagGi;
        if (edge_in) {
            const int jj = t0 + lane;
            if (jj < n2) {
                c_h = edge_in[2 * jj];
                c_e = edge_in[2 * jj + 1];
            }
        }
    };
    int chunk_sym = 0, chunk_h = 0, chunk_e = 0;
    uint8_t* tp = trace + (size_t)lane * (CL / 2);

    auto step = [&](auto ramp_tag, int t) {
        constexpr bool RAMP = decltype(ramp_tag)::value;
        // cross-lane traffic first, with all 64 lanes (lanes that have not started yet still pass the row symbol on)
        const int s_sym = __builtin_amdgcn_readlane(chunk_sym, t & 63);
        const int s_h = __builtin_amdgcn_readlane(chunk_h, t & 63);
        const int s_e = __builtin_amdgcn_readlane(chunk_e, t & 63);
        symrow = __builtin_amdgcn_update_dpp(s_sym, symrow, kDppWaveShr1, 0xf, 0xf, false);
        const int Hl = __builtin_amdgcn_update_dpp(s_h, lastH, kDppWaveShr1, 0xf, 0xf, false);
        const int El = __builtin_amdgcn_update_dpp(s_e, lastE, kDppWaveShr1, 0xf, 0xf, false);
        if (!RAMP || t >= lane) {   // first 63 steps: lane l starts at step l and keeps its initial state until then
            int w8[CL];
#pragma unroll
            for (int c = 0; c < CL; ++c) w8[c] = tab[symrow + (int)((apk[c / 4] >> (8 * (c % 4))) & 0xffu)];

            unsigned word[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) word[q] = 0;
            int diag = Hl_prev, upH = Hl, upE = El;
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                const int Dp = diag + w8[c];
                const int Ei = max(EXT0 ? (upE | kFlagGiExt) : ((upE - e16) | kFlagGiExt), upH + kOpenGi);
                const int Fj = max(EXT0 ? (Gd[c] | kFlagGdExt) : ((Gd[c] - e16) | kFlagGdExt), Hrow[c] + kOpenGd);
                const int Hp = max(max(Dp, Ei), Fj);
                const int Hc = Hp & ~15;
                const unsigned x = ((unsigned)Ei & (unsigned)kFlagGiExt) | (unsigned)Fj;   // bits 1:0 = the two flags
                unsigned nib;   // (Hp & 12) | (x & ~12); whatever lies above bit 3 is shifted out by the alignbit
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(nib) : "v"(kTagMask), "v"(Hp), "v"(x));
                word[c / 8] = __builtin_amdgcn_alignbit(nib, word[c / 8], 4);
                diag = Hrow[c];
                Hrow[c] = Hc;
                Gd[c] = Fj;
                upH = Hc;
                upE = Ei;
            }
            Hl_prev = Hl;
            lastH = upH;
            lastE = upE;
            realign_store<CL>(tp, word);
            // lane 63 holds row t - 63: a real row from t = 63 on (and t < n2 + 63 always)
            if (lane == 63) {
                if (last_tile) {
                    if (upH > best) { best = upH; bestj = t - 63; }
                } else {
                    edge_out[2 * (t - 63)] = upH;
                    edge_out[2 * (t - 63) + 1] = upE;
                }
            }
        }
        tp += 64 * (CL / 2);
    };

    for (int t0 = 0; t0 < steps; t0 += 64) {
        if ((t0 & (kRealignStrip - 1)) == 0) {
            if (a.dbg && lane == 0) a.dbg[(threadIdx.x >> 6) * 8 + 4] = t0;
            refill(t0);
        }
        load_chunk(t0, chunk_sym, chunk_h, chunk_e);
        const int tend = min(t0 + 64, steps);
        if (t0 == 0) {
            for (int t = 0; t < tend; ++t) step(std::true_type{}, t);
        } else {
            int t = t0;   // two steps per trip: the row registers swap roles instead of being copied
            for (; t + 1 < tend; t += 2) {
                step(std::false_type{}, t);
                step(std::false_type{}, t + 1);
            }
            if (t < tend) step(std::false_type{}, t);
        }
    }
}

// One launch per column class (CL) present in the chunk: the host sorts the pairs by class and gives every
// launch its slice [first, first + count) of the pair array and its own queue counter.
template <int CL, bool EXT0>
__global__ void __launch_bounds__(256) k_realign_dp(RealignArgs a, int first, int count, int qslot) {
    __shared__ int8_t tab[(kRealignPadSym + 1) * kRealignRowStride];   // 16*W + 8 per (row symbol, column symbol)
    __shared__ uint8_t s_enc[256];
    __shared__ uint8_t s2strip[4 * kRealignStrip];
    s_enc[threadIdx.x] = c_enc[threadIdx.x];
    for (int i = threadIdx.x; i < (kRealignPadSym + 1) * kRealignRowStride; i += 256) {
        const int r = i / kRealignRowStride, c = i % kRealignRowStride;
        int w = 0;
        if (r < kNSym && c < kNSym) w = c_mat[c][r];
        tab[i] = (int8_t)(16 * w + kTagDiag);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (;;) {
        // All 64 lanes must take every iteration together.  Without this convergence point the compiler may
        // merge the single-lane region that ends one iteration with the one that starts the next; the other 63
        // lanes then run ahead, readfirstlane() returns THEIR (zero) item, and the wave re-runs item 0 for ever.
        __builtin_amdgcn_wave_barrier();
        int item = 0;
        if (a.dbg && lane == 0) a.dbg[(threadIdx.x >> 6) * 8 + 0] += 1;
        if (lane == 0) item = atomicAdd(a.queue + qslot, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (a.dbg && lane == 0) a.dbg[(threadIdx.x >> 6) * 8 + 1] = item;
        if (item >= count) break;
        const RealignPair pr = a.pairs[first + item];
        if (a.dbg && lane == 0) { a.dbg[(threadIdx.x >> 6) * 8 + 2] = pr.cl; a.dbg[(threadIdx.x >> 6) * 8 + 3] = pr.n2; }
        int best = INT32_MIN, bestj = 0;
        for (int tile = 0; tile < pr.ntiles; ++tile) {
            if (tile > 0) {   // the previous tile's edge stores must be visible to this tile's loads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            realign_tile<CL, EXT0>(a, pr, tile, tab, s_enc, s2strip + (threadIdx.x >> 6) * kRealignStrip, best, bestj);
        }
        if (a.dbg && lane == 0) a.dbg[(threadIdx.x >> 6) * 8 + 5] += 1;
        __builtin_amdgcn_wave_barrier();
        if (lane == 63) {
            a.score[pr.orig] = best >> 4;
            a.end2[pr.orig] = bestj;
            atomicAdd(a.cells, (unsigned long long)pr.n1 * (unsigned long long)pr.n2);
        }
    }
}

// Trace-back: one thread per pair.  State machine of parasail's CIGAR walk (see oracle/strk_oracle.c,
// strk_o_realign): in H follow the recorded source; in a gap state emit the gap base and leave the
// state unless the cell's "extended" flag is set.
__global__ void __launch_bounds__(64) k_realign_trace(RealignArgs a) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= a.n_pairs) return;
    const RealignPair pr = a.pairs[idx];
    const uint8_t* s1 = a.s1 + pr.s1_off;
    const uint8_t* s2 = a.s2 + pr.s2_off;
    uint32_t* out = a.cigar + pr.cig_off;
    const int cl = pr.cl, half = cl / 2;
    const size_t tile_bytes = (size_t)(pr.n2 + 63) * 64 * half;
    const uint8_t* trace = a.trace + pr.trace_off;
    const int tagGi = a.gap_pref ? 0 : 4;
    int i = pr.n1 - 1, j = a.end2[pr.orig];
    int n = 0;
    bool overflow = false;
    uint32_t cur_op = 0xffffffffu, cur_len = 0;
    auto push = [&](uint32_t op) {
        if (op == cur_op) { ++cur_len; return; }
        if (cur_len) {
            if (n < pr.cig_cap) out[n] = (cur_len << 4) | cur_op; else overflow = true;
            ++n;
        }
        cur_op = op;
        cur_len = 1;
    };
    int where = 2;   // 2: H, 1: gap consuming s1 ('I'), 0: gap consuming s2 ('D')
    while (i >= 0 && j >= 0) {
        const int ci = i + pr.pad;
        const int tile = ci / (64 * cl), rem = ci % (64 * cl);
        const int l = rem / cl, c = rem % cl;
        const uint8_t byte = trace[(size_t)tile * tile_bytes + ((size_t)(j + l) * 64 + l) * half + (c >> 1)];
        const int nib = (c & 1) ? (byte >> 4) : (byte & 15);
        if (where == 2) {
            const int tag = nib & kTagMask;
            if (tag == kTagDiag) {
                push(c_enc[s1[i]] == c_enc[s2[j]] ? 7u : 8u);
                --i; --j;
            } else where = (tag == tagGi) ? 1 : 0;
        } else if (where == 1) {
            push(1u);
            if (!(nib & kFlagGiExt)) where = 2;
            --i;
        } else {
            push(2u);
            if (!(nib & kFlagGdExt)) where = 2;
            --j;
        }
    }
    while (i >= 0) { push(1u); --i; }
    if (j >= 0) {   // free leading s2 bases: one D run
        if (cur_op == 2u) cur_len += (uint32_t)(j + 1);
        else { push(2u); cur_len = (uint32_t)(j + 1); }
    }
    push(0xfffffffeu);   // flush the pending run (the sentinel itself stays pending)
    if (overflow) { a.n_cigar[pr.orig] = -1; return; }
    // runs were collected end to start
    const int total = n;
    for (int k = 0; k < total / 2; ++k) {
        const uint32_t x = out[k];
        out[k] = out[total - 1 - k];
        out[total - 1 - k] = x;
    }
    a.n_cigar[pr.orig] = total;
}

}  // namespace strk
