// strk_replay.h — k_replay: in-order search replay per locus with the start-count feedback
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

// ---------------------------------------------------------------------------------------------
// Search replay: one lane per locus walks its reads in caller order (call_locus.py:1082) with the
// start-count feedback (call_locus.py:1129-1136,1161) and replays the hill climb on the table.
// ---------------------------------------------------------------------------------------------
struct ReplayArgs {
    int32_t max_iters, lsr, step, tie_last, feedback, narrow;
    int32_t* out_cn;
    int32_t* out_score;
    int32_t* out_n;
    int32_t* out_start;
    // per-locus resume state
    int32_t* next_read;   // [n_loci] first read not yet finished (== read_off[l+1] when done)
    double* frac;         // [n_loci]
    int32_t* need_lo;     // [n_loci] window wanted by the read that missed
    int32_t* need_hi;
    // Two passes when the banded kernel took part (strk_api.hip: ... k_dp_band -> k_replay pass 1 -> exact kernels -> k_replay
    // pass 2).  Pass 1 (pre_exact = 1) runs BEFORE the exact kernels: it walks every locus as far as certified band tables
    // carry it.  At a read whose table is still to be written by the exact kernels (a band fall-back, a read the band never
    // took) it notes where the locus stopped and leaves; at a read whose banded table cannot certify the search from the
    // start the feedback gave it (the band kernel certified it from the estimate only) it first appends that read and every
    // later read of the locus that still has a banded table to the exact kernels' class lists — reads of a locus look alike:
    // what is uncertain for one usually is for the next.  Pass 2 (resume = 1) runs behind the exact kernels and takes the
    // stopped loci from there with exact tables, so that the only misses left for the host are searches that leave their
    // candidate window.  (Before: every uncertain read was a miss, and every call of the bench workload paid a host round.)
    int32_t pre_exact;     // 1: pass 1
    int32_t resume;        // 1: pass 2 — continue the loci with stop[l] == kStopRescore from next_read / frac
    int32_t* stop;         // [n_loci] kStopNone / kStopRescore / kStopMiss
};
constexpr int kStopNone = 0, kStopRescore = 1, kStopMiss = 2;

// One wave per locus: lane i holds the inputs of the locus's i-th read (coalesced loads), the
// in-order chain over the reads is wave-uniform ALU work on values fetched with v_readlane.
// (at most 96 VGPRs: a wave of this kernel then fits next to the two resident waves of a band kernel on a SIMD, so that the
// replay of one call runs inside the band pass of the next one instead of waiting for a free CU slot)
#ifndef STRK_REPLAY_WAVES
#define STRK_REPLAY_WAVES 5   // waves per SIMD the register allocator aims at (tools/exp_build.sh variants)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STRK_REPLAY_WAVES, 8))) k_replay(KArgs a, ReplayArgs p) {
    const int l = blockIdx.x;
    const int lane = threadIdx.x;
    const int r_end = a.read_off[l + 1];
    double frac = 0.0;
    int r_next = a.read_off[l];   // first read not finished yet
    if (p.resume) {
        if (p.stop[l] != kStopRescore) return;
        r_next = p.next_read[l];
        frac = p.frac[l];
    }
    bool missed = false, rescore = false;
    for (int base = r_next; base < r_end && !missed && !rescore; base += 64) {
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
                const bool is_exact = !a.band_mode || a.exact[rp];
                if (p.pre_exact && is_exact) {   // its table comes with the exact kernels: pass 2 continues here
                    rescore = true;
                    break;
                }
                if (is_exact) {
                    res = search_replay(start, p.step, p.lsr, p.max_iters, p.tie_last, a.table + a.tab_off[r], a.win_lo[r],
                                        min(a.win_n[r], 64), seen, p.narrow);
                } else {
                    // banded table: lower bounds + certificate; an ambiguous comparison asks for exact scores
                    const int nfl = a.nfl[rp], ntr = a.ntr[rp], nfr = a.nfr[rp];
                    const int m = a.motif_off[l + 1] - a.motif_off[l];
                    const int wlo = a.win_lo[r], wn = min(a.win_n[r], 64);
                    const BandGeo geo = band_geometry(nfl, ntr, nfr, m, wlo, wn, a.band_tune);
                    const int flags = a.end_flags;
                    auto ub = [&](int k) { return band_ub(geo, nfl, ntr, nfr, m, wlo + k, flags); };
                    const CertResult cr = search_replay_cert(start, p.step, p.lsr, p.max_iters, p.tie_last,
                                                             a.table + a.tab_off[r], wlo, wn, seen, ub, p.narrow);
                    res = cr.res;
                    if (cr.uncertain && p.pre_exact) {
                        // this read and the rest of the locus go to the exact kernels (every read at most once: test-and-set of
                        // its flag byte), lane j taking read r + j, r + j + 64, ...
                        for (int q0 = r; q0 < r_end; q0 += 64) {
                            const int q = q0 + lane;
                            if (q >= r_end) continue;
                            const int rq = a.rep[q];
                            unsigned* const w = reinterpret_cast<unsigned*>(a.exact + (rq & ~3));
                            const unsigned bit = 1u << (8 * (rq & 3));
                            if (atomicOr(w, bit) & (0xffu << (8 * (rq & 3)))) continue;   // exact already, or listed by another lane
                            const int cls = classify(a.nfl[rq], a.ntr[rq], a.nfr[rq], m, a.win_lo[rq], min(a.win_n[rq], kTableMax), 0, 0);
                            const int idx = atomicAdd(&a.counters[kCntClass0 + cls], 1);
                            if (idx < a.list_stride) {
                                int32_t* gl = a.cls_list + (size_t)cls * a.list_stride * 2;
                                gl[2 * idx] = rq;
                                gl[2 * idx + 1] = 0;
                            } else {
                                atomicOr(&a.counters[kCntError], kErrList);
                            }
                            atomicAdd(&a.counters[kCntBandFallback], 1);
                            if (cls != kGenericClass) {
                                const unsigned long long cc = (unsigned long long)(a.nfl[rq] + a.ntr[rq] + a.nfr[rq]) *
                                                              ((unsigned long long)a.nfl[rq] + (unsigned long long)(a.win_lo[rq] + a.win_n[rq] - 1) * m + a.nfr[rq]);
                                atomicAdd(a.cells, cc);
                                atomicAdd(a.cells + cell_slot_of_list(cls), cc);
                            }
                        }
                        rescore = true;
                        break;
                    }
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
        p.stop[l] = rescore ? kStopRescore : (missed ? kStopMiss : kStopNone);
        // loci whose search left its window, per motif-length bucket (the host adapts each bucket's default window; k_plan
        // counts the loci of a bucket)
        if (missed) atomicAdd(&a.counters[kCntMissB + win_bucket(a.motif_off[l + 1] - a.motif_off[l])], 1);
    }
}

}  // namespace strk
