// strk_replay.h — k_replay: in-order search replay per locus with the start-count feedback
// Part of strk_kernels.h: included at its end, after the shared definitions (KArgs, counters, k_hash, k_plan).
#pragma once

namespace strk {

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
// (at most 96 VGPRs: a wave of this kernel then fits next to the two resident waves of a band kernel on a SIMD, so that the
// replay of one call runs inside the band pass of the next one instead of waiting for a free CU slot)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) k_replay(KArgs a, ReplayArgs p) {
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
        // per motif-length bucket: loci, and loci whose search left its window (the host adapts each bucket's default window)
        const int bkt = win_bucket(a.motif_off[l + 1] - a.motif_off[l]);
        atomicAdd(&a.counters[kCntLociB + bkt], 1);
        if (missed) atomicAdd(&a.counters[kCntMissB + bkt], 1);
    }
}

}  // namespace strk
