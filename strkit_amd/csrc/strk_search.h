// strk_search.h — the candidate-size hill climb, replayed over a precomputed score table.
//
// Control flow follows the only in-tree statement of the search, get_ref_repeat_count
// (strkit/call/repeats.py:100-151), applied to one score per candidate size as the read-side
// contract describes (repeats.py:55-56: ((best size, best score), n_explored, best - start)):
//   * to_explore is a Python list used as a stack, seeded [(s-step,-1),(s+step,+1),(s,0)]
//     (repeats.py:100-101), so (s, 0) is visited first (:107);
//   * negative sizes are skipped (:108-109); the window is +/- local_search_range on the open
//     side(s) (:114-117); every unseen size in it is scored and counted (:120-130);
//   * the window's maximum (Python max(): first maximal element, :135) is chased by +/- step if
//     that size has not been scored yet (:136-151);
//   * the loop stops when the stack is empty or n >= max_iters (:106);
//   * the result is the first maximum of the {size: score} dict in insertion order (:154).
// The same function runs on the device (one lane per locus) and on the host (window-miss path).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STRK_HD __host__ __device__ inline
#else
#define STRK_HD inline
#endif

namespace strk {

struct SearchResult {
    int32_t cn, score, n_explored;
    int32_t miss;              // 1: a needed size lies outside the table window
    int32_t need_lo, need_hi;  // the window that was being scanned when the miss happened
    int32_t empty;             // 1: nothing was scored (Python's max() would raise)
};

// View of one read's score table: scores[k] = score of candidate size lo + k, k < n.
// `seen` must provide test(k) / set(k) for k < n, initially all clear.
template <class Seen>
STRK_HD SearchResult search_replay(int32_t start, int32_t step, int32_t lsr, int32_t max_iters,
                                   int32_t tie_last, const int32_t* scores, int32_t lo, int32_t n,
                                   Seen& seen) {
    SearchResult res = {0, 0, 0, 0, 0, 0, 0};
    int64_t st_size[4];
    int32_t st_dir[4];
    int sp = 0;
    st_size[sp] = (int64_t)start - step; st_dir[sp++] = -1;
    st_size[sp] = (int64_t)start + step; st_dir[sp++] = 1;
    st_size[sp] = start;                 st_dir[sp++] = 0;
    bool have_best = false;
    int32_t best_i = 0, best_s = 0, n_scored = 0;
    const bool widen = step > lsr;
    while (sp > 0 && n_scored < max_iters) {
        --sp;
        const int64_t size = st_size[sp];
        const int32_t dir = st_dir[sp];
        if (size < 0) continue;
        int64_t w_lo = size - ((dir < 1 || widen) ? lsr : 0);
        if (w_lo < 0) w_lo = 0;
        const int64_t w_hi = size + ((dir > -1 || widen) ? lsr : 0);
        if (w_lo < lo || w_hi >= (int64_t)lo + n) {
            res.miss = 1;
            res.need_lo = (int32_t)w_lo;
            res.need_hi = (int32_t)w_hi;
            return res;
        }
        bool have_mv = false;
        int64_t mv_i = 0;
        int32_t mv_s = 0;
        for (int64_t i = w_lo; i <= w_hi; ++i) {
            const int32_t k = (int32_t)(i - lo);
            const int32_t s = scores[k];
            if (!seen.test(k)) {
                seen.set(k);
                ++n_scored;
                if (!have_best || s > best_s || (tie_last && s == best_s)) {
                    have_best = true;
                    best_i = (int32_t)i;
                    best_s = s;
                }
            }
            if (!have_mv || s > mv_s || (tie_last && s == mv_s)) {
                have_mv = true;
                mv_i = i;
                mv_s = s;
            }
        }
        // "not in seen" for a size outside the table window is simply true: it cannot have been scored.
        if (mv_i > size) {
            const int64_t nr = mv_i + step;
            const bool in_tab = nr >= lo && nr < (int64_t)lo + n;
            if (nr >= 0 && !(in_tab && seen.test((int32_t)(nr - lo)))) { st_size[sp] = nr; st_dir[sp++] = 1; }
        }
        if (mv_i < size) {
            const int64_t nr = mv_i - step;
            const bool in_tab = nr >= lo && nr < (int64_t)lo + n;
            if (nr >= 0 && !(in_tab && seen.test((int32_t)(nr - lo)))) { st_size[sp] = nr; st_dir[sp++] = -1; }
        }
    }
    res.n_explored = n_scored;
    if (!have_best) {
        res.empty = 1;
        return res;
    }
    res.cn = best_i;
    res.score = best_s;
    return res;
}

struct SeenMask64 {
    uint64_t m = 0;
    STRK_HD bool test(int k) const { return (m >> k) & 1; }
    STRK_HD void set(int k) { m |= (uint64_t)1 << k; }
};

// Caller protocol (strkit/call/call_locus.py:1129-1136): start = est + round(frac * est) unless that
// offset is < -est, in which case frac is reset and the bare estimate is used.  Python's round()
// on a float is round-half-to-even of the exact double product = rint() in the default FP mode.
STRK_HD int32_t feedback_start(int32_t est, double* frac) {
    int64_t sc = est;
#if defined(__HIP_DEVICE_COMPILE__)
    const int64_t off = (int64_t)rint(*frac * (double)sc);
#else
    const int64_t off = (int64_t)__builtin_rint(*frac * (double)sc);
#endif
    if (off < -sc) *frac = 0.0;
    else sc += off;
    return (int32_t)sc;
}

// call_locus.py:1161: frac += new_offset / max(read_cn, 1)   (float64 true division)
STRK_HD void feedback_update(double* frac, int32_t cn, int32_t start) {
    *frac += (double)(cn - start) / (double)(cn > 1 ? cn : 1);
}

}  // namespace strk
