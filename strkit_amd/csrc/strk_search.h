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

// How local_search_range changes inside one search.  The reference calls its search parameters INITIAL values that "can be
// narrowed within the get_repeat_count fn" (strkit/call/repeat_count_params.py:14); the function itself is not in the tree, so
// which schedule it follows is unknown.  0 keeps the range fixed, as the in-tree sibling
// get_ref_repeat_count does (the default); the others are the plausible forms, selectable so that reference vectors
// (tests/test_reference_vectors.py) can name the one that fits:
//   1  one less after every explored stack entry (never below 1);
//   2  halved after every explored stack entry (never below 1);
//   3  the three seed entries [(s-step,-1),(s+step,+1),(s,0)] use the initial range, every chased entry 1.
// An explored entry is one that was popped with a size >= 0.  step_size stays fixed (the read side runs with 1).
constexpr int32_t kNarrowNone = 0, kNarrowDecrement = 1, kNarrowHalve = 2, kNarrowAfterSeed = 3, kNarrowModes = 4;
struct LsrSchedule {
    int32_t mode, lsr, cur;
    STRK_HD LsrSchedule(int32_t mode_, int32_t lsr_) : mode(mode_), lsr(lsr_), cur(lsr_) {}
    STRK_HD int32_t floor1() const { return lsr < 1 ? lsr : 1; }
    STRK_HD int32_t at(bool seed) const { return mode == kNarrowAfterSeed ? (seed ? lsr : floor1()) : cur; }
    STRK_HD void explored() {
        if (mode == kNarrowDecrement) cur = cur - 1 > floor1() ? cur - 1 : floor1();
        else if (mode == kNarrowHalve) cur = (cur >> 1) > floor1() ? (cur >> 1) : floor1();
    }
};

// View of one read's score table: scores[k] = score of candidate size lo + k, k < n.
// `seen` must provide test(k) / set(k) for k < n, initially all clear.
template <class Seen>
STRK_HD SearchResult search_replay(int32_t start, int32_t step, int32_t lsr0, int32_t max_iters,
                                   int32_t tie_last, const int32_t* scores, int32_t lo, int32_t n,
                                   Seen& seen, int32_t narrow = kNarrowNone) {
    SearchResult res = {0, 0, 0, 0, 0, 0, 0};
    int64_t st_size[4];
    int32_t st_dir[4];
    int sp = 0;
    st_size[sp] = (int64_t)start - step; st_dir[sp++] = -1;
    st_size[sp] = (int64_t)start + step; st_dir[sp++] = 1;
    st_size[sp] = start;                 st_dir[sp++] = 0;
    int seeds = 3;                       // the entries below this stack depth are seeds (nothing is pushed under them)
    bool have_best = false;
    int32_t best_i = 0, best_s = 0, n_scored = 0;
    LsrSchedule sched(narrow, lsr0);
    while (sp > 0 && n_scored < max_iters) {
        --sp;
        const int64_t size = st_size[sp];
        const int32_t dir = st_dir[sp];
        const bool seed = sp < seeds;
        if (seed) seeds = sp;
        if (size < 0) continue;
        const int32_t lsr = sched.at(seed);
        sched.explored();
        const bool widen = step > lsr;
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

// ---------------------------------------------------------------------------------------------
// Banded scoring with an exactness certificate (k_dp_band).
//
// The band kernel restricts the forward DP to the diagonals d = j - r in [dlo, dlo + Wd) and the
// backward DP (over the reversed right flank) to delta = j' - k' in [bdlo, bdlo + Wd); what it returns
// for candidate i, S_band[i], is the exact maximum over the alignments that stay inside both bands and
// end by the normal rules, so S_band[i] <= S[i].  Every other alignment touches a diagonal outside one
// of the bands (or belongs to one of the two free-end families the band kernel does not track), and an
// alignment that touches diagonal d scores at most 2 * (number of cells on d): split it at a cell on d,
// each half has no more diagonal moves than d has cells on that side, and reaching d from any other
// start/end diagonal costs 5 per step while gaining at most 2.  Hence
//     S[i] <= max(S_band[i], band_ub(i)),
// and an entry with S_band[i] >= band_ub(i) is exact.  The search is replayed on (S_band, band_ub)
// with search_replay_cert(); whenever a comparison could be changed by an inexact entry the read is
// re-scored by the exact kernels.  BASELINE's band = 64/128/512 are widths at which HiFi reads of the
// named configs certify; noisy (ONT-like) reads mostly do not and take the exact path.
// ---------------------------------------------------------------------------------------------
struct BandGeo {
    int32_t ok;      // eligible for the band kernel
    int32_t cls;     // band class 0..7 (band_class_G / band_class_D)
    int32_t G;
    int32_t wd;      // forward band width in diagonals = D * G (128, 256, 512, 1024; 96 for the narrow class)
    int32_t dlo;     // forward band: d in [dlo, dlo + wd)
    int32_t bwd;     // backward band width (= wd: in original diagonals the backward band must hold diagonal 0 AND the end
                     //   corner's, with the same slack as the forward band, or the 2 * len(diagonal) bound is void)
    int32_t bdlo;    // backward band (reversed coordinates): delta in [bdlo, bdlo + bwd)
    int32_t cmin;    // first db node column the fork rows can touch
    int32_t ncol;    // number of such columns: (n - 1) * m + wd
};

// Band classes: 0..3 = 8 << c lanes per read x 16 diagonals per lane (bands of 128, 256, 512, 1 024); 4..7 = the same lane
// counts x 12 diagonals (96, 192, 384, 768).  The candidate sizes of a window of +-W span 2 W |motif| diagonals, so the band a
// read needs follows its motif length: motifs of up to 4 bases leave a band of 96 as much slack as longer ones have in 128,
// motifs of ~17 need 330 diagonals at W = 8 (BASELINE config 4's long motifs: 384 instead of 512).
// Per-class limits (what the class's LDS layout holds: class c lives in the layout of class c & 3).  Lane counts of 32 and 64
// generate the forward row symbols on the fly (left flank <= 192 rows from LDS, then the motif, one running address).
constexpr int kNumBandClasses = 8;
STRK_HD constexpr int band_class_layout(int c) { return c & 3; }
STRK_HD constexpr int band_class_G(int c) { return 8 << (c & 3); }
STRK_HD constexpr int band_class_D(int c) { return c >= 4 ? 12 : 16; }
STRK_HD constexpr int band_class_wd(int c) { return band_class_G(c) * band_class_D(c); }
STRK_HD constexpr int band_max_db(int c) { return (c & 3) == 0 ? 448 : ((c & 3) == 1 ? 1024 : ((c & 3) == 2 ? 4096 : 12288)); }
STRK_HD constexpr int band_max_col(int c) { return (c & 3) == 0 ? 320 : ((c & 3) == 1 ? 512 : ((c & 3) == 2 ? 1024 : 1536)); }
STRK_HD constexpr bool band_class_fly(int c) { return (c & 3) >= 2; }
// classes that track the running maximum of the last column (alignments that end there above the fork row): all but the
// narrowest ones, where the bound on those alignments never reaches a good read's score (motifs of up to ~9 bases)
STRK_HD constexpr bool band_class_lmax(int c) { return (c & 3) >= 1; }
STRK_HD constexpr bool band_class_wide_kernel(int c) { return (c & 3) >= 2; }   // k_dp_band_wide's classes
constexpr int kBandMaxFlank = 127;
constexpr int kBandFlyMaxMotif = 126;   // ... and the motif at least twice (the running address is taken back by whole copies)
constexpr int kBandFlyMaxFlank = 192;   // on-the-fly rows: 256 staged bytes hold the left flank behind the 63 null rows of the last lane
constexpr int kBandRowSlack = 96;    // prefix rows a band item may have beyond |db|
constexpr int kBandNarrowSlack = 22; // diagonals a 12-diagonal class keeps free on each side of the candidates' span, at least

// Where the forward band lies (round 4): the table holds the candidates est - W .. est + W, but a search that converges at once
// only ever scores est +- 4 (seed windows [s - 3, s + 3], [s + 1, s + 4], [s - 4, s - 1]); the outer entries are there for the
// reads whose start the caller's feedback moved.  The band is therefore laid around the corner diagonals of the INNER candidates
// only — the table's middle +- span_w sizes — with the slack rule below on each side; an outer candidate still gets its fork row and
// a sound (S_band, band_ub) pair, just with less slack (its corner sits nearer the band's edge), and a search that needs it and
// cannot certify it is re-scored exactly as before.  What that buys: a window of +-6 sizes of a 6-base motif spans 73 diagonals and
// took the 128-diagonal class; its inner +-4 span 49 and fit the 96-diagonal one.
// Slack per side: |db| / 32 (a HiFi read's score deficit grows with its length), at least 12 (22 in the 12-diagonal classes),
// and slack_m8 / 8 diagonals per motif base: a search window that does not hold the best size has its maximum about 7 |motif| under
// the perfect score (one copy too many or too few), and the 2 * len(diagonal) bound of its inexact entries must stay below that.
struct BandTune {
    int32_t span_w;     // half-width (in candidate sizes) of the table's middle that the band is laid around
    int32_t slack_m8;   // slack per side >= slack_m8 * |motif| / 8 diagonals
};
constexpr BandTune kBandTuneNone = {64, 0};   // the whole table, no motif term (the geometry of rounds 2-3)

// the inner candidates [c_lo, c_hi] of a table [lo, lo + n)
STRK_HD void band_inner(int32_t lo, int32_t n, int32_t span_w, int32_t* c_lo, int32_t* c_hi) {
    int32_t t = (n - 1) / 2 - span_w;
    if (t < 0) t = 0;
    *c_lo = lo + t;
    *c_hi = lo + n - 1 - t;
}

STRK_HD BandGeo band_geometry(int32_t nfl, int32_t ntr, int32_t nfr, int32_t m, int32_t lo, int32_t n, BandTune tune = kBandTuneNone) {
    BandGeo b = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t ndb = (int64_t)nfl + ntr + nfr;
    if (nfl < 1 || nfr < 1 || nfr > kBandMaxFlank || m < 1 || m > 256 || n < 1 || n > 32) return b;
    int32_t c_lo, c_hi;
    band_inner(lo, n, tune.span_w, &c_lo, &c_hi);
    const int64_t e_lo = (int64_t)ntr - (int64_t)c_hi * m, e_hi = (int64_t)ntr - (int64_t)c_lo * m;
    const int64_t span_lo = e_lo < 0 ? e_lo : 0, span_hi = e_hi > 0 ? e_hi : 0;
    // slack on each side: the certificate needs about half the score deficit of the read, and a HiFi read's deficit grows
    // with its length (~0.3 % errors at 9-12 points each): |db| / 32 keeps the certificate failures of kilobase windows
    // (long motifs, BASELINE config 4) at a few per cent
    int64_t smin = ndb >> 5;
    if (smin < 12) smin = 12;
    if (smin < ((int64_t)tune.slack_m8 * m >> 3)) smin = (int64_t)tune.slack_m8 * m >> 3;
    const int64_t rows = (int64_t)nfl + (int64_t)(lo + n - 1) * m;
    int32_t cls = -1;
    for (int32_t k = 0; k < kNumBandClasses && cls < 0; ++k) {   // narrowest class that holds band and window: 96, 128, 192, 256, ...
        const int32_t c = (k >> 1) + ((k & 1) ? 0 : 4);
#ifdef STRK_NO_NARROW
        if (c >= 4) continue;
#endif
        const int64_t w = band_class_wd(c);
        const int64_t slack = (c >= 4 && smin < kBandNarrowSlack) ? kBandNarrowSlack : smin;
        if (span_hi - span_lo + 1 + 2 * slack > w) continue;
        if (ndb > band_max_db(c) || (int64_t)(n - 1) * m + w > band_max_col(c) || rows > band_max_db(c) + kBandRowSlack) continue;
        if (band_class_fly(c) && (nfl > kBandFlyMaxFlank || m > kBandFlyMaxMotif)) continue;
        cls = c;
    }
    if (cls < 0) return b;
    const int32_t wd = band_class_wd(cls);
    // (a band wider than the window covers the whole matrix: such short windows cost no more here than in the exact
    // kernel and spare it a launch that only a handful of reads would use)
    if ((cls & 3) >= 1 && (int64_t)wd * 5 > (ndb + 1) * 4) return b;   // a wide band must drop at least a fifth of the columns
    const int64_t extra = wd - (span_hi - span_lo + 1);
    b.ok = 1;
    b.cls = cls;
    b.G = band_class_G(cls);
    b.wd = wd;
    b.dlo = (int32_t)(span_lo - extra / 2);
    b.bwd = wd;
    b.bdlo = -(b.bwd / 2);
    b.cmin = (int32_t)(nfl + (int64_t)lo * m + b.dlo);
    b.ncol = (int32_t)((int64_t)(n - 1) * m + wd);
    return b;
}

// The geometry of an item that band_geometry has already put into class `cls` (k_plan did; the band kernel works on that class's
// list): the same numbers without the search over the classes, in 32-bit arithmetic (the class limits bound every length).
STRK_HD BandGeo band_geometry_of_class(int32_t cls, int32_t nfl, int32_t ntr, int32_t m, int32_t lo, int32_t n, BandTune tune = kBandTuneNone) {
    BandGeo b;
    int32_t c_lo, c_hi;
    band_inner(lo, n, tune.span_w, &c_lo, &c_hi);
    const int32_t e_lo = ntr - c_hi * m, e_hi = ntr - c_lo * m;
    const int32_t span_lo = e_lo < 0 ? e_lo : 0, span_hi = e_hi > 0 ? e_hi : 0;
    const int32_t wd = band_class_wd(cls);
    const int32_t extra = wd - (span_hi - span_lo + 1);
    b.ok = 1;
    b.cls = cls;
    b.G = band_class_G(cls);
    b.wd = wd;
    b.dlo = span_lo - extra / 2;
    b.bwd = wd;
    b.bdlo = -(wd / 2);
    b.cmin = nfl + lo * m + b.dlo;
    b.ncol = (n - 1) * m + wd;
    return b;
}

// (32-bit arithmetic: band_ub is only asked about items band_geometry accepted, whose lengths are bounded by the class limits)
STRK_HD int32_t band_diag_len(int32_t nc, int32_t ndb, int32_t d) {
    const int32_t v = d >= 0 ? (nc < ndb - d ? nc : ndb - d) : (nc + d < ndb ? nc + d : ndb);
    return v < 0 ? 0 : v;
}
// longest diagonal among d >= d0 (resp. d <= d0)
STRK_HD int32_t band_len_beyond_hi(int32_t nc, int32_t ndb, int32_t d0) { return d0 >= 0 ? band_diag_len(nc, ndb, d0) : (nc < ndb ? nc : ndb); }
STRK_HD int32_t band_len_beyond_lo(int32_t nc, int32_t ndb, int32_t d0) { return d0 <= 0 ? band_diag_len(nc, ndb, d0) : (nc < ndb ? nc : ndb); }

// Upper bound on the score of every alignment of candidate i the band kernel does not consider.
STRK_HD int32_t band_ub(const BandGeo& b, int32_t nfl, int32_t ntr, int32_t nfr, int32_t m, int32_t i, int32_t end_flags) {
    const int32_t ndb = nfl + ntr + nfr, R = nfl + i * m, nc = R + nfr;
    const int32_t e = ndb - nc;   // diagonal of the end corner
    int32_t L = band_len_beyond_hi(nc, ndb, b.dlo + b.wd);                     // right of the forward band
    int32_t v = band_len_beyond_lo(nc, ndb, b.dlo - 1);                       // left of it
    if (v > L) L = v;
    v = band_len_beyond_hi(nc, ndb, e - b.bdlo + 1);                          // backward band, original diagonals
    if (v > L) L = v;
    v = band_len_beyond_lo(nc, ndb, e - (b.bdlo + b.bwd - 1) - 1);
    if (v > L) L = v;
    // alignments that end in the last column at a row <= R_i (free candidate end): the classes of band_class_lmax track
    // the in-band ones exactly (the rest leaves the band: first term), the narrowest class bounds them all
    if ((end_flags & 8) && !band_class_lmax(b.cls)) { v = band_len_beyond_hi(nc, ndb, ndb - R); if (v > L) L = v; }
    if (end_flags & 4) { v = band_len_beyond_lo(nc, ndb, -R - 1); if (v > L) L = v; }      // starts on the left edge below R_i
    return 2 * L;
}

// search_replay on a banded table: scores[k] is a lower bound, ub(k) the bound on what the band kernel
// ignored.  Identical to search_replay whenever it returns with uncertain == 0; uncertain == 1 means an
// inexact entry could have changed a comparison (or been the maximum) and the read needs exact scores.
struct CertResult {
    SearchResult res;
    int32_t uncertain;
};

template <class Seen, class Ub>
STRK_HD CertResult search_replay_cert(int32_t start, int32_t step, int32_t lsr0, int32_t max_iters, int32_t tie_last,
                                      const int32_t* scores, int32_t lo, int32_t n, Seen& seen, const Ub& ub,
                                      int32_t narrow = kNarrowNone) {
    CertResult out = {{0, 0, 0, 0, 0, 0, 0}, 0};
    SearchResult& res = out.res;
    int64_t st_size[4];
    int32_t st_dir[4];
    int sp = 0;
    st_size[sp] = (int64_t)start - step; st_dir[sp++] = -1;
    st_size[sp] = (int64_t)start + step; st_dir[sp++] = 1;
    st_size[sp] = start;                 st_dir[sp++] = 0;
    bool have_best = false;
    int32_t best_i = 0, best_s = 0, n_scored = 0;
    int32_t max_inexact = -(1 << 30);   // largest upper bound among the inexact entries scored so far
    int seeds = 3;
    LsrSchedule sched(narrow, lsr0);
    while (sp > 0 && n_scored < max_iters) {
        --sp;
        const int64_t size = st_size[sp];
        const int32_t dir = st_dir[sp];
        const bool seed = sp < seeds;
        if (seed) seeds = sp;
        if (size < 0) continue;
        const int32_t lsr = sched.at(seed);
        sched.explored();
        const bool widen = step > lsr;
        int64_t w_lo = size - ((dir < 1 || widen) ? lsr : 0);
        if (w_lo < 0) w_lo = 0;
        const int64_t w_hi = size + ((dir > -1 || widen) ? lsr : 0);
        if (w_lo < lo || w_hi >= (int64_t)lo + n) {
            res.miss = 1;
            res.need_lo = (int32_t)w_lo;
            res.need_hi = (int32_t)w_hi;
            return out;
        }
        bool have_mv = false;
        int64_t mv_i = 0;
        int32_t mv_s = 0, win_inexact = -(1 << 30);
        for (int64_t i = w_lo; i <= w_hi; ++i) {
            const int32_t k = (int32_t)(i - lo);
            const int32_t s = scores[k];
            const int32_t u = ub(k);
            const bool exact = s >= u;
            if (!seen.test(k)) {
                seen.set(k);
                ++n_scored;
                if (exact) {
                    if (!have_best || s > best_s || (tie_last && s == best_s)) { have_best = true; best_i = (int32_t)i; best_s = s; }
                } else if (u > max_inexact) {
                    max_inexact = u;
                }
            }
            if (exact) {
                if (!have_mv || s > mv_s || (tie_last && s == mv_s)) { have_mv = true; mv_i = i; mv_s = s; }
            } else if (u > win_inexact) {
                win_inexact = u;
            }
        }
        // the window's maximum must be an exact entry that no inexact one can reach or tie
        if (!have_mv || win_inexact >= mv_s) { out.uncertain = 1; return out; }
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
    if (n_scored == 0) { res.empty = 1; return out; }
    if (!have_best || max_inexact >= best_s) { out.uncertain = 1; return out; }
    res.cn = best_i;
    res.score = best_s;
    return out;
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
