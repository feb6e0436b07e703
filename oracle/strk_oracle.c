/*
 * oracle/strk_oracle.c — CPU restatement of STRkit's per-read repeat-count path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under strkit_amd/ (the product) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of the read-side path lives in two third-party
 * packages that are NOT vendored under /root/reference:
 *   - strkit_rust_ext == 0.29.0   (reference pyproject.toml:18; call site
 *                                  strkit/call/repeats.py:58-68)
 *   - parasail >= 1.3.4, < 1.4    (reference pyproject.toml:14; call sites
 *                                  strkit/call/align_matrix.py:34,
 *                                  strkit/call/repeats.py:33,40,92-93,
 *                                  strkit/call/realign.py:56)
 * and no reference test holds golden vectors for it.  This file
 * restates (a) parasail's published semi-global Gotoh recurrence and (b) the search
 * control flow of the only in-tree statement of the algorithm, get_ref_repeat_count
 * (strkit/call/repeats.py:100-151), applied to ONE alignment per candidate as the
 * read-side contract (repeats.py:55-56) describes.  Every choice that cannot be read
 * from /root/reference is a named switch below with its default documented.
 *
 * Functions and the reference lines they follow:
 *   strk_o_matrix / strk_o_encode  : strkit/call/align_matrix.py:15-44, strkit/iupac.py:9-21
 *   strk_o_sg_align                : parasail sg_* semantics (gap of length k costs
 *                                    open + (k-1)*extend; end-gap flags), used by the
 *                                    reference at repeats.py:33,40 (sg_qe) and realign.py:56 (sg_dx)
 *   strk_o_repeat_count            : repeats.py:47-70 (contract) + repeats.py:100-151 (search shape)
 *   strk_o_count_locus             : strkit/call/call_locus.py:1079,1125-1161 (start-count feedback)
 *   strk_o_score_ref_boundaries    : repeats.py:23-43
 *   strk_o_ref_repeat_count        : repeats.py:73-192
 *   strk_o_realign                 : strkit/call/realign.py:56-72 (parasail sg_dx_trace + CIGAR; tie rules restated
 *                                    from memory of parasail's trace kernels — see the function's own header)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define NSYM 17 /* 16 alphabet letters + parasail's implicit '*' row/column */

/* flags for strk_o_sg_align (s1 = parasail "query"/profile side, s2 = "database") */
#define STRK_O_S1_BEG_FREE 1
#define STRK_O_S1_END_FREE 2
#define STRK_O_S2_BEG_FREE 4
#define STRK_O_S2_END_FREE 8
#define STRK_O_SG_ALL 15

#define STRK_O_TIE_FIRST 0 /* Python max(): first maximal element (repeats.py:135,154) */
#define STRK_O_TIE_LAST 1  /* Rust Iterator::max_by_key: last maximal element */

static int8_t g_mat[NSYM][NSYM];
static uint8_t g_enc[256];
static int g_ready = 0;

/* align_matrix.py:25-26 — index order "ACGT" + IUPAC keys in dict order + "X". */
static const char ALPHABET[] = "ACGTRYSWKMBDHVNX";

/* iupac.py:9-21.  NB: "D" is (A,C,T) in the reference — identical to "H" — a quirk
 * of the reference table that the scoring matrix inherits; reproduced on purpose. */
static const char* members(char c) {
    switch (c) {
    case 'R': return "AG";
    case 'Y': return "CT";
    case 'S': return "CG";
    case 'W': return "AT";
    case 'K': return "GT";
    case 'M': return "AC";
    case 'B': return "CGT";
    case 'D': return "ACT"; /* sic — reference iupac.py:17 */
    case 'H': return "ACT";
    case 'V': return "ACG";
    case 'N': return "ACGT";
    case 'X': return "ACGT"; /* align_matrix.py:29 */
    default: return "";
    }
}

static int idx_of(char c) {
    const char* p = strchr(ALPHABET, c);
    return p ? (int)(p - ALPHABET) : 16;
}

void strk_o_init(void) {
    if (g_ready) return;
    /* parasail matrix_create(alphabet, match, mismatch): diagonal = match, off-diagonal =
     * mismatch, plus one extra '*' row/column of zeros that every character outside the
     * alphabet maps to; the mapper is case-insensitive.  (parasail is not in-tree: from its
     * published source, flagged unverifiable here — SURVEY.md §7 "Case handling".) */
    for (int i = 0; i < NSYM; i++)
        for (int j = 0; j < NSYM; j++)
            g_mat[i][j] = (i == 16 || j == 16) ? 0 : (i == j ? 2 : -7);
    /* align_matrix.py:36-39: code<->member base = 2, except X<->base = 0, both ways. */
    for (int ci = 4; ci < 16; ci++) {
        char code = ALPHABET[ci];
        for (const char* m = members(code); *m; m++) {
            int bi = idx_of(*m);
            int8_t v = (code != 'X') ? 2 : 0;
            g_mat[ci][bi] = v;
            g_mat[bi][ci] = v;
        }
    }
    memset(g_enc, 16, sizeof g_enc);
    for (int i = 0; i < 16; i++) {
        g_enc[(unsigned char)ALPHABET[i]] = (uint8_t)i;
        g_enc[(unsigned char)(ALPHABET[i] | 0x20)] = (uint8_t)i; /* lower case */
    }
    g_ready = 1;
}

const int8_t* strk_o_matrix(void) {
    strk_o_init();
    return &g_mat[0][0];
}

int strk_o_encode(int c) {
    strk_o_init();
    return g_enc[(unsigned char)c];
}

/*
 * Semi-global Gotoh alignment score, exact int32 (parasail's *_sat variants retry at wider
 * integer widths, so their result equals the exact recurrence).
 *   E(i,j) = max(E(i,j-1) - ext, H(i,j-1) - open)       gap in s1 (consumes s2)
 *   F(i,j) = max(F(i-1,j) - ext, H(i-1,j) - open)       gap in s2 (consumes s1)
 *   H(i,j) = max(H(i-1,j-1) + W(s1[i], s2[j]), E, F)
 * Boundaries: H(i,0) = 0 if S2_BEG_FREE... careful with naming — we follow parasail:
 *   "s1 begin free"  = leading s1 characters may be skipped for free  -> H(i,0) = 0
 *   "s2 begin free"  = leading s2 characters may be skipped for free  -> H(0,j) = 0
 *   "s1 end free"    = trailing s1 characters may be skipped          -> max over H(i, n2)
 *   "s2 end free"    = trailing s2 characters may be skipped          -> max over H(n1, j)
 * otherwise the boundary costs open + (k-1)*ext and the alignment must reach the end.
 * End position tie rule (only matters for end1/end2, never for the score): the cell (n1,n2)
 * first, then the last column scanned for increasing i with strict '>' (smallest i wins),
 * then the last row for increasing j with strict '>'.  parasail's own rule is
 * implementation-specific and unverifiable here.
 */
int32_t strk_o_sg_align(const uint8_t* s1, int32_t n1, const uint8_t* s2, int32_t n2, int32_t open,
                        int32_t ext, int32_t flags, int32_t* end1, int32_t* end2) {
    strk_o_init();
    if (n1 <= 0 || n2 <= 0) {
        if (end1) *end1 = -1;
        if (end2) *end2 = -1;
        return 0;
    }
    const int32_t NEG = INT32_MIN / 4;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int32_t* F = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int32_t* lastcol = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 + 1));
    H[0] = 0;
    for (int32_t j = 1; j <= n2; j++) {
        H[j] = (flags & STRK_O_S2_BEG_FREE) ? 0 : -(open + (j - 1) * ext);
        F[j] = NEG;
    }
    for (int32_t i = 1; i <= n1; i++) {
        const int8_t* wrow = g_mat[g_enc[s1[i - 1]]];
        int32_t diag = H[0];
        H[0] = (flags & STRK_O_S1_BEG_FREE) ? 0 : -(open + (i - 1) * ext);
        int32_t E = NEG;
        for (int32_t j = 1; j <= n2; j++) {
            int32_t e1 = E - ext, e2 = H[j - 1] - open;
            E = e1 > e2 ? e1 : e2;
            int32_t f1 = F[j] - ext, f2 = H[j] - open;
            F[j] = f1 > f2 ? f1 : f2;
            int32_t h = diag + wrow[g_enc[s2[j - 1]]];
            if (E > h) h = E;
            if (F[j] > h) h = F[j];
            diag = H[j];
            H[j] = h;
        }
        lastcol[i] = H[n2];
    }
    int32_t best = H[n2], b1 = n1, b2 = n2;
    if (flags & STRK_O_S1_END_FREE) {
        int32_t cb = lastcol[1], ci = 1;
        for (int32_t i = 2; i <= n1; i++)
            if (lastcol[i] > cb) { cb = lastcol[i]; ci = i; }
        if (cb > best || (cb == best && ci < b1)) { best = cb; b1 = ci; b2 = n2; }
    }
    if (flags & STRK_O_S2_END_FREE) {
        for (int32_t j = 1; j <= n2; j++)
            if (H[j] > best) { best = H[j]; b1 = n1; b2 = j; }
    }
    if (end1) *end1 = b1 - 1;
    if (end2) *end2 = b2 - 1;
    free(H);
    free(F);
    free(lastcol);
    return best;
}

/* ---- candidate scoring (read side) ------------------------------------------------------
 * One alignment per candidate size i: (fl + motif*i + fr) against (fl + tr + fr), linear gap
 * 5 per base (align_matrix.py:17, passed as open=extend at repeats.py:33), all four end gaps
 * free (plain parasail "sg"; the read-side mode is inside strkit_rust_ext and unverifiable —
 * it is the `flags` argument so tests can record the alternatives). */
typedef struct {
    const uint8_t *tr, *fl, *fr, *motif;
    int32_t ntr, nfl, nfr, m;
    uint8_t* db;
    int32_t ndb;
    int32_t flags;
    int64_t cells; /* DP cells evaluated, for GCUPS reporting */
    /* inter-sequence SIMD (strk_simd.c): the scores of sixteen consecutive sizes from one pass, kept per read */
    int32_t simd_lo, simd_ok;
    int32_t simd_scores[16];
} cand_ctx;

/* strk_simd.c */
int strk_o_simd_available(void);
int64_t strk_o_simd_scores16(const uint8_t* db, int32_t ndb, const uint8_t* fl, int32_t nfl, const uint8_t* fr, int32_t nfr,
                             const uint8_t* motif, int32_t m, int32_t lo, int32_t* scores);
static int g_simd = 0;
/* 1: candidate scores come from the AVX2 inter-sequence pass where it applies (all end gaps free, scores within int16,
 * a CPU with AVX2); results are identical, only the cost differs.  Returns what is in effect. */
int strk_o_set_simd(int on) {
    g_simd = on && strk_o_simd_available();
    return g_simd;
}

static int32_t score_candidate(cand_ctx* c, int32_t i) {
    if (g_simd && c->flags == STRK_O_SG_ALL && i >= 0) {
        if (!c->simd_ok || i < c->simd_lo || i >= c->simd_lo + 16) {
            const int32_t lo = i - 4 > 0 ? i - 4 : 0;   /* the search asks for start-3 first and then climbs either way */
            const int64_t cells = strk_o_simd_scores16(c->db, c->ndb, c->fl, c->nfl, c->fr, c->nfr, c->motif, c->m, lo, c->simd_scores);
            if (cells >= 0) {
                c->simd_lo = lo;
                c->simd_ok = 1;
                c->cells += cells;
            } else {
                c->simd_ok = 0;
            }
        }
        if (c->simd_ok && i >= c->simd_lo && i < c->simd_lo + 16) return c->simd_scores[i - c->simd_lo];
    }
    int32_t nq = c->nfl + i * c->m + c->nfr;
    uint8_t* q = (uint8_t*)malloc((size_t)(nq > 0 ? nq : 1));
    memcpy(q, c->fl, (size_t)c->nfl);
    for (int32_t k = 0; k < i; k++) memcpy(q + c->nfl + k * c->m, c->motif, (size_t)c->m);
    memcpy(q + c->nfl + i * c->m, c->fr, (size_t)c->nfr);
    /* profile side (s1) = db sequence, other side (s2) = candidate (repeats.py:92, :33) */
    int32_t s = strk_o_sg_align(c->db, c->ndb, q, nq, 5, 5, c->flags, NULL, NULL);
    c->cells += (int64_t)nq * c->ndb;
    free(q);
    return s;
}

int32_t strk_o_candidate_score(const uint8_t* tr, int32_t ntr, const uint8_t* fl, int32_t nfl,
                               const uint8_t* fr, int32_t nfr, const uint8_t* motif, int32_t m,
                               int32_t i, int32_t flags) {
    cand_ctx c = {tr, fl, fr, motif, ntr, nfl, nfr, m, NULL, nfl + ntr + nfr, flags, 0, 0, 0, {0}};
    c.db = (uint8_t*)malloc((size_t)(c.ndb > 0 ? c.ndb : 1));
    memcpy(c.db, fl, (size_t)nfl);
    memcpy(c.db + nfl, tr, (size_t)ntr);
    memcpy(c.db + nfl + ntr, fr, (size_t)nfr);
    int32_t s = score_candidate(&c, i);
    free(c.db);
    return s;
}

/* insertion-ordered {size: score} map (Python dict semantics; tiny, linear probe) */
typedef struct {
    int32_t *k, *v;
    int32_t n, cap;
} omap;
static int32_t omap_find(const omap* m, int32_t key) {
    for (int32_t i = 0; i < m->n; i++)
        if (m->k[i] == key) return i;
    return -1;
}
static void omap_put(omap* m, int32_t key, int32_t val) {
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->k = (int32_t*)realloc(m->k, sizeof(int32_t) * (size_t)m->cap);
        m->v = (int32_t*)realloc(m->v, sizeof(int32_t) * (size_t)m->cap);
    }
    m->k[m->n] = key;
    m->v[m->n] = val;
    m->n++;
}

/*
 * Read-side hill-climb.  Control flow transcribed from repeats.py:100-151 with a single
 * {size: score} map, return contract from repeats.py:55-56:
 *   ((best size, best score), n_explored, best size - start_count).
 * Defaults (none confirmable from /root/reference): n_explored counts newly
 * scored sizes (repeats.py:130); window rule repeats.py:114-117; tie rule first-max.
 * Returns 0, or -1 if nothing could be scored (Python's max() of an empty dict would raise).
 *
 * `tie_rule` is a rule word: bit 0 = STRK_O_TIE_LAST; bits 8-9 = how local_search_range changes inside the search — the
 * reference calls it an INITIAL value that "can be narrowed within the get_repeat_count fn" (repeat_count_params.py:14) and
 * the function is not in the tree.  0: fixed (the in-tree sibling's way, the default); 1: one less after every explored stack
 * entry, never below 1; 2: halved after every explored entry, never below 1; 3: the three seed entries use it as given, every
 * chased entry 1.  An explored entry is one popped with size >= 0.  (Same modes as the product's STRK_NARROW_*.)
 */
#define STRK_O_NARROW_OF(rule) (((rule) >> 8) & 3)
int strk_o_repeat_count(int32_t start_count, const uint8_t* tr, int32_t ntr, const uint8_t* fl,
                        int32_t nfl, const uint8_t* fr, int32_t nfr, const uint8_t* motif,
                        int32_t m, int32_t max_iters, int32_t lsr, int32_t step, int32_t tie_rule,
                        int32_t flags, int32_t* out_cn, int32_t* out_score, int32_t* out_n,
                        int64_t* out_cells) {
    cand_ctx c = {tr, fl, fr, motif, ntr, nfl, nfr, m, NULL, nfl + ntr + nfr, flags, 0, 0, 0, {0}};
    c.db = (uint8_t*)malloc((size_t)(c.ndb > 0 ? c.ndb : 1));
    memcpy(c.db, fl, (size_t)nfl);
    memcpy(c.db + nfl, tr, (size_t)ntr);
    memcpy(c.db + nfl + ntr, fr, (size_t)nfr);

    omap map = {0, 0, 0, 0};
    /* list used as a stack: pop() takes the LAST element, so (start, 0) goes first */
    int32_t st_size[8 + 64], st_dir[8 + 64];
    int32_t sp = 0;
    st_size[sp] = start_count - step; st_dir[sp++] = -1;
    st_size[sp] = start_count + step; st_dir[sp++] = 1;
    st_size[sp] = start_count;        st_dir[sp++] = 0;
    int32_t n = 0;
    const int32_t narrow = STRK_O_NARROW_OF(tie_rule), lsr_given = lsr, lsr_floor = lsr < 1 ? lsr : 1;
    int32_t lsr_cur = lsr, seeds = 3;   /* stack entries below depth `seeds` are the initial three */
    tie_rule &= 1;
    while (sp > 0 && n < max_iters) {
        sp--;
        int32_t size = st_size[sp], dir = st_dir[sp];
        const int is_seed = sp < seeds;
        if (is_seed) seeds = sp;
        if (size < 0) continue;
        lsr = narrow == 3 ? (is_seed ? lsr_given : lsr_floor) : lsr_cur;
        if (narrow == 1) lsr_cur = lsr_cur - 1 > lsr_floor ? lsr_cur - 1 : lsr_floor;
        if (narrow == 2) lsr_cur = (lsr_cur >> 1) > lsr_floor ? (lsr_cur >> 1) : lsr_floor;
        int32_t lo = size - ((dir < 1 || step > lsr) ? lsr : 0);
        if (lo < 0) lo = 0;
        int32_t hi = size + ((dir > -1 || step > lsr) ? lsr : 0);
        int32_t mv_i = -1, mv_s = 0;
        for (int32_t i = lo; i <= hi; i++) {
            int32_t at = omap_find(&map, i);
            int32_t s;
            if (at < 0) {
                s = score_candidate(&c, i);
                omap_put(&map, i, s);
                n++;
            } else {
                s = map.v[at];
            }
            if (mv_i < 0 || s > mv_s || (tie_rule == STRK_O_TIE_LAST && s == mv_s)) {
                mv_i = i;
                mv_s = s;
            }
        }
        int32_t nr;
        if (mv_i > size && omap_find(&map, (nr = mv_i + step)) < 0 && nr >= 0) {
            st_size[sp] = nr; st_dir[sp++] = 1;
        }
        if (mv_i < size && omap_find(&map, (nr = mv_i - step)) < 0 && nr >= 0) {
            st_size[sp] = nr; st_dir[sp++] = -1;
        }
    }
    int rc = 0;
    if (map.n == 0) {
        rc = -1;
    } else {
        int32_t bi = 0;
        for (int32_t i = 1; i < map.n; i++)
            if (map.v[i] > map.v[bi] || (tie_rule == STRK_O_TIE_LAST && map.v[i] == map.v[bi])) bi = i;
        *out_cn = map.k[bi];
        *out_score = map.v[bi];
    }
    *out_n = n;
    if (out_cells) *out_cells = c.cells;
    free(map.k);
    free(map.v);
    free(c.db);
    return rc;
}

/* Python round(): round-half-to-even on the exact double value. */
static int64_t py_round(double x) { return (int64_t)nearbyint(x); }

/*
 * Caller protocol for one locus (call_locus.py:1079,1125-1161): reads are visited in order,
 * read_sc = est_cn + round(frac * est_cn) unless that offset < -est_cn (then frac := 0 and the
 * bare estimate is used); after the call frac += new_offset / max(read_cn, 1)  (float64).
 * Inputs are CSR-packed: read r has bases seqs[off[r] .. off[r+1]) laid out fl|tr|fr.
 */
int strk_o_count_locus(int32_t n_reads, const uint8_t* seqs, const int64_t* off, const int32_t* nfl,
                       const int32_t* ntr, const int32_t* nfr, const int32_t* est_cn,
                       const uint8_t* motif, int32_t m, int32_t max_iters, int32_t lsr, int32_t step,
                       int32_t tie_rule, int32_t flags, int32_t feedback, int32_t memo, int32_t* out_cn,
                       int32_t* out_score, int32_t* out_n, int32_t* out_start, int64_t* out_cells) {
    double frac = 0.0;
    int64_t cells = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        const uint8_t* base = seqs + off[r];
        int64_t read_sc = est_cn[r];
        if (feedback) {
            int64_t o = py_round(frac * (double)read_sc);
            if (o < -read_sc) frac = 0.0;
            else read_sc += o;
        }
        int32_t cn = 0, sc = 0, n = 0;
        int64_t cl = 0;
        int hit = 0;
        if (memo) {
            /* the reference memoises get_repeat_count on its full argument tuple with
             * functools.lru_cache (repeats.py:47): an identical earlier call of this locus is a hit */
            for (int32_t q = 0; q < r && !hit; q++) {
                if (out_start[q] != (int32_t)read_sc || nfl[q] != nfl[r] || ntr[q] != ntr[r] || nfr[q] != nfr[r]) continue;
                if (memcmp(seqs + off[q], base, (size_t)(nfl[r] + ntr[r] + nfr[r])) != 0) continue;
                cn = out_cn[q]; sc = out_score[q]; n = out_n[q];
                hit = 1;
            }
        }
        if (!hit) {
            int rc = strk_o_repeat_count((int32_t)read_sc, base + nfl[r], ntr[r], base, nfl[r],
                                         base + nfl[r] + ntr[r], nfr[r], motif, m, max_iters, lsr, step,
                                         tie_rule, flags, &cn, &sc, &n, &cl);
            if (rc) return rc;
        }
        cells += cl;
        out_cn[r] = cn;
        out_score[r] = sc;
        out_n[r] = n;
        out_start[r] = (int32_t)read_sc;
        int32_t new_off = cn - (int32_t)read_sc;
        frac += (double)new_off / (double)(cn > 1 ? cn : 1);
    }
    if (out_cells) *out_cells = cells;
    return 0;
}

/* ---- reference side (repeats.py:23-43) ---------------------------------------------------
 * Two sg_qe alignments per candidate: profile side (s1) = db_seq resp. reversed db_seq with
 * only its END free; other side = fl+cand resp. reversed(cand+fr), aligned globally.
 * adj = end_query + 1 - len(flank) - ref_size. */
static void reverse_into(uint8_t* dst, const uint8_t* src, int32_t n) {
    for (int32_t i = 0; i < n; i++) dst[i] = src[n - 1 - i];
}

void strk_o_score_ref_boundaries(const uint8_t* db, int32_t ndb, const uint8_t* fl, int32_t nfl,
                                 const uint8_t* fr, int32_t nfr, const uint8_t* motif, int32_t m,
                                 int32_t i, int32_t ref_size, int32_t* out4) {
    int32_t ncand = i * m;
    uint8_t* dbr = (uint8_t*)malloc((size_t)(ndb > 0 ? ndb : 1));
    reverse_into(dbr, db, ndb);
    uint8_t* ext_r = (uint8_t*)malloc((size_t)(nfl + ncand + 1));
    memcpy(ext_r, fl, (size_t)nfl);
    for (int32_t k = 0; k < i; k++) memcpy(ext_r + nfl + k * m, motif, (size_t)m);
    int32_t e1 = -1;
    int32_t s_fwd = strk_o_sg_align(db, ndb, ext_r, nfl + ncand, 5, 5, STRK_O_S1_END_FREE, &e1, NULL);
    out4[0] = s_fwd;
    out4[1] = e1 + 1 - nfl - ref_size;
    uint8_t* tmp = (uint8_t*)malloc((size_t)(ncand + nfr + 1));
    for (int32_t k = 0; k < i; k++) memcpy(tmp + k * m, motif, (size_t)m);
    memcpy(tmp + ncand, fr, (size_t)nfr);
    uint8_t* ext_l = (uint8_t*)malloc((size_t)(ncand + nfr + 1));
    reverse_into(ext_l, tmp, ncand + nfr);
    int32_t e2 = -1;
    int32_t s_rev = strk_o_sg_align(dbr, ndb, ext_l, ncand + nfr, 5, 5, STRK_O_S1_END_FREE, &e2, NULL);
    out4[2] = s_rev;
    out4[3] = e2 + 1 - nfr - ref_size;
    free(dbr);
    free(ext_r);
    free(tmp);
    free(ext_l);
}

/*
 * get_ref_repeat_count (repeats.py:73-192).  Outputs: final (cn, score), l_offset, r_offset,
 * (n_offset_scores, n_iters_final) and the adjusted split lengths (new nfl, ntr, nfr — the
 * bases themselves never change, only where the flank/TR boundaries fall).
 * tr is passed as-is to the boundary search (ref FASTA may be soft-masked lower case; the
 * encoder is case-insensitive) and the final count upper-cases it (repeats.py:183), a no-op
 * for this restatement for the same reason.
 */
int strk_o_ref_repeat_count(int32_t start_count, const uint8_t* tr, int32_t ntr, const uint8_t* fl,
                            int32_t nfl, const uint8_t* fr, int32_t nfr, const uint8_t* motif,
                            int32_t m, int32_t ref_size, int32_t vcf_anchor_size, int32_t max_iters,
                            int32_t lsr, int32_t step, int32_t respect_coords, int32_t tie_rule,
                            int32_t flags, int32_t* out /* cn, score, l_off, r_off, n_off_scores,
                            n_iters_final, new_nfl, new_ntr, new_nfr */) {
    int32_t ndb = nfl + ntr + nfr;
    uint8_t* db = (uint8_t*)malloc((size_t)(ndb > 0 ? ndb : 1));
    memcpy(db, fl, (size_t)nfl);
    memcpy(db + nfl, tr, (size_t)ntr);
    memcpy(db + nfl + ntr, fr, (size_t)nfr);
    int32_t l_offset = 0, r_offset = 0, n_off = 0;
    /* the boundary search below IS the in-tree code (fixed range): only the final read-side call takes the whole rule word */
    const int32_t rule_word = tie_rule;
    tie_rule &= 1;

    if (!respect_coords) {
        /* fwd / rev maps are always filled together (repeats.py:123-128) so one key list */
        omap fs = {0, 0, 0, 0}, fa = {0, 0, 0, 0}, rs = {0, 0, 0, 0}, ra = {0, 0, 0, 0};
        int32_t cap = 16 + 2 * (max_iters + 2 * lsr + 4);
        int32_t* st_size = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
        int32_t* st_dir = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
        int32_t sp = 0;
        st_size[sp] = start_count - step; st_dir[sp++] = -1;
        st_size[sp] = start_count + step; st_dir[sp++] = 1;
        st_size[sp] = start_count;        st_dir[sp++] = 0;
        while (sp > 0 && n_off < max_iters) {
            sp--;
            int32_t size = st_size[sp], dir = st_dir[sp];
            if (size < 0) continue;
            int32_t lo = size - ((dir < 1 || step > lsr) ? lsr : 0);
            if (lo < 0) lo = 0;
            int32_t hi = size + ((dir > -1 || step > lsr) ? lsr : 0);
            /* mv = max((*fwd_scores, *rev_scores), key=(score, adj)) — first maximal element of
             * the concatenation: all fwd entries (ascending i) then all rev entries. */
            int32_t mv_i = -1, mv_s = 0, mv_a = 0;
            for (int pass = 0; pass < 2; pass++) {
                for (int32_t i = lo; i <= hi; i++) {
                    int32_t at = omap_find(&fs, i);
                    if (at < 0) { /* only reachable in pass 0 */
                        int32_t r4[4];
                        strk_o_score_ref_boundaries(db, ndb, fl, nfl, fr, nfr, motif, m, i, ref_size, r4);
                        omap_put(&fs, i, r4[0]); omap_put(&fa, i, r4[1]);
                        omap_put(&rs, i, r4[2]); omap_put(&ra, i, r4[3]);
                        n_off++;
                        at = fs.n - 1;
                    }
                    int32_t s = pass == 0 ? fs.v[at] : rs.v[at];
                    int32_t a = pass == 0 ? fa.v[at] : ra.v[at];
                    if (mv_i < 0 || s > mv_s || (s == mv_s && a > mv_a)) { mv_i = i; mv_s = s; mv_a = a; }
                }
            }
            int32_t nr;
            if (mv_i > size && omap_find(&fs, (nr = mv_i + step)) < 0 && nr >= 0) {
                st_size[sp] = nr; st_dir[sp++] = 1;
            }
            if (mv_i < size && omap_find(&fs, (nr = mv_i - step)) < 0 && nr >= 0) {
                st_size[sp] = nr; st_dir[sp++] = -1;
            }
        }
        if (fs.n == 0) {
            free(fs.k); free(fs.v); free(fa.k); free(fa.v); free(rs.k); free(rs.v); free(ra.k); free(ra.v);
            free(st_size); free(st_dir); free(db);
            return -1;
        }
        int32_t bf = 0, br = 0;
        for (int32_t i = 1; i < fs.n; i++) {
            if (fs.v[i] > fs.v[bf]) bf = i;
            if (rs.v[i] > rs.v[br]) br = i;
        }
        l_offset = ra.v[br];
        r_offset = fa.v[bf];
        if (l_offset >= nfl - vcf_anchor_size) l_offset = 0;
        if (r_offset >= nfr) r_offset = 0;
        free(fs.k); free(fs.v); free(fa.k); free(fa.v); free(rs.k); free(rs.v); free(ra.k); free(ra.v);
        free(st_size); free(st_dir);
    }
    int32_t lo_pos = l_offset > 0 ? l_offset : 0, ro_pos = r_offset > 0 ? r_offset : 0;
    int32_t nfl2 = nfl - lo_pos, ntr2 = ntr + lo_pos + ro_pos, nfr2 = nfr - ro_pos;
    /* round(((start*m) + max(0,l) + max(0,r)) / m) — Python float division then round() */
    int32_t start2 = (int32_t)py_round(((double)((int64_t)start_count * m + lo_pos + ro_pos)) / (double)m);
    int32_t cn = 0, sc = 0, n = 0;
    int rc = strk_o_repeat_count(start2, db + nfl2, ntr2, db, nfl2, db + nfl2 + ntr2, nfr2, motif, m,
                                 max_iters, lsr, step, rule_word, flags, &cn, &sc, &n, NULL);
    out[0] = cn; out[1] = sc; out[2] = l_offset; out[3] = r_offset; out[4] = n_off; out[5] = n;
    out[6] = nfl2; out[7] = ntr2; out[8] = nfr2;
    free(db);
    return rc;
}

/* ---- realignment (strkit/call/realign.py:56-72) ------------------------------------------
 * parasail sg_dx_trace_scan_16(s1 = reference window, s2 = wildcarded read, open = 7, extend = 0,
 * dna_matrix): s1 is aligned end to end, leading and trailing s2 bases are free.  Gap of length
 * k costs open + (k-1)*extend (so with extend = 0 every gap costs 7).  The recurrence is the one
 * of strk_o_sg_align; this function also keeps the trace-back tables and emits the CIGAR the
 * reference reads as pr.cigar.seq (BAM encoding len<<4|op, ops from "MIDNSHP=X": I = 1 consumes
 * s1 only, D = 2 consumes s2 only, '=' = 7, 'X' = 8), s1 as "query", s2 as "ref".
 *
 * Choices parasail leaves to its implementation (its source is not under /root/reference — the
 * rules below restate its trace kernels from memory and are named so they can be flipped):
 *   - H tie:   diagonal first, then (gap_pref == 0) the gap that consumes s1 ('I'), then the gap
 *              that consumes s2 ('D');  gap_pref == 1 swaps the two gap kinds
 *   - gap tie: extension wins over opening (open only when strictly greater)
 *   - end:     smallest s2 end position among the maxima of the last row
 *   - '='/'X': '=' when both bases encode to the same alphabet letter (case-insensitive)
 * The CIGAR starts at s2 position 0: the free leading s2 bases appear as one D run (the
 * reference passes ref_start = 0 to get_aligned_pair_matches, realign.py:71); trailing free s2
 * bases are not part of it.  The 16-bit saturating arithmetic of the _16 kernel is not modelled:
 * scores here are exact int32 (they fit int16 whenever 2*n1 < 32 767).
 *
 * Returns the number of CIGAR runs written (<= cap), or -1 if cap is too small, -2 on empty input.
 */
#define TB_H_DIAG 2
#define TB_H_GI 1 /* gap consuming s1: 'I' */
#define TB_H_GD 0 /* gap consuming s2: 'D' */
#define TB_GI_EXT 4
#define TB_GD_EXT 8

int32_t strk_o_realign(const uint8_t* s1, int32_t n1, const uint8_t* s2, int32_t n2, int32_t open,
                       int32_t ext, int32_t gap_pref, int32_t* out_score, int32_t* out_end2,
                       uint32_t* cigar, int32_t cap) {
    strk_o_init();
    if (n1 <= 0 || n2 <= 0) return -2;
    const int32_t NEG = INT32_MIN / 4;
    const size_t W = (size_t)n2 + 1;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * W);
    int32_t* GI = (int32_t*)malloc(sizeof(int32_t) * W);           /* per column j, runs along i */
    uint8_t* T = (uint8_t*)malloc((size_t)(n1 + 1) * W);
    H[0] = 0;
    for (int32_t j = 1; j <= n2; j++) { H[j] = 0; GI[j] = NEG; }
    for (int32_t i = 1; i <= n1; i++) {
        const int8_t* wrow = g_mat[g_enc[s1[i - 1]]];
        uint8_t* trow = T + (size_t)i * W;
        int32_t diag = H[0];
        H[0] = -(open + (i - 1) * ext);
        int32_t GD = NEG;                                           /* runs along j */
        for (int32_t j = 1; j <= n2; j++) {
            uint8_t t = 0;
            int32_t d_ext = GD - ext, d_opn = H[j - 1] - open;
            if (d_opn > d_ext) GD = d_opn; else { GD = d_ext; t |= TB_GD_EXT; }
            int32_t i_ext = GI[j] - ext, i_opn = H[j] - open;
            if (i_opn > i_ext) GI[j] = i_opn; else { GI[j] = i_ext; t |= TB_GI_EXT; }
            int32_t h = diag + wrow[g_enc[s2[j - 1]]];
            int32_t first = gap_pref ? GD : GI[j], second = gap_pref ? GI[j] : GD;
            uint8_t tf = gap_pref ? TB_H_GD : TB_H_GI, ts = gap_pref ? TB_H_GI : TB_H_GD;
            uint8_t th = TB_H_DIAG;
            if (first > h) { h = first; th = tf; }
            if (second > h) { h = second; th = ts; }
            diag = H[j];
            H[j] = h;
            trow[j] = (uint8_t)(t | th);
        }
    }
    int32_t best = H[1], bj = 1;
    for (int32_t j = 2; j <= n2; j++)
        if (H[j] > best) { best = H[j]; bj = j; }
    if (out_score) *out_score = best;
    if (out_end2) *out_end2 = bj - 1;
    /* trace-back, runs collected in reverse */
    int32_t n = 0, rc = 0;
    uint32_t* rev = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(2 * n1 + 4));
    int32_t i = n1, j = bj, where = TB_H_DIAG;
#define PUSH(op)                                                          \
    do {                                                                  \
        if (n > 0 && (rev[n - 1] & 15u) == (uint32_t)(op)) rev[n - 1] += 16u; \
        else rev[n++] = 16u | (uint32_t)(op);                             \
    } while (0)
    while (i > 0 || j > 0) {
        if (i == 0) { PUSH(2); j--; continue; }
        if (j == 0) { PUSH(1); i--; continue; }
        const uint8_t t = T[(size_t)i * W + j];
        if (where == TB_H_DIAG) {
            const int th = t & 3;
            if (th == TB_H_DIAG) {
                PUSH(g_enc[s1[i - 1]] == g_enc[s2[j - 1]] ? 7 : 8);
                i--; j--;
            } else where = th;
        } else if (where == TB_H_GI) {
            PUSH(1);
            if (!(t & TB_GI_EXT)) where = TB_H_DIAG;
            i--;
        } else {
            PUSH(2);
            if (!(t & TB_GD_EXT)) where = TB_H_DIAG;
            j--;
        }
    }
#undef PUSH
    if (n > cap) rc = -1;
    else {
        for (int32_t k = 0; k < n; k++) cigar[k] = rev[n - 1 - k];
        rc = n;
    }
    free(rev); free(T); free(GI); free(H);
    return rc;
}
