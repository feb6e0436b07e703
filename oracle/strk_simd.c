/*
 * oracle/strk_simd.c — inter-sequence AVX2 (16 x int16) variant of the CPU restatement's candidate scoring.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of strk_oracle.c): used by bench.py's cpu_baseline leg as the "simd" CPU
 * baseline ("inter-sequence SIMD variant, parasail-class") and by tests/test_oracle.py, which checks it
 * against the scalar restatement.  The reference scores one candidate per parasail call (striped intra-sequence SIMD,
 * repeats.py:33,40,92-93); here the sixteen lanes of a 256-bit register hold SIXTEEN CANDIDATE SIZES of one read
 * (fl + motif*i + fr for i = lo .. lo+15) against the same window fl+tr+fr, so one pass over the matrix gives the whole
 * score window the search needs.  Same recurrence (linear gap 5 per base, dna_matrix, all four end gaps free — plain
 * parasail "sg"), exact while scores fit int16 (windows up to ~16 kb; the caller falls back to the scalar code otherwise).
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NSYM 17
extern const int8_t* strk_o_matrix(void);
extern int strk_o_encode(int c);

int strk_o_simd_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* scores[k] = sg score of (fl + motif*(lo+k) + fr) against db = fl+tr+fr for k < 16 (sizes < 0 give INT32_MIN).
 * Returns the number of DP cells evaluated (16 lanes x rows x columns), or -1 when int16 cannot hold the scores. */
__attribute__((target("avx2")))
int64_t strk_o_simd_scores16(const uint8_t* db, int32_t ndb, const uint8_t* fl, int32_t nfl, const uint8_t* fr, int32_t nfr,
                             const uint8_t* motif, int32_t m, int32_t lo, int32_t* scores) {
    const int8_t* mat = strk_o_matrix();
    int32_t nq[16], max_nq = 0;
    for (int k = 0; k < 16; k++) {
        const int32_t i = lo + k;
        nq[k] = i < 0 ? 0 : nfl + i * m + nfr;
        if (nq[k] > max_nq) max_nq = nq[k];
    }
    if (2 * (int64_t)(ndb > max_nq ? ndb : max_nq) > 30000 || ndb < 1) return -1;
    /* encoded db and the distinct symbols in it */
    uint8_t* dbe = (uint8_t*)malloc((size_t)ndb);
    int present[NSYM] = {0}, slot[NSYM], n_slot = 0;
    for (int32_t j = 0; j < ndb; j++) { dbe[j] = (uint8_t)strk_o_encode(db[j]); present[dbe[j]] = 1; }
    for (int s = 0; s < NSYM; s++) slot[s] = present[s] ? n_slot++ : -1;
    /* per row: the substitution scores of the 16 lanes' symbols against every symbol the window holds */
    __m256i* prof = (__m256i*)aligned_alloc(32, sizeof(__m256i) * (size_t)(max_nq > 0 ? max_nq : 1) * (size_t)n_slot);
    __m256i* act = (__m256i*)aligned_alloc(32, sizeof(__m256i) * (size_t)(max_nq > 0 ? max_nq : 1));
    uint8_t fle[1], dummy = 0;
    (void)fle; (void)dummy;
    for (int32_t r = 0; r < max_nq; r++) {
        int16_t sym[16], on[16];
        for (int k = 0; k < 16; k++) {
            const int32_t i = lo + k;
            int c = NSYM - 1;
            on[k] = (int16_t)(r < nq[k] ? -1 : 0);
            if (r < nq[k]) {
                if (r < nfl) c = strk_o_encode(fl[r]);
                else if (r < nfl + i * m) c = strk_o_encode(motif[(r - nfl) % m]);
                else c = strk_o_encode(fr[r - nfl - i * m]);
            }
            sym[k] = (int16_t)c;
        }
        act[r] = _mm256_loadu_si256((const __m256i*)on);
        for (int s = 0; s < NSYM; s++) {
            if (slot[s] < 0) continue;
            int16_t w[16];
            for (int k = 0; k < 16; k++) w[k] = mat[sym[k] * NSYM + s];
            prof[(size_t)r * n_slot + slot[s]] = _mm256_loadu_si256((const __m256i*)w);
        }
    }
    __m256i* H = (__m256i*)aligned_alloc(32, sizeof(__m256i) * (size_t)(ndb + 1));
    for (int32_t j = 0; j <= ndb; j++) H[j] = _mm256_setzero_si256();        /* row 0: free start along the window */
    const __m256i gap = _mm256_set1_epi16(5);
    __m256i lastcol = _mm256_set1_epi16(-32000);                              /* max of H[r][ndb] over the lanes' own rows */
    for (int32_t r = 0; r < max_nq; r++) {
        const __m256i on = act[r];
        const __m256i* pr = prof + (size_t)r * n_slot;
        __m256i diag = H[0];                                                  /* H[r-1][0] = 0 (free start along the candidate) */
        __m256i left = _mm256_setzero_si256();                                /* H[r][0] = 0 */
        for (int32_t j = 1; j <= ndb; j++) {
            const __m256i up = H[j];
            __m256i h = _mm256_adds_epi16(diag, pr[slot[dbe[j - 1]]]);
            h = _mm256_max_epi16(h, _mm256_subs_epi16(_mm256_max_epi16(up, left), gap));
            h = _mm256_blendv_epi8(up, h, on);                                /* a lane past its last row keeps that row */
            diag = up;
            left = h;
            H[j] = h;
        }
        lastcol = _mm256_max_epi16(lastcol, _mm256_or_si256(_mm256_and_si256(on, left), _mm256_andnot_si256(on, lastcol)));
    }
    __m256i best = lastcol;
    /* free end along the window: the lanes' last rows (column 0 of the last row and row 0 of the last column are boundary
     * nodes, not alignment ends: the scalar restatement and parasail leave them out too) */
    for (int32_t j = 1; j <= ndb; j++) best = _mm256_max_epi16(best, H[j]);
    int16_t out[16];
    _mm256_storeu_si256((__m256i*)out, best);
    for (int k = 0; k < 16; k++) scores[k] = lo + k < 0 ? INT32_MIN : (nq[k] == 0 ? 0 : out[k]);
    free(H); free(act); free(prof); free(dbe);
    return 16 * (int64_t)max_nq * ndb;
}
