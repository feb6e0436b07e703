// strk_scoring.h — substitution matrix and base encoding of the repeat-count path.
//
// Follows strkit/call/align_matrix.py:15-44 (alphabet order, match/mismatch/indel constants,
// IUPAC and 'X' overrides) and strkit/iupac.py:9-21 (code -> member bases, including the
// reference's "D" = (A, C, T) entry).  parasail's matrix_create adds an implicit '*' row/column of
// zeros that every byte outside the alphabet maps to, case-insensitively (parasail is not
// vendored in the reference tree; see DESIGN.md "parity unpinned").
#pragma once
#include <stdint.h>
#include <string.h>

namespace strk {

constexpr int kMatch = 2;       // align_matrix.py:15
constexpr int kMismatch = -7;   // align_matrix.py:16
constexpr int kGap = 5;         // align_matrix.py:17 (passed as open == extend, repeats.py:33)
constexpr int kNSym = 17;       // 16 letters + '*'
constexpr int kStar = 16;
constexpr int kNullSym = 17;    // kernel-internal "no row" symbol: scores 0 in G-space
constexpr int kWBias = 2 * kGap;  // G-space diagonal bias: w' = W + 2*gap, always in [0, 255]

struct ScoreTables {
    int8_t mat[kNSym][kNSym];
    uint8_t enc[256];
};

inline void build_score_tables(ScoreTables* t) {
    static const char alphabet[] = "ACGTRYSWKMBDHVNX";  // align_matrix.py:25
    static const struct { char code; const char* members; } codes[] = {
        {'R', "AG"}, {'Y', "CT"}, {'S', "CG"}, {'W', "AT"}, {'K', "GT"}, {'M', "AC"},
        {'B', "CGT"}, {'D', "ACT"} /* iupac.py:17 */, {'H', "ACT"}, {'V', "ACG"}, {'N', "ACGT"},
        {'X', "ACGT"} /* align_matrix.py:29 */};
    memset(t->enc, kStar, sizeof t->enc);
    for (int i = 0; i < 16; i++) {
        t->enc[(unsigned char)alphabet[i]] = (uint8_t)i;
        t->enc[(unsigned char)(alphabet[i] + ('a' - 'A'))] = (uint8_t)i;
    }
    for (int i = 0; i < kNSym; i++)
        for (int j = 0; j < kNSym; j++)
            t->mat[i][j] = (int8_t)((i == kStar || j == kStar) ? 0 : (i == j ? kMatch : kMismatch));
    for (const auto& c : codes) {
        const int ci = t->enc[(unsigned char)c.code];
        const int8_t v = (int8_t)(c.code == 'X' ? 0 : 2);  // align_matrix.py:38-39
        for (const char* m = c.members; *m; ++m) {
            const int bi = t->enc[(unsigned char)*m];
            t->mat[ci][bi] = v;
            t->mat[bi][ci] = v;
        }
    }
}

}  // namespace strk
