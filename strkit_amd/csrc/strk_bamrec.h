// strk_bamrec.h — BAM alignment record layout and the walk from a CIGAR to the four locus boundaries, for host and device.
//
// The reference has this in strkit_rust_ext (STRkitAlignedSegment, get_read_coords_from_matched_pairs; call sites
// strkit/call/call_locus.py:837-958,1082-1146); the readable statement of the rules is strkit_amd/frontend/extract.py.
// strk_frontend.h (host front end) and strk_dbam.inc (device front end) both build on these functions.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define STRK_FE_HD __host__ __device__ inline
#else
#define STRK_FE_HD inline
#endif

namespace strk_fe {

STRK_FE_HD int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
STRK_FE_HD uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
STRK_FE_HD uint16_t rd_u16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

STRK_FE_HD constexpr bool consumes_query(uint32_t op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
STRK_FE_HD constexpr bool consumes_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
STRK_FE_HD constexpr bool is_aligned(uint32_t op) { return op == 0 || op == 7 || op == 8; }

// Fixed part of a BAM alignment record (after block_size): refID, pos, l_read_name, mapq, bin, n_cigar_op, flag,
// l_seq, next_refID, next_pos, tlen = 32 bytes, then read_name, cigar, seq (4 bit), qual.
struct Rec {
    int32_t tid, pos, l_name, n_cigar, flag, l_seq;   // n_cigar / cigar: the real CIGAR (from the CG tag when the record holds the long-CIGAR placeholder)
    const uint8_t* name;
    const uint8_t* cigar;
    const uint8_t* seq;
    const uint8_t* qual;
};
STRK_FE_HD bool parse_rec(const uint8_t* buf, int64_t n_bytes, int64_t off, Rec* r, int64_t* next) {
    if (off + 4 > n_bytes) return false;
    const int32_t block = rd_i32(buf + off);
    if (block < 32 || off + 4 + block > n_bytes) return false;
    const uint8_t* p = buf + off + 4;
    r->tid = rd_i32(p);
    r->pos = rd_i32(p + 4);
    r->l_name = p[8];
    r->n_cigar = rd_u16(p + 12);
    r->flag = rd_u16(p + 14);
    r->l_seq = rd_i32(p + 16);
    const int64_t need = 32 + (int64_t)r->l_name + 4 * (int64_t)r->n_cigar + (r->l_seq + 1) / 2 + r->l_seq;
    if (r->l_seq < 0 || need > block) return false;
    r->name = p + 32;
    r->cigar = r->name + r->l_name;
    r->seq = r->cigar + 4 * (size_t)r->n_cigar;
    r->qual = r->seq + (r->l_seq + 1) / 2;
    *next = off + 4 + block;
    // Alignments with more than 65 535 CIGAR operations (ultralong reads) store the placeholder <l_seq>S<ref_len>N
    // and the real CIGAR in the tag CG:B,I (SAM specification, section 4.2.2).
    if (r->n_cigar == 2 && r->l_seq > 0) {
        const uint32_t c0 = rd_u32(r->cigar), c1 = rd_u32(r->cigar + 4);
        if ((c0 & 15u) == 4 && (int64_t)(c0 >> 4) == r->l_seq && (c1 & 15u) == 3) {
            const uint8_t* t = r->qual + r->l_seq;
            const uint8_t* const tend = p + block;
            while (t + 3 <= tend) {
                const char ty = (char)t[2];
                const uint8_t* v = t + 3;
                int64_t sz = -1;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') {
                    const uint8_t* z = v;
                    while (z < tend && *z) ++z;
                    if (z >= tend) return false;
                    sz = z - v + 1;
                } else if (ty == 'B') {
                    if (v + 5 > tend) return false;
                    const char sub = (char)v[0];
                    const int64_t cnt = rd_u32(v + 1);
                    const int64_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
                    sz = 5 + cnt * es;
                    if (t[0] == 'C' && t[1] == 'G' && sub == 'I' && v + sz <= tend) {
                        r->cigar = v + 5;
                        r->n_cigar = (int32_t)cnt;
                        break;
                    }
                }
                if (sz < 0 || v + sz > tend) return false;
                t = v + sz;
            }
        }
    }
    return true;
}

// Reference span and the two soft clips of an alignment (what the record scan keeps per record).
STRK_FE_HD void cigar_span(const uint8_t* cigar, int32_t n_cigar, int64_t* ref_len, int32_t* clip_l, int32_t* clip_r) {
    int64_t rl = 0;
    int32_t cl = 0, cr = 0;
    for (int32_t i = 0; i < n_cigar; ++i) {
        const uint32_t c = rd_u32(cigar + 4 * (size_t)i), op = c & 15u;
        if (consumes_ref(op)) rl += c >> 4;
        if (op == 4 && i == 0) cl = (int32_t)(c >> 4);
        if (op == 4 && i == n_cigar - 1) cr = (int32_t)(c >> 4);
    }
    *ref_len = rl; *clip_l = cl; *clip_r = cr;
}

// Read positions of the four locus boundaries (extract.py: get_read_coords_from_cigar) in ONE pass over the CIGAR, no
// per-alignment arrays: the same answers as strk_frontend.h's Runs::build + read_coords (tests/test_frontend.py compares
// them on random alignments), for code that cannot allocate (a GPU lane).  false = the read does not span the locus.
//   aligned pairs are numbered along the alignment; idx(c) = first pair whose reference coordinate is >= c;
//   out = { q(idx(lfc)), q(idx(lc) - 1) + 1, q(idx(rc)), q(idx(rfc)) or one past the last pair }.
STRK_FE_HD bool read_coords_linear(const uint8_t* cigar, int32_t n_cigar, int64_t start, int64_t lfc, int64_t lc, int64_t rc,
                                   int64_t rfc, int64_t out[4]) {
    const int64_t want[4] = {lfc, lc, rc, rfc};
    int64_t idx[4] = {-1, -1, -1, -1}, qat[4] = {0, 0, 0, 0};
    int64_t q_before_l = -1;        // query position of pair idx(lc) - 1
    int64_t q = 0, r = start, n_pairs = 0, first_r0 = 0, last_q = -1, last_r = -1;
    bool any = false;
    for (int32_t i = 0; i < n_cigar; ++i) {
        const uint32_t c = rd_u32(cigar + 4 * (size_t)i), op = c & 15u;
        const int64_t len = c >> 4;
        if (is_aligned(op) && len > 0) {
            if (!any) { first_r0 = r; any = true; }
            for (int t = 0; t < 4; ++t) {
                if (idx[t] < 0 && r + len > want[t]) {      // the first run that ends past want[t]
                    const int64_t d = want[t] > r ? want[t] - r : 0;
                    idx[t] = n_pairs + d;
                    qat[t] = q + d;
                    if (t == 1) q_before_l = d > 0 ? q + d - 1 : last_q;   // same run, or the last pair of the previous run
                }
            }
            n_pairs += len;
            last_q = q + len - 1;
            last_r = r + len - 1;
        }
        if (consumes_query(op)) q += len;
        if (consumes_ref(op)) r += len;
    }
    if (n_pairs == 0) return false;
    // call_locus.py:907-909 skips a read when left_flank_coord < segment.start or right_flank_coord >= segment.end:
    // the last aligned reference base must be at or right of right_flank_coord
    if (!(first_r0 <= lfc && last_r >= rfc)) return false;
    const int64_t i_l = idx[1] < 0 ? n_pairs : idx[1], i_r = idx[2] < 0 ? n_pairs : idx[2];
    if (i_l == 0 || i_r >= n_pairs) return false;
    out[0] = idx[0] >= 0 ? qat[0] : last_q;                     // min(idx, n - 1)
    out[1] = (idx[1] >= 0 ? q_before_l : last_q) + 1;           // bases inserted at a tract boundary belong to the tract
    out[2] = qat[2];
    out[3] = idx[3] >= 0 ? qat[3] : last_q + 1;
    if (out[1] > out[2]) out[2] = out[1];
    return true;
}

}  // namespace strk_fe
