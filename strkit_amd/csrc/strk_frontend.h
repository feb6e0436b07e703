// strk_frontend.h — host-side (CPU) front end: BAM record scan and read -> (left flank, tract, right flank)
// extraction for blocks of (read, locus) pairs.  No device code.
//
// The reference does this in strkit_rust_ext (STRkitBAMReader, STRkitAlignedSegment.get_sequence_data_for_locus,
// get_read_coords_from_matched_pairs; call sites strkit/call/call_locus.py:837-958,1082-1146); the readable
// statement of the same rules is strkit_amd/frontend/extract.py, and tests/test_frontend.py checks this file
// against it record by record.
#pragma once
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include <zlib.h>

#include "strk_bamrec.h"

namespace strk_fe {

// Aligned (M, =, X) runs of one alignment: read position, reference position, length, index of the first pair.
struct Runs {
    std::vector<int64_t> q0, r0, ln, k0;
    int64_t n_pairs = 0;
    void build(const uint8_t* cigar, int32_t n_cigar, int64_t start) {
        q0.clear(); r0.clear(); ln.clear(); k0.clear();
        n_pairs = 0;
        int64_t q = 0, r = start;
        for (int32_t i = 0; i < n_cigar; ++i) {
            const uint32_t c = rd_u32(cigar + 4 * (size_t)i), op = c & 15u;
            const int64_t len = c >> 4;
            if (is_aligned(op) && len > 0) {
                q0.push_back(q); r0.push_back(r); ln.push_back(len); k0.push_back(n_pairs);
                n_pairs += len;
            }
            if (consumes_query(op)) q += len;
            if (consumes_ref(op)) r += len;
        }
    }
    // index of the first aligned pair whose reference coordinate is >= c (n_pairs if none)
    int64_t first_pair_at_or_after(int64_t c) const {
        size_t lo = 0, hi = ln.size();
        while (lo < hi) {   // first run that ends past c
            const size_t mid = (lo + hi) / 2;
            if (r0[mid] + ln[mid] > c) hi = mid; else lo = mid + 1;
        }
        if (lo >= ln.size()) return n_pairs;
        return k0[lo] + (c > r0[lo] ? c - r0[lo] : 0);
    }
    int64_t query_at(int64_t k) const {
        size_t lo = 0, hi = k0.size();
        while (lo < hi) {   // last run with k0 <= k
            const size_t mid = (lo + hi) / 2;
            if (k0[mid] <= k) lo = mid + 1; else hi = mid;
        }
        const size_t i = lo - 1;
        return q0[i] + (k - k0[i]);
    }
};

// Read positions of the four locus boundaries (extract.py: get_read_coords_from_cigar).  false = incomplete.
inline bool read_coords(const Runs& ix, int64_t lfc, int64_t lc, int64_t rc, int64_t rfc, int64_t out[4]) {
    const int64_t n = ix.n_pairs;
    if (n == 0) return false;
    // call_locus.py:907-909 skips a read when left_flank_coord < segment.start or right_flank_coord >= segment.end:
    // the last aligned reference base (end - 1) must be at or right of right_flank_coord
    const bool full_l = ix.r0.front() <= lfc, full_r = ix.r0.back() + ix.ln.back() - 1 >= rfc;
    if (!(full_l && full_r)) return false;
    const int64_t i_lfs = ix.first_pair_at_or_after(lfc), i_l = ix.first_pair_at_or_after(lc);
    const int64_t i_r = ix.first_pair_at_or_after(rc), i_rfe = ix.first_pair_at_or_after(rfc);
    if (i_l == 0 || i_r >= n) return false;
    out[0] = ix.query_at(i_lfs < n - 1 ? i_lfs : n - 1);
    out[1] = ix.query_at(i_l - 1) + 1;     // bases inserted at a tract boundary belong to the tract
    out[2] = ix.query_at(i_r);
    out[3] = i_rfe < n ? ix.query_at(i_rfe) : ix.query_at(n - 1) + 1;
    if (out[1] > out[2]) out[2] = out[1];
    return true;
}


// ---- BGZF (the block-gzip container of BAM): independent deflate blocks, so they inflate in parallel ----------------
struct BgzfBlock {
    int64_t in_off;    // start of the deflate payload
    int32_t in_len;    // payload bytes
    int32_t out_len;   // ISIZE
    uint32_t crc;
    int64_t out_off;
};

// Walks the block headers.  Returns 0 and fills `blocks`, or -1 if the stream is not BGZF / is truncated.
inline int bgzf_index(const uint8_t* p, int64_t n, std::vector<BgzfBlock>* blocks, int64_t* total) {
    int64_t off = 0, out = 0;
    while (off < n) {
        if (off + 18 > n || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8 || !(p[off + 3] & 4)) return -1;
        const int xlen = rd_u16(p + off + 10);
        int64_t x = off + 12;
        const int64_t xend = x + xlen;
        int bsize = -1;
        if (xend > n) return -1;
        while (x + 4 <= xend) {
            const int slen = rd_u16(p + x + 2);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = rd_u16(p + x + 4);
            x += 4 + slen;
        }
        if (bsize < 0) return -1;
        const int64_t next = off + bsize + 1;
        if (next > n || next - 8 < xend) return -1;
        BgzfBlock b;
        b.in_off = xend;
        b.in_len = (int32_t)(next - 8 - xend);
        b.crc = rd_u32(p + next - 8);
        b.out_len = (int32_t)rd_u32(p + next - 4);
        b.out_off = out;
        if (b.out_len < 0 || b.out_len > 65536) return -1;
        out += b.out_len;
        blocks->push_back(b);
        off = next;
    }
    *total = out;
    return 0;
}

inline bool bgzf_inflate_block(const uint8_t* p, const BgzfBlock& b, uint8_t* out) {
    if (b.out_len == 0) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(p + b.in_off);
    zs.avail_in = (uInt)b.in_len;
    zs.next_out = out + b.out_off;
    zs.avail_out = (uInt)b.out_len;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == (uLong)b.out_len;
    inflateEnd(&zs);
    return ok && crc32(crc32(0L, Z_NULL, 0), out + b.out_off, (uInt)b.out_len) == b.crc;
}

}  // namespace strk_fe
