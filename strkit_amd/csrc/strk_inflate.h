// strk_inflate.h — DEFLATE (RFC 1951) decoder for BGZF blocks, written to run one block per GPU lane.
//
// The reference reads its alignment files through htslib (pysam / strkit_rust_ext's STRkitBAMReader, call sites
// strkit/call/call_sample.py:103-121); BGZF — the container of BAM — is a series of independent gzip members of at most
// 64 KiB each (SAM specification section 4.1), so a file is tens of thousands of independent deflate streams: byte work
// that is bound by memory, not by arithmetic, and that HBM serves far faster than the host's cores serve zlib.
//
// Decoder, per stream (one lane):
//   * canonical Huffman decoding WITHOUT a per-bit loop and WITHOUT a big look-up table: the next 15 stream bits are
//     bit-reversed into a left-justified value v; the code length L is 1 + the number of per-length limits v has reached
//     (limit[L] = left-justified end of the codes of length L; fifteen compare-and-select steps against registers,
//     branch-free: each step that holds replaces a word that carries L, the base and the split of that length), the symbol
//     is sym[base[L] + (v >> (15 - L))].  Per-lane state: 2 x (15 limits + 16 words) in registers, the sorted symbol arrays
//     in a 320-byte slot of LDS (strk_inf::Tables) — one LDS read per symbol.  A lane's time per block does not depend on
//     how many other lanes run (it waits: for LDS, for its own stores), so the kernel's rate is the number of resident
//     lanes, which LDS sets: the symbols are kept as bytes; the ninth bit of a literal/length symbol is not stored, it
//     follows from the position (the symbols of one length are sorted, the literals come first: one split index per length).
//   * a 64-bit bit buffer refilled when fewer than 48 bits are left, from eight bytes that were loaded when the previous
//     refill moved the read position (the caller pads the input by 16 bytes): a refill never waits for memory;
//   * literals are collected in a register and stored eight at a time (a lane's loads wait for its older stores: the memory
//     counter retires in order);
//   * the token loop is a state machine — every iteration a lane either decodes one token or copies up to eight bytes of
//     a pending match — so that lanes in different states of different streams share one loop body.
// The same functions compile for the host (tests/test_inflate.py checks them against zlib on every block of a synthetic
// BAM and on streams of every block type).
#pragma once
#include <stdint.h>
#include <string.h>
#include <utility>

#if defined(__HIPCC__)
#define STRK_INF_HD __host__ __device__ inline
#else
#define STRK_INF_HD inline
#endif
// The loops of a decoder are short, data-dependent and run by one lane each: nothing for a loop vectoriser (and the ROCm 7.2
// compiler's post-RA scheduler crashes on what -O3 makes of them when it tries).
#if defined(__clang__)
#define STRK_INF_LOOP _Pragma("clang loop vectorize(disable) interleave(disable)")
#define STRK_INF_UNROLL _Pragma("unroll")
#else
#define STRK_INF_LOOP
#define STRK_INF_UNROLL
#endif

namespace strk_inf {

constexpr int kErrNone = 0, kErrBadBlockType = 1, kErrBadCode = 2, kErrBadLengths = 3, kErrOverrun = 4, kErrBadDistance = 5,
              kErrStored = 6, kErrSize = 7, kErrCrc = 8;

struct Tables {           // 320 bytes per stream: LDS on the device
    uint8_t lsym[288];    // literal/length symbols sorted by (code length, symbol), low eight bits
    uint8_t dsym[32];     // distance symbols, likewise (whole); the code-length code's while a block header is read
};
#ifndef STRK_INF_LITERAL_RUN
#define STRK_INF_LITERAL_RUN 4
#endif
constexpr int kLiteralRun = STRK_INF_LITERAL_RUN;
constexpr int kLensBytes = 19 + 286 + 30 + 1;   // code lengths while a block header is being read (global scratch on the device)

struct Code {             // one canonical code: registers on the device (every index is a constant after unrolling)
    uint32_t lim[15];     // lim[L - 1], L = 1..15: left-justified (15-bit) end of the codes of length L
    uint32_t word[16];    // word[L - 1]: bits 0-15 = (int16) index of the first symbol of length L - first code of length L,
                          // bits 16-24 = index of the first symbol >= 256 of that length (the split), bits 28-31 = L;
                          // word[15] = 0: no code starts like that
};

STRK_INF_HD uint32_t bitrev15(uint32_t x) {   // the low 15 bits of x, reversed
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x) >> 17;
#else
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
    x = ((x & 0x00ffu) << 8) | ((x >> 8) & 0x00ffu);
    return (x & 0xffffu) >> 1;
#endif
}

// Canonical code from `n` code lengths: sorted symbols, bases and limits.  false: over-subscribed or (when more than one
// code is used) incomplete set of lengths.
template <class F, int... K>
STRK_INF_HD void for_each_length_impl(F&& f, std::integer_sequence<int, K...>) { (f(std::integral_constant<int, K>()), ...); }
template <class F>
STRK_INF_HD void for_each_length(F&& f) { for_each_length_impl(f, std::make_integer_sequence<int, 15>()); }   // f(0) ... f(14)

STRK_INF_HD bool build(const uint8_t* lens, int n, uint8_t* sym, Code* c) {
    int count[16];   // low half: symbols of that length; high half: those below 256 among them
    STRK_INF_LOOP
    for (int l = 0; l < 16; ++l) count[l] = 0;
    STRK_INF_LOOP
    for (int i = 0; i < n; ++i) count[lens[i] & 15] += i < 256 ? 0x10001 : 1;
    int offs[16];
    int code = 0, used = 0, left = 1;
    offs[1] = 0;
    bool over = false;
    // one step per length, written out at compile time: every index into `c` is a constant from the start, so the code
    // never exists in memory (a loop that is unrolled later leaves the compiler time to turn the select chain of code_word
    // into one load from a selected address of a scratch copy)
    for_each_length([&](auto lc) {
        constexpr int l = decltype(lc)::value + 1;
        const int cnt = count[l] & 0xffff, lits = count[l] >> 16;
        left *= 2;                                       // (may be negative once over-subscribed: not a shift)
        left -= cnt;
        over |= left < 0;                                // over-subscribed
        code <<= 1;                                      // first code of length l
        c->word[l - 1] = (uint32_t)(uint16_t)(int16_t)(offs[l] - code) | ((uint32_t)(offs[l] + lits) << 16) | ((uint32_t)l << 28);
        code += cnt;
        c->lim[l - 1] = (uint32_t)code << (15 - l);
        used += cnt;
        if (l < 15) offs[l + 1] = offs[l] + cnt;
    });
    c->word[15] = 0;
    if (over) return false;
    if (left > 0 && used > 1) return false;              // incomplete (a single code of length 1 is allowed: RFC 1951 3.2.7)
    STRK_INF_LOOP
    for (int i = 0; i < n; ++i) {
        const int l = lens[i] & 15;
        if (l) sym[offs[l]++] = (uint8_t)i;
    }
    return true;
}

// the word of the code that the left-justified 15-bit value v starts with (0: no code).  The limits do not decrease with
// the length, so the last step that holds is the one of the code's own length.
STRK_INF_HD uint32_t code_word(uint32_t v, const Code& c) {
    uint32_t w = c.word[0];
    for_each_length([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        w = v >= c.lim[k] ? c.word[k + 1] : w;
    });
    return w;
}
// index of the symbol in the sorted array, given the word
STRK_INF_HD int code_at(uint32_t v, uint32_t w) { return (int)(int16_t)(w & 0xffffu) + (int)(v >> (15 - (int)(w >> 28))); }

struct Stream {
    const uint8_t* p;     // next input byte to load
    const uint8_t* end;   // end of the payload (loads may run up to 16 bytes past it: the caller pads)
    uint64_t buf;
    uint64_t ahead;       // the eight bytes at p, loaded when p last moved: a refill never waits for memory
    int cnt;              // valid bits in buf
};

// (never further than the padding: a stream that has run past its payload keeps re-loading the last padded bytes until the
// overrun check behind the refill reports it — a truncated or crafted block cannot make the decoder read elsewhere)
STRK_INF_HD void load_ahead(Stream& s) {
    const uint8_t* q = s.p <= s.end + 8 ? s.p : s.end + 8;
    memcpy(&s.ahead, q, 8);
}
STRK_INF_HD void refill(Stream& s) {
    s.buf |= s.ahead << s.cnt;
    s.p += (63 - s.cnt) >> 3;
    s.cnt |= 56;
    load_ahead(s);
}
STRK_INF_HD uint32_t take(Stream& s, int n) {   // n <= 32 bits, LSB first
    const uint32_t v = (uint32_t)(s.buf & ((1ull << n) - 1));
    s.buf >>= n;
    s.cnt -= n;
    return v;
}

// Reads a dynamic block header (HLIT, HDIST, HCLEN, the code-length code, the two sets of lengths) and builds the tables.
STRK_INF_HD int read_dynamic(Stream& s, Tables* t, uint8_t* lens, Code* ll, Code* dl) {
    // (every refill of the header is followed by the check the literal loop has: a truncated or crafted header must not walk
    // the stream further than the 16 bytes of padding the caller promises behind the payload)
    refill(s);
    if (s.p - 8 > s.end) return kErrOverrun;
    const int nlen = (int)take(s, 5) + 257, ndist = (int)take(s, 5) + 1, ncode = (int)take(s, 4) + 4;
    if (nlen > 286 || ndist > 30) return kErrBadLengths;
    // the order of the code-length code's lengths: 16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15, five bits each
    const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                              10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    STRK_INF_LOOP
    for (int i = 0; i < 19; ++i) lens[i] = 0;
    STRK_INF_LOOP
    for (int i = 0; i < ncode; ++i) {
        if (s.cnt < 3) {
            refill(s);
            if (s.p - 8 > s.end) return kErrOverrun;
        }
        const int at = (int)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31);
        lens[at] = (uint8_t)take(s, 3);
    }
    // the code-length code: its words in the distance code's registers, its symbols in the distance slots, until the real
    // distance code is built
    Code& cl = *dl;
    if (!build(lens, 19, t->dsym, &cl)) return kErrBadLengths;
    int i = 0;
    STRK_INF_LOOP
    while (i < nlen + ndist) {
        refill(s);
        if (s.p - 8 > s.end) return kErrOverrun;
        const uint32_t v = bitrev15((uint32_t)s.buf);
        const uint32_t w = code_word(v, cl);
        const int len = (int)(w >> 28);
        if (len == 0 || len > 7) return kErrBadCode;
        const int sym = t->dsym[code_at(v, w)];
        s.buf >>= len; s.cnt -= len;
        if (sym < 16) { lens[19 + i++] = (uint8_t)sym; continue; }
        int prev = 0, rep;
        if (sym == 16) {
            if (i == 0) return kErrBadLengths;
            prev = lens[19 + i - 1];
            rep = 3 + (int)take(s, 2);
        } else if (sym == 17) rep = 3 + (int)take(s, 3);
        else rep = 11 + (int)take(s, 7);
        if (i + rep > nlen + ndist) return kErrBadLengths;
        STRK_INF_LOOP
        while (rep--) lens[19 + i++] = (uint8_t)prev;
    }
    if (lens[19 + 256] == 0) return kErrBadLengths;   // no end-of-block code
    if (!build(lens + 19, nlen, t->lsym, ll)) return kErrBadLengths;
    if (!build(lens + 19 + nlen, ndist, t->dsym, dl)) return kErrBadLengths;
    return kErrNone;
}

STRK_INF_HD int set_fixed(Tables* t, uint8_t* lens, Code* ll, Code* dl) {
    STRK_INF_LOOP
    for (int i = 0; i < 144; ++i) lens[i] = 8;
    STRK_INF_LOOP
    for (int i = 144; i < 256; ++i) lens[i] = 9;
    STRK_INF_LOOP
    for (int i = 256; i < 280; ++i) lens[i] = 7;
    STRK_INF_LOOP
    for (int i = 280; i < 288; ++i) lens[i] = 8;
    if (!build(lens, 288, t->lsym, ll)) return kErrBadLengths;
    STRK_INF_LOOP
    for (int i = 0; i < 30; ++i) lens[i] = 5;
    // (the fixed distance code uses 30 of its 32 five-bit codes: build() would call it incomplete)
    STRK_INF_LOOP
    for (int i = 30; i < 32; ++i) lens[i] = 5;
    if (!build(lens, 32, t->dsym, dl)) return kErrBadLengths;
    return kErrNone;
}

// Inflates one raw deflate stream of `in_len` bytes into exactly `out_len` bytes.  `in` must be readable up to in_len + 16.
// `lens`: kLensBytes of scratch.
STRK_INF_HD int inflate_block(const uint8_t* in, int in_len, uint8_t* out, int out_len, Tables* t, uint8_t* lens) {
    // The base values and extra-bit counts of the length and distance symbols (RFC 1951 3.2.5) are computed, not looked up:
    // a table indexed by a lane's own symbol would be a load from memory, two of them in a row for every match.
    Stream s;
    s.p = in; s.end = in + in_len; s.buf = 0; s.cnt = 0;
    load_ahead(s);
    Code ll, dl;
    int pos = 0;          // bytes written to `out`
    uint64_t ob = 0;      // literals not yet written: `on` bytes (out[pos .. pos + on)), stored eight at a time
    int on = 0;
    auto flush = [&]() {
        STRK_INF_LOOP
        for (int i = 0; i < on; ++i) out[pos + i] = (uint8_t)(ob >> (8 * i));
        pos += on; ob = 0; on = 0;
    };
    bool last = false, in_block = false;
    int copy_len = 0, copy_dist = 0;
    // Every round of the outer loop a lane (1) finishes its pending match, (2) decodes literals up to the next match or the
    // end of the deflate block, (3) decodes that match.  The lanes of a wave run the three phases together (each phase lasts
    // as long as the slowest lane needs): a loop whose every iteration offers all the token kinds would make every lane pay
    // for every kind, every time.
    for (;;) {
        while (copy_len > 0) {
            // up to 32 bytes per iteration when the source lies at least that far back, else eight whatever the match's
            // length and distance: the bytes at the source are loaded, a source that overlaps the destination (distance < 8:
            // a run of a short pattern) is completed by doubling the pattern, and whole words are stored even when fewer
            // bytes belong to the match — what lies behind it is written again by the tokens that follow (never past the
            // end of the block)
            if (copy_dist >= 264 && copy_len > 128 && pos + 264 <= out_len) {
                // (the longest matches in one trip as well: the slowest lane of a wave sets the pace of this loop)
                uint64_t w[33];
                const uint8_t* src = out + pos - copy_dist;
                STRK_INF_UNROLL
                for (int k = 0; k < 33; ++k) memcpy(&w[k], src + 8 * k, 8);
                STRK_INF_UNROLL
                for (int k = 0; k < 33; ++k) memcpy(out + pos + 8 * k, &w[k], 8);
                pos += copy_len; copy_len = 0;            // (a match has at most 258 bytes)
            } else if (copy_dist < 8 && pos + 8 <= out_len) {
                // a run of a short pattern: one load, then every further word comes from the word before it (its last
                // `distance` bytes, doubled up to eight) — stores only, nothing to wait for
                uint64_t w;
                memcpy(&w, out + pos - copy_dist, 8);
                const int keep = 8 * copy_dist, drop = 64 - keep;
                w <<= drop;
                for (;;) {
                    w >>= drop;                               // the pattern's `distance` bytes, low
                    w |= w << keep;                           // 2 x distance bytes (distance >= 4: done)
                    if (copy_dist < 4) {
                        w |= w << (2 * keep);                 // 4 x distance
                        if (copy_dist < 2) w |= w << 32;      // 8 x 1
                    }
                    memcpy(out + pos, &w, 8);
                    const int n = copy_len < 8 ? copy_len : 8;
                    pos += n; copy_len -= n;
                    if (copy_len <= 0 || pos + 8 > out_len) break;
                }
            } else if (copy_dist >= 128 && copy_len > 32 && pos + 128 <= out_len) {
                // (a long match: sixteen loads in flight, one trip to memory for 128 bytes)
                uint64_t w[16];
                const uint8_t* src = out + pos - copy_dist;
                STRK_INF_UNROLL
                for (int k = 0; k < 16; ++k) memcpy(&w[k], src + 8 * k, 8);
                STRK_INF_UNROLL
                for (int k = 0; k < 16; ++k) memcpy(out + pos + 8 * k, &w[k], 8);
                const int n = copy_len < 128 ? copy_len : 128;
                pos += n; copy_len -= n;
            } else if (copy_dist >= 32 && copy_len > 8 && pos + 32 <= out_len) {
                uint64_t w0, w1, w2, w3;
                const uint8_t* src = out + pos - copy_dist;
                memcpy(&w0, src, 8); memcpy(&w1, src + 8, 8); memcpy(&w2, src + 16, 8); memcpy(&w3, src + 24, 8);
                memcpy(out + pos, &w0, 8); memcpy(out + pos + 8, &w1, 8); memcpy(out + pos + 16, &w2, 8); memcpy(out + pos + 24, &w3, 8);
                const int n = copy_len < 32 ? copy_len : 32;
                pos += n; copy_len -= n;
            } else if (pos + 8 <= out_len) {
                uint64_t w;
                memcpy(&w, out + pos - copy_dist, 8);
                memcpy(out + pos, &w, 8);
                const int n = copy_len < 8 ? copy_len : 8;
                pos += n; copy_len -= n;
            } else {
                out[pos] = out[pos - copy_dist];
                ++pos; --copy_len;
            }
        }
        if (!in_block) {
            flush();
            if (last) break;
            if (s.cnt < 48) {
                refill(s);
                if (s.p - 8 > s.end) return kErrOverrun;
            }
            last = take(s, 1) != 0;
            const int type = (int)take(s, 2);
            if (type == 0) {
                // stored: skip to the byte boundary, LEN / NLEN, raw bytes
                const int drop = s.cnt & 7;
                s.buf >>= drop; s.cnt -= drop;
                const uint8_t* q = s.p - (s.cnt >> 3);       // first byte not consumed
                if (q + 4 > s.end) return kErrStored;
                const int len = q[0] | (q[1] << 8), nlen = q[2] | (q[3] << 8);
                if ((len ^ nlen) != 0xffff || q + 4 + len > s.end || pos + len > out_len) return kErrStored;
                STRK_INF_LOOP
                for (int i = 0; i < len; ++i) out[pos + i] = q[4 + i];
                pos += len;
                s.p = q + 4 + len; s.buf = 0; s.cnt = 0;
                load_ahead(s);
                continue;
            }
            int rc;
            if (type == 1) rc = set_fixed(t, lens, &ll, &dl);
            else if (type == 2) rc = read_dynamic(s, t, lens, &ll, &dl);
            else return kErrBadBlockType;
            if (rc) return rc;
            in_block = true;
            continue;
        }
        int sym = -1, len;
        uint32_t v, w;
        // (at most kLiteralRun literals per round: a lane with a long run goes round again — its phases (1) and (3) are
        // empty — instead of keeping the lanes that have reached their next match waiting)
        for (int lit = 0; lit < kLiteralRun; ++lit) {
            if (s.cnt < 48) {
                refill(s);
                if (s.p - 8 > s.end) return kErrOverrun;     // the stream ran past its payload
            }
            v = bitrev15((uint32_t)s.buf);
            w = code_word(v, ll);
            len = (int)(w >> 28);
            if (len == 0) return kErrBadCode;
            const int at = code_at(v, w);
            sym = t->lsym[at] | (at >= (int)((w >> 16) & 0x1ffu) ? 256 : 0);
            s.buf >>= len; s.cnt -= len;
            if (sym >= 256) break;
            if (pos + on >= out_len) return kErrSize;
            ob |= (uint64_t)sym << (8 * on);
            if (++on == 8) {
                memcpy(out + pos, &ob, 8);
                pos += 8; ob = 0; on = 0;
            }
        }
        if (sym < 256) continue;                          // the run goes on
        if (sym == 256) { in_block = false; continue; }
        flush();
        if (sym > 285) return kErrBadCode;
        // lengths: 257..264 -> 3..10; from 265 on four symbols share a number of extra bits e = 1, 2, ...: base =
        // ((4 + (sym - 265) % 4) << e) + 3; 285 -> 258
        const int le = sym < 265 || sym == 285 ? 0 : (sym - 261) >> 2;
        const int lb = sym < 265 ? sym - 254 : sym == 285 ? 258 : ((4 + ((sym - 265) & 3)) << le) + 3;
        const int mlen = lb + (int)take(s, le);
        v = bitrev15((uint32_t)s.buf);
        w = code_word(v, dl);
        len = (int)(w >> 28);
        if (len == 0) return kErrBadCode;
        const int dsymv = t->dsym[code_at(v, w)];
        s.buf >>= len; s.cnt -= len;
        if (dsymv > 29) return kErrBadCode;
        // distances: 0..3 -> 1..4; from 4 on two symbols share e = 1, 2, ... extra bits: base = ((2 + d % 2) << e) + 1
        const int de = dsymv < 4 ? 0 : (dsymv >> 1) - 1;
        const int db = dsymv < 4 ? dsymv + 1 : ((2 + (dsymv & 1)) << de) + 1;
        const int dist = db + (int)take(s, de);
        if (dist > pos) return kErrBadDistance;
        if (pos + mlen > out_len) return kErrSize;
        copy_len = mlen; copy_dist = dist;
    }
    return pos == out_len ? kErrNone : kErrSize;
}

// CRC-32 (gzip) of `n` bytes, byte at a time over a 256-entry table (crc_table: shared, LDS on the device).
STRK_INF_HD void crc_table_entry(uint32_t* tab, int i) {
    uint32_t c = (uint32_t)i;
    STRK_INF_LOOP
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    tab[i] = c;
}
STRK_INF_HD uint32_t crc32_bytes(const uint32_t* tab, const uint8_t* p, int n) {
    uint32_t c = 0xffffffffu;
    STRK_INF_LOOP
    for (int i = 0; i < n; ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

}  // namespace strk_inf
