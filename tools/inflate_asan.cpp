// The device inflater's code (strk_inflate.h compiles for the host too) under AddressSanitizer / UBSan: every BGZF block of the
// files given is inflated into a heap buffer of exactly its size from a heap copy of its payload padded by the 16 bytes the
// decoder may read ahead — a store or a load one byte outside either is reported.  Build and run: tools/inflate_asan.sh
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "../strkit_amd/csrc/strk_inflate.h"

int main(int argc, char** argv) {
    long blocks = 0, bytes = 0, variants = 0;
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> c((size_t)n);
        if (fread(c.data(), 1, (size_t)n, f) != (size_t)n) return 2;
        fclose(f);
        long off = 0;
        while (off + 18 <= n) {
            const int xlen = c[off + 10] | (c[off + 11] << 8);
            const int bsize = c[off + 16] | (c[off + 17] << 8);
            const long next = off + bsize + 1;
            uint32_t crc, isize;
            memcpy(&crc, &c[next - 8], 4);
            memcpy(&isize, &c[next - 4], 4);
            const int in_len = (int)(next - 8 - (off + 12 + xlen));
            uint8_t* in = (uint8_t*)malloc((size_t)in_len + 16);
            memcpy(in, &c[off + 12 + xlen], (size_t)in_len);
            memset(in + in_len, 0, 16);
            uint8_t* out = (uint8_t*)malloc(isize ? isize : 1);
            strk_inf::Tables* t = (strk_inf::Tables*)malloc(sizeof(strk_inf::Tables));
            uint8_t* lens = (uint8_t*)malloc(strk_inf::kLensBytes);
            const int rc = isize ? strk_inf::inflate_block(in, in_len, out, (int)isize, t, lens) : 0;
            if (rc) { fprintf(stderr, "%s: block at %ld: error %d\n", argv[a], off, rc); return 1; }
            if (crc32(0, out, isize) != crc) { fprintf(stderr, "%s: block at %ld: CRC mismatch\n", argv[a], off); return 1; }
            // truncated and corrupted payloads (the dynamic-block header above all): whatever the decoder returns, it must
            // stay inside the payload + 16 bytes of padding and inside the output buffer
            if (isize && in_len > 4) {
                const int cuts[] = {1, 2, 3, 5, 9, 17, 33, 64, in_len / 2, in_len - 1};
                for (int cut : cuts) {
                    if (cut <= 0 || cut >= in_len) continue;
                    uint8_t* tr = (uint8_t*)malloc((size_t)cut + 16);
                    memcpy(tr, in, (size_t)cut);
                    memset(tr + cut, 0xff, 16);                     // (ones: every length code of a header read from them is long)
                    (void)strk_inf::inflate_block(tr, cut, out, (int)isize, t, lens);
                    free(tr);
                    ++variants;
                }
                for (int k = 0; k < 24 && k < in_len; ++k) {         // flip bits inside the first bytes (block header, code lengths)
                    uint8_t* cr = (uint8_t*)malloc((size_t)in_len + 16);
                    memcpy(cr, in, (size_t)in_len);
                    memset(cr + in_len, 0, 16);
                    cr[k] ^= (uint8_t)(0x5a + 7 * k);
                    (void)strk_inf::inflate_block(cr, in_len, out, (int)isize, t, lens);
                    free(cr);
                    ++variants;
                }
            }
            free(in); free(out); free(t); free(lens);
            ++blocks; bytes += isize;
            off = next;
        }
    }
    printf("%ld blocks, %ld bytes, %ld truncated / corrupted variants: clean\n", blocks, bytes, variants);
    return 0;
}
