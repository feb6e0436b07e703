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
    long blocks = 0, bytes = 0;
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
            free(in); free(out); free(t); free(lens);
            ++blocks; bytes += isize;
            off = next;
        }
    }
    printf("%ld blocks, %ld bytes: clean\n", blocks, bytes);
    return 0;
}
