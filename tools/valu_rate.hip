// Micro-benchmark: issue rate of the VALU instructions the DP kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ void __launch_bounds__(256) k(int* out, int iters, int a0) {
    int a = a0 + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a ^ 11, f = a ^ 13, g = a + 17, h = a + 19;
    float fa = a, fb = b, fc = c, fd = d, fe = e, ff = f, fg = g, fh = h;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 1) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %2, %2, %3, %4\n v_max3_i32 %4, %4, %5, %6\n v_max3_i32 %6, %6, %7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 2) { REP64(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd), "+v"(fe), "+v"(ff), "+v"(fg), "+v"(fh));) }
        if (KIND == 3) { REP64(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %6, %6, %7, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd), "+v"(fe), "+v"(ff), "+v"(fg), "+v"(fh));) }
        if (KIND == 4) { REP64(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 5) { REP64(asm volatile("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %2, %2, %3\n v_pk_max_u16 %4, %4, %5\n v_pk_max_u16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 6) { REP64(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %6, %6, %7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 7) { REP64(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 8) { REP64(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %5\n v_max_i32 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 9) { REP64(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd), "+v"(fe), "+v"(ff), "+v"(fg), "+v"(fh));) }
        if (KIND == 10) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 11) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %1\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte0 %4, %5\n v_cvt_f32_ubyte3 %6, %7" : "+v"(fa), "+v"(b), "+v"(fc), "+v"(d), "+v"(fe), "+v"(f), "+v"(fg), "+v"(h));) }
        if (KIND == 12) { REP64(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_max_i16 %4, %4, %5\n v_pk_add_i16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 13) { REP64(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        // dependency distance 1 (every instruction needs the previous result) and 2
        if (KIND == 14) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %0, %0, %3, %4\n v_max3_i32 %0, %0, %5, %6\n v_max3_i32 %0, %0, %7, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 15) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_add_u32 %3, %3, %4\n v_max3_i32 %0, %0, %5, %6\n v_add_u32 %7, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 16) { REP64(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %0, %0, %3\n v_max_i32 %0, %0, %5\n v_max_i32 %0, %0, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (int)(fa + fb + fc + fd + fe + ff + fg + fh);
}

template <int KIND>
void run(const char* name, int* d_out) {
    const int iters = 200;
    const double n_inst = 4.0 * 64 * iters;  // per wave
    for (int wps : {1, 2, 4}) {  // waves per SIMD
        const int blocks = 256 * wps;  // 256 CUs x (wps blocks of 4 waves)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = wps * n_inst ; cycles at 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-18s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, wps, ms,
               cyc / (wps * n_inst));
    }
}

int main() {
    int* d_out;
    hipMalloc(&d_out, 256 * 4 * 256 * 4);
    run<0>("v_add_u32", d_out);
    run<1>("v_max3_i32", d_out);
    run<8>("v_max_i32", d_out);
    run<2>("v_add_f32", d_out);
    run<3>("v_max3_f32", d_out);
    run<9>("v_max_f32", d_out);
    run<4>("v_pk_add_u16", d_out);
    run<5>("v_pk_max_u16", d_out);
    run<12>("v_pk_max/add_i16", d_out);
    run<6>("v_perm_b32", d_out);
    run<7>("v_add_u32_sdwa", d_out);
    run<10>("v_mov_b32_dpp", d_out);
    run<11>("v_cvt_f32_ubyteN", d_out);
    run<13>("v_add3_u32", d_out);
    run<14>("max3 chain dist 1", d_out);
    run<15>("max3 chain dist 2", d_out);
    run<16>("max chain dist 1", d_out);
    return 0;
}
