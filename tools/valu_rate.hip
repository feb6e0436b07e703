// Micro-benchmark: issue rate of the VALU instructions the DP kernels are made of (gfx950), at 1, 2, 4 and 8 waves per
// SIMD, in shader cycles measured INSIDE the kernel (s_memtime) and at the clock the chip actually held
// (s_memtime / s_memrealtime), not at a nominal clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip ; run on the GPU box:
//        tools/valu_rate > gpurun_out/valu_rate.txt   (the committed copy is profiles/r02_valu_rate.txt)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// every KIND issues 4 instructions per asm statement, 64 statements per loop iteration
template <int KIND>
__global__ void __launch_bounds__(256) k(int* out, unsigned long long* stamps, int iters, int a0) {
    int a = a0 + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a ^ 11, f = a ^ 13, g = a + 17, h = a + 19;
    float fa = a, fb = b, fc = c, fd = d, fe = e, ff = f, fg = g, fh = h;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
        if (KIND == 17) { REP64(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_add_i16 %4, %4, %5\n v_pk_add_i16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 18) { REP64(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %4, %4, %5\n v_pk_max_i16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 19) { REP64(asm volatile("v_alignbyte_b32 %0, %1, %0, 1\n v_alignbyte_b32 %2, %3, %2, 1\n v_alignbyte_b32 %4, %5, %4, 1\n v_alignbyte_b32 %6, %7, %6, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 20) { REP64(asm volatile("v_max3_i16 %0, %0, %1, %2\n v_max3_i16 %2, %2, %3, %4\n v_max3_i16 %4, %4, %5, %6\n v_max3_i16 %6, %6, %7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        // the DP cell as the band kernel issues it: t = diag + byte(w) ; cell = max3(up, left = previous cell, t)
        if (KIND == 21) { REP64(asm volatile("v_add_u32_sdwa %1, %2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_max3_i32 %0, %3, %0, %1\n v_add_u32_sdwa %1, %4, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %0, %5, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        // the same with the adds of two cells issued ahead of the dependent max3 chain
        if (KIND == 22) { REP64(asm volatile("v_add_u32_sdwa %1, %2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %6, %4, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %0, %3, %0, %1\n v_max3_i32 %0, %5, %0, %6" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        // packed 16-bit cell pair (two reads per register): t = diag + w ; cell = max(max(up, left), t)
        if (KIND == 23) { REP64(asm volatile("v_pk_add_i16 %1, %2, %7\n v_pk_max_i16 %6, %3, %0\n v_pk_max_i16 %0, %6, %1\n v_perm_b32 %7, %4, %5, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 24) { REP64(asm volatile("v_and_b32 %0, %0, %1\n v_lshrrev_b32 %2, 8, %3\n v_bfe_u32 %4, %5, 8, 8\n v_and_or_b32 %6, %7, %0, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (int)(fa + fb + fc + fd + fe + ff + fg + fh);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int KIND>
void run(const char* name, int* d_out, unsigned long long* d_st, std::vector<unsigned long long>& h_st) {
    const int iters = 200;
    const double n_inst = 4.0 * 64 * iters;  // per wave
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        const int blocks = 256 * wps;  // 256 CUs x (wps blocks of 4 waves: one wave per SIMD per block)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, 10, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const size_t nw = (size_t)blocks * 4;
        hipMemcpy(h_st.data(), d_st, nw * 16, hipMemcpyDeviceToHost);
        std::vector<double> cyc(nw), mhz(nw);
        for (size_t w = 0; w < nw; ++w) {
            cyc[w] = (double)h_st[2 * w];
            mhz[w] = (double)h_st[2 * w] / (double)std::max<unsigned long long>(h_st[2 * w + 1], 1) * 100.0;   // s_memrealtime ticks at 100 MHz
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(mhz.begin(), mhz.end());
        const double cyc_med = cyc[nw / 2], clk = mhz[nw / 2];
        // in-kernel: the median wave's loop took cyc_med shader cycles while wps waves shared its SIMD
        // event-based: the whole launch at the measured clock (includes launch ramp and tail)
        printf("%-22s waves/SIMD=%d  %8.3f ms  clock %6.0f MHz  cycles per wave-instruction per SIMD: %5.2f in-kernel, %5.2f from the launch time\n",
               name, wps, ms, clk, cyc_med / (wps * n_inst), ms * 1e-3 * clk * 1e6 / (wps * n_inst));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    int* d_out;
    unsigned long long* d_st;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_st, 256 * 8 * 4 * 16);
    std::vector<unsigned long long> h_st(256 * 8 * 4 * 2);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# device %s, %d CUs, clockRate %d kHz; 4 x 64 x 200 instructions per wave; one 256-thread block = one wave per SIMD\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    run<0>("v_add_u32", d_out, d_st, h_st);
    run<1>("v_max3_i32", d_out, d_st, h_st);
    run<8>("v_max_i32", d_out, d_st, h_st);
    run<2>("v_add_f32", d_out, d_st, h_st);
    run<3>("v_max3_f32", d_out, d_st, h_st);
    run<9>("v_max_f32", d_out, d_st, h_st);
    run<4>("v_pk_add_u16", d_out, d_st, h_st);
    run<5>("v_pk_max_u16", d_out, d_st, h_st);
    run<17>("v_pk_add_i16", d_out, d_st, h_st);
    run<18>("v_pk_max_i16", d_out, d_st, h_st);
    run<12>("v_pk_max/add_i16 mix", d_out, d_st, h_st);
    run<20>("v_max3_i16", d_out, d_st, h_st);
    run<6>("v_perm_b32", d_out, d_st, h_st);
    run<19>("v_alignbyte_b32", d_out, d_st, h_st);
    run<7>("v_add_u32_sdwa", d_out, d_st, h_st);
    run<10>("v_mov_b32_dpp", d_out, d_st, h_st);
    run<11>("v_cvt_f32_ubyteN", d_out, d_st, h_st);
    run<13>("v_add3_u32", d_out, d_st, h_st);
    run<24>("and/lshr/bfe/and_or", d_out, d_st, h_st);
    run<14>("max3_i32 chain dist 1", d_out, d_st, h_st);
    run<15>("max3_i32 chain dist 2", d_out, d_st, h_st);
    run<16>("max_i32 chain dist 1", d_out, d_st, h_st);
    run<21>("DP cell add,max3 chain", d_out, d_st, h_st);
    run<22>("DP cell 2add,2max3", d_out, d_st, h_st);
    run<23>("pk16 cell add,max,max,perm", d_out, d_st, h_st);
    return 0;
}
