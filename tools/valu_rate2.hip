// Opcode survey (round 3): which gfx950 VALU opcodes issue at the SIMD-32 rate (one wave64 instruction per ~2 cycles once
// two or more waves share a SIMD) and which at ~4.2.  Method (round 2's tools/valu_rate.hip, removed in round 4): s_memtime inside the kernel,
// the clock the chip held from s_memrealtime, HIP events around the launch), four INDEPENDENT instructions per asm
// statement, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate2 tools/valu_rate2.hip ; run on the GPU box:
//        tools/valu_rate2 > gpurun_out/valu_rate2.txt   (the committed copy is profiles/r03_valu_rate2.txt)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define REGS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1), "s"(s2) : "vcc"
// two-operand form  op D, D, S  on four independent register pairs; three-operand form  op D, D, S, T
#define OP2(op) REP64(asm volatile(op " %0, %0, %1\n " op " %2, %2, %3\n " op " %4, %4, %5\n " op " %6, %6, %7" REGS);)
#define OP3(op) REP64(asm volatile(op " %0, %0, %1, %2\n " op " %2, %2, %3, %4\n " op " %4, %4, %5, %6\n " op " %6, %6, %7, %0" REGS);)
#define OPI(op, imm) REP64(asm volatile(op " %0, " imm ", %1\n " op " %2, " imm ", %3\n " op " %4, " imm ", %5\n " op " %6, " imm ", %7" REGS);)
#define RAW(txt) REP64(asm volatile(txt REGS);)

template <int KIND>
__global__ void __launch_bounds__(256) k(int* out, unsigned long long* stamps, int iters, int a0, int s1, int s2) {
    int a = a0 + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a ^ 11, f = a ^ 13, g = a + 17, h = a + 19;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { OP2("v_add_u32") }
        if (KIND == 1) { OP2("v_max_i32") }
        if (KIND == 2) { OP2("v_max_u32") }
        if (KIND == 3) { OP2("v_min_i32") }
        if (KIND == 4) { OP2("v_min_u32") }
        if (KIND == 5) { OP2("v_sub_u32") }
        if (KIND == 6) { OP2("v_subrev_u32") }
        if (KIND == 7) { RAW("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc") }
        if (KIND == 8) { RAW("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %2, %2, 1, %3\n v_lshl_add_u32 %4, %4, 1, %5\n v_lshl_add_u32 %6, %6, 1, %7") }
        if (KIND == 9) { OP3("v_mad_u32_u24") }
        if (KIND == 10) { OP3("v_mad_i32_i24") }
        if (KIND == 11) { OP3("v_sad_u8") }
        if (KIND == 12) { OP3("v_msad_u8") }
        if (KIND == 13) { OP3("v_med3_i32") }
        if (KIND == 14) { OP3("v_min3_i32") }
        if (KIND == 15) { OP3("v_max3_u32") }
        if (KIND == 16) { OP2("v_and_b32") }
        if (KIND == 17) { OP2("v_or_b32") }
        if (KIND == 18) { OP2("v_xor_b32") }
        if (KIND == 19) { OPI("v_lshrrev_b32", "8") }
        if (KIND == 20) { OPI("v_lshlrev_b32", "1") }
        if (KIND == 21) { OPI("v_ashrrev_i32", "1") }
        if (KIND == 22) { RAW("v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %2, %3, 8, 8\n v_bfe_u32 %4, %5, 16, 8\n v_bfe_u32 %6, %7, 16, 8") }
        if (KIND == 23) { OP3("v_and_or_b32") }
        if (KIND == 24) { OP3("v_or3_b32") }
        if (KIND == 25) { OP3("v_xad_u32") }
        if (KIND == 26) { RAW("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7") }
        if (KIND == 27) { OP3("v_fma_f32") }
        if (KIND == 28) { OP2("v_mul_f32") }
        if (KIND == 29) { OP2("v_min_f32") }
        if (KIND == 30) { OP3("v_maximum3_f32") }
        if (KIND == 31) { OP3("v_pk_maximum3_f16") }
        if (KIND == 32) { OP2("v_pk_max_f16") }
        if (KIND == 33) { OP2("v_pk_add_f16") }
        if (KIND == 34) { OP3("v_pk_fma_f16") }
        if (KIND == 35) { OP2("v_max_f16") }
        if (KIND == 36) { OP2("v_max_u16") }
        if (KIND == 37) { OP2("v_max_i16") }
        if (KIND == 38) { OP2("v_add_u16") }
        if (KIND == 39) { RAW("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %6, vcc, %6, %7") }
        if (KIND == 40) { OP2("v_mul_lo_u32") }
        if (KIND == 41) { OP2("v_mul_u32_u24") }
        // byte k of S0 added to the accumulator in one instruction: dot product with a one-hot byte vector held in an SGPR
        if (KIND == 42) { RAW("v_dot4_u32_u8 %0, %1, %8, %0\n v_dot4_u32_u8 %2, %3, %9, %2\n v_dot4_u32_u8 %4, %5, %8, %4\n v_dot4_u32_u8 %6, %7, %9, %6") }
        if (KIND == 43) { RAW("v_dot4_i32_i8 %0, %1, %8, %0\n v_dot4_i32_i8 %2, %3, %9, %2\n v_dot4_i32_i8 %4, %5, %8, %4\n v_dot4_i32_i8 %6, %7, %9, %6") }
        if (KIND == 44) { RAW("v_dot2_u32_u16 %0, %1, %8, %0\n v_dot2_u32_u16 %2, %3, %9, %2\n v_dot2_u32_u16 %4, %5, %8, %4\n v_dot2_u32_u16 %6, %7, %9, %6") }
        if (KIND == 45) { OP3("v_add3_u32") }
        if (KIND == 46) { RAW("v_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_max_i32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_max_i32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_max_i32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") }
        if (KIND == 47) { RAW("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf") }
        if (KIND == 48) { RAW("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf") }
        if (KIND == 49) { RAW("v_cmp_gt_i32 vcc, %0, %1\n v_cmp_gt_i32 vcc, %2, %3\n v_cmp_gt_i32 vcc, %4, %5\n v_cmp_gt_i32 vcc, %6, %7") }
        if (KIND == 50) { OP2("v_cvt_pk_u16_u32") }
        if (KIND == 51) { OP3("v_bfi_b32") }
        if (KIND == 52) { RAW("v_alignbit_b32 %0, %0, %1, 8\n v_alignbit_b32 %2, %2, %3, 8\n v_alignbit_b32 %4, %4, %5, 8\n v_alignbit_b32 %6, %6, %7, 8") }
        if (KIND == 53) { OP2("v_pk_max_i16") }
        if (KIND == 54) { OP2("v_pk_add_u16") }
        if (KIND == 55) { RAW("v_pk_sub_u16 %0, %0, %1 clamp\n v_pk_sub_u16 %2, %2, %3 clamp\n v_pk_sub_u16 %4, %4, %5 clamp\n v_pk_sub_u16 %6, %6, %7 clamp") }
        if (KIND == 56) { OP3("v_max3_i32") }
        if (KIND == 57) { OP3("v_perm_b32") }
        if (KIND == 58) { RAW("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3") }
        // mixes: does a fast-class instruction between two slow ones hide?  add_u32 / max3 alternating, and the DP cell with
        // a plain add (w as a dword): add, max3
        if (KIND == 59) { RAW("v_add_u32 %1, %2, %7\n v_max3_i32 %0, %3, %0, %1\n v_add_u32 %6, %4, %7\n v_max3_i32 %0, %5, %0, %6") }
        if (KIND == 60) { RAW("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_max3_i32 %4, %4, %5, %6\n v_add_u32 %6, %6, %7") }
        if (KIND == 61) { OP2("v_add_f32") }
        if (KIND == 62) { OP2("v_max_f32") }
        if (KIND == 63) { OP2("v_sub_f32") }
        if (KIND == 64) { RAW("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf") }
        if (KIND == 65) { RAW("v_and_b32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_and_b32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
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
    printf("%-24s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, 10, 1, 0x0100, 0x010000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 1, 0x0100, 0x010000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const size_t nw = (size_t)blocks * 4;
        hipMemcpy(h_st.data(), d_st, nw * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz(nw);
        for (size_t w = 0; w < nw; ++w)
            mhz[w] = (double)h_st[2 * w] / (double)std::max<unsigned long long>(h_st[2 * w + 1], 1) * 100.0;
        std::sort(mhz.begin(), mhz.end());
        const double clk = mhz[nw / 2];
        printf("  %dw: %5.2f (%4.0f MHz)", wps, ms * 1e-3 * clk * 1e6 / (wps * n_inst), clk);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
    fflush(stdout);
}

#define RUN(K, N) run<K>(N, d_out, d_st, h_st);
int main() {
    int* d_out;
    unsigned long long* d_st;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_st, 256 * 8 * 4 * 16);
    std::vector<unsigned long long> h_st(256 * 8 * 4 * 2);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# device %s, %d CUs; cycles per wave-instruction per SIMD from the launch time at the measured clock, 1 / 2 / 4 waves per SIMD\n",
           prop.gcnArchName, prop.multiProcessorCount);
    RUN(0, "v_add_u32") RUN(5, "v_sub_u32") RUN(6, "v_subrev_u32") RUN(39, "v_add_co_u32") RUN(45, "v_add3_u32") RUN(8, "v_lshl_add_u32")
    RUN(1, "v_max_i32") RUN(2, "v_max_u32") RUN(3, "v_min_i32") RUN(4, "v_min_u32") RUN(56, "v_max3_i32") RUN(15, "v_max3_u32")
    RUN(14, "v_min3_i32") RUN(13, "v_med3_i32")
    RUN(16, "v_and_b32") RUN(17, "v_or_b32") RUN(18, "v_xor_b32") RUN(19, "v_lshrrev_b32") RUN(20, "v_lshlrev_b32") RUN(21, "v_ashrrev_i32")
    RUN(22, "v_bfe_u32") RUN(23, "v_and_or_b32") RUN(24, "v_or3_b32") RUN(25, "v_xad_u32") RUN(51, "v_bfi_b32") RUN(52, "v_alignbit_b32")
    RUN(57, "v_perm_b32") RUN(26, "v_mov_b32") RUN(7, "v_cndmask_b32") RUN(49, "v_cmp_gt_i32")
    RUN(9, "v_mad_u32_u24") RUN(10, "v_mad_i32_i24") RUN(40, "v_mul_lo_u32") RUN(41, "v_mul_u32_u24")
    RUN(11, "v_sad_u8") RUN(12, "v_msad_u8") RUN(42, "v_dot4_u32_u8") RUN(43, "v_dot4_i32_i8") RUN(44, "v_dot2_u32_u16")
    RUN(61, "v_add_f32") RUN(63, "v_sub_f32") RUN(28, "v_mul_f32") RUN(27, "v_fma_f32") RUN(62, "v_max_f32") RUN(29, "v_min_f32") RUN(30, "v_maximum3_f32")
    RUN(31, "v_pk_maximum3_f16") RUN(32, "v_pk_max_f16") RUN(33, "v_pk_add_f16") RUN(34, "v_pk_fma_f16") RUN(35, "v_max_f16")
    RUN(36, "v_max_u16") RUN(37, "v_max_i16") RUN(38, "v_add_u16") RUN(53, "v_pk_max_i16") RUN(54, "v_pk_add_u16") RUN(55, "v_pk_sub_u16 clamp")
    RUN(50, "v_cvt_pk_u16_u32")
    RUN(58, "v_add_u32_sdwa") RUN(46, "v_max_i32_sdwa") RUN(47, "v_max_i32_dpp") RUN(48, "v_add_u32_dpp") RUN(64, "v_mov_b32_dpp") RUN(65, "v_and_b32_dpp")
    RUN(59, "cell: add_u32,max3 chain") RUN(60, "3 add_u32 : 1 max3")
    return 0;
}
