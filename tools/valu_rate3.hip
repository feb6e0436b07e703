// Round 4, VERDICT r3 item 3: the two open questions of tools/valu_rate2.hip.
//  (a) v_cndmask_b32 measured 23.5 cycles per wave-instruction there, with VCC never written and the destination equal to
//      src0.  Here: VCC written once in front of the loop (VOP2 form), an SGPR-pair mask (VOP3 form), distinct destination.
//  (b) the issue cost of the ACTUAL band step (strk_dp_band.h, D = 16: 16 x {v_add_u32_sdwa, v_max3_i32} in a dependent chain,
//      4 v_perm_b32, 4 v_alignbyte_b32, v_and_b32_dpp + v_mov_b32_dpp + v_and for the two edge exchanges, one address add,
//      ds_read_u8 x 2 + ds_read_b64 per step, the s_waitcnt and the loop's scalar tail) at 1 and 2 waves per SIMD: cycles per
//      step and per VALU instruction of the mix — what bench.py's floor model should charge instead of a flat 4.2.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate3 tools/valu_rate3.hip ; run: tools/valu_rate3 > gpurun_out/valu_rate3.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// ---- (a) v_cndmask ----
template <int KIND>
__global__ void __launch_bounds__(256) k_cnd(int* out, unsigned long long* stamps, int iters, int a0, unsigned long long mask) {
    int a = a0 + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a ^ 11, f = a ^ 13, g = a + 17, h = a + 19;
    int p = 0, q = 0, r = 0, s = 0;
    if (KIND == 0) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");   // VCC written once, outside the loop
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (KIND == 1) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %8\n v_cndmask_b32_e64 %2, %2, %3, %8\n v_cndmask_b32_e64 %4, %4, %5, %8\n v_cndmask_b32_e64 %6, %6, %7, %8"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(mask));) }
        if (KIND == 2) { REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %5, %8\n v_cndmask_b32_e64 %1, %5, %6, %8\n v_cndmask_b32_e64 %2, %6, %7, %8\n v_cndmask_b32_e64 %3, %7, %4, %8"
                                            : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(a), "v"(b), "v"(c), "v"(d), "s"(mask));) }
        // what replaces a select where one side is 0: an AND with a per-lane mask register
        if (KIND == 3) { REP64(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        // a select as the compiler emits it in a loop: compare + cndmask pairs
        if (KIND == 4) { REP64(asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_i32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");) }
        // VCC written by the SCALAR unit (a wave-uniform condition the compiler turned into a select): s_cselect_b64 vcc + v_cndmask
        if (KIND == 5) { REP64(asm volatile("s_cmp_lg_u32 %8, 29\n s_cselect_b64 vcc, -1, 0\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(i) : "vcc", "scc");) }
        // v_cmp, three unrelated instructions, then the v_cndmask that reads its VCC
        if (KIND == 6) { REP64(asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n v_cndmask_b32 %0, %0, %1, vcc"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");) }
        // the same select without VCC: mask = (x - y) >> 31 (arithmetic), d = (d & ~mask) | (e & mask) ... here: d -= e & mask (four fast-class instructions)
        if (KIND == 7) { REP64(asm volatile("v_sub_u32 %2, %0, %1\n v_ashrrev_i32 %2, 31, %2\n v_and_b32 %4, %2, %5\n v_sub_u32 %6, %6, %4"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + p + q + r + s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// ---- (b) the band step ----
// H v0..v15 | selectors v16..v19 | w quads v20..v23 | row word v24:v25 | left v26 | up v27 | psym v28 | pnb v29 | sym v30 | nb v31 |
// masks v32, v33 | scratch v34 | running address v35
#define CELL(k, wq, byte, up, left) \
    "v_add_u32_sdwa v" #k ", v" #k ", v" #wq " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #byte "\n" \
    "v_max3_i32 v" #k ", v" #up ", v" #left ", v" #k "\n"
#define STEP_VALU(LDS1, LDS2) \
    "v_perm_b32 v20, v25, v24, v16\n v_perm_b32 v21, v25, v24, v17\n v_perm_b32 v22, v25, v24, v18\n v_perm_b32 v23, v25, v24, v19\n" \
    "v_and_b32_dpp v26, v15, v32 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    CELL(0, 20, 0, 1, 26) CELL(1, 20, 1, 2, 0) CELL(2, 20, 2, 3, 1) CELL(3, 20, 3, 4, 2) \
    CELL(4, 21, 0, 5, 3) CELL(5, 21, 1, 6, 4) CELL(6, 21, 2, 7, 5) CELL(7, 21, 3, 8, 6) \
    CELL(8, 22, 0, 9, 7) CELL(9, 22, 1, 10, 8) \
    "v_add_u32 v35, 2, v35\n" \
    LDS1 \
    CELL(10, 22, 2, 11, 9) CELL(11, 22, 3, 12, 10) CELL(12, 23, 0, 13, 11) CELL(13, 23, 1, 14, 12) \
    "v_mov_b32_dpp v27, v0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    CELL(14, 23, 2, 15, 13) \
    "v_add_u32_sdwa v34, v15, v23 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n" \
    "v_and_b32 v27, v27, v33\n" \
    LDS2 \
    "v_alignbyte_b32 v16, v17, v16, 1\n v_alignbyte_b32 v17, v18, v17, 1\n v_alignbyte_b32 v18, v19, v18, 1\n v_alignbyte_b32 v19, v31, v19, 1\n" \
    "v_max3_i32 v15, v27, v14, v34\n"
#define LDS_A "ds_read_u8 v30, v28 offset:3\n ds_read_u8 v31, v29\n"
#define LDS_B "s_waitcnt lgkmcnt(0)\n ds_read_b64 v[24:25], v30 offset:256\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","s20","scc","memory"

template <int KIND>   // 0: the step with its LDS reads; 1: VALU only; 2: only the 32 cell instructions
__global__ void __launch_bounds__(256) k_step(int* out, unsigned long long* stamps, int iters) {
    __shared__ unsigned char lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (unsigned char)((i * 8) & 0xf8);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;   // (address space 3 offset)
    int res;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile(
        "v_mov_b32 v28, %1\n v_add_u32 v29, 64, %1\n v_mov_b32 v32, -1\n v_mov_b32 v33, -1\n v_mov_b32 v35, 0\n"
        "v_mov_b32 v0, 1\n v_mov_b32 v1, 2\n v_mov_b32 v2, 3\n v_mov_b32 v3, 4\n v_mov_b32 v4, 5\n v_mov_b32 v5, 6\n v_mov_b32 v6, 7\n v_mov_b32 v7, 8\n"
        "v_mov_b32 v8, 1\n v_mov_b32 v9, 2\n v_mov_b32 v10, 3\n v_mov_b32 v11, 4\n v_mov_b32 v12, 5\n v_mov_b32 v13, 6\n v_mov_b32 v14, 7\n v_mov_b32 v15, 8\n"
        "v_mov_b32 v16, 0x03020100\n v_mov_b32 v17, 0x07060504\n v_mov_b32 v18, 0x03020100\n v_mov_b32 v19, 0x07060504\n"
        "v_mov_b32 v24, 0x01020304\n v_mov_b32 v25, 0x05060708\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
        "s_mov_b32 s20, %2\n"
        "1:\n"
        : "=v"(res) : "v"(base), "s"(iters) : CLOB);
    // (the loop body is a separate statement so that the three kinds share the prologue)
    if (KIND == 0) asm volatile("2:\n" STEP_VALU(LDS_A, LDS_B) STEP_VALU(LDS_A, LDS_B) "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : : : CLOB);
    if (KIND == 1) asm volatile("2:\n" STEP_VALU("", "") STEP_VALU("", "") "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : : : CLOB);
    if (KIND == 2) asm volatile("2:\n"
        CELL(0, 20, 0, 1, 26) CELL(1, 20, 1, 2, 0) CELL(2, 20, 2, 3, 1) CELL(3, 20, 3, 4, 2) CELL(4, 21, 0, 5, 3) CELL(5, 21, 1, 6, 4) CELL(6, 21, 2, 7, 5) CELL(7, 21, 3, 8, 6)
        CELL(8, 22, 0, 9, 7) CELL(9, 22, 1, 10, 8) CELL(10, 22, 2, 11, 9) CELL(11, 22, 3, 12, 10) CELL(12, 23, 0, 13, 11) CELL(13, 23, 1, 14, 12) CELL(14, 23, 2, 15, 13) CELL(15, 23, 3, 27, 14)
        CELL(0, 20, 0, 1, 26) CELL(1, 20, 1, 2, 0) CELL(2, 20, 2, 3, 1) CELL(3, 20, 3, 4, 2) CELL(4, 21, 0, 5, 3) CELL(5, 21, 1, 6, 4) CELL(6, 21, 2, 7, 5) CELL(7, 21, 3, 8, 6)
        CELL(8, 22, 0, 9, 7) CELL(9, 22, 1, 10, 8) CELL(10, 22, 2, 11, 9) CELL(11, 22, 3, 12, 10) CELL(12, 23, 0, 13, 11) CELL(13, 23, 1, 14, 12) CELL(14, 23, 2, 15, 13) CELL(15, 23, 3, 27, 14)
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : : : CLOB);
    asm volatile("v_add_u32 %0, v0, v15\n v_add_u32 %0, %0, v7" : "=v"(res) : : CLOB);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

static double clock_mhz(const std::vector<unsigned long long>& h_st, size_t nw) {
    std::vector<double> mhz(nw);
    for (size_t w = 0; w < nw; ++w) mhz[w] = (double)h_st[2 * w] / (double)std::max<unsigned long long>(h_st[2 * w + 1], 1) * 100.0;
    std::sort(mhz.begin(), mhz.end());
    return mhz[nw / 2];
}

template <class F>
void measure(const char* name, double insts_per_wave, double steps_per_wave, F launch, int* d_out, unsigned long long* d_st,
             std::vector<unsigned long long>& h_st) {
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        launch(blocks, true);
        hipEventRecord(e0);
        launch(blocks, false);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const size_t nw = (size_t)blocks * 4;
        hipMemcpy(h_st.data(), d_st, nw * 16, hipMemcpyDeviceToHost);
        const double clk = clock_mhz(h_st, nw);
        const double cyc = ms * 1e-3 * clk * 1e6 / wps;   // SIMD cycles per wave's share
        if (steps_per_wave > 0) printf("  %dw: %6.1f /step %5.2f /inst (%4.0f MHz)", wps, cyc / steps_per_wave, cyc / insts_per_wave, clk);
        else printf("  %dw: %5.2f (%4.0f MHz)", wps, cyc / insts_per_wave, clk);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    int* d_out;
    unsigned long long* d_st;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_st, 256 * 8 * 4 * 16);
    std::vector<unsigned long long> h_st(256 * 8 * 4 * 2);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# device %s, %d CUs; SIMD cycles per wave-instruction (and per band step) from the launch time at the measured clock, 1 / 2 / 4 waves per SIMD\n",
           prop.gcnArchName, prop.multiProcessorCount);
    const int iters = 200;
    const double n4 = 4.0 * 64 * iters;
#define CND(K, N) measure(N, n4, 0, [&](int blocks, bool warm) { hipLaunchKernelGGL(k_cnd<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, warm ? 10 : iters, 1, 0x5555555555555555ull); }, d_out, d_st, h_st);
    CND(0, "v_cndmask_b32 (VOP2, vcc set once, dst = src0)")
    CND(1, "v_cndmask_b32_e64 (SGPR-pair mask, dst = src0)")
    CND(2, "v_cndmask_b32_e64 (SGPR-pair mask, dst != srcs)")
    CND(3, "v_and_b32 with a mask register (replacement)")
    CND(4, "v_cmp_gt_i32 + v_cndmask_b32 pairs (per instr.)")
    CND(5, "s_cselect_b64 vcc + v_cndmask + 3 v_add (x4/grp)")
    CND(6, "v_cmp, 3 v_add, v_cndmask (5 instr. per 4 counted)")
    CND(7, "sub, ashr, and, sub: select without VCC")
    const int siters = 4000;   // pairs of steps
#define STEP(K, N, INSTS) measure(N, (INSTS) * 2.0 * siters, 2.0 * siters, [&](int blocks, bool warm) { hipLaunchKernelGGL(k_step<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, warm ? 50 : siters); }, d_out, d_st, h_st);
    STEP(0, "band step, 44 VALU + 3 LDS reads + waitcnt", 44.0)
    STEP(1, "band step, VALU only (44 instructions)", 44.0)
    STEP(2, "16 cells only (32 instructions, chain)", 32.0)
    return 0;
}
