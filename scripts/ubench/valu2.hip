// VALU issue-rate micro-benchmarks for gfx950 with exact instructions (inline asm).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 1000
#define REP8(x) x x x x x x x x

#define KERNEL(name, body)                                                        \
    __global__ void name(unsigned* out, unsigned seed) {                          \
        unsigned v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7,  \
                 v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;          \
        unsigned b = seed | 1, c = seed * 3 + 2;                                  \
        for (int it = 0; it < ITER; it++) {                                       \
            asm volatile(REP8(body)                                               \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(b), "v"(c), "s"(seed));                            \
        }                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7; \
    }

// each body = 8 independent instructions on v0..v7 (operands %0..%7), %8/%9 = VGPR inputs, %10 = SGPR
#define B8(ins, tail) ins " %0, %0, " tail "\n" ins " %1, %1, " tail "\n" ins " %2, %2, " tail "\n" ins " %3, %3, " tail "\n" \
                      ins " %4, %4, " tail "\n" ins " %5, %5, " tail "\n" ins " %6, %6, " tail "\n" ins " %7, %7, " tail "\n"

KERNEL(k_max_e32, B8("v_max_i32_e32", "%8"))
KERNEL(k_max_e64, B8("v_max_i32_e64", "%8"))
KERNEL(k_max_sgpr, B8("v_max_i32_e64", "%10"))
KERNEL(k_max3, B8("v_max3_i32", "%8, %9"))
KERNEL(k_add_e32, B8("v_add_u32_e32", "%8"))
KERNEL(k_sub_clamp, B8("v_sub_u32_e64", "%8 clamp"))
KERNEL(k_add3, B8("v_add3_u32", "%8, %9"))
KERNEL(k_pk_max, B8("v_pk_max_i16", "%8"))
KERNEL(k_perm, B8("v_perm_b32", "%8, %9"))
KERNEL(k_mad24, B8("v_mad_i32_i24", "%8, %9"))
KERNEL(k_med3, B8("v_med3_i32", "%8, %9"))
KERNEL(k_max_i16, B8("v_max_i16_e64", "%8"))
KERNEL(k_add_sdwa, B8("v_add_u32_sdwa", "%8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"))
KERNEL(k_add_sdwa_sext, "v_add_u32_sdwa %0, %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %1, %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %2, %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %3, %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %4, %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %5, %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %6, %6, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_add_u32_sdwa %7, %7, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
// dependent chain: 8 dependent v_max_i32 on one register
KERNEL(k_dep_max, "v_max_i32_e32 %0, %0, %8\n v_max_i32_e32 %0, %0, %9\n v_max_i32_e32 %0, %0, %8\n v_max_i32_e32 %0, %0, %9\n"
                  "v_max_i32_e32 %0, %0, %8\n v_max_i32_e32 %0, %0, %9\n v_max_i32_e32 %0, %0, %8\n v_max_i32_e32 %0, %0, %9\n")
KERNEL(k_dep_max3, "v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %9, %8\n v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %9, %8\n"
                   "v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %9, %8\n v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %9, %8\n")

typedef void (*Fn)(unsigned*, unsigned);
static void run(const char* name, Fn f, int wavesPerSimd) {
    unsigned* d; (void)hipMalloc(&d, 256 * 4 * 64 * 8 * 4 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    int blocks = 256 * wavesPerSimd;
    f<<<blocks, 256>>>(d, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    f<<<blocks, 256>>>(d, 3);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double instr = (double)ITER * 64 * wavesPerSimd;
    printf("%-22s waves/SIMD=%d : %.3f ms  %.2f ns per wave-instr per SIMD  (x2.4GHz = %.2f clk)\n", name, wavesPerSimd, ms,
           ms * 1e6 / instr, ms * 1e-3 * 2.4e9 / instr);
    (void)hipFree(d);
}
#define RUN(k) run(#k, k, 1); run(#k, k, 4);
int main() {
    RUN(k_max_e32) RUN(k_max_e64) RUN(k_max_sgpr) RUN(k_max3) RUN(k_add_e32) RUN(k_sub_clamp) RUN(k_add3) RUN(k_pk_max) RUN(k_perm)
    RUN(k_mad24) RUN(k_med3) RUN(k_max_i16) RUN(k_add_sdwa) RUN(k_add_sdwa_sext) RUN(k_mov_dpp) RUN(k_dep_max) RUN(k_dep_max3)
    return 0;
}
