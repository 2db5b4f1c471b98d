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

KERNEL(k0, B8("v_add_u32_e32", "%8"))
KERNEL(k1, B8("v_sub_u32_e32", "%8"))
KERNEL(k2, B8("v_max_i32_e32", "%8"))
KERNEL(k3, B8("v_min_i32_e32", "%8"))
KERNEL(k4, B8("v_max_u32_e32", "%8"))
KERNEL(k5, B8("v_and_b32_e32", "%8"))
KERNEL(k6, B8("v_or_b32_e32", "%8"))
KERNEL(k7, B8("v_xor_b32_e32", "%8"))
KERNEL(k8, B8("v_lshlrev_b32_e32", "%8"))
KERNEL(k9, B8("v_ashrrev_i32_e32", "%8"))
KERNEL(k10, B8("v_add_f32_e32", "%8"))
KERNEL(k11, B8("v_sub_f32_e32", "%8"))
KERNEL(k12, B8("v_mul_f32_e32", "%8"))
KERNEL(k13, B8("v_max_f32_e32", "%8"))
KERNEL(k14, B8("v_min_f32_e32", "%8"))
KERNEL(k15, B8("v_fma_f32", "%8, %9"))
KERNEL(k16, B8("v_max3_f32", "%8, %9"))
KERNEL(k17, B8("v_med3_f32", "%8, %9"))
KERNEL(k18, B8("v_min3_f32", "%8, %9"))
KERNEL(k19, B8("v_add_f16_e32", "%8"))
KERNEL(k20, B8("v_max_f16_e32", "%8"))
KERNEL(k21, B8("v_pk_add_f16", "%8"))
KERNEL(k22, B8("v_pk_max_f16", "%8"))
KERNEL(k24, B8("v_add_u16_e32", "%8"))
KERNEL(k25, B8("v_sub_u16_e32", "%8"))
KERNEL(k26, B8("v_max_i16_e32", "%8"))
KERNEL(k27, B8("v_max_u16_e32", "%8"))
KERNEL(k28, B8("v_min_i16_e32", "%8"))
KERNEL(k29, B8("v_add_i16", "%8 clamp"))
KERNEL(k30, B8("v_sub_i16", "%8 clamp"))
KERNEL(k31, B8("v_add_u16_e64", "%8 clamp"))
KERNEL(k32, B8("v_sub_u16_e64", "%8 clamp"))
KERNEL(k33, B8("v_max3_i16", "%8, %9"))
KERNEL(k34, B8("v_mul_lo_u32", "%8"))
KERNEL(k35, B8("v_mul_u32_u24_e32", "%8"))
KERNEL(k36, B8("v_add_i32", "%8 clamp"))
KERNEL(k37, B8("v_sub_i32", "%8 clamp"))
KERNEL(k39, B8("v_cndmask_b32_e32", "%8, vcc"))
KERNEL(k40, B8("v_bfe_i32", "%8, 16"))
KERNEL(k41, B8("v_lshl_add_u32", "%8, %9"))
KERNEL(k42, B8("v_add_lshl_u32", "%8, %9"))
KERNEL(k43, B8("v_sad_u32", "%8, %9"))

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
    printf("%-40s w/SIMD=%d : %.2f ns/instr/SIMD\n", name, wavesPerSimd, ms * 1e6 / instr);
    (void)hipFree(d);
}
int main() {
    run("v_add_u32_e32 %%8", k0, 4); run("v_add_u32_e32 %%8", k0, 8);
    run("v_sub_u32_e32 %%8", k1, 4); run("v_sub_u32_e32 %%8", k1, 8);
    run("v_max_i32_e32 %%8", k2, 4); run("v_max_i32_e32 %%8", k2, 8);
    run("v_min_i32_e32 %%8", k3, 4); run("v_min_i32_e32 %%8", k3, 8);
    run("v_max_u32_e32 %%8", k4, 4); run("v_max_u32_e32 %%8", k4, 8);
    run("v_and_b32_e32 %%8", k5, 4); run("v_and_b32_e32 %%8", k5, 8);
    run("v_or_b32_e32 %%8", k6, 4); run("v_or_b32_e32 %%8", k6, 8);
    run("v_xor_b32_e32 %%8", k7, 4); run("v_xor_b32_e32 %%8", k7, 8);
    run("v_lshlrev_b32_e32 %%8", k8, 4); run("v_lshlrev_b32_e32 %%8", k8, 8);
    run("v_ashrrev_i32_e32 %%8", k9, 4); run("v_ashrrev_i32_e32 %%8", k9, 8);
    run("v_add_f32_e32 %%8", k10, 4); run("v_add_f32_e32 %%8", k10, 8);
    run("v_sub_f32_e32 %%8", k11, 4); run("v_sub_f32_e32 %%8", k11, 8);
    run("v_mul_f32_e32 %%8", k12, 4); run("v_mul_f32_e32 %%8", k12, 8);
    run("v_max_f32_e32 %%8", k13, 4); run("v_max_f32_e32 %%8", k13, 8);
    run("v_min_f32_e32 %%8", k14, 4); run("v_min_f32_e32 %%8", k14, 8);
    run("v_fma_f32 %%8, %%9", k15, 4); run("v_fma_f32 %%8, %%9", k15, 8);
    run("v_max3_f32 %%8, %%9", k16, 4); run("v_max3_f32 %%8, %%9", k16, 8);
    run("v_med3_f32 %%8, %%9", k17, 4); run("v_med3_f32 %%8, %%9", k17, 8);
    run("v_min3_f32 %%8, %%9", k18, 4); run("v_min3_f32 %%8, %%9", k18, 8);
    run("v_add_f16_e32 %%8", k19, 4); run("v_add_f16_e32 %%8", k19, 8);
    run("v_max_f16_e32 %%8", k20, 4); run("v_max_f16_e32 %%8", k20, 8);
    run("v_pk_add_f16 %%8", k21, 4); run("v_pk_add_f16 %%8", k21, 8);
    run("v_pk_max_f16 %%8", k22, 4); run("v_pk_max_f16 %%8", k22, 8);
    run("v_add_u16_e32 %%8", k24, 4); run("v_add_u16_e32 %%8", k24, 8);
    run("v_sub_u16_e32 %%8", k25, 4); run("v_sub_u16_e32 %%8", k25, 8);
    run("v_max_i16_e32 %%8", k26, 4); run("v_max_i16_e32 %%8", k26, 8);
    run("v_max_u16_e32 %%8", k27, 4); run("v_max_u16_e32 %%8", k27, 8);
    run("v_min_i16_e32 %%8", k28, 4); run("v_min_i16_e32 %%8", k28, 8);
    run("v_add_i16 %%8 clamp", k29, 4); run("v_add_i16 %%8 clamp", k29, 8);
    run("v_sub_i16 %%8 clamp", k30, 4); run("v_sub_i16 %%8 clamp", k30, 8);
    run("v_add_u16_e64 %%8 clamp", k31, 4); run("v_add_u16_e64 %%8 clamp", k31, 8);
    run("v_sub_u16_e64 %%8 clamp", k32, 4); run("v_sub_u16_e64 %%8 clamp", k32, 8);
    run("v_max3_i16 %%8, %%9", k33, 4); run("v_max3_i16 %%8, %%9", k33, 8);
    run("v_mul_lo_u32 %%8", k34, 4); run("v_mul_lo_u32 %%8", k34, 8);
    run("v_mul_u32_u24_e32 %%8", k35, 4); run("v_mul_u32_u24_e32 %%8", k35, 8);
    run("v_add_i32 %%8 clamp", k36, 4); run("v_add_i32 %%8 clamp", k36, 8);
    run("v_sub_i32 %%8 clamp", k37, 4); run("v_sub_i32 %%8 clamp", k37, 8);
    run("v_cndmask_b32_e32 %%8, vcc", k39, 4); run("v_cndmask_b32_e32 %%8, vcc", k39, 8);
    run("v_bfe_i32 %%8, 16", k40, 4); run("v_bfe_i32 %%8, 16", k40, 8);
    run("v_lshl_add_u32 %%8, %%9", k41, 4); run("v_lshl_add_u32 %%8, %%9", k41, 8);
    run("v_add_lshl_u32 %%8, %%9", k42, 4); run("v_add_lshl_u32 %%8, %%9", k42, 8);
    run("v_sad_u32 %%8, %%9", k43, 4); run("v_sad_u32 %%8, %%9", k43, 8);
    return 0;
}
