// VALU micro-benchmarks for gfx950: issue rate of the instructions the DP kernel is made of.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define BC(T, x) __builtin_bit_cast(T, x)
#define ITER 2000
#define UNR 16

template <int OP> __device__ __forceinline__ unsigned op(unsigned a, unsigned b) {
    if (OP == 0) return (unsigned)max((int)a, (int)b);                                                  // v_max_i32
    if (OP == 1) return BC(unsigned, __builtin_elementwise_max(BC(s16x2, a), BC(s16x2, b)));            // v_pk_max_i16
    if (OP == 2) return BC(unsigned, __builtin_elementwise_add_sat(BC(s16x2, a), BC(s16x2, b)));        // v_pk_add_i16 clamp
    if (OP == 3) return BC(unsigned, __builtin_elementwise_sub_sat(BC(u16x2, a), BC(u16x2, b)));        // v_pk_sub_u16 clamp
    if (OP == 4) return __builtin_amdgcn_perm(a, b, 0x05040100u);                                       // v_perm_b32
    if (OP == 5) return a + b;                                                                          // v_add_u32
    if (OP == 6) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xf, 0xf, false) + b;   // dpp add
    return a;
}

// INDEP independent chains per lane
template <int OP, int INDEP> __global__ void k(unsigned* out, unsigned seed) {
    unsigned v[INDEP];
    for (int i = 0; i < INDEP; i++) v[i] = threadIdx.x * 7 + i + seed;
    unsigned b = seed | 1;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int i = 0; i < INDEP; i++) v[i] = op<OP>(v[i], b + u);
    }
    unsigned r = 0;
    for (int i = 0; i < INDEP; i++) r ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP, int INDEP> void run(const char* name, int wavesPerSimd) {
    unsigned* d; hipMalloc(&d, 256 * 4 * 64 * 8 * 4 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = one per SIMD
    k<OP, INDEP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP, INDEP><<<blocks, 256>>>(d, 3);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)ITER * UNR * INDEP * wavesPerSimd;  // wave-instructions per SIMD
    double clk = ms * 1e-3 * 2.4e9;
    printf("%-16s indep=%d waves/SIMD=%d : %.3f ms, %.2f clk/wave-instr (at 2.4 GHz)\n", name, INDEP, wavesPerSimd, ms, clk / instr);
    hipFree(d);
}

int main() {
    run<0, 8>("v_max_i32", 1); run<0, 8>("v_max_i32", 4);
    run<1, 8>("v_pk_max_i16", 1); run<1, 8>("v_pk_max_i16", 4); run<1, 1>("v_pk_max_i16", 1); run<1, 1>("v_pk_max_i16", 4); run<1, 2>("v_pk_max_i16", 4);
    run<2, 8>("v_pk_add_i16 sat", 4); run<3, 8>("v_pk_sub_u16 sat", 4);
    run<4, 8>("v_perm_b32", 4); run<5, 8>("v_add_u32", 4); run<5, 1>("v_add_u32", 1); run<6, 8>("dpp wave_shr add", 4);
    return 0;
}
