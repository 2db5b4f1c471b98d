// Two independent row streams interleaved inside one asm block: does intra-wave ILP lift the issue rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned reg;
#define S 19
#define ITER 400
__device__ __forceinline__ void row2(reg dA, reg& eA, reg& fA, reg& accA, reg& hA, reg dB, reg& eB, reg& fB, reg& accB, reg& hB, reg Qv, reg Rv) {
    reg hqA, hqB;
    asm("v_max_i16_e32 %0, %10, %1\n\t"
        "v_max_i16_e32 %5, %11, %6\n\t"
        "v_max_i16_e32 %0, %0, %2\n\t"
        "v_max_i16_e32 %5, %5, %7\n\t"
        "v_sub_u16_e32 %1, %1, %13\n\t"
        "v_sub_u16_e32 %6, %6, %13\n\t"
        "v_sub_u16_e64 %4, %0, %12 clamp\n\t"
        "v_sub_u16_e64 %9, %5, %12 clamp\n\t"
        "v_sub_u16_e32 %2, %2, %13\n\t"
        "v_sub_u16_e32 %7, %7, %13\n\t"
        "v_max_i16_e32 %3, %3, %0\n\t"
        "v_max_i16_e32 %8, %8, %5\n\t"
        "v_max_i16_e32 %1, %1, %4\n\t"
        "v_max_i16_e32 %6, %6, %9\n\t"
        "v_max_i16_e32 %2, %2, %4\n\t"
        "v_max_i16_e32 %7, %7, %9"
        : "=&v"(hA), "+v"(eA), "+v"(fA), "+v"(accA), "=&v"(hqA), "=&v"(hB), "+v"(eB), "+v"(fB), "+v"(accB), "=&v"(hqB)
        : "v"(dA), "v"(dB), "v"(Qv), "v"(Rv));
}
__device__ __forceinline__ reg add16(reg a, reg b) { reg r; asm("v_add_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, int iters) {
    reg HpA[S], EnA[S], HpB[S], EnB[S];
    for (int j = 0; j < S; j++) { HpA[j] = (threadIdx.x * 7 + j) & 63; EnA[j] = (threadIdx.x + j * 3) & 31; HpB[j] = (threadIdx.x * 5 + j) & 63; EnB[j] = (threadIdx.x + j * 9) & 31; }
    reg mxA = 0, fA = 1, diagA = 0, hA = 0, mxB = 0, fB = 2, diagB = 1, hB = 0;
    const reg Qv = 11, Rv = 1;
    reg P = seed & 7;
    for (int it = 0; it < iters; it++) {
        reg dA = add16(diagA, P), dB = add16(diagB, P + 1);
#pragma unroll
        for (int j = 0; j < S; j++) {
            reg dnA = 0, dnB = 0;
            if (j + 1 < S) { dnA = add16(HpA[j], P + j); dnB = add16(HpB[j], P + j + 2); }
            row2(dA, EnA[j], fA, mxA, hA, dB, EnB[j], fB, mxB, hB, Qv, Rv);
            HpA[j] = hA; HpB[j] = hB;
            dA = dnA; dB = dnB;
        }
        diagA = hA & 15; diagB = hB & 15;
    }
    reg r = mxA + fA + mxB + fB;
    for (int j = 0; j < S; j++) r += HpA[j] + EnA[j] + HpB[j] + EnB[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 1; w <= 5; w++) {
        int blocks = 256 * w;
        k<<<blocks, 256>>>(d, 1, ITER);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<<<blocks, 256>>>(d, 3, ITER);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        double rows = (double)ITER * S * w * 2;  // row blocks (single-sequence rows) per SIMD
        printf("2-stream waves/SIMD=%d: %.3f ms, %.2f ns per row per SIMD (= %.1f clk at 2.25 GHz; ideal 18)\n", w, ms,
               ms * 1e6 / rows, ms * 1e6 / rows * 2.25);
    }
    return 0;
}
