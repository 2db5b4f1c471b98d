// How fast can the SW row block (8 dependent-ish full-rate 16-bit ops + 1 add) issue, by waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned reg;
#define S 38
#define ITER 400
__device__ __forceinline__ void row_sw(reg d, reg& e, reg& f, reg& acc, reg& h, reg Qv, reg Rv) {
    reg hq;
    asm("v_max_i16_e32 %0, %5, %1\n\t"
        "v_max_i16_e32 %0, %0, %2\n\t"
        "v_sub_u16_e32 %1, %1, %7\n\t"
        "v_sub_u16_e64 %4, %0, %6 clamp\n\t"
        "v_sub_u16_e32 %2, %2, %7\n\t"
        "v_max_i16_e32 %3, %3, %0\n\t"
        "v_max_i16_e32 %1, %1, %4\n\t"
        "v_max_i16_e32 %2, %2, %4"
        : "=&v"(h), "+v"(e), "+v"(f), "+v"(acc), "=&v"(hq)
        : "v"(d), "v"(Qv), "v"(Rv));
}
__device__ __forceinline__ reg add16(reg a, reg b) { reg r; asm("v_add_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, int iters) {
    reg Hp[S], En[S];
    for (int j = 0; j < S; j++) { Hp[j] = (threadIdx.x * 7 + j) & 63; En[j] = (threadIdx.x + j * 3) & 31; }
    reg mx = 0, f = 1, diag = 0, h = 0;
    const reg Qv = 11, Rv = 1;
    reg P = seed & 7;
    for (int it = 0; it < iters; it++) {
        reg d = add16(diag, P);
#pragma unroll
        for (int j = 0; j < S; j++) {
            reg dn = 0;
            if (j + 1 < S) dn = add16(Hp[j], P + j);
            row_sw(d, En[j], f, mx, h, Qv, Rv);
            Hp[j] = h;
            d = dn;
        }
        diag = h & 15;
    }
    reg r = mx + f;
    for (int j = 0; j < S; j++) r += Hp[j] + En[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 1; w <= 4; w++) {
        int blocks = 256 * w;  // 256-thread blocks: one wave per SIMD each
        k<<<blocks, 256>>>(d, 1, ITER);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<<<blocks, 256>>>(d, 3, ITER);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        double rows = (double)ITER * S * w;  // row blocks per SIMD
        printf("waves/SIMD=%d: %.3f ms, %.2f ns per row block per SIMD (= %.1f clk at 2.25 GHz; ideal 9 x 2 = 18)\n", w, ms,
               ms * 1e6 / rows, ms * 1e6 / rows * 2.25);
    }
    return 0;
}
