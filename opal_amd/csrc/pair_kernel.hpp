// pair_kernel.hpp -- ONE query x target pair spread over SEVERAL waves (NW / HW / OV, int32): the longest targets of a
// global-mode search (reference recurrence src/opal.cpp:748-771, boundary conditions :671-689, result rules :844-904).
//
// What the other kernels do with such a target: one wave, 64 lanes x ceil(Q / 64) rows, one column step after the other --
// 35 276 dependent steps of 32 int32 rows for a 35 213-residue target against a 2000-aa query, 22 ms of a lone wave on a
// search whose bulk takes 9 (the role-swapped kernels do not help there: (Q + 63) steps per 512 target residues). Here the
// query's rows are spread over W waves of ONE workgroup (W x 64 lanes x S rows, S = 2 / 4 / 8): a step is 2-8 rows instead of
// 32, and the waves form one systolic array 64 W lanes long -- wave w + 1 runs D = 160 steps behind wave w and takes the
// bottom row of wave w's rows (H and the F input of the row below, one 8-byte word per column) from a boundary row in
// global memory, as the row chunks of a query beyond one wave do between launches (dp_kernel.hpp, FLAG_MULTI) -- only that
// all chunks run at once. All waves step through a common clock tau with a workgroup barrier every 8 steps: wave w works on
// its step tau - w D. The boundary words reach lane 0 the way the residues do: every 64 steps lane k loads the word of column
// s + k (one coalesced read; a load per step sat on lane 0's critical path: ~600 clocks of L2 latency per 240-clock step), and
// the queue moves one lane towards lane 0 per step; a word is read at least 32 steps (four barriers) after it was written, by a
// wave of the same CU.
//
//   * rows: lane g = 64 w + k of the array holds the S query rows g S + j - padTop; HW / OV pad ABOVE the query (virtual rows
//     with profile 0 reproduce the zero top boundary), NW pads BELOW it (pad rows never feed a real one);
//   * the query profile -- 2-byte entries (the host checks that the matrix fits), [letter][lane g][row j], staged once per
//     workgroup for all W waves: (A + 1) x 64 W S x 2 bytes (2000-aa query: 102 KB);
//   * every wave reads the target's residues itself and runs its own lane queue of profile addresses (64 columns per refill);
//   * int32 arithmetic, no saturation: the host has proven every value inside +-2^30 (the 32-bit rung of the ladder);
//   * results: NW the final cell; HW the last row's maximum and its first column; OV also the last column's maximum with the
//     smallest row, reduced over the lanes of a wave and then over the waves (upper rows win ties).
#pragma once
#include "dp_kernel.hpp"

#define PAIR_MAX_WAVES 16
#define PAIR_LAG 160    // steps between consecutive waves: 64 (lane pipeline) + 64 (boundary words are read a batch of 64 columns at a time) + 32 (four barriers between a word's store and its load)
#define PAIR_NEG (-(1 << 30))

struct PairParams {
    const unsigned char* query;
    const int* matrix;
    int Q, A, Go, Ge;
    const unsigned char* res;
    const uint64_t* seqOff;
    const int* pairLen;
    const int* pairSeq;
    int firstItem, numItems;  // items [firstItem, +numItems): one workgroup each
    int* outScore;
    int* outEndQ;
    int* outEndT;
    unsigned long long* bnd;      // boundary rows: row w (written by wave w, read by wave w + 1) at bnd + w * bndStride
    unsigned long long bndBase;   // residue offset of the first item of the launch (a column's word sits at its residue offset - bndBase)
    unsigned long long bndStride; // words per boundary row
    int* started;                 // (may be null) every workgroup adds one when it starts: opal_hip_wait_started
    unsigned long long* clk;
};
typedef void (*PairFn)(const PairParams);

__host__ __device__ inline size_t opal_pair_lds_bytes(int waves, int S, int A) {
    return (size_t)(A + 1) * (size_t)waves * 64 * (size_t)S * 2 + 16 * PAIR_MAX_WAVES + 64;
}

template <int MODE, int S>
__global__ void __launch_bounds__(PAIR_MAX_WAVES * 64) opal_pair_kernel(const PairParams p) {
    static_assert(MODE != MODE_SW, "local alignment: swap_kernel.hpp / the streaming kernels");
    static_assert(S == 2 || S == 4 || S == 8, "rows per lane: one ds_read of 4 / 8 / 16 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int W = (int)(blockDim.x >> 6);
    const int R = W * 64 * S;  // rows of the whole array
    const int Q = p.Q, A = p.A, Go = p.Go, Ge = p.Ge;
    const int padTop = (MODE == MODE_NW) ? 0 : R - Q;
    const unsigned stride = (unsigned)R * 2u;  // one letter's profile
    int* const waveBest = reinterpret_cast<int*>(lds + (size_t)(A + 1) * stride);  // [W][2]: last-column maximum and its row, per wave (OV)
    if (threadIdx.x == 0 && p.started) atomicAdd(p.started, 1);

    // ---- the profile of all W waves: QP[letter][g][j] = M[query[g S + j - padTop]][letter]; letter A = beyond the target
    for (int e = threadIdx.x; e < R; e += blockDim.x) {
        const int row = e - padTop;
        const bool real = row >= 0 && row < Q;
        const int* mrow = p.matrix + (real ? (int)p.query[row] * A : 0);
        const int fill = row < 0 ? 0 : PAIR_NEG >> 16;  // virtual rows above the query: transparent; below it (NW): never the maximum
        unsigned char* dst = lds + (size_t)e * 2;
        for (int letter = 0; letter < A; letter++) *reinterpret_cast<short*>(dst + (size_t)letter * stride) = (short)(real ? mrow[letter] : fill);
        *reinterpret_cast<short*>(dst + (size_t)A * stride) = (short)(row < 0 ? 0 : PAIR_NEG >> 16);
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(2);  // these workgroups are the critical path of their search

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k = lane;
    const int g = w * 64 + k;
    const int item = p.firstItem + (int)blockIdx.x;
    int len = p.pairLen[item];
    const int seq = p.pairSeq[item];
    if (seq < 0) len = 0;
    const uint64_t tokBase = p.seqOff[item];
    const int nsteps = len > 0 ? len + 63 : 0;
    const int total = len > 0 ? nsteps + (W - 1) * PAIR_LAG : 0;
    const bool hasPrev = w > 0, hasNext = w < W - 1;
    unsigned long long* const bndRead = p.bnd + (size_t)(w > 0 ? w - 1 : 0) * p.bndStride + (tokBase - p.bndBase);
    unsigned long long* const bndWrite = p.bnd + (size_t)w * p.bndStride + (tokBase - p.bndBase);

    // ---- state: H[r][c-1] and E[r][c] of my rows (column -1: opal.cpp:671-682)
    const int r0 = g * S - padTop;  // query row in register 0 (negative: virtual)
    int Hp[S], En[S];
#pragma unroll
    for (int j = 0; j < S; j++) {
        if (MODE == MODE_OV || r0 + j < 0) { Hp[j] = 0; En[j] = -Go; }
        else { Hp[j] = -Go - (r0 + j) * Ge; En[j] = Hp[j] - Go; }
    }
    // what the lane below reads before my first column: H[my last row][-1]
    int outH = ((MODE == MODE_NW || MODE == MODE_HW) && r0 + S - 1 >= 0) ? -Go - (r0 + S - 1) * Ge : 0, outF = 0;
    // H[r0 - 1][-1]: the diagonal input of my first column
    int diagIn = 0;
    if ((MODE == MODE_NW || MODE == MODE_HW) && r0 - 1 >= 0) diagIn = -Go - (r0 - 1) * Ge;
    int topHv = -Go;  // NW: H[-1][c], kept incrementally (wave 0, lane 0)

    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned laneOff = (unsigned)g * (unsigned)(2 * S);
    const unsigned padOff = ldsBase + (unsigned)A * stride;
    unsigned offv = padOff + laneOff;
    unsigned tokq = padOff;      // lane queue of the profile rows of this batch's columns (lane 0 consumes)
    unsigned tokNext = 0;
    if (k < len) tokNext = p.res[tokBase + (uint64_t)k];
    unsigned bndqLo = 0, bndqHi = 0;  // lane queue of the boundary words {H, F} above this batch's columns (lane 0 of the waves below the first consumes)

    // results
    int best = INT_MIN, bestCol = -1, prevBest = INT_MIN;  // HW / OV: last-row maximum (first column), and the same over columns 0 .. L-2
    int lcMax = INT_MIN, lcRow = -1;                        // OV: last column's maximum over my rows, smallest row
    const int gLast = (MODE == MODE_NW) ? (Q - 1) / S : W * 64 - 1;   // array lane holding query row Q-1 ...
    const int jLast = (MODE == MODE_NW) ? (Q - 1) - gLast * S : S - 1; // ... and its register
    int jlo = padTop - g * S, jhi = Q + padTop - g * S;                // my registers [jlo, jhi) hold real query rows
    jlo = jlo < 0 ? 0 : (jlo > S ? S : jlo);
    jhi = jhi < 0 ? 0 : (jhi > S ? S : jhi);

    for (int tau = 0; tau < total; tau++) {
        const int s = tau - w * PAIR_LAG;  // my wave's step (wave-uniform)
        if (s >= 0 && s < nsteps) {
            // ---- profile address of this step (lane queue, refilled every 64 steps) and the boundary word of the next one
            if ((s & 63) == 0) {
                const unsigned tok = tokNext;
                const int col = s + k;
                if (col + 64 < len) tokNext = p.res[tokBase + (uint64_t)(col + 64)];
                tokq = (col < len) ? ldsBase + tok * stride : padOff;
                if (hasPrev && col < len) {  // (the wave above wrote column s + 63 at its step s + 126 = 34 of my steps ago)
                    const unsigned long long word = __hip_atomic_load(bndRead + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    bndqLo = (unsigned)(word & 0xffffffffull); bndqHi = (unsigned)(word >> 32);
                }
            }
            {
                const unsigned shifted = shr1(offv) + (unsigned)(2 * S);
                offv = (k == 0) ? tokq + (unsigned)(w * 64) * (unsigned)(2 * S) : shifted;
                tokq = shl1(tokq);
            }
            const unsigned bndH = bndqLo, bndF = bndqHi;  // lane 0: the word of column s
            bndqLo = shl1(bndqLo); bndqHi = shl1(bndqHi);

            // ---- inputs from the row above: the lane above, the wave above (boundary row), or the top boundary
            int inH = (int)shr1((unsigned)outH), inF = (int)shr1((unsigned)outF);
            if (k == 0) {
                if (hasPrev) { inH = (int)bndH; inF = (int)bndF; }
                else if (MODE == MODE_NW) { inH = topHv; inF = topHv - Go; topHv -= Ge; }  // H[-1][c] = -Go - c Ge (opal.cpp:686-689)
                else { inH = 0; inF = -Go; }
            }
            const int c = s - k;  // my column
            if (c >= 0) {
                int P[S];
                if constexpr (S == 8) {
                    const int4 a = lds_ld16(offv);
                    P[0] = (int)(short)a.x; P[1] = a.x >> 16; P[2] = (int)(short)a.y; P[3] = a.y >> 16;
                    P[4] = (int)(short)a.z; P[5] = a.z >> 16; P[6] = (int)(short)a.w; P[7] = a.w >> 16;
                } else if constexpr (S == 4) {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    const v2i a = *(__attribute__((address_space(3))) const v2i*)(uintptr_t)offv;
                    P[0] = (int)(short)a.x; P[1] = a.x >> 16; P[2] = (int)(short)a.y; P[3] = a.y >> 16;
                } else {
                    const int a = *(__attribute__((address_space(3))) const int*)(uintptr_t)offv;
                    P[0] = (int)(short)a; P[1] = a >> 16;
                }
                int diag = diagIn, f = inF, h = 0;
#pragma unroll
                for (int j = 0; j < S; j++) {  // reference core loop opal.cpp:748-771
                    const int e = En[j];
                    const int d = diag + P[j];
                    h = max(max(d, e), f);
                    const int hq = h - Go;
                    En[j] = max(e - Ge, hq);
                    f = max(f - Ge, hq);
                    diag = Hp[j];
                    Hp[j] = h;
                }
                outH = h;
                outF = f;
                if (hasNext && k == 63 && c < len)
                    __hip_atomic_store(bndWrite + c, (unsigned long long)(unsigned)outH | ((unsigned long long)(unsigned)outF << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == MODE_NW) {
                    if (g == gLast && c == len - 1) {  // final cell, once per target
                        int hl = Hp[0];
#pragma unroll
                        for (int j = 1; j < S; j++) hl = (jLast == j) ? Hp[j] : hl;
                        best = hl;
                    }
                } else if (g == gLast && c < len) {
                    // last query row = last register of the last lane of the array: strict >, the first column wins (opal.cpp:824-834)
                    if (c == len - 1) prevBest = best;
                    if (outH > best) { best = outH; bestCol = c; }
                }
                if (MODE == MODE_OV && c == len - 1) {  // the last column over my real rows: the value, then the smallest row
                    int bv = INT_MIN, bj = -1;
#pragma unroll
                    for (int j = S - 1; j >= 0; j--)
                        if (j >= jlo && j < jhi && Hp[j] >= bv) { bv = Hp[j]; bj = j; }
                    if (bj >= 0) { lcMax = bv; lcRow = r0 + bj; }
                }
            }
            diagIn = inH;
        }
        if ((tau & 7) == 7) __syncthreads();
    }

    // ---- results (reference opal.cpp:844-904)
    if (MODE == MODE_OV) {
        int v = lcMax, r = lcRow;
        for (int d = 1; d < 64; d <<= 1) {
            const int ov = __shfl_xor(v, d, 64), orow = __shfl_xor(r, d, 64);
            if (ov > v || (ov == v && orow >= 0 && (r < 0 || orow < r))) { v = ov; r = orow; }
        }
        __syncthreads();
        if (k == 0) { waveBest[2 * w] = v; waveBest[2 * w + 1] = r; }
        __syncthreads();
        lcMax = INT_MIN; lcRow = -1;
        for (int ww = 0; ww < W; ww++)  // upper waves hold the smaller rows: they win ties
            if (waveBest[2 * ww + 1] >= 0 && waveBest[2 * ww] > lcMax) { lcMax = waveBest[2 * ww]; lcRow = waveBest[2 * ww + 1]; }
    }
    if (g == gLast && seq >= 0 && len > 0) {
        int sc = best, eq = Q - 1, et = (MODE == MODE_NW) ? len - 1 : bestCol;
        if (MODE == MODE_OV && lcRow >= 0 && lcMax > prevBest) { sc = lcMax; et = len - 1; eq = lcRow; }
        p.outScore[seq] = sc;
        if (p.outEndQ) { p.outEndQ[seq] = eq; p.outEndT[seq] = et; }
    }
}

PairFn opal_pair_fn(int mode, int rows);  // inst_pair.hip: rows 2 / 4 / 8, NW / HW / OV
