// rev_kernel.hip -- start locations of alignments on the GPU.
//
// What it replaces: the score-limited reverse pass of the reference's OPAL_SEARCH_ALIGNMENT,
// reference src/opal.cpp:1475-1507 (driver) and the DP part of findAlignment :1236-1362. Given the score
// and the end cell (endQ, endT) of a sequence -- both already in device memory after the search -- the
// reversed query prefix [0..endQ] is aligned with the reversed target prefix [0..endT] from the corner
// (leading gaps charged as in NW, :1266-1269 / :1283-1287); the first column in which a cell the mode allows
// as an end reaches the score, then the first row of that column, is where the forward alignment starts
// (:1275, :1340-1362, :1498-1499). Only that cell is produced here; the operations come from the host
// traceback over the (start..end) box.
//
// Mapping: one wavefront per sequence; lane k owns S consecutive rows of the reversed prefix in VGPRs,
// columns travel down the lanes one step apart (the same systolic hand-off as dp_kernel.hpp: bottom H and
// next-row F through one wave_shr:1 DPP move each). The prefix differs per sequence (it ends at ITS endQ),
// so there is no shared query profile: the score matrix sits in LDS once per block and a lane keeps the
// matrix-row offsets of its S query letters in registers. Plain int32 arithmetic (the search has already
// returned without overflow for these sequences). The wave stops 64 steps after the first lane has seen
// the score: later columns cannot hold an earlier cell.
//
// The reference picks the ROW by scanning the whole stop column for the first cell equal to the score
// (:1353-1357), also over cells the mode does not allow as an end (possible in OV only). Such a coincidence
// is reported through `flags` and the host repeats that sequence with its exact full pass.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "opal_hip.h"

#define REV_MODE_NW 0
#define REV_MODE_HW 1
#define REV_MODE_OV 2
#define REV_MODE_SW 3
#define REV_NEG (-(1 << 28))
#define REV_WAVES_PER_BLOCK 4

struct RevParams {
    const unsigned char* query;
    int Q, A, Go, Ge, mode;
    const int* matrix;
    const unsigned char* res;   // residues, one byte per column (item order)
    const uint64_t* seqOff;     // [items] start of each sequence
    const int* seqItem;   // caller index -> pair*2 + half
    const int* items;     // caller indices to process (NULL: 0..numItems-1)
    int numItems;
    const int* score; const int* endQ; const int* endT;
    int* startQ; int* startT;
    int* flags;           // [2]: [0] sequences whose start must be recomputed by the host (OV coincidence), [1] not found
    int* redo;            // list of caller indices counted in flags[0]
    int exactRow;         // second pass: the stop column is known (startT), find the first row of it holding the score
};

__device__ __forceinline__ int rev_shr1(int v) {  // lane i receives lane i-1's value (lane 0: 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}

template <int S>
__global__ void __launch_bounds__(REV_WAVES_PER_BLOCK * 64) opal_rev_kernel(const RevParams p) {
    extern __shared__ int mlds[];  // (A+1) x A score matrix, row A = padding (REV_NEG)
    const int A = p.A;
    for (int i = threadIdx.x; i < (A + 1) * A; i += blockDim.x) mlds[i] = i < A * A ? p.matrix[i] : REV_NEG;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int w = (int)blockIdx.x * REV_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
    if (w >= p.numItems) return;
    const int seq = p.items ? p.items[w] : w;
    const int score = p.score[seq];
    const int eQ = p.endQ[seq], eT = p.endT[seq];
    const int aq = eQ + 1, at = eT + 1;  // lengths of the reversed prefixes
    const int mode = p.mode;
    if ((mode == REV_MODE_SW && score == 0) || aq <= 0 || at <= 0) {  // no alignment (opal.cpp:1479-1483) / empty shapes
        if (lane == 0) { p.startQ[seq] = -1; p.startT[seq] = -1; }
        return;
    }
    if (mode == REV_MODE_NW) {  // the reverse pass ends in the far corner: the alignment starts at (0, 0)
        if (lane == 0) { p.startQ[seq] = 0; p.startT[seq] = 0; }
        return;
    }
    const int Go = p.Go, Ge = p.Ge;
    // second pass: column (of the reversed target prefix) in which the first pass stopped
    const int exactC = p.exactRow ? eT - p.startT[seq] : -1;
    if (p.exactRow && (p.startT[seq] < 0 || exactC < 0 || exactC >= at)) {  // no stop column: the host's full pass
        if (lane == 0) p.redo[atomicAdd(p.flags, 1)] = seq;
        return;
    }
    const int item = p.seqItem[seq];
    const unsigned char* tok = p.res + p.seqOff[item];

    const int r0 = lane * S;  // my first row of the reversed prefix
    int Hp[S], En[S], qoff[S];
#pragma unroll
    for (int j = 0; j < S; j++) {
        const int r = r0 + j;
        const int letter = r < aq ? (int)p.query[eQ - r] : A;  // rows beyond the prefix: padding, never feed a real row
        qoff[j] = letter * A;
        Hp[j] = -Go - r * Ge;      // H[r][-1] (opal.cpp:1266-1269)
        En[j] = Hp[j] - Go;        // E[r][0] = max(H[r][-1] - Go, -inf)
    }
    const int jl = aq - 1 - r0;    // register holding the last row (in [0,S) on exactly one lane)
    int outH = -Go - (r0 + S - 1) * Ge;  // what the lane below reads before my first column: H[last row][-1]
    int outF = 0;
    int diagIn = 0;
    int found = 0, bestC = INT_MAX, bestR = INT_MAX;
    // OV only: one bit per step, newest in bit 0 of hist0 -- "a real cell that is NOT an allowed end held the score in
    // the column I computed that step". 128 steps of history cover the stop column on every lane (see below).
    unsigned long long hist0 = 0, hist1 = 0;
    int sLast = 0;
    int tLetter = 0;
    int limit = p.exactRow ? exactC + 64 : at + 63;  // steps: lane k computes column s - k
    int exactR = INT_MAX;  // second pass: smallest row of mine holding the score in column exactC
    for (int s = 0; s < limit; s++) {
        // target letters travel down the lanes with the columns; lane 0 takes column s from memory
        int fresh = 0;
        if (s < at) {
            const unsigned t = tok[eT - s];
            fresh = (int)t;
        }
        const int shifted = rev_shr1(tLetter);
        tLetter = lane == 0 ? fresh : shifted;
        int inH = rev_shr1(outH);
        int inF = rev_shr1(outF);
        const int c = s - lane;
        int coincidence = 0;
        if (lane == 0) {  // top boundary: H[-1][c] = -Go - c*Ge, F[0][c] = H[-1][c] - Go (opal.cpp:1283-1287)
            inH = -Go - c * Ge;
            inF = inH - Go;
        }
        if (c >= 0 && c < at) {
            int diag = lane == 0 ? (c == 0 ? 0 : -Go - (c - 1) * Ge) : diagIn;
            int f = inF, h = 0, cm = REV_NEG;
            const bool lastCol = (c == at - 1);
#pragma unroll
            for (int j = 0; j < S; j++) {
                const int P = mlds[qoff[j] + tLetter];
                const int e = En[j];
                const int d = diag + P;
                h = max(d, max(e, f));
                const bool real = r0 + j < aq;
                const bool allowed = mode == REV_MODE_SW ? real : (j == jl || (mode == REV_MODE_OV && lastCol && real));
                cm = max(cm, allowed ? h : REV_NEG);
                if (mode == REV_MODE_OV) coincidence |= (h == score && real && !allowed);
                if (c == exactC && h == score && real && exactR == INT_MAX) exactR = r0 + j;
                const int hq = h - Go;
                En[j] = max(e - Ge, hq);
                f = max(f - Ge, hq);
                diag = Hp[j];
                Hp[j] = h;
            }
            outH = h;
            outF = f;
            if (!found && cm >= score) {  // first column of mine in which an allowed cell reaches the score (opal.cpp:1275)
                int r = -1;
#pragma unroll
                for (int j = S - 1; j >= 0; j--) {
                    const bool real = r0 + j < aq;
                    const bool allowed = mode == REV_MODE_SW ? real : (j == jl || (mode == REV_MODE_OV && lastCol && real));
                    if (allowed && Hp[j] >= score) r = r0 + j;
                }
                found = 1; bestC = c; bestR = r;
            }
        }
        diagIn = inH;
        hist1 = (hist1 << 1) | (hist0 >> 63);
        hist0 = (hist0 << 1) | (unsigned long long)(coincidence & 1);
        sLast = s;
        // The first hit in TIME (step s0) need not be the smallest column, but a smaller one, c* <= s0, is computed by
        // lane k at step c* + k <= s0 + 63: run 64 more steps. (And c* + k >= s0 - 63, so 128 bits of history suffice.)
        if (!p.exactRow && __ballot(found) != 0ull) limit = min(limit, s + 65);
    }
    if (p.exactRow) {  // the reference's row scan of the stop column (opal.cpp:1353-1357): first row holding the score
        int er = exactR;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(er, d, 64); er = o < er ? o : er; }
        if (lane == 0) {
            if (er == INT_MAX) p.redo[atomicAdd(p.flags, 1)] = seq;  // (cannot happen: the first pass found the score there)
            else p.startQ[seq] = eQ - er;
        }
        return;
    }
    // the slot's answer: smallest column, then smallest row
    int bc = bestC, br = bestR;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oc = __shfl_xor(bc, d, 64), orow = __shfl_xor(br, d, 64);
        if (oc < bc || (oc == bc && orow < br)) { bc = oc; br = orow; }
    }
    // Did a cell that is not an allowed end hold the score in the stop column bc? (The reference's row scan,
    // opal.cpp:1353-1357, would pick it.) Lane k computed column bc at step bc + k.
    int co = 0;
    if (mode == REV_MODE_OV && bc != INT_MAX) {
        const int age = sLast - (bc + lane);
        if (age >= 0 && age < 64) co = (int)((hist0 >> age) & 1ull);
        else if (age >= 64 && age < 128) co = (int)((hist1 >> (age - 64)) & 1ull);
        else if (age >= 128) co = 1;  // cannot happen (see the loop); be safe
    }
    co = __ballot(co != 0) != 0ull;
    if (lane == 0) {
        if (bc == INT_MAX) {
            p.startQ[seq] = -1; p.startT[seq] = -1;
            atomicAdd(p.flags + 1, 1);
        } else {
            p.startQ[seq] = eQ - br;   // opal.cpp:1498-1499
            p.startT[seq] = eT - bc;
        }
        if (co || bc == INT_MAX) p.redo[atomicAdd(p.flags, 1)] = seq;
    }
}

typedef void (*RevFn)(const RevParams);

extern "C" int opal_hip_rev_rows_supported(int query_length) {
    static const int rows[] = {4, 8, 16, 24, 32, 40};
    for (int r : rows)
        if (64 * r >= query_length) return r;
    return 0;
}

extern "C" int opal_hip_rev_launch(const OpalHipRevLaunch* a) {
    const int S = opal_hip_rev_rows_supported(a->query_length);
    RevFn fn = nullptr;
    switch (S) {
    case 4: fn = opal_rev_kernel<4>; break;
    case 8: fn = opal_rev_kernel<8>; break;
    case 16: fn = opal_rev_kernel<16>; break;
    case 24: fn = opal_rev_kernel<24>; break;
    case 32: fn = opal_rev_kernel<32>; break;
    case 40: fn = opal_rev_kernel<40>; break;
    }
    if (!fn || a->mode < 0 || a->mode > 3) return 2;
    if (a->num_items <= 0) return 0;
    const size_t lds = (size_t)(a->alphabet_length + 1) * a->alphabet_length * sizeof(int);
    if (lds > 64 * 1024) return 2;
    RevParams p;
    p.query = a->query; p.Q = a->query_length; p.A = a->alphabet_length; p.Go = a->gap_open; p.Ge = a->gap_ext; p.mode = a->mode;
    p.matrix = a->score_matrix; p.res = a->residues; p.seqOff = a->seq_offset; p.seqItem = a->seq_item;
    p.items = a->items; p.numItems = a->num_items;
    p.score = a->score; p.endQ = a->end_query; p.endT = a->end_target;
    p.startQ = a->start_query; p.startT = a->start_target; p.flags = a->flags; p.redo = a->redo; p.exactRow = a->exact_row;
    const int blocks = (a->num_items + REV_WAVES_PER_BLOCK - 1) / REV_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(REV_WAVES_PER_BLOCK * 64), lds, (hipStream_t)a->stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
