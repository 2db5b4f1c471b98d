// trace_kernel.hip -- alignment operations on the GPU: the traceback of OPAL_SEARCH_ALIGNMENT.
//
// What it replaces: the traceback part of the reference's findAlignment, reference src/opal.cpp:1236-1431 (flags
// :1378-1394, walk :1369-1409), and this repo's host version of it (opal_traceback.cpp), for alignments whose start and
// end cells are known (the search gives the end, rev_kernel.hip the start). Only the box between the two cells is
// computed: the reversed query prefix against the reversed target prefix from the corner, global (NW) recurrences, one
// byte of flags per cell -- where H came from (E, then F, then the diagonal: opal.cpp:1378-1386) and whether the E / F
// gap of this cell opens here (:1389, :1394) -- then a walk from the far corner back to the corner. The decisions are the
// host version's, so the operation strings are identical to the reference's.
//
// Mapping: opal_trace_kernel, one wavefront per alignment, the systolic layout of rev_kernel.hip (lane k owns S
// consecutive rows of the box in VGPRs, columns one step apart, int32 arithmetic); every lane packs the flags of its rows
// four to a dword and stores them into the alignment's flag matrix in HBM (column-major, column stride padded to 4
// bytes, so a column is one contiguous run over the lanes). opal_walk_kernel, one thread per alignment: the state machine
// of the walk, one dependent byte load per step (latency-bound, but tens of thousands of walks run side by side); the
// operations come out in forward order (the walk runs from the reversed box's far corner = the alignment's start).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "opal.h"
#include "opal_hip.h"

#define TR_NEG (-(1 << 28))
#define TR_WAVES_PER_BLOCK 4

struct TraceParams {
    const unsigned char* query;
    int A, Go, Ge;
    const int* matrix;
    const unsigned char* res;   // residues, one byte per column (item order)
    const uint64_t* seqOff;     // [items] start of each sequence
    const int* seqItem;          // caller index -> pair*2 + half
    const int* items;            // caller indices of this batch
    int numItems;
    const int* endQ; const int* endT; const int* startQ; const int* startT;   // caller order
    const uint64_t* dirOff;      // per batch entry: start of its flag matrix in `dirs`
    unsigned char* dirs;
    const uint64_t* opsOff;      // per batch entry: start of its operations in `ops` (capacity rows + columns)
    unsigned char* ops;
    int* opsLen;                 // per batch entry
    int* boxScore;               // per batch entry: H of the far corner (must equal the alignment's score)
};

__device__ __forceinline__ int tr_shr1(int v) {  // lane i receives lane i-1's value (lane 0: 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}

template <int S>
__global__ void __launch_bounds__(TR_WAVES_PER_BLOCK * 64) opal_trace_kernel(const TraceParams p) {
    extern __shared__ int mlds[];  // (A+1) x A score matrix, row A = padding
    const int A = p.A;
    for (int i = threadIdx.x; i < (A + 1) * A; i += blockDim.x) mlds[i] = i < A * A ? p.matrix[i] : TR_NEG;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int w = (int)blockIdx.x * TR_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
    if (w >= p.numItems) return;
    const int seq = p.items[w];
    const int eQ = p.endQ[seq], eT = p.endT[seq];
    const int bq = eQ - p.startQ[seq] + 1, bt = eT - p.startT[seq] + 1;  // rows and columns of the box
    const int ld = (bq + 3) & ~3;
    const int Go = p.Go, Ge = p.Ge;
    const int item = p.seqItem[seq];
    const unsigned char* tok = p.res + p.seqOff[item];
    unsigned char* const dirs = p.dirs + p.dirOff[w];

    const int r0 = lane * S;
    int Hp[S], En[S], qoff[S];
#pragma unroll
    for (int j = 0; j < S; j++) {
        const int r = r0 + j;
        const int letter = r < bq ? (int)p.query[eQ - r] : A;
        qoff[j] = letter * A;
        Hp[j] = -Go - r * Ge;   // H[r][-1] (opal.cpp:1266-1269)
        En[j] = Hp[j] - Go;     // E[r][0] = H[r][-1] - Go: the gap opens there
    }
    unsigned long long eOpen = ~0ull;  // bit j: the E of my row j (next column to compute) opens at that cell
    const int jl = bq - 1 - r0;        // register holding the last row (in [0,S) on exactly one lane)
    int outH = -Go - (r0 + S - 1) * Ge;
    int outF = 0, diagIn = 0, tLetter = 0;
    const bool mine = r0 < bq;  // lanes below the box only pass values along (nobody reads them)
    const int limit = bt + 63;
    for (int s = 0; s < limit; s++) {
        int fresh = 0;
        if (s < bt) {
            const unsigned t = tok[eT - s];
            fresh = (int)t;
        }
        const int shifted = tr_shr1(tLetter);
        tLetter = lane == 0 ? fresh : shifted;
        int inH = tr_shr1(outH);
        int inF = tr_shr1(outF);
        const int c = s - lane;
        if (lane == 0) {  // top boundary: H[-1][c] = -Go - c*Ge, F[0][c] = H[-1][c] - Go (opal.cpp:1283-1287)
            inH = -Go - c * Ge;
            inF = inH - Go;
        }
        if (c >= 0 && c < bt && mine) {
            int diag = lane == 0 ? (c == 0 ? 0 : -Go - (c - 1) * Ge) : diagIn;
            int f = inF, h = 0;
            unsigned fOpen = (inF == inH - Go) ? 1u : 0u;  // F of my first row: max(H[up] - Go, F[up] - Ge)
            unsigned word = 0;
            uint32_t* const dst = reinterpret_cast<uint32_t*>(dirs + (size_t)c * ld + r0);
#pragma unroll
            for (int j = 0; j < S; j++) {
                const int P = mlds[qoff[j] + tLetter];
                const int e = En[j];
                const int d = diag + P;
                h = max(d, max(e, f));
                // H prefers E, then F, then the diagonal (opal.cpp:1378-1386); bit 2: E opens here, bit 3: F opens here
                const unsigned src = (h == e) ? 0u : ((h == f) ? 1u : 2u);
                const unsigned flag = src | ((unsigned)((eOpen >> j) & 1ull) << 2) | (fOpen << 3);
                word |= flag << (8 * (j & 3));
                if ((j & 3) == 3 || j == S - 1) {
                    if (r0 + (j & ~3) < ld) dst[j >> 2] = word;
                    word = 0;
                }
                const int hq = h - Go;
                const int en = max(e - Ge, hq);
                eOpen = (eOpen & ~(1ull << j)) | ((unsigned long long)(en == hq) << j);
                En[j] = en;
                const int fn = max(f - Ge, hq);
                fOpen = (fn == hq) ? 1u : 0u;
                f = fn;
                diag = Hp[j];
                Hp[j] = h;
            }
            outH = h;
            outF = f;
            if (c == bt - 1 && jl >= 0 && jl < S) {
                int v = Hp[0];
#pragma unroll
                for (int j = 1; j < S; j++) v = (j == jl) ? Hp[j] : v;
                p.boxScore[w] = v;
            }
        }
        diagIn = inH;
    }
}

// One thread per alignment: the walk of opal.cpp:1369-1409 over the flags, from the far corner of the box back to its
// corner; operations are written in walk order, which is the forward order of the alignment.
__global__ void __launch_bounds__(256) opal_walk_kernel(const TraceParams p) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= p.numItems) return;
    const int seq = p.items[w];
    const int eQ = p.endQ[seq], eT = p.endT[seq];
    const int bq = eQ - p.startQ[seq] + 1, bt = eT - p.startT[seq] + 1;
    const int ld = (bq + 3) & ~3;
    const int item = p.seqItem[seq];
    const unsigned char* tok = p.res + p.seqOff[item];
    const unsigned char* const dirs = p.dirs + p.dirOff[w];
    unsigned char* const out = p.ops + p.opsOff[w];
    int ri = bq - 1, ci = bt - 1, n = 0;
    int state = 0;  // 0: in H, 1: in E (gap in the query = insertion), 2: in F (deletion)
    while (ri >= 0 && ci >= 0) {
        const unsigned f = dirs[(size_t)ci * ld + ri];
        if (state == 0) {
            const unsigned src = f & 3u;
            if (src == 0u) state = 1;
            else if (src == 1u) state = 2;
            else {
                const unsigned t = tok[eT - ci];
                const unsigned tl = t;
                out[n++] = (unsigned)p.query[eQ - ri] == tl ? OPAL_ALIGN_MATCH : OPAL_ALIGN_MISMATCH;
                ri--; ci--;
            }
        } else if (state == 1) {
            state = (f & 4u) ? 0 : 1;
            out[n++] = OPAL_ALIGN_INS;
            ci--;
        } else {
            state = (f & 8u) ? 0 : 2;
            out[n++] = OPAL_ALIGN_DEL;
            ri--;
        }
    }
    while (ri >= 0) { out[n++] = OPAL_ALIGN_DEL; ri--; }  // leading gap: only one of the two loops runs
    while (ci >= 0) { out[n++] = OPAL_ALIGN_INS; ci--; }
    p.opsLen[w] = n;
}

typedef void (*TraceFn)(const TraceParams);

extern "C" int opal_hip_trace_rows_supported(int rows) {
    static const int sup[] = {4, 8, 16, 24, 32, 40};
    for (int r : sup)
        if (64 * r >= rows) return r;
    return 0;
}

extern "C" int opal_hip_trace_launch(const OpalHipTraceLaunch* a) {
    const int S = opal_hip_trace_rows_supported(a->max_rows);
    TraceFn fn = nullptr;
    switch (S) {
    case 4: fn = opal_trace_kernel<4>; break;
    case 8: fn = opal_trace_kernel<8>; break;
    case 16: fn = opal_trace_kernel<16>; break;
    case 24: fn = opal_trace_kernel<24>; break;
    case 32: fn = opal_trace_kernel<32>; break;
    case 40: fn = opal_trace_kernel<40>; break;
    }
    if (!fn) return 2;
    if (a->num_items <= 0) return 0;
    const size_t lds = (size_t)(a->alphabet_length + 1) * a->alphabet_length * sizeof(int);
    if (lds > 64 * 1024) return 2;
    TraceParams p;
    p.query = a->query; p.A = a->alphabet_length; p.Go = a->gap_open; p.Ge = a->gap_ext;
    p.matrix = a->score_matrix; p.res = a->residues; p.seqOff = a->seq_offset; p.seqItem = a->seq_item;
    p.items = a->items; p.numItems = a->num_items;
    p.endQ = a->end_query; p.endT = a->end_target; p.startQ = a->start_query; p.startT = a->start_target;
    p.dirOff = a->dir_offset; p.dirs = a->dirs; p.opsOff = a->ops_offset; p.ops = a->ops; p.opsLen = a->ops_length;
    p.boxScore = a->box_score;
    const int blocks = (a->num_items + TR_WAVES_PER_BLOCK - 1) / TR_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(TR_WAVES_PER_BLOCK * 64), lds, (hipStream_t)a->stream, p);
    if (hipGetLastError() != hipSuccess) return 2;
    hipLaunchKernelGGL(opal_walk_kernel, dim3((a->num_items + 255) / 256), dim3(256), 0, (hipStream_t)a->stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
