// swap_kernel.hpp -- gfx950: local alignment (SW) with the ROLES SWAPPED, for the few longest targets of a search with a
// short query.
//
// Role: the same cells as dp_kernel.hpp's SW kernels (reference src/opal.cpp:164-470 searchDatabaseSW_; recurrence
// :280-327), for the pairs those kernels are worst at. A class kernel puts the QUERY on the rows of a wave (lanes x rows
// per lane) and walks the target column by column: a 35 213-residue target against a 110-aa query is 35 276 dependent
// steps of one lone wave holding 2 rows per lane -- 4.8 ms for a search whose DP work is 3 ms on the whole chip
// (Swiss-Prot-shaped databases, the reference's published queries O74807 / P19930: aligner_comparison.md:27-34).
// Local alignment is symmetric under transposition (gap penalties are the same for both sequences; S(a, b) keeps a = query
// letter, b = target letter), so here
//   * the TARGET is the rows: one wave holds 64 lanes x S consecutive target residues (a "row chunk" of 64*S residues),
//     H[r][c-1] and E[r][c] of its rows in VGPRs as in dp_kernel.hpp, and works through the target's row chunks from
//     the top; a chunk's bottom row (H and the F input of the row below, one word per QUERY column) stays in LDS for
//     the chunk below -- it is Q words;
//   * the QUERY is the columns: Q + 63 steps per row chunk, i.e. ceil(L / (64 S)) * (Q + 63) steps of S rows each instead of
//     L + 63 steps of ceil(Q / 64) rows: 14 x 173 long steps instead of 35 276 short ones, and a step's fixed cost
//     (lane hand-off, address, LDS round trip) is paid 15x less often;
//   * the profile is the TARGET's: TP[query letter a][lane k][row j] = scoreMatrix[a][target[t0 + k S + j]] (2-byte
//     entries, the layout and the aligned ds_read_b128 of Vec<18>), staged per row chunk by the whole workgroup (4 waves);
//     the DP itself runs on wave 0 (the other waves wait at the barrier and cost nothing);
//   * the column's query letter reaches lane k at step c + k as an LDS address read from a per-search table (qoff), one
//     step ahead, instead of through a lane queue.
// One workgroup per target (the profile of a 2560-row chunk fills most of a CU's LDS); the host sends only the handful of
// targets here whose lone-wave time on the class kernels would exceed the search (planner.cpp: swap_prefix).
// Scores, and end cells at 8 / 16 rows per lane; scores at the 16-bit limit go to the 32-bit re-run list like everywhere else.
#pragma once
#include "dp_kernel.hpp"

struct SwapParams {
    const unsigned char* query;
    const int* matrix;
    int Q, A, Go, Ge;
    const unsigned char* res;
    const uint64_t* seqOff;
    const int* pairLen;
    const int* pairSeq;
    int firstItem, numItems;  // items [firstItem, +numItems): one workgroup each
    int* outScore;
    int* outEndQ;   // END kernels: where the score is first reached (opal.h:44-45): smallest TARGET index, then smallest query index
    int* outEndT;
    int* ovfCount;
    int* ovfItems;
    int satLimit, negPad;
    unsigned long long* clk;  // clock probe (dp_kernel.hpp)
    int* started;             // (may be null) every workgroup adds one when it starts: opal_hip_wait_started
};
typedef void (*SwapFn)(const SwapParams);

#define SWAP_WAVES 4          // waves per workgroup: all stage the chunk profile, wave 0 runs the DP
#define SWAP_MAX_QUERY 1024   // longest query the host sends here (LDS tables: 8 bytes per query column)

// LDS: the target profile of one row chunk, the column address table (Q + 128 words: 63 pad columns in front, 65 behind),
// the chunk boundary row (Q + 66 words), the score matrix as 16-bit entries, the probe slot.
// (no odd-unit padding of a lane's segment as the throughput kernels have it: a lone wave does not care about a two-way
// bank conflict, and 16 rows per lane then take 52 KB instead of 80)
template <int S>
__host__ __device__ inline int opal_swap_stride() { return opal_profile_stride(2, 64, S, 0); }
template <int S>
__host__ __device__ inline size_t opal_swap_lds_bytes(int Q, int A) {
    return (size_t)(A + 1) * (size_t)opal_swap_stride<S>() + 4 * (size_t)(Q + 128) + 4 * (size_t)(Q + 66) + (((size_t)A * A * 2 + 15) & ~(size_t)15) + 48;
}

// END: also the end cell. In the swapped frame the target index is the ROW and the query index the column, so the cell
// wanted is the one with the smallest row, then the smallest column, among those that hold the score (opal.h:44-45,
// opal.cpp:385-401 pick the smallest target index first). Every lane keeps the best cell of its own rows: a column whose
// maximum over the lane's rows improves the lane's running maximum replaces it (smallest row of that column), a column
// that TIES with it replaces it only if it holds the value in a smaller row (it comes later, so its column is larger);
// both only inside the query (a pad column can repeat a value, never create one, and never exceeds what the same row
// held in column Q-1: it got there by one step to the right and then down, column Q-1 gets it straight down) and only if the
// value reaches what the whole wave had a few steps ago. The lanes hold ascending rows and the chunks run top to bottom, so the final choice is
// the highest score, then the smallest global row.
template <int S, bool END = false>
__global__ void __launch_bounds__(SWAP_WAVES * 64, 1) opal_swap_sw(const SwapParams p) {
    using V = Vec<18>;
    using reg = typename V::reg;
    constexpr int ES = 2, RB = 8;
    constexpr int SPAD = (S + RB - 1) / RB * RB;
    constexpr int LANE_BYTES = opal_lane_bytes(ES, S, 0);
    constexpr int NBLK = SPAD / RB;
    static_assert(S % 2 == 0, "rows per lane: even (the staging writes row pairs)");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int stride = opal_swap_stride<S>();
    const int Q = p.Q, A = p.A;
    unsigned* const qoff = reinterpret_cast<unsigned*>(lds + (size_t)(A + 1) * stride);
    unsigned* const bnd = qoff + (Q + 128);
    unsigned short* const mat16 = reinterpret_cast<unsigned short*>(bnd + (Q + 66));
    unsigned char* const clkSlot = reinterpret_cast<unsigned char*>(mat16) + (((size_t)A * A * 2 + 15) & ~(size_t)15);
    clk_begin(p.clk, nullptr, clkSlot);
    if (threadIdx.x == 0 && p.started) atomicAdd(p.started, 1);  // this workgroup has its CU

    const int item = p.firstItem + (int)blockIdx.x;
    const int seq = p.pairSeq[item];
    const int L = seq >= 0 ? p.pairLen[item] : 0;
    const uint64_t tb = p.seqOff[item];
    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned padOff = ldsBase + (unsigned)(A * stride);
    const int negPad = p.negPad;

    // ---- once per target: matrix as 16-bit entries, column address table, zero boundary (= the top boundary of local
    // alignment for the first chunk), the pad letter's profile row (pad COLUMNS: before column 0 and behind column Q-1)
    for (int i = threadIdx.x; i < A * A; i += blockDim.x) mat16[i] = (unsigned short)p.matrix[i];
    for (int i = threadIdx.x; i < Q + 128; i += blockDim.x) {
        const int c = i - 63;
        qoff[i] = (c >= 0 && c < Q) ? ldsBase + (unsigned)p.query[c] * (unsigned)stride : padOff;
    }
    for (int i = threadIdx.x; i < Q + 66; i += blockDim.x) bnd[i] = 0u;
    for (int i = threadIdx.x; i < 64 * SPAD / 2; i += blockDim.x) {
        const int k = i / (SPAD / 2), j2 = i - k * (SPAD / 2);
        const unsigned v = (unsigned)negPad & 0xffffu;
        *reinterpret_cast<unsigned*>(lds + (size_t)A * stride + k * LANE_BYTES + j2 * 4) = v | (v << 16);
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = lane;
    const unsigned Qs = (unsigned)p.Go & 0xffffu, Rs = (unsigned)p.Ge & 0xffffu;
    reg mx = 0;
    int bestRow = INT_MAX, bestCol = -1;  // END: my best cell (global target row, query column)
    int waveMax = 0;                      // END: what the whole wave had reached a few steps ago (scan filter)
    const int chunkRows = 64 * S;
    const int nchunks = (L + chunkRows - 1) / chunkRows;
    const unsigned qoffLane = (unsigned)(uintptr_t)(opal_lds_u8*)reinterpret_cast<unsigned char*>(qoff) + 4u * (unsigned)(63 - k);
    const unsigned bndBase = (unsigned)(uintptr_t)(opal_lds_u8*)reinterpret_cast<unsigned char*>(bnd);
    typedef __attribute__((address_space(3))) unsigned lds_u32;

    for (int ch = 0; ch < nchunks; ch++) {
        const int t0 = ch * chunkRows;
        const int rowsHere = min(chunkRows, L - t0);
        __syncthreads();  // the previous chunk's DP is over (and, first time round, the tables above are written)
        // ---- stage TP for rows [t0, t0 + 64 S): a thread owns a pair of rows (one 32-bit store per letter)
        for (int i = threadIdx.x; i < 64 * S / 2; i += blockDim.x) {
            const int kk = i / (S / 2), j2 = i - kk * (S / 2);
            const int t = t0 + kk * S + 2 * j2;
            const int b0 = t < L ? (int)p.res[tb + (uint64_t)t] : -1;
            const int b1 = t + 1 < L ? (int)p.res[tb + (uint64_t)t + 1] : -1;
            unsigned char* dst = lds + kk * LANE_BYTES + j2 * 4;
            const unsigned np = (unsigned)negPad & 0xffffu;
            for (int a = 0; a < A; a++) {
                const unsigned v0 = b0 >= 0 ? (unsigned)mat16[a * A + b0] : np;
                const unsigned v1 = b1 >= 0 ? (unsigned)mat16[a * A + b1] : np;
                *reinterpret_cast<unsigned*>(dst + (size_t)a * stride) = v0 | (v1 << 16);
            }
        }
        __syncthreads();
        if (wave != 0) continue;
        __builtin_amdgcn_s_setprio(3);  // this wave is the critical path of the search: ahead of whatever shares its SIMD

        // ---- the DP of this row chunk on wave 0: Q + usedLanes - 1 steps, lane k on column s - k
        const int usedLanes = (rowsHere + S - 1) / S;
        const int nsteps = Q + usedLanes - 1;
        const bool hasNext = ch + 1 < nchunks;
        reg Hp[S], En[S];
#pragma unroll
        for (int j = 0; j < S; j++) { Hp[j] = 0; En[j] = 0; }
        reg outH = 0, outF = 0, diagIn = 0;
        // addresses and boundary words are read one step ahead; the first row block of a step's profile column too
        unsigned offCur = *(lds_u32*)(uintptr_t)(qoffLane) + (unsigned)(k * LANE_BYTES);
        unsigned offNxt = *(lds_u32*)(uintptr_t)(qoffLane + 4u) + (unsigned)(k * LANE_BYTES);
        unsigned bndCur = *(lds_u32*)(uintptr_t)(bndBase);
        int4 pfirst = lds_ld16(offCur);
        for (int s = 0; s < nsteps; s++) {
            const unsigned offc = offCur;
            const int4 pcur = pfirst;
            const unsigned off2 = *(lds_u32*)(uintptr_t)(qoffLane + 4u * (unsigned)(s + 2)) + (unsigned)(k * LANE_BYTES);
            const unsigned bndNxt = *(lds_u32*)(uintptr_t)(bndBase + 4u * (unsigned)(s + 1));
            pfirst = lds_ld16(offNxt);

            reg inH = (reg)shr1((unsigned)outH);
            reg inF = (reg)shr1((unsigned)outF);
            inH = (k == 0) ? (reg)(bndCur & 0xffffu) : inH;
            inF = (k == 0) ? (reg)(bndCur >> 16) : inF;
            reg f = inF;
            reg cm = 0;  // END: this column's maximum over my rows
            int4 pc = pcur;
            reg d = V::adds(diagIn, (reg)pc.x), t = 0;
            static_for<NBLK>([&](auto rbc) {
                constexpr int rb = decltype(rbc)::value;
                int4 pn = pc;
                if constexpr (rb + 1 < NBLK) pn = lds_ld16(offc + (rb + 1) * RB * ES);
                reg Pv[RB + 1];
                Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.x >> 16; Pv[2] = (reg)pc.y; Pv[3] = (reg)pc.y >> 16;
                Pv[4] = (reg)pc.z; Pv[5] = (reg)pc.z >> 16; Pv[6] = (reg)pc.w; Pv[7] = (reg)pc.w >> 16;
                Pv[RB] = (reg)pn.x;
                static_for<RB / 4>([&](auto subc) {
                    constexpr int j0 = rb * RB + decltype(subc)::value * 4;
                    if constexpr (j0 < S) {
                        constexpr int n = (S - j0) < 4 ? (S - j0) : 4;
                        constexpr bool last = j0 + n == S;
                        const reg* Pn = &Pv[j0 - rb * RB + 1];
                        V::template rows<ROWS_SW, n, last>(&Hp[j0], &En[j0], Pn, f, END ? cm : mx, d, t, Qs, Rs);
                    }
                });
                pc = pn;
            });
            outH = Hp[S - 1];
            outF = f;
            if constexpr (END) {
                if ((s & 7) == 0) waveMax = wave_max_i32(V::half(mx, 0));
                const int cv = V::half(cm, 0), mv = V::half(mx, 0);
                const int c = s - k;
                if (cv >= mv && cv > 0 && cv >= waveMax && (unsigned)c < (unsigned)Q) {
                    int r = 0;
#pragma unroll
                    for (int j = S - 1; j >= 0; j--)
                        if (V::same(Hp[j], cm, 0)) r = j;  // smallest row of the column holding it
                    const int row = t0 + k * S + r;
                    if (cv > mv || row < bestRow) { bestRow = row; bestCol = c; }
                }
                mx = V::maxs(mx, cm);
            }
            // the bottom row of a full chunk: lane 63 finishes column s - 63
            if (hasNext && k == 63 && s >= 63) {
                const unsigned w = ((unsigned)outH & 0xffffu) | ((unsigned)outF << 16);
                *(lds_u32*)(uintptr_t)(bndBase + 4u * (unsigned)(s - 63)) = w;
            }
            diagIn = inH;
            offCur = offNxt; offNxt = off2; bndCur = bndNxt;
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- result: the maximum over the lanes of wave 0 (every chunk accumulated into mx)
    if (wave == 0) {
        int sc = V::half(mx, 0), row = bestRow, col = bestCol;
        if constexpr (END) {  // highest score, then smallest target index (= row)
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) {
                const int osc = __shfl_xor(sc, dd, 64), orow = __shfl_xor(row, dd, 64), ocol = __shfl_xor(col, dd, 64);
                if (osc > sc || (osc == sc && orow < row)) { sc = osc; row = orow; col = ocol; }
            }
        } else {
            sc = wave_max_i32(sc);
        }
        if (lane == 0 && seq >= 0) {
            if (sc >= p.satLimit) {  // (opal.cpp:364-370): hand over to width 32
                const int at = atomicAdd(p.ovfCount, 1);
                p.ovfItems[at] = item;
            } else {
                p.outScore[seq] = sc;
                if constexpr (END) {
                    const bool none = sc <= 0 || col < 0;  // score 0: no cell (test.cpp:242-243)
                    p.outEndQ[seq] = none ? -1 : col;
                    p.outEndT[seq] = none ? -1 : row;
                }
            }
        }
    }
    clk_end(clkSlot);
}

SwapFn opal_swap_kernel(int rows, int want_end);  // inst_swap.hip: rows per lane 8 / 16 / 24 / 40, with end cells 8 / 16 (nullptr: not instantiated)
size_t opal_swap_lds(int rows, int Q, int A);  // LDS bytes of that shape

// ------------------------------------------------------------------ NW / HW / OV with the roles swapped (int32)
// The same idea for the global modes (reference src/opal.cpp:594-977 searchDatabase_; boundary conditions :671-689, core
// :748-771, results :844-904): the longest targets of a search -- the ones whose a-priori bounds push them onto the 32-bit
// kernels (a 35 213-residue target: 35 276 dependent steps of int32 rows, 8 ms against a 110-aa query and 17 ms against a
// 513-aa one, on a search whose bulk takes 3 / 13 ms) and the ones that would sit on the widest 16-bit class as lone waves --
// run with the TARGET on the rows of a wave (64 lanes x S residues per row chunk) and the query on the columns. 32-bit
// arithmetic throughout: no range analysis decides anything here, and a lone wave is bound by latency, not by the half-rate
// v_max_i32. In the swapped frame H'[t][q] = H[q][t]:
//   column -1 (before the query):  H'[t][-1] = -Go - t*Ge (NW), 0 (HW, OV: leading target residues are free)
//   row -1 (before the target):    H'[-1][q] = 0 (OV), -Go - q*Ge (NW, HW)
//   NW  score H'[L-1][Q-1], end (Q-1, L-1)
//   HW  score max_t H'[t][Q-1] (the LAST COLUMN), endQ = Q-1, endT = the smallest t holding it (opal.cpp:824-834: first column)
//   OV  A = max_q H'[L-1][q] (the last ROW; the smallest q holding it), B = max_{t <= L-2} H'[t][Q-1] (smallest t);
//       A > B: score A, end (q, L-1); otherwise score B, end (Q-1, t)   (opal.cpp:881-904)
// Lanes hold ascending rows and the chunks run top to bottom, so "smallest t" is the first lane / first chunk that holds the
// value, and "smallest q" the first step. The chunk's bottom row (H and the F input of the row below, two words per query
// column) waits in LDS for the chunk below, as in opal_swap_sw.
// ES: bytes per profile entry. 4: any matrix that passed the 32-bit argument check. 2: matrices whose entries fit 16 bits
// (sign-extended on the way into the registers: one more instruction per row) -- twice the rows per lane in the same LDS,
// half the steps per target; a lone wave's step is bound by its latency chain, not by its rows.
template <int S, int ES>
__host__ __device__ inline int opal_swapgl_stride() { return opal_profile_stride(ES, 64, S, 0); }
template <int S, int ES>
__host__ __device__ inline size_t opal_swapgl_lds_bytes(int Q, int A) {
    return (size_t)(A + 1) * (size_t)opal_swapgl_stride<S, ES>() + 4 * (size_t)(Q + 128) + 8 * (size_t)(Q + 66) + 16 + 48;  // (+16: the clock probe's slot is 16-byte aligned)
}

template <int MODE, int S, int ES>
__global__ void __launch_bounds__(SWAP_WAVES * 64, 1) opal_swap_gl(const SwapParams p) {
    constexpr int RB = 16 / ES;  // rows per LDS read
    constexpr int SPAD = (S + RB - 1) / RB * RB;
    constexpr int LANE_BYTES = opal_lane_bytes(ES, S, 0);
    constexpr int NBLK = SPAD / RB;
    static_assert(ES == 4 || (ES == 2 && S % 2 == 0), "2-byte entries: the staging writes row pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int stride = opal_swapgl_stride<S, ES>();
    const int Q = p.Q, A = p.A, Go = p.Go, Ge = p.Ge;
    unsigned* const qoff = reinterpret_cast<unsigned*>(lds + (size_t)(A + 1) * stride);
    int* const bndH = reinterpret_cast<int*>(qoff + (Q + 128));
    int* const bndF = bndH + (Q + 66);
    // (the probe's slot holds 64-bit words: bndF ends on a 4-byte boundary when Q is odd)
    unsigned char* const clkSlot = lds + ((((size_t)(A + 1) * stride + 4 * (size_t)(Q + 128) + 8 * (size_t)(Q + 66)) + 15) & ~(size_t)15);
    clk_begin(p.clk, nullptr, clkSlot);
    if (threadIdx.x == 0 && p.started) atomicAdd(p.started, 1);  // this workgroup has its CU

    const int item = p.firstItem + (int)blockIdx.x;
    const int seq = p.pairSeq[item];
    const int L = seq >= 0 ? p.pairLen[item] : 0;
    const uint64_t tb = p.seqOff[item];
    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned padOff = ldsBase + (unsigned)(A * stride);
    constexpr int NEGINF = -(1 << 30);

    // ---- once per target: column address table, the top boundary row (t = -1) of the first chunk, the pad letter's row
    for (int i = threadIdx.x; i < Q + 128; i += blockDim.x) {
        const int c = i - 63;
        qoff[i] = (c >= 0 && c < Q) ? ldsBase + (unsigned)p.query[c] * (unsigned)stride : padOff;
    }
    for (int i = threadIdx.x; i < Q + 66; i += blockDim.x) {
        // H'[-1][q] and the F input of row 0: F'[0][q] = H'[-1][q] - Go (F'[-1][q] = -inf)
        const long long h = (MODE == MODE_OV) ? 0LL : -(long long)Go - (long long)i * Ge;
        bndH[i] = (int)(h < NEGINF ? NEGINF : h);
        bndF[i] = (int)(h - Go < NEGINF ? NEGINF : h - Go);
    }
    for (int i = threadIdx.x; i < 64 * (SPAD * ES / 4); i += blockDim.x) {
        const int k = i / (SPAD * ES / 4), w = i - k * (SPAD * ES / 4);
        *reinterpret_cast<int*>(lds + (size_t)A * stride + k * LANE_BYTES + w * 4) = 0;  // (pad columns: nothing reads their results)
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = lane;
    const int chunkRows = 64 * S;
    const int nchunks = (L + chunkRows - 1) / chunkRows;
    const unsigned qoffLane = (unsigned)(uintptr_t)(opal_lds_u8*)reinterpret_cast<unsigned char*>(qoff) + 4u * (unsigned)(63 - k);
    const unsigned bndHBase = (unsigned)(uintptr_t)(opal_lds_u8*)reinterpret_cast<unsigned char*>(bndH);
    const unsigned bndFBase = (unsigned)(uintptr_t)(opal_lds_u8*)reinterpret_cast<unsigned char*>(bndF);
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) int lds_i32;

    // results (per lane; reduced over the wave at the end)
    int colBest = INT_MIN, colRow = INT_MAX;  // HW / OV: maximum of the last column over my rows (OV: rows <= L-2), smallest row holding it
    int rowBest = INT_MIN, rowCol = -1;       // OV: maximum of the last row (t = L-1) over the columns, smallest column; NW: the final cell

    for (int ch = 0; ch < nchunks; ch++) {
        const int t0 = ch * chunkRows;
        const int rowsHere = min(chunkRows, L - t0);
        __syncthreads();
        // ---- stage TP for rows [t0, t0 + 64 S): TP[a][k][j] = M[a][target[t]] (rows beyond the target: 0, they feed nothing real)
        if constexpr (ES == 4) {
            for (int i = threadIdx.x; i < 64 * S; i += blockDim.x) {
                const int kk = i / S, j = i - kk * S;
                const int t = t0 + i;
                const int b = t < L ? (int)p.res[tb + (uint64_t)t] : -1;
                unsigned char* dst = lds + kk * LANE_BYTES + j * 4;
                for (int a = 0; a < A; a++) *reinterpret_cast<int*>(dst + (size_t)a * stride) = b >= 0 ? p.matrix[a * A + b] : 0;
            }
        } else {  // a thread owns a pair of rows: one 32-bit store per letter
            for (int i = threadIdx.x; i < 64 * S / 2; i += blockDim.x) {
                const int kk = i / (S / 2), j2 = i - kk * (S / 2);
                const int t = t0 + kk * S + 2 * j2;
                const int b0 = t < L ? (int)p.res[tb + (uint64_t)t] : -1;
                const int b1 = t + 1 < L ? (int)p.res[tb + (uint64_t)t + 1] : -1;
                unsigned char* dst = lds + kk * LANE_BYTES + j2 * 4;
                for (int a = 0; a < A; a++) {
                    const unsigned v0 = b0 >= 0 ? (unsigned)p.matrix[a * A + b0] & 0xffffu : 0u;
                    const unsigned v1 = b1 >= 0 ? (unsigned)p.matrix[a * A + b1] & 0xffffu : 0u;
                    *reinterpret_cast<unsigned*>(dst + (size_t)a * stride) = v0 | (v1 << 16);
                }
            }
        }
        __syncthreads();
        if (wave != 0) continue;
        __builtin_amdgcn_s_setprio(3);

        const int usedLanes = (rowsHere + S - 1) / S;
        const int nsteps = Q + usedLanes - 1;
        const bool hasNext = ch + 1 < nchunks;
        const int r0 = t0 + k * S;  // my first row
        // column -1: H'[t][-1] and E'[t][0] = H'[t][-1] - Go
        int Hp[S], En[S];
#pragma unroll
        for (int j = 0; j < S; j++) {
            const long long hb = (MODE == MODE_NW) ? -(long long)Go - (long long)(r0 + j) * Ge : 0LL;
            Hp[j] = (int)(hb < NEGINF ? NEGINF : hb);
            En[j] = (int)(hb - Go < NEGINF ? NEGINF : hb - Go);
        }
        // what the lane below reads before my first column: H' of my last row at column -1
        int outH = Hp[S - 1], outF = 0;
        // H'[r0 - 1][-1]: the diagonal input of my first column (lane 0 of the first chunk: H'[-1][-1] = 0)
        int diagIn = 0;
        if (MODE == MODE_NW && r0 > 0) { const long long hb = -(long long)Go - (long long)(r0 - 1) * Ge; diagIn = (int)(hb < NEGINF ? NEGINF : hb); }
        // the last row of the target, if it lies in this chunk: lane kL, register jL
        const bool lastChunk = !hasNext;
        const int kL = lastChunk ? (L - 1 - t0) / S : -1;
        const int jL = lastChunk ? (L - 1 - t0) - kL * S : -1;
        // my registers [0, jhi) hold rows of the target that count for the last column (OV: rows <= L-2)
        const int lastCount = (MODE == MODE_OV) ? L - 1 : L;
        int jhi = lastCount - r0;
        jhi = jhi < 0 ? 0 : (jhi > S ? S : jhi);

        unsigned offCur = *(lds_u32*)(uintptr_t)(qoffLane) + (unsigned)(k * LANE_BYTES);
        unsigned offNxt = *(lds_u32*)(uintptr_t)(qoffLane + 4u) + (unsigned)(k * LANE_BYTES);
        int bhCur = *(lds_i32*)(uintptr_t)(bndHBase), bfCur = *(lds_i32*)(uintptr_t)(bndFBase);
        int4 pfirst = lds_ld16(offCur);
        for (int s = 0; s < nsteps; s++) {
            const unsigned offc = offCur;
            const int4 pcur = pfirst;
            const unsigned off2 = *(lds_u32*)(uintptr_t)(qoffLane + 4u * (unsigned)(s + 2)) + (unsigned)(k * LANE_BYTES);
            const int bhNxt = *(lds_i32*)(uintptr_t)(bndHBase + 4u * (unsigned)(s + 1));
            const int bfNxt = *(lds_i32*)(uintptr_t)(bndFBase + 4u * (unsigned)(s + 1));
            pfirst = lds_ld16(offNxt);

            int inH = (int)shr1((unsigned)outH);
            int inF = (int)shr1((unsigned)outF);
            inH = (k == 0) ? bhCur : inH;
            inF = (k == 0) ? bfCur : inF;
            const int c = s - k;  // my column
            if (c >= 0) {  // (before my first column my boundary values must stay as they are)
                int f = inF;
                int diag = diagIn;
                int4 pc = pcur;
                static_for<NBLK>([&](auto rbc) {
                    constexpr int rb = decltype(rbc)::value;
                    int4 pn = pc;
                    if constexpr (rb + 1 < NBLK) pn = lds_ld16(offc + (rb + 1) * RB * ES);
                    int P[RB];
                    if constexpr (ES == 4) { P[0] = pc.x; P[1] = pc.y; P[2] = pc.z; P[3] = pc.w; }
                    else {
                        P[0] = (pc.x << 16) >> 16; P[1] = pc.x >> 16; P[2] = (pc.y << 16) >> 16; P[3] = pc.y >> 16;
                        P[4] = (pc.z << 16) >> 16; P[5] = pc.z >> 16; P[6] = (pc.w << 16) >> 16; P[7] = pc.w >> 16;
                    }
#pragma unroll
                    for (int i = 0; i < RB; i++) {
                        constexpr int jb = rb * RB;
                        const int j = jb + i;
                        if (j < S) {
                            // (reference core loop opal.cpp:748-771; E along the columns, F down the rows)
                            const int e = En[j];
                            const int d = diag + P[i];
                            int h = d > e ? d : e;
                            h = h > f ? h : f;
                            const int hq = h - Go;
                            const int e2 = e - Ge, f2 = f - Ge;
                            En[j] = e2 > hq ? e2 : hq;
                            f = f2 > hq ? f2 : hq;
                            diag = Hp[j];
                            Hp[j] = h;
                        }
                    }
                    pc = pn;
                });
                outH = Hp[S - 1];
                outF = f;
                if (hasNext && k == 63 && c < Q) {
                    *(lds_i32*)(uintptr_t)(bndHBase + 4u * (unsigned)c) = outH;
                    *(lds_i32*)(uintptr_t)(bndFBase + 4u * (unsigned)c) = outF;
                }
                // ---- results
                if (MODE != MODE_NW && c == Q - 1) {  // my rows of the LAST COLUMN: the smallest row holding their maximum
                    int bv = INT_MIN, bj = -1;
#pragma unroll
                    for (int j = S - 1; j >= 0; j--)
                        if (j < jhi && Hp[j] >= bv) { bv = Hp[j]; bj = j; }
                    if (bj >= 0 && bv > colBest) { colBest = bv; colRow = r0 + bj; }  // (strict: earlier chunks hold smaller rows)
                }
                if (MODE != MODE_HW && k == kL && c < Q) {  // the LAST ROW of the target
                    const int v = SelectRow<int, S, 0, S>::get(Hp, jL);
                    if (MODE == MODE_NW) { if (c == Q - 1) { rowBest = v; rowCol = c; } }
                    else if (v > rowBest) { rowBest = v; rowCol = c; }  // strict: the first column wins
                }
            }
            diagIn = inH;
            offCur = offNxt; offNxt = off2; bhCur = bhNxt; bfCur = bfNxt;
        }
        __builtin_amdgcn_s_setprio(0);
    }

    if (wave == 0) {
        // last column: the highest value, then the smallest row, over the lanes; last row: one lane holds it
        int cb = colBest, cr = colRow, rbv = rowBest, rc = rowCol;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const int ocb = __shfl_xor(cb, dd, 64), ocr = __shfl_xor(cr, dd, 64);
            if (ocb > cb || (ocb == cb && ocr < cr)) { cb = ocb; cr = ocr; }
            const int orb = __shfl_xor(rbv, dd, 64), orc = __shfl_xor(rc, dd, 64);
            if (orc >= 0 && (rc < 0 || orb > rbv)) { rbv = orb; rc = orc; }
        }
        if (lane == 0 && seq >= 0 && L > 0) {
            int sc, eq, et;
            if (MODE == MODE_NW) { sc = rbv; eq = Q - 1; et = L - 1; }
            else if (MODE == MODE_HW) { sc = cb; eq = Q - 1; et = cr; }
            else {  // OV (opal.cpp:881-904): the last target residue (our last row) wins only if strictly better
                if (rc >= 0 && (cr == INT_MAX || rbv > cb)) { sc = rbv; eq = rc; et = L - 1; }
                else { sc = cb; eq = Q - 1; et = cr; }
            }
            p.outScore[seq] = sc;
            if (p.outEndQ) { p.outEndQ[seq] = eq; p.outEndT[seq] = et; }
        }
    }
    clk_end(clkSlot);
}

// inst_swap.hip: NW / HW / OV; rows per lane 8 with 4-byte profile entries, 16 with 2-byte ones (nullptr: not instantiated)
SwapFn opal_swap_gl_kernel(int mode, int rows);
size_t opal_swap_gl_lds(int rows, int Q, int A);
