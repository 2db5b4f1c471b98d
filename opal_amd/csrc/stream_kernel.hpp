// stream_kernel.hpp -- gfx950: the database search as a STREAM of sequences through persistent waves.
//
// Role: the same as dp_kernel.hpp's kernels (reference src/opal.cpp:164-470 searchDatabaseSW_, :594-977 searchDatabase_,
// with loadNextSequence :472-490 -- the reference, too, hands a lane its next sequence as soon as the previous one ends),
// for searches in which every group of lanes sees many sequences: long queries (32 or 64 lanes per sequence) in all four
// modes, SW score searches of large databases at any query length. The class kernels of dp_kernel.hpp lose
// (G-1)/(L+G-1) of their steps to filling and draining the lane pipeline for every sequence (18 % at G = 64 on a
// 350-residue database) and restage a ~100 KB query profile for every 16 sequences with nothing else resident on
// the CU. Here
//   * workgroups are PERSISTENT: one per CU, 16 waves, the profile is staged once;
//   * every slot (G lanes) works through a STREAM of sequences handed out by a device-wide counter (longest
//     first, so the slots finish together), the next sequence entering lane 0 while the previous one is still
//     draining through the upper lanes: no fill/drain between sequences, no tail of half-empty workgroups
//     (SlotStream / stream_fetch: the descriptor ring of a slot and its refill);
//   * SW (opal_sw_stream): sequences are separated by two separator columns whose "letters" carry gap penalties of
//     0xFFFF: the unsigned saturating subtractions of the SW row block (E = max(E -sat Ge, H -sat Go)) then zero E and F
//     in the first column and H in the second (profile value -32768), which is exactly the state a new sequence starts
//     from -- no register is reset by hand and no step is special-cased. The gap operands therefore live in VGPRs, loaded
//     per step from a 16-byte info block that sits behind each letter's profile row in LDS. A slot's score is the
//     maximum over its lanes' running maxima, collected at the second separator column as it passes down the lanes
//     (one DPP move per separator step) and written by the last lane; with end locations the best cell travels the
//     same way;
//   * NW/HW/OV (opal_gl_stream, further down): diagonal-skewed arithmetic with a per-sequence local step, the boundary
//     column reloaded from LDS at every sequence start.
// Plain 16-bit arithmetic (Vec<17>: 4-byte profile entries, Vec<18>: 2-byte entries for queries beyond ~1400
// residues); SW scores at the saturation limit go to the 32-bit re-run list like everywhere else.
#pragma once
#include "dp_kernel.hpp"

struct StreamParams {
    KParams base;  // work == nullptr; firstItem/numItems: the item range (pair*2+half); log2G: lanes per sequence
    int* next;     // device counter (zeroed before the launch): next item of the range to hand out
    // The first `sideBlocks` workgroups of the grid run the class body on the item range `side`, STREAM_SIDE_WAVES waves each;
    // the streaming workgroups follow. NW/HW/OV: the unskewed body, same geometry (sequences too long for the skewed
    // arithmetic); SW, width 17: 4G lanes x S/4 rows (the longest sequences of a long-tailed database)
    KClass side;
    int sideBlocks;
    // The first `prioItems` sequences handed out (the longest ones of the range) raise the issue priority of the wave that works
    // on them: every wave of a persistent launch is equally old, so nothing else would let the wave holding a 20 000-residue
    // sequence run ahead of its three SIMD neighbours -- and the search ends when that wave does.
    int prioItems;
    // NW / HW / OV: waves of a side workgroup that take work (0 = STREAM_SIDE_WAVES, one per SIMD: a handful of sequences that
    // are the critical path; 16 = all of them when the side range holds hundreds of sequences -- then it is throughput that counts)
    int sideWaves;
};
#define STREAM_SW_SIDE_MAX_ROWS 39  // SW streams instantiated with the class body for side items: rows 1..39 (40: one register short)
#define STREAM_SIDE_WAVES 4  // one wave per SIMD: these few sequences are the critical path, they run at the lone-wave rate
typedef void (*StreamFn)(const StreamParams);

#define STREAM_WAVES 16          // waves per (persistent) workgroup: 4 per SIMD at <= 128 registers
#define STREAM_RING 4            // descriptors kept per slot (the newest two can be referenced by a refill, the
                                 // last lane may still be flushing the one before)
#define STREAM_DESC_BYTES 32

// one letter's profile row: G lane segments, then the 16-byte info block {Go, Ge, flush flag, 0} sixteen times over -- lane l
// reads copy l & 15, so that the sixteen lanes the LDS serves together, each on the block of ITS letter, fall on sixteen
// different bank slots (the letter stride is a multiple of 256 bytes from 8 lanes per sequence on: dp_kernel.hpp, LDS geometry)
#define STREAM_INFO_BYTES 256
template <int WIDTH>
__host__ __device__ inline int opal_stream_stride(int G, int S) {
    constexpr int ES = WIDTH == 18 ? 2 : 4;
    return opal_letter_stride(G * opal_lane_bytes(ES, S, 1) + STREAM_INFO_BYTES, G);
}
template <int WIDTH>
__host__ __device__ inline size_t opal_stream_lds_bytes(int G, int S, int A) {
    return (size_t)(A + 3) * (size_t)opal_stream_stride<WIDTH>(G, S) + (size_t)STREAM_WAVES * (64 / G) * STREAM_RING * STREAM_DESC_BYTES + 48  // (+48: probe slot)
           + (size_t)STREAM_WAVES * (64 / G) * 2 * opal_swend_ckpt_words(S) * 4;  // end locations: the kept half column of a slot's two sequences in flight
}


// ---- the stream of one slot (slot-uniform values, kept by every lane of the slot) and its refill
struct SlotStream {
    int nEntries = 0;     // descriptors appended so far
    int curStart = 0;     // stream position where the newest entry starts
    int coveredEnd = 0;   // stream position where it ends (INT_MAX once the item range is exhausted)
    int lastEnd = 0;      // end of the last real entry
    bool done = false;
    bool hot = false;      // the newest entry is one of the launch's longest sequences (StreamParams::prioItems)
    unsigned tokNext = 0;  // letter of the position fetched last
    unsigned bndNext = 0;  // row-chunked queries: the chunk boundary {H, F input} above that position (0: none)
};

// Token of stream position x for lane k of a slot (x - k is a multiple of G; called by all lanes of the wave together):
// appends the next item of the range to the slot's descriptor ring when the positions [x-k, x-k+G) are not covered yet
// (one atomic on the device-wide counter per appended sequence), looks the position up in the newest two descriptors and
// issues the token load, which is consumed a batch later. SEPS separator columns (letters PAD+1 .. PAD+SEPS) follow every
// sequence; an entry spans at least 2G positions (pad columns after short sequences), which is what bounds the look-up
// to two descriptors and the ring to four.
// Descriptor: {start, len, caller index, -, residue offset lo, hi, item, running maximum (SW with end locations)}.
// The FIRST sequence of every slot is not drawn from the counter but dealt (round 6): slot (workgroup b, wave w, slot sl) starts on
// item (w * workgroups + b) * slots + sl, so the launch's longest sequences -- the first items -- sit on wave 0 of every persistent
// workgroup (one wave per CU; the slots of a wave take neighbours in length, which end together), the next ones on another SIMD
// of every CU, and so on; the counter then hands out the items from the
// number of slots on. Drawn from the counter, the sixteen longest sequences of a tailed database went to whichever waves won the
// race, now and then four of them to one SIMD: a 1700-aa query on the first 100 000 sequences of the published shape took 17.4 to
// 28.7 ms from search to search on one and the same plan (the waves on a launch's longest sequences run at raised priority --
// among themselves that is no priority).
template <int SEPS, bool BND = false>
__device__ __forceinline__ void stream_fetch(SlotStream& st, const KParams& p, int* next, int* ring, const int x, const int k,
                                             const int G, const int slotLane0, const int PAD, const int prioItems = 0,
                                             const int firstIdx = -1, const int totalSlots = 0) {
    const int base = x - k;
    const bool need = !st.done && st.coveredEnd <= base + G - 1;  // slot-uniform
    if (__ballot(need) != 0ull) {  // (wave-uniform branch, every lane inside: the broadcast below reads lane 0 of the slot)
        int idx = 0;
        if (need && k == 0) idx = (firstIdx >= 0 && st.nEntries == 0) ? firstIdx : atomicAdd(next, 1) + totalSlots;
        idx = __shfl(idx, slotLane0, 64);
        if (need) {
            if (idx < p.numItems) {
                const int item = p.firstItem + idx;
                int len = p.pairLen[item];
                const int seq = p.pairSeq[item];
                if (seq < 0) len = 0;
                const uint64_t tb = p.seqOff[item];
                const int span = max(len + SEPS, 2 * G);
                if (k == 0) {
                    int* e = ring + (st.nEntries & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4);
                    e[0] = st.coveredEnd; e[1] = len; e[2] = seq; e[3] = 0;
                    e[4] = (int)(unsigned)(tb & 0xffffffffu); e[5] = (int)(unsigned)(tb >> 32); e[6] = item; e[7] = 0;
                }
                st.curStart = st.coveredEnd;
                st.coveredEnd += span;
                st.lastEnd = st.coveredEnd;
                st.nEntries++;
                st.hot = idx < prioItems;
            } else {
                st.done = true;
                st.coveredEnd = INT_MAX;
                st.hot = false;
            }
        }
        if (prioItems > 0) {  // (wave-uniform: some slot of this wave is on one of the longest sequences, or none is any more)
            if (__ballot(st.hot) != 0ull) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ring writes by lane 0 of a slot are read by its other lanes below (same wave: LDS keeps program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    unsigned tok = (unsigned)PAD, bnd = 0;
    if (st.nEntries > 0 && x < st.lastEnd) {
        const int e = (x >= st.curStart) ? (st.nEntries - 1) : (st.nEntries - 2);
        const int* d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4);
        const int4 d0 = *reinterpret_cast<const int4*>(d);
        const int col = x - d0.x;
        if (col < d0.y) {
            const uint64_t tb = (uint64_t)(unsigned)d[4] | ((uint64_t)(unsigned)d[5] << 32);
            tok = p.res[tb + (uint64_t)col];
            if constexpr (BND) { if (p.chunkFlags & 1) bnd = reinterpret_cast<const unsigned*>(p.bnd)[tb + (uint64_t)col]; }
        } else if (SEPS >= 1 && col == d0.y) tok = (unsigned)(PAD + 1);
        else if (SEPS >= 2 && col == d0.y + 1) tok = (unsigned)(PAD + 2);
    }
    st.tokNext = tok;
    st.bndNext = bnd;
}

// CHUNK: the launch covers query rows [q0, q0 + chunkRows) of a query beyond one wave (more than 64 lanes
// x 40 rows, or a profile beyond LDS), one launch per row chunk from the top. The chunk's bottom row -- H and the F input
// of the row below, 16 bits each in one word per column, at the column's residue offset -- goes through p.bnd: lane 0
// takes its top inputs from the word the chunk above left (fetched with the letter, a batch ahead, and handed to lane 0
// through the same lane queue), the last lane's outputs are collected in a second lane queue and stored once per batch,
// every lane one column. Separator and pad columns carry zeros, as the top boundary of local alignment does everywhere.
// A sequence's score is the maximum over the chunks (read-modify-write of outScore: one chunk at a time); with end
// locations the best cell so far stays unless this chunk's is better (higher score, or the same in a smaller column).
template <int WIDTH, int S, bool SWEND = false, bool CHUNK = false, bool SIDE = false>
__global__ void __launch_bounds__(STREAM_WAVES * 64) opal_sw_stream(const StreamParams sp) {
    static_assert(!SIDE || (WIDTH == 17 && !CHUNK), "side items: single-launch SW streams of width 17");
    using V = Vec<WIDTH>;
    using reg = typename V::reg;
    using elem = typename V::elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int RB = 16 / ES;
    constexpr int SPAD = (S + RB - 1) / RB * RB;
    constexpr int LANE_BYTES = opal_lane_bytes(ES, S, 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const KParams& p = sp.base;
    count_started(p.started);
    if constexpr (SIDE) {  // (instantiations of their own, inst_stream17head.hip: inside the plain kernels the extra body cost C4 1.1 %)
        if ((int)blockIdx.x < sp.sideBlocks) {
            // The head of a long-tailed database (Swiss-Prot: 35 000 residues among 350): sequences that would pin their G lanes
            // for longer than the whole search takes run on the class body instead, on 4G lanes x S/4 rows, as the FIRST
            // workgroups of this very grid -- inside one launch the dispatch order is ours; as a launch of their own on a
            // side stream they started before or after the persistent workgroups from run to run (13.9 / 15.4 / 18.9 ms).
            opal_dp_body<17, MODE_SW, SWEND ? (FLAG_SWEND | FLAG_NOCKPT) : 0, (S + 3) / 4>(p, sp.side, (int)blockIdx.x, STREAM_SIDE_WAVES);
            return;
        }
    }

    const int log2G = p.log2G;
    const int G = 1 << log2G;
    const int stride = opal_stream_stride<WIDTH>(G, S);
    const int A = p.A;
    const int q0 = CHUNK ? p.q0 : 0;              // first query row of this launch
    const int Q = CHUNK ? p.chunkRows : p.Q;      // ... and how many it covers
    const bool hasPrev = CHUNK && (p.chunkFlags & 1), hasNext = CHUNK && (p.chunkFlags & 2);
    const int padTop = G * S - Q;  // virtual rows above the query (profile 0: H, E, F stay 0 there; first chunk only)
    const int PAD = A, SEP1 = A + 1, SEP2 = A + 2;
    unsigned char* const clkSlot = lds + (size_t)(A + 3) * stride + (size_t)STREAM_WAVES * (64 >> log2G) * STREAM_RING * STREAM_DESC_BYTES;
    clk_begin(p.clk, p.tl, clkSlot);

    // ---- stage the profile: QP[letter][k][j]; letters A (beyond a slot's last sequence), A+1, A+2 (separators)
    {
        const int perLetter = G * SPAD;
        for (int rem = threadIdx.x; rem < perLetter; rem += blockDim.x) {
            const int k = rem / SPAD;
            const int j = rem - k * SPAD;
            const int lr = k * S + j - padTop;
            const bool real = lr >= 0 && j < S;
            unsigned char* dst = lds + k * LANE_BYTES + j * ES;
            if (real) {
                const int* mrow = p.matrix + (int)p.query[q0 + lr] * A;
                for (int letter = 0; letter < A; letter += 2) {
                    const int l1 = min(letter + 1, A - 1);
                    const int v0 = mrow[letter], v1 = mrow[l1];
                    *reinterpret_cast<elem*>(dst + letter * stride) = (elem)v0;
                    *reinterpret_cast<elem*>(dst + l1 * stride) = (elem)v1;
                }
            } else {
                for (int letter = 0; letter < A; letter++) *reinterpret_cast<elem*>(dst + letter * stride) = (elem)(lr < 0 ? 0 : p.negPad);
            }
            // H is in [0, 32767] at all times, so H - 32768 < 0: a separator column cannot extend anything
            const int sepv = lr < 0 ? 0 : -32768;
            *reinterpret_cast<elem*>(dst + PAD * stride) = (elem)(lr < 0 ? 0 : p.negPad);
            *reinterpret_cast<elem*>(dst + SEP1 * stride) = (elem)sepv;
            *reinterpret_cast<elem*>(dst + SEP2 * stride) = (elem)sepv;
        }
        for (int letter = threadIdx.x; letter < A + 3; letter += blockDim.x) {
            int4 info;
            const bool sep = letter > A;
            info.x = sep ? 0xffff : (p.Go & 0xffff);
            info.y = sep ? 0xffff : (p.Ge & 0xffff);
            info.z = letter == SEP2 ? 1 : 0;
            info.w = 0;
            for (int c = 0; c < STREAM_INFO_BYTES / 16; c++) *reinterpret_cast<int4*>(lds + letter * stride + G * LANE_BYTES + 16 * c) = info;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = lane & (G - 1);
    const int slot = lane >> log2G;
    const int slots = 64 >> log2G;
    // this slot's descriptor ring: {start, len, caller index, -, residue offset lo, hi, item, running maximum (SWEND)}
    int* const ring = reinterpret_cast<int*>(lds + (size_t)(A + 3) * stride + (size_t)((wave * slots + slot) * STREAM_RING) * STREAM_DESC_BYTES);
    const int slotLane0 = lane & ~(G - 1);

    reg Hp[S], En[S];
#pragma unroll
    for (int j = 0; j < S; j++) { Hp[j] = 0; En[j] = 0; }
    reg outH = 0, outF = 0, diagIn = 0, mx = 0;
    int outM = 0;
    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned padOff = ldsBase + (unsigned)(PAD * stride);
    const unsigned infoOff = (unsigned)((G - k) * LANE_BYTES + 16 * (lane & 15));  // from my profile address to my copy of my letter's info block
    unsigned offv = padOff + (unsigned)(k * LANE_BYTES);
    unsigned tokq = padOff;

    SlotStream st;  // the stream of this slot
    int flushCount = 0;        // entries this lane has seen the end of (the last lane: whose score has been written)
    // SWEND: the best cell of my sequence so far over ALL lanes of the slot, exactly (dp_kernel.hpp, CKPT): its score and the
    // step at which I am (was, would be) on its column -- and the same for the sequence behind mine in the stream, which the
    // lanes above me have entered already (a slot has at most two sequences in flight: an entry spans at least 2G positions).
    // The column that holds the best cell waits in LDS, one buffer per sequence in flight (by the parity of its number).
    int curV = 0, curS = INT_MIN, nxtV = 0, nxtS = INT_MIN;  // (INT_MIN: a tie at score 0 replaces nothing)
    // The sequences a slot has in flight (two; three when a sequence shorter than G - 2 follows a long one, see the settlement)
    // settle separately, GROUPED by (sequences finished by the lane - those lane 0 of the slot had finished at the last refill)
    // modulo 3: in the middle of a sequence every lane of every slot is in group 0 and one all-reduce serves the wave (by the
    // sequence number alone, four slots would need several groups almost every time).
    int refCount = 0;
    constexpr int CKW = opal_swend_ckpt_words(S);
    auto ckpt_base = [&]() { return (unsigned)(uintptr_t)(opal_lds_u8*)(clkSlot + 48) + (unsigned)((wave * slots + slot) * 2 * CKW * 4); };  // (made when needed: no register across the loop)

    // (this slot's first item and the number of slots of the launch: see stream_fetch)
    const int streamBlocks = (int)gridDim.x - (SIDE ? sp.sideBlocks : 0), streamBlock = (int)blockIdx.x - (SIDE ? sp.sideBlocks : 0);
    // (only where the persistent workgroups start together: behind side workgroups they take their CUs one by one as those end, and
    // a workgroup that starts late must not be bound to a long sequence -- drawn from the counter, the late ones get what is left:
    // 390 aa on the published shape 10.5 ms drawn, 11.1 dealt)
    const bool deal = !SIDE || sp.sideBlocks == 0;
    const int totalSlots = deal ? streamBlocks * STREAM_WAVES * slots : 0, firstIdx = deal ? (wave * streamBlocks + streamBlock) * slots + slot : -1;
    // (the dealt item is the slot's FIRST fetch only: the step loop's fetches do not keep its index alive -- a register the 38- and
    // 39-row shapes do not have)
    auto fetch = [&](const int x) { stream_fetch<2, CHUNK>(st, p, sp.next, ring, x, k, G, slotLane0, PAD, sp.prioItems, -1, totalSlots); };
    stream_fetch<2, CHUNK>(st, p, sp.next, ring, k, k, G, slotLane0, PAD, sp.prioItems, firstIdx, totalSlots);
    unsigned bndq = 0;  // CHUNK: lane queue of the boundary words above the columns of this batch (lane 0 consumes)
    unsigned outq = 0;  // CHUNK: lane queue of the last lane's outputs, oldest towards lane 0

#ifndef OPAL_STREAM_PF_LATE
#define OPAL_STREAM_PF_LATE 1
#endif
    // LATE (single-launch streams): the profile address of the NEXT step, its gap block and its first row block are made and
    // read inside the last row block of the current step -- behind the batch refill when the next step starts a batch --, so
    // a step no longer begins by waiting out an LDS round trip (dp_kernel.hpp, PF_LATE: the registers that carry the three
    // reads are live across the step's tail only).
    // (... not at 33 rows per lane and more with 2-byte profile entries, score only: measured on the published shape, 2200 aa = 64 x 35
    // rows 68.6 ms with it, 63.5 without, 2432 aa = 64 x 38 rows 75.2 / 66.5; at 32 rows and fewer it gains 0-17 %, and with 4-byte
    // entries at every height -- profiles/r06_ab_stream_late_prefetch.txt)
    constexpr bool LATE = !CHUNK && S <= (SWEND ? 33 : (SIDE ? 37 : 38)) && !(ES == 2 && !SWEND && S > 32) && OPAL_STREAM_PF_LATE != 0;  // (the tallest shapes are two registers short)
    bool finished = false;
    int4 infoNext = make_int4(0, 0, 0, 0), pfirst = make_int4(0, 0, 0, 0);
    // start of a batch of G steps (wave-uniform): refill the queue with columns s .. s+G-1, fetch those of the next batch
    auto refill = [&](const int s) {
        {
            const unsigned letter = st.tokNext;
            const unsigned above = st.bndNext;
            if constexpr (CHUNK) {
                // The last lane's outputs were pushed at the start of each of the G steps behind us (position x leaves it in step
                // x+G-1): lane j holds the word of position s-2G+j. The positions [s-2G, s+G) touch at most three ring entries (an
                // entry spans at least 2G positions): the newest three.
                const int xo = s - 2 * G + k;
                if (hasNext && xo >= 0 && st.nEntries > 0 && xo < st.lastEnd) {
                    int e = st.nEntries - 1;
                    const int* d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4);
                    if (xo < d[0] && e > 0) { e--; d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4); }
                    if (xo < d[0] && e > 0) { e--; d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4); }
                    const int col = xo - d[0];
                    if (col >= 0 && col < d[1]) {
                        const uint64_t tb = (uint64_t)(unsigned)d[4] | ((uint64_t)(unsigned)d[5] << 32);
                        reinterpret_cast<unsigned*>(p.bnd)[tb + (uint64_t)col] = outq;
                    }
                }
            }
            fetch(s + G + k);
            if constexpr (SWEND) refCount = __shfl(flushCount, slotLane0, 64);
            tokq = ldsBase + letter * (unsigned)stride;
            bndq = above;
            // every slot has run dry and its last separator has left the last lane (CHUNK: the store above covered the positions
            // up to s-G-1, i.e. everything before lastEnd)
            if (__ballot(!(st.done && s >= st.lastEnd + G)) == 0ull) finished = true;
        }
    };
    auto next_address = [&]() {  // offv <- the profile address of the step that follows
        const unsigned shifted = shr1(offv) + (unsigned)LANE_BYTES;
        offv = (k == 0) ? tokq : shifted;
        tokq = shl1(tokq);
    };
    if constexpr (LATE) {
        refill(0);
        next_address();
        infoNext = lds_ld16(offv + infoOff);
        pfirst = lds_ld16(offv);
    }

    for (int s = 0; !finished; s++) {
        if constexpr (!LATE) {
            if ((s & (G - 1)) == 0) { refill(s); if (finished) break; }
        }
        if constexpr (CHUNK) {  // the last lane's outputs of the step behind us join the queue
            const unsigned mine = ((unsigned)outH & 0xffffu) | ((unsigned)outF << 16);
            const unsigned shifted = shl1(outq);
            outq = (k == G - 1) ? mine : shifted;
        }
        reg inH = (reg)shr1((unsigned)outH);
        reg inF = (reg)shr1((unsigned)outF);
        if constexpr (CHUNK) {  // top boundary: what the chunk above left for this column (zeros in the first chunk)
            inH = (k == 0) ? (reg)(bndq & 0xffffu) : inH;
            inF = (k == 0) ? (reg)(bndq >> 16) : inF;
            bndq = shl1(bndq);
        } else {
            inH = (k == 0) ? (reg)0 : inH;  // top boundary of local alignment
            inF = (k == 0) ? (reg)0 : inF;
        }
        if constexpr (!LATE) next_address();
        const int4 info = LATE ? infoNext : lds_ld16(offv + infoOff);
        const reg Qv = (reg)info.x, Rv = (reg)info.y;

        reg f = inF;
        constexpr int NQ = opal_swend_groups(S);
        reg aq[NQ];  // SWEND: the column maximum per group of row blocks (dp_kernel.hpp, opal_swend_groups), each started by its first block
        {
            constexpr int NBLK = SPAD / RB;
            int4 pc = LATE ? pfirst : lds_ld16(offv);
            reg d = V::adds(diagIn, (reg)pc.x), t = 0;
            static_for<NBLK>([&](auto rbc) {
                constexpr int rb = decltype(rbc)::value;
                int4 pn = pc;
                if constexpr (rb + 1 < NBLK) pn = lds_ld16(offv + (rb + 1) * RB * ES);
                else if constexpr (LATE) {  // every read of this step has been issued: the next step's address and reads
                    if (((s + 1) & (G - 1)) == 0) refill(s + 1);
                    next_address();
                    infoNext = lds_ld16(offv + infoOff);
                    pn = lds_ld16(offv);
                }
                reg Pv[RB + 1];
                if constexpr (ES == 4) {
                    Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.y; Pv[2] = (reg)pc.z; Pv[3] = (reg)pc.w;
                } else {
                    Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.x >> 16; Pv[2] = (reg)pc.y; Pv[3] = (reg)pc.y >> 16;
                    Pv[4] = (reg)pc.z; Pv[5] = (reg)pc.z >> 16; Pv[6] = (reg)pc.w; Pv[7] = (reg)pc.w >> 16;
                }
                Pv[RB] = (reg)pn.x;
                static_for<RB / 4>([&](auto subc) {
                    constexpr int j0 = rb * RB + decltype(subc)::value * 4;
                    if constexpr (j0 < S) {
                        constexpr int n = (S - j0) < 4 ? (S - j0) : 4;
                        constexpr bool last = j0 + n == S;
                        constexpr int grp = opal_swend_group_of_block(S, j0 / 4);
                        constexpr bool starts = SWEND && j0 / 4 == opal_swend_group_first_block(S, grp);
                        V::template rows<starts ? ROWS_SWVI : ROWS_SWV, n, last>(&Hp[j0], &En[j0], &Pv[j0 - rb * RB + 1], f, SWEND ? aq[grp] : mx, d, t, Qv, Rv);
                    }
                });
                pc = pn;
            });
            if constexpr (LATE) pfirst = pc;
        }
        outH = Hp[S - 1];
        outF = f;
        diagIn = inH;

        if constexpr (SWEND) {
            // End location in the same pass, exactly and per SLOT, as in the single-launch class kernels (dp_kernel.hpp, CKPT): a
            // lane's column maximum matters only if it beats the best cell of its sequence so far -- a higher value, or the same
            // value in a smaller column. Only when some lane of the WAVE has such a column do the slots settle who holds their
            // best (one all-reduce of (value, lane) keys per sequence in flight); the winner leaves the group of its rows that
            // holds the maximum in LDS, where the last lane looks the row up when the sequence has passed it. Lanes of the
            // winner's sequence take the new best as theirs, lanes still on the sequence before it as their next one's.
            // (Before: per-lane row scans filtered by two running bounds, one LDS atomic per 8 steps; 14 % over the score-only
            // stream at a 2000-aa query, 26 % with cheap gaps.)
            reg cm = aq[0];
            if constexpr (NQ == 2) cm = V::maxs(aq[0], aq[1]);
            if constexpr (NQ == 3) cm = V::maxs(V::maxs(aq[0], aq[1]), aq[2]);
            if constexpr (NQ == 4) cm = V::maxs(V::maxs(aq[0], aq[1]), V::maxs(aq[2], aq[NQ - 1]));
            const int cv = (int)((unsigned)cm & 0xffffu);  // (SW values are >= 0)
            const bool cand = OPAL_ABL >= 3 ? false : (cv > curV || (cv == curV && s < curS));
            if (__ballot(cand) != 0ull && (OPAL_ABL < 2 || p.satLimit < 0)) {
                // (THREE groups, round 6: the lanes move on to their next sequence at its predecessor's second separator, and two of
                // those can be three positions apart -- a sequence shorter than G - 2 behind a long one, whose entry is padded
                // BEHIND its separators -- so the G lanes of a slot can be on three sequences: n, the short n+1, and, in its pad
                // columns, already n+2. By parity alone the lanes on n+2 took the settlements of n for their own and never raised a
                // smaller best of n+2 again: a wrong end cell among equal scores, a lower score otherwise -- rare: found by the ties
                // test once in 25 runs (which slot a sequence lands in depends on timing). The two LDS buffers by parity are safe: n+2 has no
                // real column before every lane has left n.)
                const int par = flushCount & 1, grp = (int)((unsigned)(flushCount - refCount + 3000) % 3u);
                static_for<3>([&](auto pgc) {
                    constexpr int pg = decltype(pgc)::value;
                    const bool mine = cand && grp == pg;
                    if (__ballot(mine) != 0ull) {
                        const unsigned key = slot_max_u32(mine ? ((unsigned)(cv << 6) | (unsigned)k) + 1u : 0u, log2G);
                        if (key) {  // (the same in every lane of a slot)
                            const int kw = (int)((key - 1u) & 63u);
                            const int v = (int)((key - 1u) >> 6), at = s - kw + k;
                            if (grp == pg) {
                                curV = v; curS = at;
                                if (k == kw && (OPAL_ABL < 1 || p.satLimit < 0))
                                    swend_keep_column<V, S>(ckpt_base() + (unsigned)(par * CKW * 4), Hp, aq, cm, k);
                            } else if (kw < k) {  // (a winner BELOW me is on the sequence I have left)
                                nxtV = v; nxtS = at;
                            }
                        }
                    }
                });
            }
        }

        // ---- second separator column: hand the slot's maximum down the lanes, the last lane writes it
        // (SWEND: nothing to hand down -- every lane knows its sequence's best cell; the last lane looks its row up)
        if (__ballot(info.z != 0) != 0ull) {
            int inM = 0;
            if constexpr (!SWEND) {
                inM = (int)shr1((unsigned)outM);
                inM = (k == 0) ? 0 : inM;
            }
            int outCol = INT_MAX, outRow = -1;
            if constexpr (SWEND) {
                if (__ballot(info.z != 0 && k == G - 1) != 0ull) {  // (wave-uniform: once per sequence and slot)
                    if (info.z != 0 && k == G - 1 && curV > 0) {
                        int kw = 0;
                        const int r = swend_kept_row<S>(ckpt_base() + (unsigned)((flushCount & 1) * CKW * 4), curV, kw);
                        outRow = q0 + kw * S - padTop + r;
                    }
                }
            }
            if (info.z != 0) {
                const int* d = ring + (flushCount & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4);
                if constexpr (SWEND) {
                    outM = curV;
                    if (k == G - 1 && curV > 0) outCol = (curS - k) - d[0];  // (the entry's first stream position)
                    curV = nxtV; curS = nxtS; nxtV = 0; nxtS = INT_MIN;
                    flushCount++;
                } else {
                    const int mine = V::half(mx, 0);
                    outM = mine > inM ? mine : inM;
                }
                mx = 0;
                if (k == G - 1) {
                    const int seq = d[2];
                    if constexpr (!SWEND) flushCount++;
                    if (CHUNK && seq >= 0) {
                        // maximum over the chunks; a score at the saturation limit is listed for the 32-bit re-run once, by the
                        // first chunk that sees it, and marked for the chunks below (as the class kernels do, dp_kernel.hpp)
                        const int prev = hasPrev ? p.outScore[seq] : 0;
                        int sc = prev > outM ? prev : outM;
                        int col = outCol, row = outRow;
                        if constexpr (SWEND) {
                            // the chunks above hold smaller rows: on equal scores they keep the cell unless ours has a smaller column
                            if (hasPrev) {
                                int pc = p.outEndT[seq];
                                if (pc < 0) pc = INT_MAX;
                                if (prev > outM || (prev == outM && pc <= outCol)) { col = pc; row = p.outEndQ[seq]; }
                            }
                        }
                        if (sc >= p.satLimit) {
                            if (!(hasPrev && prev >= p.satLimit)) {
                                const int at = atomicAdd(p.ovfCount, 1);
                                p.ovfItems[at] = d[6];
                            }
                            int lim = p.satLimit;
                            asm volatile("" : "+s"(lim));  // (made here: as a lane value kept across the column loop it was spilled)
                            p.outScore[seq] = lim;
                        } else {
                            p.outScore[seq] = sc;
                            if constexpr (SWEND) {
                                const bool none = sc <= 0 || col == INT_MAX;  // score 0: no cell (test.cpp:242-243)
                                p.outEndQ[seq] = none ? -1 : row;
                                p.outEndT[seq] = none ? -1 : col;
                            }
                        }
                    } else if (seq >= 0) {
                        if (outM >= p.satLimit) {  // (opal.cpp:364-370): hand over to width 32
                            const int at = atomicAdd(p.ovfCount, 1);
                            p.ovfItems[at] = d[6];
                        } else {
                            p.outScore[seq] = outM;
                            if constexpr (SWEND) {
                                const bool none = outM <= 0 || outCol == INT_MAX;  // score 0: no cell (test.cpp:242-243)
                                p.outEndQ[seq] = none ? -1 : outRow;
                                p.outEndT[seq] = none ? -1 : outCol;
                            }
                        }
                    }
                }
            }
        }
    }
    clk_end(clkSlot);
}

// ------------------------------------------------------------------ NW / HW / OV as a stream of sequences
// The same persistent-workgroup scheme for the global modes (reference src/opal.cpp:594-977 searchDatabase_, with the
// lane re-initialisation of :921-961), in the diagonal-skewed representation of dp_kernel.hpp (6 instructions per cell:
// X' = X + (row register + local step) * Ge). There is no zero clamp to build a separator from, so a lane that reaches
// the first column of its next sequence RELOADS its boundary column: H[r][-1] and E[r][0] of its rows (in its frame at
// local step k) are per-lane constants, staged once per workgroup in LDS as 4-byte entries and fetched with
// ds_read_b128 straight into the DP registers -- no arithmetic. The local step (column + lane) restarts with every
// sequence, so the skew stays as small as in the class kernels; the host sends only sequences whose skewed values stay
// inside 16 bits (the others take the 32-bit path with those that need it anyway).
// Results: NW the final cell on the lane holding query row Q-1; HW the last-row maximum tracked by the last lane; OV
// also the last-column maximum, handed down the lanes as each reaches its last column (value, then smallest row).
template <int WIDTH>
__host__ __device__ inline int opal_gl_stream_stride(int G, int S) {
    constexpr int ES = WIDTH == 18 ? 2 : 4;
    return opal_letter_stride(G * opal_lane_bytes(ES, S, 1), G);
}
__host__ __device__ inline int opal_gl_init_entries(int S) { return 2 * ((S + 3) / 4 * 4) + 4; }  // Hp[], En[], diagonal input
template <int WIDTH>
__host__ __device__ inline size_t opal_gl_stream_lds_bytes(int G, int S, int A) {
    return (size_t)(A + 1) * (size_t)opal_gl_stream_stride<WIDTH>(G, S) + (size_t)G * opal_gl_init_entries(S) * 4 +
           (size_t)STREAM_WAVES * (64 / G) * STREAM_RING * STREAM_DESC_BYTES + 48;  // (+48: probe slot)
}

// CHUNK: one row chunk [q0, q0 + chunkRows) of a query beyond one wave, as in opal_sw_stream: the chunk's bottom row goes
// through p.bnd as TRUE values (H, and the F input of the row below; 16 bits each -- the host has proven that every true
// value fits), moved out of the last lane's skewed frame when it is queued for the store and into lane 0's when it is
// consumed (two additions each). NW pads below the query (the last chunk is the partial one), HW/OV above (the first);
// results are written by the last chunk only, OV's last-column maximum travels from chunk to chunk through p.scratch.
template <int WIDTH, int MODE, int S, bool CHUNK = false>
__global__ void __launch_bounds__(STREAM_WAVES * 64) opal_gl_stream(const StreamParams sp) {
    using V = Vec<WIDTH>;
    using reg = typename V::reg;
    using elem = typename V::elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int RB = 16 / ES;
    constexpr int SPAD = (S + RB - 1) / RB * RB;
    constexpr int SP4 = (S + 3) / 4 * 4;
    constexpr int INIT = 2 * SP4 + 4;  // entries of one lane's boundary block
    constexpr int LANE_BYTES = opal_lane_bytes(ES, S, 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const KParams& p = sp.base;
    count_started(p.started);
    if ((int)blockIdx.x < sp.sideBlocks) {
        // The sequences whose skewed values could leave 16 bits (the longest ones, a handful): the class body with plain
        // arithmetic, in the same launch and ahead of the streaming workgroups in dispatch order. (As a separate launch on
        // another stream they overlapped with the stream only when HIP happened to give the two streams different hardware
        // queues: 9 ms or 13 ms per search from run to run.)
        opal_dp_body<WIDTH, MODE, 0, S>(p, sp.side, (int)blockIdx.x, sp.sideWaves > 0 ? sp.sideWaves : STREAM_SIDE_WAVES);
        return;
    }

    const int log2G = p.log2G;
    const int G = 1 << log2G;
    const int stride = opal_gl_stream_stride<WIDTH>(G, S);
    const int A = p.A;
    const int q0 = CHUNK ? p.q0 : 0;           // first query row of this launch
    const int Q = CHUNK ? p.chunkRows : p.Q;   // ... and how many it covers
    const bool hasPrev = CHUNK && (p.chunkFlags & 1), hasNext = CHUNK && (p.chunkFlags & 2);
    const int Go = p.Go, Ge = p.Ge;
    const int padTop = (MODE == MODE_NW) ? 0 : (G * S - Q);  // NW pads below the query, HW/OV above (dp_kernel.hpp)
    const int PAD = A;
    const int sk = 2 * Ge;  // the skewed diagonal: H'(j,s) = H'(j-1,s-1) + P + 2*Ge
    unsigned char* const initBase = lds + (size_t)(A + 1) * stride;
    unsigned char* const clkSlot = initBase + (size_t)G * INIT * 4 + (size_t)STREAM_WAVES * (64 >> log2G) * STREAM_RING * STREAM_DESC_BYTES;
    clk_begin(p.clk, p.tl, clkSlot);

    // ---- stage the profile (+2*Ge) and every lane's boundary block
    {
        const int perLetter = G * SPAD;
        for (int rem = threadIdx.x; rem < perLetter; rem += blockDim.x) {
            const int k = rem / SPAD;
            const int j = rem - k * SPAD;
            const int lr = k * S + j - padTop;
            const bool real = lr >= 0 && j < S && lr < Q;
            unsigned char* dst = lds + k * LANE_BYTES + j * ES;
            if (real) {
                const int* mrow = p.matrix + (int)p.query[q0 + lr] * A;
                for (int letter = 0; letter < A; letter += 2) {
                    const int l1 = min(letter + 1, A - 1);
                    const int v0 = mrow[letter], v1 = mrow[l1];
                    *reinterpret_cast<elem*>(dst + letter * stride) = (elem)(v0 + sk);
                    *reinterpret_cast<elem*>(dst + l1 * stride) = (elem)(v1 + sk);
                }
                *reinterpret_cast<elem*>(dst + PAD * stride) = (elem)(p.negPad + sk);
            } else {
                const int fill = (lr < 0 ? 0 : p.negPad) + sk;
                for (int letter = 0; letter <= A; letter++) *reinterpret_cast<elem*>(dst + letter * stride) = (elem)fill;
            }
        }
        auto hbound = [&](int lr) -> long long {  // H[r][-1] of chunk row lr = query row q0 + lr (opal.cpp:671-679); rows above the query: 0
            return (MODE == MODE_OV || q0 + lr < 0) ? 0LL : -(long long)Go - (long long)(q0 + lr) * Ge;
        };
        for (int idx = threadIdx.x; idx < G * INIT; idx += blockDim.x) {
            const int k = idx / INIT;
            const int e = idx - k * INIT;
            const int r0k = k * S - padTop;
            int v = 0;
            // (frame: X' = X + (j + local step) * Ge. The class kernels moved to the row position k * S + j in round 4 and save the two
            // subtractions at the lane hand-off; here that costs more than it saves: with a 2000-aa query the frame would grow by
            // 2000 * Ge, every sequence beyond 1 870 instead of 1 993 residues would leave the 16-bit range and join the side
            // workgroups, which displace streaming ones -- C5 NW 7 900 -> 7 700 GCUPS, profiles/r04_ab_glstream.txt)
            if (e < SP4) {  // H[r][-1] in the frame of my step before my first column
                const int j = e;
                v = V::clampv(hbound(r0k + j)) + (j + k - 1) * Ge;
            } else if (e < 2 * SP4) {  // E[r][0] = H[r][-1] - Go in the frame of my first column
                const int j = e - SP4;
                v = V::clampv(hbound(r0k + j) - Go) + (j + k) * Ge;
            } else if (e == 2 * SP4) {  // H[r0-1][-1] as the lane above would have handed it over
                v = V::clampv(hbound(r0k - 1)) + (k - 2) * Ge;
            }
            reinterpret_cast<int*>(initBase)[idx] = v & 0xffff;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = lane & (G - 1);
    const int slot = lane >> log2G;
    const int slots = 64 >> log2G;
    int* const ring = reinterpret_cast<int*>(initBase + (size_t)G * INIT * 4 + (size_t)((wave * slots + slot) * STREAM_RING) * STREAM_DESC_BYTES);
    const int slotLane0 = lane & ~(G - 1);
    const unsigned myInit = (unsigned)(uintptr_t)(opal_lds_u8*)initBase + (unsigned)(k * INIT * 4);

    reg Hp[S], En[S];
#pragma unroll
    for (int j = 0; j < S; j++) { Hp[j] = 0; En[j] = 0; }
    reg outF = 0, diagIn = 0;
    const reg Qv = V::splat(Go), Rv = V::splat(Ge);
    const unsigned GQs = (unsigned)(Go - Ge) & 0xffffu;
    const reg SR1v = V::splat((S - 1) * Ge);
    const unsigned SR1s = (unsigned)((S - 1) * Ge) & 0xffffu;
    reg sRv = 0;  // (my column + k) * Ge: the local step of my current sequence
    const unsigned Qs = (unsigned)Go & 0xffffu, Rs = (unsigned)Ge & 0xffffu;  // (wave-uniform operands straight from SGPRs)
    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned padOff = ldsBase + (unsigned)(PAD * stride);
    unsigned offv = padOff + (unsigned)(k * LANE_BYTES);
    unsigned tokq = padOff;

    // where query row Q-1 lives: NW lane kLast, register jLast; HW/OV the last register of the last lane
    const int kLast = (MODE == MODE_NW) ? (Q - 1) / S : G - 1;
    const int jLast = (MODE == MODE_NW) ? (Q - 1) - kLast * S : S - 1;
    const int r0 = q0 + k * S - padTop;
    int jlo = padTop - k * S, jhi = Q + padTop - k * S;  // my registers [jlo, jhi) hold real query rows
    jlo = jlo < 0 ? 0 : (jlo > S ? S : jlo);
    jhi = jhi < 0 ? 0 : (jhi > S ? S : jhi);

    // The lane hand-off and the profile address as ONE DPP instruction each (shr1_sub16 / shr1_add32: their second operand must be a
    // register) -- where two more long-lived registers fit: the tallest shapes keep the shift and the arithmetic apart.
    constexpr bool FUSE = S <= 35;
    unsigned laneBytesV = (unsigned)LANE_BYTES, sr1V = (unsigned)SR1v;
    if constexpr (FUSE) asm volatile("" : "+v"(laneBytesV), "+v"(sr1V));
    SlotStream st;  // the stream of this slot
    int started = 0;                    // entries this lane has begun
    int myStart = 0, myNext = 0;  // my current sequence: first stream position; where my next one begins (its last column: xLast)
    int xLast = INT_MIN;                // stream position of its last column (INT_MIN: it has none)
    reg bestF = V::splat(-32768);  // HW/OV: last-row maximum so far, in the skewed frame of the step at hand
    int bestCol = -1;              // its first column
    int outLcV = INT_MIN, outLcRow = -1;                    // OV: last-column maximum handed down the lanes

    const int streamBlocks = (int)gridDim.x - sp.sideBlocks, streamBlock = (int)blockIdx.x - sp.sideBlocks;
    const bool deal = sp.sideBlocks == 0;  // (as in opal_sw_stream)
    const int totalSlots = deal ? streamBlocks * STREAM_WAVES * slots : 0, firstIdx = deal ? (wave * streamBlocks + streamBlock) * slots + slot : -1;
    auto fetch = [&](const int x) { stream_fetch<0, CHUNK>(st, p, sp.next, ring, x, k, G, slotLane0, PAD, sp.prioItems, -1, totalSlots); };
    stream_fetch<0, CHUNK>(st, p, sp.next, ring, k, k, G, slotLane0, PAD, sp.prioItems, firstIdx, totalSlots);
    unsigned bndq = 0;     // CHUNK: lane queue of the boundary words above the columns of this batch (lane 0 consumes)
    unsigned outq = 0;     // CHUNK: lane queue of the last lane's outputs (true values), oldest towards lane 0
    unsigned outPack = 0;  // CHUNK: what I produced in the step behind us, out of my frame

    // (batches of G steps: the refill -- and the test for the end of the stream -- sit between them, the inner loop has a plain trip
    // count; as ONE loop with the refill under a test every step paid a scalar test, a branch and what the compiler kept live for
    // the exit path)
    // (... where the nested loops fit the register file: at 36 rows per lane and more the compiler keeps more alive across them and
    // spills; those shapes run the single loop with the refill under a test, as in round 5)
    constexpr bool BATCHED = S <= 35;
    auto refill = [&](const int s) -> bool {  // start of a batch of G steps; true: the stream has ended
        {
            const unsigned letter = st.tokNext;
            const unsigned above = st.bndNext;
            if constexpr (CHUNK) {  // store the last lane's words of positions s-2G .. s-G-1 (see opal_sw_stream)
                const int xo = s - 2 * G + k;
                if (hasNext && xo >= 0 && st.nEntries > 0 && xo < st.lastEnd) {
                    int e = st.nEntries - 1;
                    const int* d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4);
                    if (xo < d[0] && e > 0) { e--; d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4); }
                    if (xo < d[0] && e > 0) { e--; d = ring + (e & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4); }
                    const int col = xo - d[0];
                    if (col >= 0 && col < d[1]) {
                        const uint64_t tb = (uint64_t)(unsigned)d[4] | ((uint64_t)(unsigned)d[5] << 32);
                        reinterpret_cast<unsigned*>(p.bnd)[tb + (uint64_t)col] = outq;
                    }
                }
            }
            fetch(s + G + k);
            tokq = ldsBase + letter * (unsigned)stride;
            bndq = above;
            if (__ballot(!(st.done && s >= st.lastEnd + G)) == 0ull) return true;
        }
        return false;
    };
    auto column_step = [&](const int s) {
        {
        if constexpr (CHUNK) {  // the last lane's outputs of the step behind us join the queue
            const unsigned shifted = shl1(outq);
            outq = (k == G - 1) ? outPack : shifted;
        }
        const int x = s - k;  // my stream position
        // ---- inputs of this column from the lane above (frame change of S-1 registers): its last row's H and F of the step behind
        // us, both still in their registers -- read BEFORE a lane that starts a sequence in this step reloads its rows
        reg inH, inF;
        if constexpr (FUSE) {
            inH = (reg)shr1_sub16((unsigned)Hp[S - 1], sr1V);
            inF = (reg)shr1_sub16((unsigned)outF, sr1V);
        } else {
            inH = V::subs_s((reg)shr1((unsigned)Hp[S - 1]), SR1s);
            inF = V::subs_s((reg)shr1((unsigned)outF), SR1s);
        }
        // ---- this step's profile address, and its first read, BEFORE the reload below: a lane that starts a sequence in this step
        // waits for sixteen boundary reads there, and the profile read rides along instead of starting its own round trip behind them
        // (the address depends on tokens only)
        int4 pfirst = make_int4(0, 0, 0, 0);
        if constexpr (BATCHED) {
            const unsigned shifted = FUSE ? shr1_add32(offv, laneBytesV) : shr1(offv) + (unsigned)LANE_BYTES;
            offv = (k == 0) ? tokq : shifted;
            tokq = shl1(tokq);
            pfirst = lds_ld16(offv);
        }
        // ---- first column of my next sequence: reload the boundary column
        if (__ballot(x == myNext) != 0ull) {
            if (x == myNext) {
                if (started < st.nEntries) {
                    const int4 d0 = *reinterpret_cast<const int4*>(ring + (started & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4));
                    myStart = d0.x;  // == myNext
                    myNext = d0.x + max(d0.y, 2 * G);
                    xLast = d0.y > 0 ? d0.x + d0.y - 1 : INT_MIN;
                    started++;
                } else {  // the stream has run dry: nothing of mine is a result any more
                    myNext = INT_MAX;
                    xLast = INT_MIN;
                }
                static_for<SP4 / 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    const int4 h4 = lds_ld16(myInit + b * 16);
                    const int4 e4 = lds_ld16(myInit + (SP4 + b * 4) * 4);
                    if constexpr (4 * b + 0 < S) { Hp[4 * b + 0] = (reg)h4.x; En[4 * b + 0] = (reg)e4.x; }
                    if constexpr (4 * b + 1 < S) { Hp[4 * b + 1] = (reg)h4.y; En[4 * b + 1] = (reg)e4.y; }
                    if constexpr (4 * b + 2 < S) { Hp[4 * b + 2] = (reg)h4.z; En[4 * b + 2] = (reg)e4.z; }
                    if constexpr (4 * b + 3 < S) { Hp[4 * b + 3] = (reg)h4.w; En[4 * b + 3] = (reg)e4.w; }
                });
                diagIn = (reg)lds_ld16(myInit + 2 * SP4 * 4).x;
                sRv = V::splat(k * Ge);
                bestF = V::splat(-32768); bestCol = INT_MIN;
            }
        }
        // (my column is x - myStart; the per-step code works with stream positions -- the last column is x == xLast, the best
        // column is kept as a position and turned into a column when it is written: three instructions per step less)

        // ---- ... or the top boundary
        if (k == 0) {
            // H[-1][c] + (c-1)*Ge and F[0][c] + c*Ge = H[-1][c] - Go + c*Ge; NW: H[-1][c] = -Go - c*Ge, HW/OV: 0
            if (hasPrev) {  // ... a lower chunk: what the chunk above left for this column, into my frame (k = 0: sRv = c*Ge)
                inH = V::adds((reg)(bndq & 0xffffu), V::subs_s(sRv, Rs));
                inF = V::adds((reg)(bndq >> 16), sRv);
            } else if (MODE == MODE_NW) { inH = V::splat(-Go - Ge); inF = V::splat(-2 * Go); }
            else { inH = V::subs_s(sRv, Rs); inF = V::subs_s(sRv, Qs); }
        }
        if constexpr (CHUNK) bndq = shl1(bndq);
        if constexpr (!BATCHED) {
            const unsigned shifted = FUSE ? shr1_add32(offv, laneBytesV) : shr1(offv) + (unsigned)LANE_BYTES;
            offv = (k == 0) ? tokq : shifted;
            tokq = shl1(tokq);
        }
        reg f = inF;
        reg unused = 0;
        {
            constexpr int NBLK = SPAD / RB;
            int4 pc = BATCHED ? pfirst : lds_ld16(offv);
            reg d = V::adds(diagIn, (reg)pc.x), t = 0;
            static_for<NBLK>([&](auto rbc) {
                constexpr int rb = decltype(rbc)::value;
                int4 pn = pc;
                if constexpr (rb + 1 < NBLK) pn = lds_ld16(offv + (rb + 1) * RB * ES);
                reg Pv[RB + 1];
                if constexpr (ES == 4) {
                    Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.y; Pv[2] = (reg)pc.z; Pv[3] = (reg)pc.w;
                } else {
                    Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.x >> 16; Pv[2] = (reg)pc.y; Pv[3] = (reg)pc.y >> 16;
                    Pv[4] = (reg)pc.z; Pv[5] = (reg)pc.z >> 16; Pv[6] = (reg)pc.w; Pv[7] = (reg)pc.w >> 16;
                }
                Pv[RB] = (reg)pn.x;
                static_for<RB / 4>([&](auto subc) {
                    constexpr int j0 = rb * RB + decltype(subc)::value * 4;
                    if constexpr (j0 < S) {
                        constexpr int n = (S - j0) < 4 ? (S - j0) : 4;
                        constexpr bool last = j0 + n == S;
                        V::template rows<ROWS_SKEW, n, last>(&Hp[j0], &En[j0], &Pv[j0 - rb * RB + 1], f, unused, d, t, GQs, 0u);
                    }
                });
                pc = pn;
            });
        }
        outF = f;
        diagIn = inH;
        if constexpr (CHUNK) {
            // my outputs as true values: H of my last row is in frame (S-1 + local step), the F input of the row below in
            // frame (S + local step); sRv = local step * Ge
            const reg fr = V::adds(sRv, SR1v);
            outPack = ((unsigned)V::subs(Hp[S - 1], fr) & 0xffffu) | ((unsigned)V::subs(outF, V::adds(fr, Rv)) << 16);
        }

        // ---- results (caller index and length of my sequence: descriptor started-1)
        if (MODE == MODE_NW && !hasNext) {
            if (k == kLast && x == xLast) {  // the final cell: once per sequence
                const reg hl = SelectRow<reg, S, 0, S>::get(Hp, jLast);
                const int v = V::half(V::subs(hl, V::adds(sRv, V::splat(jLast * Ge))), 0);
                const int seq = ring[((started - 1) & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4) + 2];
                if (seq >= 0) {
                    p.outScore[seq] = v;
                    if (p.outEndQ) { p.outEndQ[seq] = p.Q - 1; p.outEndT[seq] = x - myStart; }  // (x == xLast: the last column)
                }
            }
        }
        if (MODE == MODE_HW || MODE == MODE_OV) {
            const bool lastc = x == xLast;
            int lcV = INT_MIN, lcRow = -1;
            if (MODE == MODE_OV) {
                // the last column: its maximum over the query rows (value, then smallest row), lane by lane as the column
                // passes down (opal.cpp:881-904)
                if (__ballot(lastc) != 0ull) {
                    int inV = (int)shr1((unsigned)outLcV), inRow = (int)shr1((unsigned)outLcRow);
                    if (k == 0) { inV = INT_MIN; inRow = -1; }
                    if (lastc) {
                        int bv = INT_MIN, bj = -1;
                        if (jlo == 0 && jhi == S) {
                            // all my rows are query rows: the maximum in the skewed frame, the running best moving from the frame
                            // of row j+1 to that of row j with one subtraction
                            reg bf = Hp[S - 1];
                            if constexpr (S > 1) OvColumnMax<V, S, S - 2>::run(Hp, Rv, bf);  // rows S-2 .. 0: the value, 2 instructions per row
                            bv = V::half(V::subs(bf, sRv), 0);
                            if (bv > inV) {  // (rare: it beats the lanes above) ... then the smallest row holding it
                                reg t = V::adds(bf, SR1v);
                                bj = 0;
                                OvColumnRow<V, S, S - 1>::run(Hp, Rv, t, bj);
                            }
                        } else {
#pragma unroll
                            for (int j = S - 1; j >= 0; j--) {
                                const int v = V::half(V::subs(Hp[j], V::adds(sRv, V::splat(j * Ge))), 0);
                                if (j >= jlo && j < jhi && v >= bv) { bv = v; bj = j; }
                                asm volatile("" : "+v"(bv), "+v"(bj));
                            }
                        }
                        lcV = inV; lcRow = inRow;
                        if (bj >= 0 && bv > inV) { lcV = bv; lcRow = r0 + bj; }  // the lanes above hold smaller rows: they keep ties
                        outLcV = lcV; outLcRow = lcRow;
                    }
                }
            }
            // Last-row maximum and its first column (last query row = last register of the last lane = this column's outH),
            // tracked by EVERY lane in the skewed frame with full-rate 16-bit instructions -- no branch, no unskewing per
            // step: the running best moves into the frame of the new step with one addition (only the last lane's value is
            // ever used; columns beyond the sequence come after its result has been written).
            const reg prevF = bestF;                      // OV: the maximum over columns 0 .. c-1, in this step's frame
            V::track_best_any(Hp[S - 1], bestF, bestCol, x);  // strict >: first column wins (opal.cpp:824-834)
            if (k == kLast && lastc) {
                const reg frame = V::adds(sRv, SR1v);
                int sc = V::half(V::subs(bestF, frame), 0), eq = p.Q - 1, et = bestCol == INT_MIN ? -1 : bestCol - myStart;
                const int seq = ring[((started - 1) & (STREAM_RING - 1)) * (STREAM_DESC_BYTES / 4) + 2];
                if constexpr (CHUNK && MODE == MODE_OV) {  // the last-column maximum over the chunks (the upper rows win ties)
                    if (seq >= 0) {
                        if (hasPrev) {
                            const int pv = p.scratch[2 * seq], pr = p.scratch[2 * seq + 1];
                            if (pv >= lcV) { lcV = pv; lcRow = pr; }
                        }
                        if (hasNext) { p.scratch[2 * seq] = lcV; p.scratch[2 * seq + 1] = lcRow; }
                    }
                }
                if (MODE == MODE_OV) {
                    const int prevBest = (x > myStart) ? V::half(V::subs(prevF, frame), 0) : INT_MIN;  // (x == xLast here: a sequence of more than one column)
                    if (lcV > prevBest) { sc = lcV; et = x - myStart; eq = lcRow; }
                }
                if (seq >= 0 && !hasNext) {
                    p.outScore[seq] = sc;
                    if (p.outEndQ) { p.outEndQ[seq] = eq; p.outEndT[seq] = et; }
                }
            }
            bestF = V::adds_s(bestF, Rs);  // into the frame of the next step
        }
        sRv = V::adds_s(sRv, Rs);
        }
    };
    if constexpr (BATCHED) {
        for (int s0 = 0;; s0 += G) {
            if (refill(s0)) break;
            for (int s = s0; s < s0 + G; s++) column_step(s);
        }
    } else {
        for (int s = 0;; s++) {
            if ((s & (G - 1)) == 0 && refill(s)) break;
            column_step(s);
        }
    }
    clk_end(clkSlot);
}

StreamFn opal_stream_s17(int rows);     // width 17 (4-byte profile entries), score only: rows 1..40
StreamFn opal_stream_s17head(int rows, int want_end);  // with the class body for side items (SIDE): rows 1..39, with end locations 1..36
StreamFn opal_stream_s17chunk(int rows);  // one row chunk of a longer query (CHUNK), width 17, score only: rows 16..40
StreamFn opal_stream_s17chunkend(int rows);  // ... with end locations: rows 16..32
StreamFn opal_stream_s18(int rows);     // width 18 (2-byte entries), score only: rows 20..40
StreamFn opal_stream_s17end(int rows);  // with end locations: rows 1..36
StreamFn opal_stream_s18end(int rows);  // rows 20..36
StreamFn opal_stream_gl17(int mode, int rows);  // NW/HW/OV, width 17: rows 1..40
StreamFn opal_stream_gl18(int mode, int rows);  // width 18: rows 20..40
StreamFn opal_stream_gl17chunk(int mode, int rows);  // one row chunk of a longer query (CHUNK), width 17: rows 24..40
