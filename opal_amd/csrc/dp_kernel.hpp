// dp_kernel.hpp -- gfx950 (MI355X, CDNA4) device code of the affine-gap database search.
//
// Replaces the reference's SIMD kernels (reference src/opal.cpp:164-470 searchDatabaseSW_,
// :594-977 searchDatabase_, :472-490 loadNextSequence). Not a translation: the reference keeps
// one database sequence per SIMD lane and streams a whole H/E column through memory; here a
// wavefront is a 64-lane systolic array and the column never leaves the register file.
//
// Mapping:
//   * A wave holds 64/G "slots"; a slot is G adjacent lanes working on ONE database sequence
//     (width 17: plain 16-bit arithmetic on full-rate VALU instructions, the default; width 32:
//     int32) or on a PAIR of sequences packed 2 x int16 per VGPR (width 16, v_pk_* math; used when
//     the 4-byte profile entries of width 17 do not fit LDS). G = 1 is the pure
//     lane-owns-a-sequence mapping.
//   * Lane k of a slot owns S consecutive query rows and keeps their H and E values in VGPRs for
//     the whole sequence (S is a template parameter: the row loop is straight-line code).
//   * At step s lane k computes column c = s-k. The bottom H and F of lane k-1 and the LDS
//     profile address of that column's residues arrive through one wave_shr:1 DPP shift each.
//   * The query profile QP[target letter][query row] = scoreMatrix[query[row]][letter] is staged
//     once per block in LDS, laid out [letter][lane k][row j]; every read is one naturally aligned
//     ds_read_b128 (8 rows of int16, 4 rows of 4-byte entries); lanes that hold the same letter
//     broadcast, different letters fall in different 16-byte bank slots (odd letter stride).
//   * Database residues stream from HBM one byte per column (the sequences sorted by length, back to back); only the
//     packed kernels of width 16 read 16-bit tokens (low byte: sequence A, high byte: B of a pair).
//   * The 16-bit kernels hand scores at their limit to a re-run list that the host schedules at
//     width 32 (the 16 -> 32 bit rung of the reference's precision ladder, opal.cpp:496-535 /
//     :983-1021; gfx950 has no packed 8-bit VALU, so an 8-bit rung would only add work).
//   * FLAG_MULTI: queries longer than G*S rows run as several launches over consecutive row chunks;
//     the bottom row of a chunk travels through HBM (KParams::bnd).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include "opal_hip.h"  // OPAL_ZERO_COPY_MAX_ROWS*

#include <type_traits>
#include <utility>

#define MODE_NW 0
#define MODE_HW 1
#define MODE_OV 2
#define MODE_SW 3

#ifndef OPAL_ABL
#define OPAL_ABL 0  // (measurement only: 1 = no column store, 2 = no event at all, 3 = no end-cell code; results wrong)
#endif
#define FLAG_SWEND 1  // SW with end location: the lane also records where its running maximum was first reached
#define FLAG_WRAP 2   // width 32 SW: detect 32-bit wrap of ulH + P (reference opal.cpp:302-303,339-351)
#define FLAG_MULTI 4  // the query is processed in several row chunks; the boundary row travels through HBM
#define FLAG_SKEW 8   // width 17, NW/HW/OV: diagonal-skewed values, no gap-extension subtractions (see row_skew)
#define FLAG_NOCKPT 32  // SW with end location without the column checkpoint in LDS (class bodies inside a streaming launch: its LDS is laid out otherwise)
#define FLAG_ZC 16    // zero-copy search: the residues are read from pinned host memory and every one is also stored to
                      // KParams::resOut, the device copy (own instantiations: the store costs the SW score kernel 2.2 %)

#define WAVES_PER_BLOCK 4      // default waves per block (the launch may use fewer or more, up to MAX)
#define MAX_WAVES_PER_BLOCK 8
// Width 18 serves long queries whose profile fills most of the LDS: one workgroup per CU, so the workgroup itself
// must bring 4 waves per SIMD (16 waves; its kernels stay below the 128 registers that allows).
#define MAX_WAVES_PER_BLOCK_W18 16
__host__ __device__ constexpr int opal_max_waves_per_block(int width) { return width == 18 ? MAX_WAVES_PER_BLOCK_W18 : MAX_WAVES_PER_BLOCK; }

struct KParams {
    const unsigned char* query;
    const int* matrix;
    int Q, A, Go, Ge;
    const unsigned char* res;  // residues of every sequence (item order = sorted by length, longest first), one byte each
    unsigned char* resOut;     // not null: `res` is the pinned HOST staging buffer of a database seen for the first time (read
                               // over PCIe while the DP runs); every token a lane decodes is also written here, the device copy
    const uint64_t* seqOff;    // [items] where each one starts
    const uint16_t* tokens;    // width 16 only: per pair (items 2p, 2p+1), column by column, residue of A | residue of B << 8
    const uint64_t* pairOff;   // width 16 only: [pairs] start of each pair's token stream
    const int* pairLen;        // [items] lengths (item = pair * 2 + half)
    const int* pairSeq;        // [items] caller indices (-1: absent)
    const int* work;
    int firstItem, numItems, log2G;
    int* outScore;
    int* outEndQ;
    int* outEndT;
    int* ovfCount;
    int* ovfItems;
    int prio;
    // FLAG_MULTI only: this launch covers query rows [q0, q0+chunkRows); chunkFlags bit0 = a chunk above
    // exists (read its bottom row from `bnd`), bit1 = a chunk below exists (write ours to `bnd`).
    int q0, chunkRows, chunkFlags;
    uint2* bnd;    // per column {H, F-input} of the chunk boundary: [token offset + c] (width 16: of the pair's stream, else of the sequence)
    int* scratch;  // [2N] OV only: last-column maximum and its row carried across chunks
    int satLimit, negPad;  // width 17 only
    // Clock probe (bench.py, off = nullptr): wave 0 of every workgroup adds the shader cycles (s_memtime) and the
    // constant-rate ticks (s_memrealtime) of its own lifetime to clk[0] / clk[1]; their ratio is the clock the chip
    // sustained under THIS kernel (it clocks to its power budget: MI355X_MICROARCH.md, DVFS).
    unsigned long long* clk;
    // Timeline probe (off = nullptr): every workgroup writes the constant-rate ticks at its start and end to
    // tl[2 * blockIdx.x], tl[2 * blockIdx.x + 1] (scripts/gpu_timeline.py: where the time of a launch goes), and the shader
    // cycles at the same moments to tl[2 * 65536 + ...] (scripts/gpu_clock_timeline.py: the clock over the life of a launch).
    unsigned long long* tl;
    // (may be null) every workgroup of the launch adds one when it starts, before anything else: a cumulative counter that is
    // never zeroed. The next search queued on the handle waits -- on its own stream, opal_hip_wait_count -- until the last
    // workgroup of this grid has been dispatched, then fills the CUs this one leaves as it drains (ladder.cpp, SearchPipe).
    int* started;
};
__device__ __forceinline__ void count_started(int* started) {
    if (threadIdx.x == 0 && started) atomicAdd(started, 1);
}

// The two start stamps -- and the pointer itself -- wait in 48 bytes behind the workgroup's LDS data, so that nothing of
// the probe is live in a register across the column loop (the 40-row kernels sit at 128 VGPRs).
__device__ __forceinline__ void clk_begin(unsigned long long* clk, unsigned long long* tl, unsigned char* slot, unsigned char* resOut = nullptr) {
    if (threadIdx.x == 0) {
        unsigned long long* s = reinterpret_cast<unsigned long long*>(slot);
        s[2] = (unsigned long long)(uintptr_t)clk;
        s[3] = (unsigned long long)(uintptr_t)tl;
        s[4] = (unsigned long long)(uintptr_t)resOut;  // (zero-copy search: see advance())
        if (clk || tl) { s[0] = (unsigned long long)clock64(); s[1] = (unsigned long long)wall_clock64(); }
    }
}
__device__ __forceinline__ void clk_end(const unsigned char* slot) {
    if (threadIdx.x == 0) {
        const unsigned long long* s = reinterpret_cast<const unsigned long long*>(slot);
        unsigned long long* clk = reinterpret_cast<unsigned long long*>((uintptr_t)s[2]);
        if (clk) {
            atomicAdd(clk, (unsigned long long)clock64() - s[0]);
            atomicAdd(clk + 1, (unsigned long long)wall_clock64() - s[1]);
        }
        unsigned long long* tl = reinterpret_cast<unsigned long long*>((uintptr_t)s[3]);
        if (tl && blockIdx.x < 65536u) {
            tl[2 * (size_t)blockIdx.x] = s[1]; tl[2 * (size_t)blockIdx.x + 1] = (unsigned long long)wall_clock64();
            // ... and the shader cycles at the same two moments (second half of the buffer): the clock this workgroup saw
            tl[2 * 65536 + 2 * (size_t)blockIdx.x] = s[0]; tl[2 * 65536 + 2 * (size_t)blockIdx.x + 1] = (unsigned long long)clock64();
        }
    }
}

// What differs between the length classes of one search (the rest of KParams is shared).
struct KClass {
    int log2G, firstItem, numItems, prio;
};

typedef void (*KernelFn)(const KParams);

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ lane-vector arithmetic
template <int WIDTH> struct Vec;

template <> struct Vec<16> {  // two int16 scores per register
    typedef unsigned int reg;
    typedef short elem;
    static constexpr int NH = 2;        // sequences per lane
    static constexpr int NEG = -32768;  // profile value of pad residues / rows below the query
    static constexpr int SATMAX = 32767;
    static __device__ __forceinline__ reg splat(int v) { return ((unsigned)v & 0xffffu) | ((unsigned)v << 16); }
    static __device__ __forceinline__ reg adds(reg a, reg b) {
        return __builtin_bit_cast(reg, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
    }
    static __device__ __forceinline__ reg subs(reg a, reg b) {
        return __builtin_bit_cast(reg, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
    }
    static __device__ __forceinline__ reg subu(reg a, reg b) {  // unsigned saturating: max(a-b, 0)
        return __builtin_bit_cast(reg, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
    }
    static __device__ __forceinline__ reg maxs(reg a, reg b) {
        return __builtin_bit_cast(reg, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
    }
    static __device__ __forceinline__ int half(reg a, int h) { return h ? ((int)a >> 16) : (int)(short)(a & 0xffffu); }
    static __device__ __forceinline__ bool same(reg a, reg b, int h) { return half(a, h) == half(b, h); }
    static __device__ __forceinline__ int clampv(long long v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : (int)v); }
};

template <> struct Vec<32> {  // one int32 score per register
    typedef int reg;
    typedef int elem;
    static constexpr int NH = 1;
    static constexpr int NEG = -(1 << 30);
    static constexpr int SATMAX = 0x7fffffff;
    static __device__ __forceinline__ reg splat(int v) { return v; }
    static __device__ __forceinline__ reg adds(reg a, reg b) { return (int)((unsigned)a + (unsigned)b); }  // wrapping, like the reference
    static __device__ __forceinline__ reg subs(reg a, reg b) { return (int)((unsigned)a - (unsigned)b); }
    static __device__ __forceinline__ reg subu(reg a, reg b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
    static __device__ __forceinline__ reg maxs(reg a, reg b) { return a > b ? a : b; }
    static __device__ __forceinline__ int half(reg a, int) { return a; }
    static __device__ __forceinline__ bool same(reg a, reg b, int) { return a == b; }
    static __device__ __forceinline__ int clampv(long long v) {
        return v < -(1LL << 30) ? -(1 << 30) : (v > (1LL << 30) ? (1 << 30) : (int)v);
    }
};

// Width "17": ONE sequence per lane in plain (non-packed) 16-bit arithmetic. On gfx950 the packed
// v_pk_*_i16 instructions and v_perm_b32 issue every 4 clocks, the scalar 16-bit v_add_u16 / v_sub_u16
// (clamp) / v_max_i16 every 2 (profiles/r01_ubench_valu_survey.txt), and one sequence per lane needs no
// v_perm to pair profile values: 9 full-rate instructions per cell instead of 10 half-rate ones per two.
// The instructions are pinned with inline asm because hipcc would fuse two v_max_i16 into one
// v_max3_i16, which issues every 8 clocks. Values live in the low 16 bits of a VGPR. Additions wrap, so
// the host passes a limit below which no wrap can have happened (satLimit = 32767 - max matrix entry);
// scores at or above it are re-run at 32 bits.
template <> struct Vec<17> {
    typedef unsigned int reg;
    typedef int elem;  // LDS profile entries are 4 bytes wide (value in the low half): no unpacking
    static constexpr int NH = 1;
    static constexpr int NEG = -32768;  // replaced by KParams::negPad
    static constexpr int SATMAX = 32767;
    static __device__ __forceinline__ reg splat(int v) { return (unsigned)v & 0xffffu; }
    static __device__ __forceinline__ reg adds(reg a, reg b) { reg r; asm("v_add_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ reg subs(reg a, reg b) { reg r; asm("v_sub_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ reg subu(reg a, reg b) { reg r; asm("v_sub_u16_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ reg maxs(reg a, reg b) { reg r; asm("v_max_i16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    // the same with a wave-uniform second operand straight from an SGPR (no per-step v_mov to make a VGPR copy of it)
    static __device__ __forceinline__ reg adds_s(reg a, unsigned s) { reg r; asm("v_add_u16_e32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(s)); return r; }
    static __device__ __forceinline__ reg subs_s(reg a, unsigned s) { reg r; asm("v_subrev_u16_e32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(s)); return r; }  // a - s
    // HW / OV, skewed frame: the running last-row maximum bf and its first column. A column c < limit whose last-row value h
    // is STRICTLY larger takes over (first column wins, opal.cpp:824-834); bf then moves into the next step's frame (+ Ge).
    // One statement: 16-bit compare (hipcc shifts both sides up and compares 32 bits), the gap as an SGPR operand, and the two
    // wait states gfx950 wants between a compare and the select that reads its mask.
    static __device__ __forceinline__ void track_best(reg h, reg& bf, int& col, int c, unsigned limit, unsigned Rs) {
        unsigned long long m;
        reg cand;
        asm("v_cmp_lt_u32_e64 %[m], %[c], %[lim]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[cand], %[bf], %[h], %[m]\n\t"   // (outside the sequence: the running best itself, which is not > itself)
            "v_cmp_gt_i16_e64 %[m], %[cand], %[bf]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[col], %[col], %[c], %[m]\n\t"
            "v_cndmask_b32_e64 %[bf], %[bf], %[cand], %[m]\n\t"
            "v_add_u16_e32 %[bf], %[rs], %[bf]"
            : [m] "=&s"(m), [cand] "=&v"(cand), [col] "+v"(col), [bf] "+v"(bf)
            : [c] "v"(c), [lim] "v"(limit), [h] "v"(h), [rs] "s"(Rs));
    }
    // ... the same without a column limit (streaming kernels: columns beyond the sequence come after its result was written)
    static __device__ __forceinline__ void track_best_any(reg h, reg& bf, int& col, int c) {
        unsigned long long m;
        asm("v_cmp_gt_i16_e64 %[m], %[h], %[bf]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[col], %[col], %[c], %[m]\n\t"
            "v_cndmask_b32_e64 %[bf], %[bf], %[h], %[m]"
            : [m] "=&s"(m), [col] "+v"(col), [bf] "+v"(bf)
            : [c] "v"(c), [h] "v"(h));
    }
    static __device__ __forceinline__ int half(reg a, int) { return (int)(short)(a & 0xffffu); }
    static __device__ __forceinline__ bool same(reg a, reg b, int) { return (unsigned short)a == (unsigned short)b; }  // v_cmp_eq_u16
    static __device__ __forceinline__ int clampv(long long v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : (int)v); }
    // One row of one column as ONE asm block (hipcc pads every asm statement with a wait state and cannot
    // coalesce registers through it, so the block is as large as the data flow allows). d = H(diag) + P.
    // SW: E and F stay >= 0 through the unsigned clamp, so H >= 0 without a max against 0.
    // The gap penalties are wave-uniform: they enter as SGPR operands (no VGPR copies to keep or re-make per step).
    static __device__ __forceinline__ void row_sw(reg d, reg& e, reg& f, reg& acc, reg& h, unsigned Qs, unsigned Rs) {
        reg hq;
        asm("v_max_i16_e32 %0, %5, %1\n\t"
            "v_max_i16_e32 %0, %0, %2\n\t"
            "v_sub_u16_e64 %1, %1, %7 clamp\n\t"
            "v_sub_u16_e64 %4, %0, %6 clamp\n\t"
            "v_sub_u16_e64 %2, %2, %7 clamp\n\t"
            "v_max_i16_e32 %3, %3, %0\n\t"
            "v_max_i16_e32 %1, %1, %4\n\t"
            "v_max_i16_e32 %2, %2, %4"
            : "=&v"(h), "+v"(e), "+v"(f), "+v"(acc), "=&v"(hq)
            : "v"(d), "s"(Qs), "s"(Rs));
    }
    // NW/HW/OV in the skewed representation X' = X + (j + s)*Ge (j: register index, s: step): a gap
    // extension costs nothing any more, E' = max(E', g), F' = max(F', g) with g = H' - (Go - Ge); the profile
    // carries the +2*Ge the diagonal needs. 6 instructions per cell with the add, instead of 8.
    static __device__ __forceinline__ void row_skew(reg d, reg& e, reg& f, reg& h, unsigned GQs) {
        reg g;
        asm("v_max_i16_e32 %0, %4, %1\n\t"
            "v_max_i16_e32 %0, %0, %2\n\t"
            "v_subrev_u16_e32 %3, %5, %0\n\t"
            "v_max_i16_e32 %1, %1, %3\n\t"
            "v_max_i16_e32 %2, %2, %3"
            : "=&v"(h), "+v"(e), "+v"(f), "=&v"(g)
            : "v"(d), "s"(GQs));
    }
    // End-location row scan, N <= 4 rows per statement: r = (H[j0+i] == cm) ? j0+i : r, highest row first so that the
    // smallest matching row wins. v_cmp_eq_u16 + v_cndmask = 2 instructions per row; the compares of a block come first
    // so that two instructions separate a compare from the select that reads its SGPR mask (gfx950 hazard).
    template <int N, int J0>
    static __device__ __forceinline__ void scan_rows(const reg* H, reg cm, int& r) {
        unsigned long long m0, m1, m2, m3;
        if constexpr (N == 4) {
            asm("v_cmp_eq_u16_e64 %[m3], %[h3], %[cm]\n\tv_cmp_eq_u16_e64 %[m2], %[h2], %[cm]\n\t"
                "v_cmp_eq_u16_e64 %[m1], %[h1], %[cm]\n\tv_cmp_eq_u16_e64 %[m0], %[h0], %[cm]\n\t"
                "v_cndmask_b32_e64 %[r], %[r], %[j3], %[m3]\n\tv_cndmask_b32_e64 %[r], %[r], %[j2], %[m2]\n\t"
                "v_cndmask_b32_e64 %[r], %[r], %[j1], %[m1]\n\tv_cndmask_b32_e64 %[r], %[r], %[j0], %[m0]"
                : [r] "+v"(r), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
                : [h0] "v"(H[0]), [h1] "v"(H[1]), [h2] "v"(H[2]), [h3] "v"(H[3]), [cm] "v"(cm), [j0] "n"(J0), [j1] "n"(J0 + 1),
                  [j2] "n"(J0 + 2), [j3] "n"(J0 + 3));
        } else if constexpr (N == 3) {
            asm("v_cmp_eq_u16_e64 %[m2], %[h2], %[cm]\n\tv_cmp_eq_u16_e64 %[m1], %[h1], %[cm]\n\t"
                "v_cmp_eq_u16_e64 %[m0], %[h0], %[cm]\n\t"
                "v_cndmask_b32_e64 %[r], %[r], %[j2], %[m2]\n\tv_cndmask_b32_e64 %[r], %[r], %[j1], %[m1]\n\t"
                "v_cndmask_b32_e64 %[r], %[r], %[j0], %[m0]"
                : [r] "+v"(r), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
                : [h0] "v"(H[0]), [h1] "v"(H[1]), [h2] "v"(H[2]), [cm] "v"(cm), [j0] "n"(J0), [j1] "n"(J0 + 1), [j2] "n"(J0 + 2));
        } else if constexpr (N == 2) {
            asm("v_cmp_eq_u16_e64 %[m1], %[h1], %[cm]\n\tv_cmp_eq_u16_e64 %[m0], %[h0], %[cm]\n\ts_nop 0\n\t"
                "v_cndmask_b32_e64 %[r], %[r], %[j1], %[m1]\n\tv_cndmask_b32_e64 %[r], %[r], %[j0], %[m0]"
                : [r] "+v"(r), [m0] "=&s"(m0), [m1] "=&s"(m1)
                : [h0] "v"(H[0]), [h1] "v"(H[1]), [cm] "v"(cm), [j0] "n"(J0), [j1] "n"(J0 + 1));
        } else {
            asm("v_cmp_eq_u16_e64 %[m0], %[h0], %[cm]\n\ts_nop 1\n\tv_cndmask_b32_e64 %[r], %[r], %[j0], %[m0]"
                : [r] "+v"(r), [m0] "=&s"(m0)
                : [h0] "v"(H[0]), [cm] "v"(cm), [j0] "n"(J0));
        }
    }
    // Blocks of up to 4 rows as one asm statement each (generated: scripts/gen_row_blocks.py)
#include "row_blocks.inc"
    // NW/HW/OV: signed, wrapping (the host proves a-priori that nothing leaves the 16-bit range)
    static __device__ __forceinline__ void row_global(reg d, reg& e, reg& f, reg& h, unsigned Qs, unsigned Rs) {
        reg hq;
        asm("v_max_i16_e32 %0, %4, %1\n\t"
            "v_max_i16_e32 %0, %0, %2\n\t"
            "v_subrev_u16_e32 %1, %6, %1\n\t"
            "v_subrev_u16_e32 %3, %5, %0\n\t"
            "v_subrev_u16_e32 %2, %6, %2\n\t"
            "v_max_i16_e32 %1, %1, %3\n\t"
            "v_max_i16_e32 %2, %2, %3"
            : "=&v"(h), "+v"(e), "+v"(f), "=&v"(hq)
            : "v"(d), "s"(Qs), "s"(Rs));
    }
};

// Width "18": the same arithmetic as width 17 with 2-byte profile entries (one ds_read_b128 feeds 8 rows; the odd
// rows' values are the high halves, moved down by one full-rate v_lshrrev_b32 each): half the LDS, so queries of
// ~1600-2560 residues, whose 4-byte profile exceeds 160 KiB, still run one sequence per lane on full-rate
// instructions (+0.5 instructions per cell) instead of falling back to the packed pairs of width 16.
template <> struct Vec<18> : Vec<17> {
    typedef short elem;
};

// wave_shr:1 -- lane i receives lane i-1's value; lane 0 receives 0 (bound_ctrl: no `old` register to prepare).
__device__ __forceinline__ unsigned shr1(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true);
}
// ... fused with the arithmetic that always follows it in the streaming kernels (a DPP operand costs the instruction what the
// separate v_mov_dpp cost: one instruction instead of two). `s_nop 1`: the two wait states between a VALU write of the shifted
// register and its DPP read -- the compiler cannot see into the statement.
__device__ __forceinline__ unsigned shr1_sub16(unsigned v, unsigned c) {  // lane i: (lane i-1's v) - c, 16 bit (lane 0: -c)
    unsigned r;
    asm("s_nop 1\n\tv_sub_u16_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned shr1_add32(unsigned v, unsigned c) {  // lane i: (lane i-1's v) + c (lane 0: c)
    unsigned r;
    asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(c));
    return r;
}
// wave_shl:1 -- lane i receives lane i+1's value; lane 63 receives 0.
__device__ __forceinline__ unsigned shl1(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true);
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int o = __shfl_xor(v, d, 64);
        v = v > o ? v : o;
    }
    return v;
}

// Register select with a wave-uniform runtime index (used once per sequence, NW only). A select chain
// with an opaque barrier per step: a plain indexed read -- or a balanced tree of selects, which LLVM
// turns back into one -- would move the whole register array to scratch memory.
template <typename T, int S, int LO, int HI> struct SelectRow {
    static __device__ __forceinline__ T get(const T (&a)[S], int idx) {
        T r = a[LO];
#pragma unroll
        for (int j = LO + 1; j < HI; j++) {
            r = (idx == j) ? a[j] : r;
            asm volatile("" : "+v"(r));
        }
        return r;
    }
};

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One aligned 16-byte LDS read at an absolute LDS byte address (profile offsets carry the LDS base, so no address
// arithmetic is left per read: ds_read_b128 v, vaddr offset:imm).
typedef int opal_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const opal_v4i opal_lds_v4i;
typedef __attribute__((address_space(3))) unsigned char opal_lds_u8;
__device__ __forceinline__ int4 lds_ld16(unsigned addr) {
    const opal_v4i v = *(opal_lds_v4i*)(uintptr_t)addr;
    return make_int4(v.x, v.y, v.z, v.w);
}

// LDS geometry shared by kernel and host-side sizing
// Every LDS read is one naturally aligned 16-byte ds_read_b128 (a misaligned one is replayed at ~64 cycles). The LDS serves
// such a read in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same + 32; MI355X_MICROARCH.md,
// LDS): one cycle per group when the 16 lanes fall on 16 different 16-byte bank slots (byte address / 16 mod 16), one more
// for every further address on a slot. Lane k of a slot reads QP[its letter][k][rows]: with the letters of a group's lanes
// all different (the lanes of a slot sit on different columns) the slot must not depend on the LETTER, or sixteen
// random slots collide 2.8-fold (round 5: 11.3 LDS cycles per read instead of 4; the LDS array 76 % busy under the
// 6-instruction kernels). Hence, from 8 lanes per sequence on:
//   * the letter stride is a multiple of 256 bytes, and
//   * a lane's segment is U 16-byte units with U odd -- k * U mod 16 then differs for the 16 values of k mod 16: with 16 or more
//     lanes per sequence every hardware group holds 16 different k mod 16, no conflict at all -- or U = 2 mod 4 (4-byte entries
//     only: 21-24 and 37-40 rows, where one more unit would cost the 8-lane shapes of a 300-aa query a workgroup per CU): the
//     eight k of an 8-lane slot still fall on eight different slots, 16 lanes collide in pairs.
// 8 lanes per sequence: a hardware group holds every k twice (two sequences): two cycles per group instead of 2.8 on average.
// Fewer lanes: the letter stride stays an odd multiple of 16 bytes (different letters on different slots: the random spread is
// the better one there).
__host__ __device__ inline int opal_rows_per_read(int es) { return 16 / es; }
// Bytes of one lane's segment of a letter's profile (padOdd: the rule above; the role-swapped kernels lay their profile out
// otherwise and pass 0).
__host__ __device__ constexpr int opal_lane_bytes(int es, int S, int padOdd) {
    const int rpr = 16 / es;
    int b = (S + rpr - 1) / rpr * rpr * es;
    const int u = b / 16;
    if (padOdd && ((u & 3) == 0 || (es == 2 && (u & 1) == 0))) b += 16;
    return b;
}
// Letter stride for `bytes` of payload per letter at G lanes per sequence
__host__ __device__ inline int opal_letter_stride(int bytes, int G, int padOdd = 1) {
    if (padOdd && G >= 8) return (bytes + 255) & ~255;
    int stride = bytes;
    if (((stride / 16) & 1) == 0) stride += 16;
    return stride;
}
__host__ __device__ inline int opal_profile_stride(int es, int G, int S, int padOdd = 1) {
    return opal_letter_stride(G * opal_lane_bytes(es, S, padOdd), G, padOdd);  // one target letter's profile
}

// OV, last column of a sequence: the maximum over a lane's rows in the skewed frame, and the smallest row attaining it.
// Two parts: the VALUE alone costs two full-rate instructions per row (best = max(best - Ge, H[J])
// walks the running best from the frame of row J+1 into that of row J); the ROW holding it -- the smallest one -- is looked
// for only when the value beats what the lanes above have found, which it rarely does (three instructions per row then:
// the target moves from the frame of row J+1 to that of row J with one subtraction).
template <typename V, int S, int J> struct OvColumnMax {
    static __device__ __forceinline__ void run(const typename V::reg (&Hp)[S], typename V::reg Rv, typename V::reg& bf) {
        asm("v_sub_u16_e32 %[bf], %[bf], %[rv]\n\t"
            "v_max_i16_e32 %[bf], %[bf], %[h]"
            : [bf] "+v"(bf)
            : [h] "v"(Hp[J]), [rv] "v"(Rv));
        if constexpr (J > 0) OvColumnMax<V, S, J - 1>::run(Hp, Rv, bf);
    }
};
template <typename V, int S, int J> struct OvColumnRow {
    // t: the column maximum in the frame of row J on entry, of row J-1 on exit
    static __device__ __forceinline__ void run(const typename V::reg (&Hp)[S], typename V::reg Rv, typename V::reg& t, int& bj) {
        unsigned long long m;
        asm("v_cmp_eq_u16_e64 %[m], %[h], %[t]\n\t"
            "v_sub_u16_e32 %[t], %[t], %[rv]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[bj], %[bj], %[j], %[m]"
            : [t] "+v"(t), [bj] "+v"(bj), [m] "=&s"(m)
            : [h] "v"(Hp[J]), [rv] "v"(Rv), [j] "n"(J));
        if constexpr (J > 0) OvColumnRow<V, S, J - 1>::run(Hp, Rv, t, bj);
    }
};

// Words of a slot's column buffer (CKPT): two rows per word, then the lane number, padded to whole 16-byte stores.
// SW with end locations, slot-exact (CKPT): a lane's rows fall into up to four GROUPS of 4-row blocks, each with a column-maximum
// accumulator of its own (which accumulator a row block feeds costs nothing, and the group's first block starts it: ROWS_SWI).
// When a lane holds its slot's new best cell it leaves the rows of ONE group -- the first that holds the column maximum -- in
// the slot's LDS buffer: one word per row as the registers hold them, then {lane, first row of the group}.
__host__ __device__ constexpr int opal_swend_groups(int S) {
    // (fewer where four accumulators cost a class kernel a wave per SIMD: the tallest shapes, and 15 / 16 / 22 / 23 / 24 rows, which sit
    // just below a register boundary -- `make check-scratch` and the occupancy remarks of inst_s16swend.hip)
    return (S > 38 || S == 24) ? 1 : ((S > 36 || S == 15 || S == 16 || S == 22 || S == 23) ? 2 : ((S + 3) / 4 < 4 ? (S + 3) / 4 : 4));
}
__host__ __device__ constexpr int opal_swend_group_of_block(int S, int b) { return b * opal_swend_groups(S) / ((S + 3) / 4); }
__host__ __device__ constexpr int opal_swend_group_first_block(int S, int g) {  // (g == groups: one past the last block)
    int b = 0;
    while (b < (S + 3) / 4 && opal_swend_group_of_block(S, b) < g) b++;
    return b;
}
__host__ __device__ constexpr int opal_swend_group_rows_max(int S) {
    return 4 * (((S + 3) / 4 + opal_swend_groups(S) - 1) / opal_swend_groups(S));
}
__host__ __device__ constexpr int opal_swend_ckpt_words(int S) { return (opal_swend_group_rows_max(S) + 2 + 3) / 4 * 4; }

// Maximum of an unsigned value over the 2^log2G lanes of a slot, left in every lane of it (all lanes of the wave active).
// Lanes 1, 2, 4, 8 apart through DPP (quad permutes, then the two mirrors: once the quads agree, mirroring eight lanes
// pairs every quad with its neighbour, mirroring sixteen pairs the two halves); 16 and 32 apart through the LDS crossbar.
// (`s_nop 1`: the two wait states between a VALU write and a DPP read of the same register -- inside an asm statement the
// compiler does not add them.)
__device__ __forceinline__ unsigned slot_max_u32(unsigned x, const int log2G) {
    if (log2G >= 1) {
        asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
        if (log2G >= 2) {
            asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
            if (log2G >= 3) {
                asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
                if (log2G >= 4) {
                    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
                    if (log2G >= 5) {
                        // every lane of a 16-lane row holds its row's maximum now; across rows through SGPRs (four v_readlane and
                        // scalar maxima: no LDS round trip -- two ds_bpermute cost a 64-lane slot ~250 clocks of latency per settlement)
                        const unsigned r0 = __builtin_amdgcn_readlane(x, 0), r1 = __builtin_amdgcn_readlane(x, 16);
                        const unsigned r2 = __builtin_amdgcn_readlane(x, 32), r3 = __builtin_amdgcn_readlane(x, 48);
                        const unsigned lo = r0 > r1 ? r0 : r1, hi = r2 > r3 ? r2 : r3;
                        if (log2G >= 6) x = lo > hi ? lo : hi;
                        else x = (__lane_id() & 32u) ? hi : lo;
                    }
                }
            }
        }
    }
    return x;
}

// The winner's store (see opal_swend_groups): ds_write2_b32, two rows per store from whatever registers hold them (a 16-byte
// store wants four consecutive registers, i.e. one v_mov per row; packing two rows per word cost a half-rate v_perm per pair).
// Inline asm with a different text per group: identical stores in several branches would be sunk behind them, with a register
// copy per row to feed them. The LDS pipe is paid per instruction whatever the number of lanes writing, so the fewer rows the
// better: a quarter of the column, not all of it (round 4) or half of it.
template <class V, int S, int GRP>
__device__ __forceinline__ void swend_put_group(const unsigned ca, const typename V::reg* Hp, const int k) {
    constexpr int first = 4 * opal_swend_group_first_block(S, GRP);
    constexpr int end = 4 * opal_swend_group_first_block(S, GRP + 1) < S ? 4 * opal_swend_group_first_block(S, GRP + 1) : S;
    constexpr int rows = end - first, RM = opal_swend_group_rows_max(S);
    static_assert(rows >= 1 && rows <= RM, "group geometry");
#define OPAL_PUT2(a, b, o0, o1)                                                                                                       \
    do {                                                                                                                            \
        if constexpr (GRP == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4 ; group 0" :: "v"(ca), "v"(a), "v"(b), "n"(o0), "n"(o1) : "memory"); \
        else if constexpr (GRP == 1) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4 ; group 1" :: "v"(ca), "v"(a), "v"(b), "n"(o0), "n"(o1) : "memory"); \
        else if constexpr (GRP == 2) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4 ; group 2" :: "v"(ca), "v"(a), "v"(b), "n"(o0), "n"(o1) : "memory"); \
        else asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4 ; group 3" :: "v"(ca), "v"(a), "v"(b), "n"(o0), "n"(o1) : "memory"); \
    } while (0)
    static_for<rows / 2>([&](auto jc) {
        constexpr int j = 2 * decltype(jc)::value;
        OPAL_PUT2(Hp[first + j], Hp[first + j + 1], j, j + 1);
    });
    const unsigned kk = (unsigned)k, ff = (unsigned)first;
    if constexpr (rows & 1) {  // the odd row shares its store with the lane number
        OPAL_PUT2(Hp[first + rows - 1], ff, rows - 1, RM + 1);
        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(ca), "v"(kk), "n"(4 * RM) : "memory");
    } else {
        OPAL_PUT2(kk, ff, RM, RM + 1);
    }
#undef OPAL_PUT2
}
template <class V, int S>
__device__ __forceinline__ void swend_keep_column(const unsigned ca, const typename V::reg* Hp, const typename V::reg* aq, const typename V::reg cm, const int k) {
    constexpr int NQ = opal_swend_groups(S);
    if (NQ == 1 || V::same(aq[0], cm, 0)) swend_put_group<V, S, 0>(ca, Hp, k);
    else if (NQ == 2 || V::same(aq[NQ > 1 ? 1 : 0], cm, 0)) swend_put_group<V, S, (NQ > 1 ? 1 : 0)>(ca, Hp, k);
    else if (NQ == 3 || V::same(aq[NQ > 2 ? 2 : 0], cm, 0)) swend_put_group<V, S, (NQ > 2 ? 2 : 0)>(ca, Hp, k);
    else swend_put_group<V, S, (NQ > 3 ? 3 : 0)>(ca, Hp, k);
}
// ... and the look-up, once per sequence: the smallest row of the kept group that holds the score (a stale word behind a shorter
// group's rows cannot win: the group's own maximum sits below it), and the lane that kept it
template <int S>
__device__ __forceinline__ int swend_kept_row(const unsigned ca, const int sc, int& lane) {
    typedef __attribute__((address_space(3))) unsigned ckpt_word;
    constexpr int RM = opal_swend_group_rows_max(S);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the rows were stored by inline asm)
    const ckpt_word* ck = (const ckpt_word*)(uintptr_t)ca;
    const int first = (int)ck[RM + 1];
    lane = (int)ck[RM];
    int r = 0;
#pragma unroll
    for (int j = RM - 1; j >= 0; j--)
        if (first + j < S && (ck[j] & 0xffffu) == (unsigned)sc) r = first + j;
    return r;
}

// ------------------------------------------------------------------ the DP kernel
// (the second launch bound asks the register allocator for 4 waves per SIMD, i.e. at most 128 VGPRs, on the single-launch
// plain-16-bit kernels: OV at 38-39 rows per lane compiled to 129-131 registers without it, one wave per SIMD less)
template <int WIDTH, int MODE, int FLAGS, int S>
__device__ __forceinline__ void opal_dp_body(const KParams& p, const KClass cls, const int blockId, const int activeWaves = 0) {
    using V = Vec<WIDTH>;
    using reg = typename V::reg;
    using elem = typename V::elem;
    constexpr int NH = V::NH;
    constexpr int ES = (int)sizeof(elem);  // bytes per profile entry
    constexpr int RB = 16 / ES;            // rows fetched per LDS read (one ds_read_b128)
    constexpr int SPAD = (S + RB - 1) / RB * RB;
    constexpr int LANE_BYTES = opal_lane_bytes(ES, S, 1);  // one lane's segment of a profile row
    constexpr bool SWEND = (FLAGS & FLAG_SWEND) != 0;
    constexpr bool WRAP = (FLAGS & FLAG_WRAP) != 0;
    constexpr bool MULTI = (FLAGS & FLAG_MULTI) != 0;
    constexpr bool PLAIN = WIDTH == 17 || WIDTH == 18;  // one sequence per lane, plain 16-bit instructions
    constexpr bool SKEW = (FLAGS & FLAG_SKEW) != 0 && WIDTH == 17 && MODE != MODE_SW && !MULTI;
#ifndef OPAL_SWEND_CKPT
#define OPAL_SWEND_CKPT 1
#endif
    // SW with end location, single launch: the best cell of a SLOT is kept exactly, step by step (two DPP reductions over the
    // slot's lanes), and the column that holds it waits in LDS -- its rows are scanned once per sequence, not at every
    // improvement (see the column loop).
    constexpr bool CKPT = OPAL_SWEND_CKPT != 0 && MODE == MODE_SW && (FLAGS & FLAG_SWEND) != 0 && WIDTH == 17 && !MULTI && (FLAGS & FLAG_NOCKPT) == 0;
    // NW and OV read the first row block of the NEXT step's profile column at the top of the current step (measured on
    // the C3 workload: NW 1.183 -> 1.117 ms, OV 1.266 -> 1.248 ms; HW and SW gain nothing from it and the 4 extra live
    // registers push their 39/40-row shapes over 128, so they read it at the start of their own step; OV had it in round 2
    // and measures 1.1 % faster without since round 3)
    // Shapes with FEW rows per lane (S <= 12: the 32- and 64-lane classes of the longest sequences, whose lone waves are the
    // critical path of a search with a real length tail -- a 35 213-residue target at 2 rows per lane is 35 000 steps of
    // pure latency) do it in every mode: with so few rows the LDS round trip is most of a step.
#ifndef OPAL_PF_LATE_MASK
// bit MODE = that mode's tall shapes prefetch late. SW: C2 6 940 -> 7 010 GCUPS, SW + end 6 160 -> 6 280 (same box, interleaved:
// profiles/r04_ab_prefetch_split.txt). NW had it in rounds 4 and 5 (+3 % then); with the letter-independent LDS layout of round 6
// (a read is 7.8 LDS cycles at 8 lanes per sequence instead of 11.2) NW measures 1.6-2 % FASTER without, HW and OV 2-3 % slower with
// (profiles/r06_ab_prefetch_mask.txt): SW only.
#define OPAL_PF_LATE_MASK 8
#endif
#ifndef OPAL_SPLIT_FILL
#define OPAL_SPLIT_FILL 1    // NW/HW/OV run their first G-1 steps in a loop of their own (with the round-4 frame: HW +3 %, OV +4 %, NW +3 % with the late prefetch)
#endif
    // NW/HW/OV: a lane must not touch its boundary column before its first column (step k), so every step tests c >= 0 and
    // runs under an execution mask -- needed during the first G-1 steps of a sequence only. SPLIT: those steps get a loop of
    // their own (the same code with the test), the rest runs unmasked.
    constexpr bool SPLIT = SKEW && OPAL_SPLIT_FILL != 0;  // (the skewed single-launch kernels: the others have no register left for the second loop's state)
    constexpr bool PF_LATE_WANTED = WIDTH == 17 && S > 12 && ((OPAL_PF_LATE_MASK >> MODE) & 1) != 0 && (MODE == MODE_SW || SPLIT);  // (width 18 at 40 rows has no register left for it)
#ifndef OPAL_NW_PF_TOP
#define OPAL_NW_PF_TOP 0  // (round 3: NW prefetched at the top of the step; with SPLIT the late prefetch measures 3 % faster)
#endif
    constexpr bool PF_TOP = PLAIN && !PF_LATE_WANTED && ((MODE == MODE_NW && (OPAL_NW_PF_TOP != 0 || !SPLIT)) || S <= 12);  // (OV with it: 1.1 % slower since round 3's leaner last-column code)
    // PF_LATE: the same read, issued as late as it can be -- in place of the (absent) block prefetch of the LAST row block,
    // into the registers that prefetch would have used, behind the address update of the coming step: the four registers are
    // live across the step's fixed work only (hand-off, boundary, column test), not across the row blocks, and the first row
    // block of a step no longer waits out an LDS round trip.
    constexpr bool PF_LATE = PF_LATE_WANTED;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int log2G = cls.log2G;
    const int G = 1 << log2G;
    const int stride = opal_profile_stride(ES, G, S);
    const int Q = p.Q, A = p.A;
    // Query rows are spread over G*S >= Q lane registers. NW pads BELOW the query (pad rows never
    // feed a real row). SW/HW/OV pad ABOVE it: a virtual row with profile 0, H[r][-1] = 0 and
    // E = -Go reproduces the zero top boundary exactly (H = 0, F passed down = -Go), and puts query row
    // Q-1 in the last register of the last lane, where it is read without a register select.
    // With FLAG_MULTI the launch covers CR query rows starting at q0; only the first chunk (SW/HW/OV) or the
    // last one (NW) is partial, so the padding rule is unchanged.
    const int satLimit = PLAIN ? p.satLimit : V::SATMAX;  // scores >= this are re-run at 32 bits
    const int negPad = PLAIN ? p.negPad : V::NEG;         // profile value of pad residues / pad rows
    const int q0 = MULTI ? p.q0 : 0;
    const int CR = MULTI ? p.chunkRows : Q;
    const bool hasPrev = MULTI && (p.chunkFlags & 1), hasNext = MULTI && (p.chunkFlags & 2);
    const int padTop = (MODE == MODE_NW || hasPrev) ? 0 : (G * S - CR);
    unsigned char* const clkSlot = lds + (size_t)(A + 1) * stride;  // (the launch reserves 48 bytes behind the profile)
    clk_begin(p.clk, p.tl, clkSlot, p.resOut);

    // ---- stage the query profile in LDS: QP[letter][k][j] = M[query[k*S+j-padTop]][letter]; letter A = padding.
    // A thread owns (lane k, row j) positions and walks the letters: one query byte per position, then matrix reads
    // that do not depend on each other (8 in flight), instead of a dependent query->matrix load pair and two integer
    // divisions per ENTRY (a block spent ~30 us here; every block of a launch pays it before its first column).
    if constexpr (PLAIN) {
        const int perLetter = G * SPAD;
        for (int rem = threadIdx.x; rem < perLetter; rem += blockDim.x) {
            const int k = rem / SPAD;
            const int j = rem - k * SPAD;
            const int lr = k * S + j - padTop;  // row within this chunk
            const bool real = lr >= 0 && j < S && lr < CR;
            const int* mrow = p.matrix + (real ? (int)p.query[q0 + lr] * A : 0);
            // virtual rows above the query are transparent (0); pad rows (NW: below the query) and the pad letter: negPad
            const int fill = (lr < 0 ? 0 : negPad) + (SKEW ? 2 * p.Ge : 0);
            unsigned char* dst = lds + k * LANE_BYTES + j * ES;
            if (real) {
                const int sk = SKEW ? 2 * p.Ge : 0;  // skewed diagonal: H'(j,s) = H'(j-1,s-1) + P + 2*Ge
                for (int letter = 0; letter < A; letter += 2) {  // two independent loads in flight per position
                    const int l1 = min(letter + 1, A - 1);
                    const int v0 = mrow[letter], v1 = mrow[l1];
                    *reinterpret_cast<elem*>(dst + letter * stride) = (elem)(v0 + sk);
                    *reinterpret_cast<elem*>(dst + l1 * stride) = (elem)(v1 + sk);
                }
                *reinterpret_cast<elem*>(dst + A * stride) = (elem)(negPad + sk);
            } else {
                for (int letter = 0; letter <= A; letter++) *reinterpret_cast<elem*>(dst + letter * stride) = (elem)fill;
            }
        }
    } else {  // the packed and 32-bit kernels keep the entry-wise loop (their register budget is tighter)
        const int perLetter = G * SPAD;
        const int total = (A + 1) * perLetter;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int letter = idx / perLetter;
            const int rem = idx - letter * perLetter;
            const int k = rem / SPAD;
            const int j = rem - k * SPAD;
            const int lr = k * S + j - padTop;  // row within this chunk
            int v = negPad;     // pad residue, and (NW) rows below the query
            if (lr < 0) v = 0;  // virtual rows above the query: transparent
            else if (letter < A && j < S && lr < CR) v = p.matrix[(int)p.query[q0 + lr] * A + letter];
            *reinterpret_cast<elem*>(lds + letter * stride + k * LANE_BYTES + j * ES) = (elem)v;
        }
    }
    __syncthreads();

    // waves of the long-sequence classes take issue priority over co-resident waves of other launches
    if (cls.prio > 0) __builtin_amdgcn_s_setprio(2);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slots = 64 >> log2G;
    // (activeWaves: only the first few waves of a workgroup take work -- one per SIMD for the long sequences that run inside
    // a streaming launch, stream_kernel.hpp; 0 = all of them)
    const int wpbUsed = activeWaves ? activeWaves : (int)(blockDim.x >> 6);
    const int job = blockId * wpbUsed + wave;
    const int k = lane & (G - 1);
    const int slot = lane >> log2G;
    const int itemIdx = job * slots + slot;
    const bool valid = wave < wpbUsed && itemIdx < cls.numItems;
    // (idle slots and whole idle waves run the loop below with no columns of their own)

    int len[NH], seq[NH];
    uint64_t tokBase = 0;
    int half0 = 0, pairIdx = 0;
#pragma unroll
    for (int h = 0; h < NH; h++) { len[h] = 0; seq[h] = -1; }
    if (valid) {
        const int w = cls.firstItem + itemIdx;
        const int item = p.work ? p.work[w] : w;
        pairIdx = (NH == 2) ? item : (item >> 1);
        half0 = (NH == 2) ? 0 : (item & 1);
        tokBase = (NH == 2) ? p.pairOff[pairIdx] : p.seqOff[item];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            len[h] = p.pairLen[2 * pairIdx + half0 + h];
            seq[h] = p.pairSeq[2 * pairIdx + half0 + h];
            if (seq[h] < 0) len[h] = 0;
        }
    }
    int slotLen = len[0];
    if (NH == 2 && len[NH - 1] > slotLen) slotLen = len[NH - 1];
    const int maxLen = __builtin_amdgcn_readfirstlane(wave_max_i32(slotLen));
    // the first step at which some lane of the wave is on the LAST column of its sequence (lane 0 of the shortest one): before it
    // the last-column test of OV is a scalar compare, not a vector compare + ballot + branch every step (sorted sequences: the
    // slots of a wave differ by a residue or two, so that is all but the last G + a few steps)
    const int negFirstLast = __builtin_amdgcn_readfirstlane(wave_max_i32(slotLen > 0 ? 1 - slotLen : INT_MIN));
    const int firstLastStep = negFirstLast == INT_MIN ? INT_MAX : -negFirstLast;
    const int nsteps = maxLen > 0 ? maxLen + G - 1 : 0;

    // ---- per-lane DP state
    const int Go = p.Go, Ge = p.Ge;
    const reg Qv = V::splat(Go), Rv = V::splat(Ge);
    // the same as wave-uniform 16-bit patterns for the asm row blocks (SGPR operands)
    const unsigned Qs = (unsigned)Go & 0xffffu, Rs = (unsigned)Ge & 0xffffu, GQs = (unsigned)(Go - Ge) & 0xffffu;
    // SKEW: X' = X + (rho + c) * Ge with rho = k * S + j, the row's position among the G * S register rows of the slot (round 4;
    // before: (j + s), which differs from lane to lane by (S - 1) * Ge and cost two subtractions per step at the hand-off from
    // lane k-1 to lane k). The code below still writes its frame terms as (j + s) = (j + c + k) and adds kf = k * (S - 1) * Ge where
    // a value enters or leaves the frame -- outside the column loop only.
    const reg SR1v = V::splat((S - 1) * Ge);     // SKEW: from the frame of a lane's register 0 to that of its last register
    const int kf = SKEW ? k * (S - 1) * Ge : 0;
    // (inside the column loop kf is needed in rare blocks only -- last columns, final cells: made again there instead of kept in a register)
    auto kf_now = [&]() -> int {
        int kk = k;
        asm volatile("" : "+v"(kk));
        return SKEW ? kk * (S - 1) * Ge : 0;
    };
    const int r0 = q0 + k * S - padTop;  // query row held in register 0 (negative: virtual rows)
    reg Hp[S], En[S];               // H[r][c-1] and E[r][c] for my rows
    // (NW, HW: the boundary value runs down the rows, one 64-bit value alive -- computed per row from scratch, the compiler
    // interleaved the 64-bit clamps of all rows: 128 registers and two to four spills in this prologue at 28-30 rows)
    long long hbRun = -(long long)Go - (long long)r0 * Ge;
#pragma unroll
    for (int j = 0; j < S; j++, hbRun -= Ge) {
        if (MODE == MODE_SW) { Hp[j] = 0; En[j] = 0; }
        else if (MODE == MODE_OV || r0 + j < 0) { Hp[j] = 0; En[j] = V::splat(V::clampv(-(long long)Go)); }
        else {  // NW, HW: H[r][-1] = -Go - r*Ge (opal.cpp:671-679), E[r][0] = H[r][-1] - Go
            const long long hb = hbRun;
            Hp[j] = V::splat(V::clampv(hb));
            En[j] = V::splat(V::clampv(hb - Go));
        }
        if (SKEW) {  // my first column runs at step k: H(j) is read as the diagonal of step k, E(j) at step k
            Hp[j] = V::adds(Hp[j], V::splat((j + k - 1) * Ge + kf));
            En[j] = V::adds(En[j], V::splat((j + k) * Ge + kf));
        }
        if constexpr ((MODE == MODE_NW || MODE == MODE_HW) && PLAIN) {
            // (row j finished before row j+1's boundary value exists: otherwise all S 64-bit clamps are in flight at once)
            unsigned lo = (unsigned)hbRun, hi = (unsigned)((unsigned long long)hbRun >> 32);
            asm volatile("" : "+v"(lo), "+v"(hi), "+v"(Hp[j]), "+v"(En[j]));
            hbRun = (long long)(((unsigned long long)hi << 32) | lo);
        }
    }
    // what the lane below reads: H and next-row F of my last row. Before my first column it must be
    // H[last row][-1] (the diagonal input of the lane below at its first column).
    reg outH = 0, outF = 0;
    if ((MODE == MODE_NW || MODE == MODE_HW) && r0 + S - 1 >= 0)
        outH = V::splat(V::clampv(-(long long)Go - (long long)(r0 + S - 1) * Ge));
    if (SKEW) outH = V::adds(outH, V::splat((S - 1 + k - 1) * Ge + kf));  // as if produced at the end of step k-1
    reg diagIn = SKEW ? V::splat(-2 * Ge) : 0;  // H[r0-1][c-1]; H[-1][-1] = 0 in every mode (frame (-1)+(-1) when skewed)
    if (hasPrev && (MODE == MODE_NW || MODE == MODE_HW))  // H[q0-1][-1] for the first lane of a lower chunk
        diagIn = V::splat(V::clampv(-(long long)Go - (long long)(q0 - 1) * Ge));
    const unsigned ldsBase = (unsigned)(uintptr_t)(opal_lds_u8*)lds;
    const unsigned padOff = ldsBase + (unsigned)(A * stride);  // profile offsets are absolute LDS addresses
    unsigned offv[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) offv[h] = padOff + (unsigned)(k * LANE_BYTES);  // pad columns until real ones arrive

    // results
    reg mx = 0;                 // SW: running maximum
    int best[NH], bestCol[NH];  // HW/OV: last-row maximum and its first column; NW: final cell
    int prevBest[NH];           // OV: last-row maximum over columns 0..L-2
    int lcMax[NH], lcRow[NH];   // OV: maximum of the last column over my rows, smallest row
    int endCol[NH], endRow[NH]; // SW second pass
    int wrapped = 0;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        best[h] = INT_MIN; bestCol[h] = -1; prevBest[h] = INT_MIN; lcMax[h] = (SKEW && MODE == MODE_OV) ? 0 : INT_MIN; lcRow[h] = -1;  // (skewed OV: lcMax is a key, see the loop)
        endCol[h] = INT_MAX; endRow[h] = -1;
    }
    // where query row Q-1 lives: NW: lane kLast, register jLast; otherwise last register of lane G-1
    const int kLast = (MODE == MODE_NW) ? (CR - 1) / S : G - 1;
    const int jLast = (MODE == MODE_NW) ? (CR - 1) - kLast * S : S - 1;
    // my registers [jlo, jhi) hold real query rows
    int jlo = padTop - k * S, jhi = CR + padTop - k * S;
    jlo = jlo < 0 ? 0 : (jlo > S ? S : jlo);
    jhi = jhi < 0 ? 0 : (jhi > S ? S : jhi);

    // Residue tokens -> LDS profile offsets, G columns at a time: at the start of every batch of G steps lane k of a
    // slot decodes the token of column s+k (letter * stride, or the padding letter beyond the sequence); the queue
    // then moves one lane towards lane 0 per step (one wave_shl DPP move), so lane 0 of a slot always holds the
    // offset of the column it starts. (Lane G-1 receives the next slot's lane 0: that value would reach my lane 0
    // only after G steps, when the queue has been refilled.) The tokens of the next batch are loaded a batch ahead.
    unsigned tokq[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) tokq[h] = padOff;
    auto token = [&](int col) -> unsigned { return NH == 2 ? (unsigned)p.tokens[tokBase + (uint64_t)col] : (unsigned)p.res[tokBase + (uint64_t)col]; };
    unsigned tokNext = 0;
    if (k < slotLen) tokNext = token(k);
    // chunk boundary prefetch (one step ahead): {H, F-input} of the row above my chunk at the next column
    auto bndIdx = [&](int c) -> uint64_t { return tokBase + (uint64_t)c; };
    uint2 bndNext = make_uint2(0u, 0u);
    if (hasPrev && 0 < slotLen) bndNext = p.bnd[bndIdx(0)];

    // The profile address of a step is formed ONE STEP AHEAD (it depends on tokens only, not on DP values), and the
    // PF_TOP kernels also issue the LDS read of the next step's first row block then: such a wave no longer begins
    // every step by waiting out an LDS round trip (DPP shift -> address -> ds_read -> first row).
    auto advance = [&](const int sn) {  // offv <- profile address of step sn (my column sn - k)
        if ((sn & (G - 1)) == 0) {  // wave-uniform: refill the queue with columns sn .. sn+G-1
            const unsigned tok = tokNext;
            const int col = sn + k;
            if constexpr (NH == 1 && (FLAGS & FLAG_ZC) != 0) {
                // Zero-copy search (KParams::resOut): the kernel is the copy engine. The pointer is read from the LDS slot
                // every batch (volatile: not hoisted) -- as a kernel argument kept across the loop it costs the tall
                // shapes two registers they do not have.
                unsigned char* const ro = *reinterpret_cast<unsigned char* volatile*>(clkSlot + 32);
                if (ro && col < slotLen) ro[tokBase + (uint64_t)col] = (unsigned char)tok;
            }
            if (col + G < slotLen) tokNext = token(col + G);
#pragma unroll
            for (int h = 0; h < NH; h++) {
                unsigned mine = tok;
                if (NH == 2) mine = (h == 0) ? (tok & 0xffu) : ((tok >> 8) & 0xffu);
#ifdef OPAL_ABL_SAME_LETTER  // (measurement only, results wrong: every lane reads letter 0 -- no LDS bank conflicts between letters)
                mine &= 0u;
#endif
                tokq[h] = (col < len[h]) ? ldsBase + mine * (unsigned)stride : padOff;
            }
        }
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const unsigned shifted = shr1(offv[h]) + (unsigned)LANE_BYTES;
            offv[h] = (k == 0) ? tokq[h] : shifted;
            tokq[h] = shl1(tokq[h]);
        }
    };
    int4 pfirst = make_int4(0, 0, 0, 0);  // PF_TOP: first row block of the profile column of the coming step
    if constexpr (PF_TOP || (PF_LATE && !SPLIT)) { advance(0); pfirst = lds_ld16(offv[0]); }

    reg bestF = V::splat(-32768);            // SKEW, HW/OV: last-row maximum so far, in the skewed frame of the step at hand
    int slotMax = 0;                         // SW end location: slot-wide running maximum (see the scan filter)
    // CKPT: the slot's best cell so far (the same in every lane of the slot) and its column buffer behind the profile and the
    // probe slot: two rows per word + the lane number, a multiple of four words per slot (opal_shim.hip adds them to the
    // launch's LDS request)
    int bestV = 0, bestC = INT_MIN;  // (INT_MIN: a tie at score 0 replaces nothing)
    // (slot index = thread index >> log2G. The 14-, 17- and 23- to 25-row shapes make the address where it is used: as a lane value kept
    // across the column loop it cost them their fifth wave per SIMD; everywhere else it stays in its register -- with cheap gaps
    // it is used in every step)
    auto ckpt_addr = [&]() -> unsigned {
        unsigned t = threadIdx.x;
        if constexpr ((S >= 23 && S <= 25) || S == 14 || S == 17) asm volatile("" : "+v"(t));
        return (unsigned)(uintptr_t)(opal_lds_u8*)(clkSlot + 48) + (t >> log2G) * (unsigned)(opal_swend_ckpt_words(S) * 4);
    };
    int topHv = V::clampv(-(long long)Go);  // NW: H[-1][0]
    reg sRv = 0;                             // SKEW: s*Ge as a 16-bit value
    // One column step. MASKED: lanes before their first column sit the step out (NW/HW/OV; SW needs no test: its state is
    // all zeros until real columns arrive).
    auto column_step = [&](const int s, auto maskedc) {
        constexpr bool MASKED = decltype(maskedc)::value;
        constexpr bool LATE = PF_LATE && !MASKED;
        if constexpr (!PF_TOP && !LATE) advance(s);
        unsigned offc[NH];  // this step's profile address
#pragma unroll
        for (int h = 0; h < NH; h++) offc[h] = offv[h];
        const int4 pcur = pfirst;
        if constexpr (PF_TOP) { advance(s + 1); pfirst = lds_ld16(offv[0]); }
        if (MODE == MODE_SW && SWEND && !CKPT && NH == 1 && (s & 7) == 0) {
            int v = V::half(mx, 0);
            for (int d = 1; d < G; d <<= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
            slotMax = v;
        }
        const uint2 bndCur = bndNext;
        if (hasPrev && s + 1 < slotLen) bndNext = p.bnd[bndIdx(s + 1)];

        // ---- inputs of this column: from the boundary (lane 0 of the slot) or from the lane above
        reg topH = 0;
        if (MODE == MODE_NW) {  // H[-1][c] = -Go - c*Ge (opal.cpp:686-689), kept incrementally and clamped to the type
            topH = V::splat(topHv);
            const int nxt = topHv - Ge;
            topHv = (nxt < V::clampv(LLONG_MIN / 2) || nxt > topHv) ? V::clampv(LLONG_MIN / 2) : nxt;
        }
        reg inH = (reg)shr1((unsigned)outH);
        reg inF = (reg)shr1((unsigned)outF);
        if constexpr (SKEW) {
            // (lane k-1's outputs are in lane k's frame already: rows rho - 1 and rho, same column)
            if (k == 0) {
                // top boundary in my frame: H[-1][c] + (s-1)*Ge (read as a diagonal one step later) and
                // F[0][c] + s*Ge = H[-1][c] - Go + s*Ge; NW: H[-1][c] = -Go - s*Ge, HW/OV: 0
                if (MODE == MODE_NW) { inH = V::splat(-Go - Ge); inF = V::splat(-2 * Go); }
                else { inH = Vec<17>::subs_s(sRv, Rs); inF = Vec<17>::subs_s(sRv, Qs); }
            }
        } else if (MODE == MODE_SW && !MULTI) {
            // top boundary of local alignment: H = 0 and F input = 0 for lane 0 of every slot (two selects, no branch)
            inH = (k == 0) ? (reg)0 : inH;
            inF = (k == 0) ? (reg)0 : inF;
        } else if (k == 0) {
            if (hasPrev) {
                inH = (reg)bndCur.x;
                inF = (reg)bndCur.y;
            } else {
                inH = topH;
                inF = (MODE == MODE_SW) ? V::subu(topH, Qv) : V::subs(topH, Qv);
            }
        }
        const int c = s - k;  // my column

        if (MODE == MODE_SW || !MASKED || c >= 0) {
            reg diag = diagIn;
            reg f = inF;
            reg h = 0;
            reg cm = 0;  // SW with end location: this column's maximum over my rows
            // ... kept per HALF of the rows during the row blocks of the plain kernels (which accumulator a row block uses
            // costs nothing), so that a row scan only visits the first half that holds the column maximum
            constexpr int NBLK4H = (S + 3) / 4;
            reg ch[2] = {0, 0};
            constexpr int NQ = opal_swend_groups(S);
            reg aq[NQ];  // CKPT: the column maximum per group of row blocks (opal_swend_groups)
            if constexpr (NQ == 1) aq[0] = 0;  // (one group: zeroed and accumulated as in round 4 -- one register less at 24 rows)
            if constexpr (PLAIN) {
                // Plain 16-bit path: one asm statement per block of 4 rows (one ds_read_b128 of profile entries), the
                // profile read of the next block issued before the current block's arithmetic. Inside a block the
                // diagonal add of row j+1 comes before row j's recurrence, so H[j][c-1]'s register is free for H[j][c].
                constexpr int KIND = MODE == MODE_SW ? ROWS_SW : (SKEW ? ROWS_SKEW : ROWS_GLOBAL);
                constexpr int NBLK = SPAD / RB;   // LDS reads per column (RB rows each: 4 with 4-byte entries, 8 with 2-byte ones)
                int4 pc = pcur;
                if constexpr (!PF_TOP && !LATE) pc = lds_ld16(offc[0]);
                reg d = V::adds(diag, (reg)pc.x), t = 0;  // (a 16-bit add reads the low half only)
                // (two row blocks of profile entries in flight instead of one were tried for the 6-instruction kernels in
                // round 3: NW unchanged, HW 2.8 % slower)
                static_for<NBLK>([&](auto rbc) {
                    constexpr int rb = decltype(rbc)::value;
                    int4 pn = pc;
                    if constexpr (rb + 1 < NBLK) pn = lds_ld16(offc[0] + (rb + 1) * RB * ES);
                    else if constexpr (LATE) { advance(s + 1); pn = lds_ld16(offv[0]); }  // every read of this step has been issued
                    reg Pv[RB + 1];  // profile values of this block's rows, then of the next block's first row
                    if constexpr (ES == 4) {
                        Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.y; Pv[2] = (reg)pc.z; Pv[3] = (reg)pc.w;
                    } else {
                        Pv[0] = (reg)pc.x; Pv[1] = (reg)pc.x >> 16; Pv[2] = (reg)pc.y; Pv[3] = (reg)pc.y >> 16;
                        Pv[4] = (reg)pc.z; Pv[5] = (reg)pc.z >> 16; Pv[6] = (reg)pc.w; Pv[7] = (reg)pc.w >> 16;
                    }
                    Pv[RB] = (reg)pn.x;
                    static_for<RB / 4>([&](auto subc) {
                        constexpr int j0 = rb * RB + decltype(subc)::value * 4;  // first row of this asm block
                        if constexpr (j0 < S) {
                            constexpr int n = (S - j0) < 4 ? (S - j0) : 4;
                            constexpr bool last = j0 + n == S;
                            const reg* Pn = &Pv[j0 - rb * RB + 1];  // profile of the row BELOW each row
                            constexpr int grp = opal_swend_group_of_block(S, j0 / 4);
                            constexpr bool starts = CKPT && NQ > 1 && j0 / 4 == opal_swend_group_first_block(S, grp);
                            reg& acc = CKPT ? aq[grp] : (SWEND ? ch[(j0 / 4) * 2 / NBLK4H] : mx);
                            V::template rows<starts ? ROWS_SWI : KIND, n, last>(&Hp[j0], &En[j0], Pn, f, acc, d, t, SKEW ? GQs : Qs, Rs);
                        }
                    });
                    pc = pn;
                });
                if constexpr (LATE) pfirst = pc;
                h = Hp[S - 1];
                if constexpr (CKPT) {
                    cm = aq[0];
                    if constexpr (NQ == 2) cm = V::maxs(aq[0], aq[1]);
                    if constexpr (NQ == 3) cm = V::maxs(V::maxs(aq[0], aq[1]), aq[2]);
                    if constexpr (NQ == 4) cm = V::maxs(V::maxs(aq[0], aq[1]), V::maxs(aq[2], aq[NQ - 1]));
                } else if constexpr (SWEND) cm = V::maxs(ch[0], ch[1]);
            } else {
#pragma unroll
            for (int rb = 0; rb < SPAD / RB; rb++) {
                reg P[RB];
                if constexpr (WIDTH == 16) {
                    const int4 ai = lds_ld16(offc[0] + rb * RB * ES), bi = lds_ld16(offc[1] + rb * RB * ES);
                    const uint4 a = make_uint4((unsigned)ai.x, (unsigned)ai.y, (unsigned)ai.z, (unsigned)ai.w);
                    const uint4 b = make_uint4((unsigned)bi.x, (unsigned)bi.y, (unsigned)bi.z, (unsigned)bi.w);
                    P[0] = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u);
                    P[1] = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
                    P[2] = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u);
                    P[3] = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
                    P[4] = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u);
                    P[5] = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
                    P[6] = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u);
                    P[7] = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
                } else {
                    const int4 a = lds_ld16(offc[0] + rb * RB * ES);
                    P[0] = a.x; P[1] = a.y; P[2] = a.z; P[3] = a.w;
                }
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int j = rb * RB + i;
                    if (j < S) {
                        // E = max(H<- - Go, E<- - Ge) was formed one column ago; F likewise one row ago
                        // (reference core loop opal.cpp:280-327 / :748-771)
                        const reg e = En[j];
                        const reg d = V::adds(diag, P[i]);
                        if (WRAP) wrapped |= (d <= INT_MIN / 2);
                        h = V::maxs(V::maxs(d, e), f);
                        reg hq;
                        if (MODE == MODE_SW) {
                            // E, F are kept >= 0 by unsigned saturation, so H >= 0 needs no extra max
                            if (SWEND) cm = V::maxs(cm, h); else mx = V::maxs(mx, h);
                            hq = V::subu(h, Qv);
                            En[j] = V::maxs(V::subu(e, Rv), hq);
                            f = V::maxs(V::subu(f, Rv), hq);
                        } else {
                            hq = V::subs(h, Qv);
                            En[j] = V::maxs(V::subs(e, Rv), hq);
                            f = V::maxs(V::subs(f, Rv), hq);
                        }
                        diag = Hp[j];
                        Hp[j] = h;
                    }
                }
            }
            }
            outH = h;
            outF = f;
            if (hasNext && k == G - 1 && c >= 0 && c < slotLen) p.bnd[bndIdx(c)] = make_uint2((unsigned)outH, (unsigned)outF);

            if constexpr (CKPT) {
                // End location in the same pass, exactly and per SLOT. The cell wanted is the one with the highest score, then
                // the smallest column, then the smallest row (opal.h:44-45); bestV / bestC hold the slot's best score so far
                // and its column, the same in every lane of the slot. A lane's column maximum matters only if it beats
                // that -- a higher value, or the same value in a smaller column (the lanes of a slot sit on different
                // columns, lane k on s - k; the same column comes to larger lanes = larger rows later, which never wins).
                // Only when some lane of the WAVE has such a column do the slots settle who holds their best: one
                // all-reduce of (value, lane) keys -- among the lanes of a slot that hold the step's highest value the
                // LARGEST lane has the smallest column -- and the winner writes its column (H of its S rows) and its lane
                // number to the slot's LDS buffer. The smallest row holding the score is looked up there once, after the
                // last column. Before: a row scan, 2 instructions per row, for every improvement of any lane's own maximum
                // -- with cheap gaps (scores grow along the whole sequence) in every step.
                const unsigned cv = (unsigned)cm & 0xffffu;  // (SW values are >= 0)
                const bool cand = OPAL_ABL >= 3 ? false : ((int)cv > bestV || ((int)cv == bestV && c < bestC));
                if (__builtin_amdgcn_ballot_w64(cand) && (OPAL_ABL < 2 || p.satLimit < 0)) {
                    const unsigned key = slot_max_u32(cand ? ((cv << 6) | (unsigned)k) + 1u : 0u, log2G);
                    if (key) {  // (the same in every lane of a slot)
                        const unsigned kw = (key - 1u) & 63u;
                        bestV = (int)((key - 1u) >> 6); bestC = s - (int)kw;
                        if ((unsigned)k == kw && (OPAL_ABL < 1 || p.satLimit < 0)) {
                            swend_keep_column<V, S>(ckpt_addr(), Hp, aq, cm, k);
                        }
                    }
                }
            } else if (MODE == MODE_SW && SWEND) {
                // End location in the same pass. cm = this column's maximum over my rows. Only a STRICT
                // improvement of my running maximum can be the first cell holding the final score
                // (smallest column first, opal.h:44-45); the row scan below runs for those columns only.
                // (Some lane of a wave improves in most columns, so in practice the scan is paid per step: measured
                // +2.6 instructions per cell, the same relative price the reference pays for end locations. A scan
                // restricted to 8-row blocks was tried in round 2 and was no faster: lanes improve in different blocks.)
                const reg nm = V::maxs(mx, cm);
                auto scan = [&](int hh) {
                    int r = 0;
                    if constexpr (PLAIN) {
                        // the first half whose maximum is the column's, then the smallest row in it
                        constexpr int NBLK4 = (S + 3) / 4;
                        const int qsel = V::same(ch[0], cm, 0) ? 0 : 1;
                        static_for<2>([&](auto qc) {
                            constexpr int qi = decltype(qc)::value;
                            if (qsel == qi) {
                                static_for<NBLK4>([&](auto bc) {  // highest block first: the smallest row holding it wins
                                    constexpr int b = NBLK4 - 1 - decltype(bc)::value;
                                    if constexpr (b * 2 / NBLK4 == qi) {
                                        constexpr int n = (S - 4 * b) < 4 ? (S - 4 * b) : 4;
                                        V::template scan_rows<n, 4 * b>(&Hp[4 * b], cm, r);
                                    }
                                });
                            }
                        });
                    } else {
#pragma unroll
                        for (int j = S - 1; j >= 0; j--)
                            if (V::same(Hp[j], cm, hh)) r = j;  // smallest row of the column holding it
                    }
                    endRow[hh] = r0 + r;
                    endCol[hh] = c;
                };
                if constexpr (NH == 1) {
                    // One sequence per lane: a strict improvement is worth scanning only if it reaches the SLOT-wide
                    // running maximum (slotMax, refreshed every 8 steps from all lanes of the slot; a stale = smaller
                    // value only lets more scans through). A value below it can never be the final score, and the cell
                    // that holds the final score passes the filter whenever it occurs (ties included: the final
                    // reduction breaks them by column, then lane). This cuts the scans from most steps to about a third.
                    const int cv = V::half(cm, 0);
                    if (cv > V::half(mx, 0) && cv >= slotMax) scan(0);
                    mx = nm;
                } else {
                    bool improved = false;
#pragma unroll
                    for (int hh = 0; hh < NH; hh++) improved |= V::half(nm, hh) != V::half(mx, hh);
                    if (improved) {
#pragma unroll
                        for (int hh = 0; hh < NH; hh++)
                            if (V::half(cm, hh) > V::half(mx, hh)) scan(hh);
                        mx = nm;
                    }
                }
            }
            if (MODE == MODE_NW) {
                // final cell: once per sequence, on the lane/register holding query row Q-1
                bool fin = false;
#pragma unroll
                for (int hh = 0; hh < NH; hh++) fin |= (k == kLast && c == len[hh] - 1 && !hasNext);
                if (fin) {
                    const reg hl = SelectRow<reg, S, 0, S>::get(Hp, jLast);
#pragma unroll
                    for (int hh = 0; hh < NH; hh++)
                        if (c == len[hh] - 1) best[hh] = V::half(SKEW ? V::subs(hl, V::splat((jLast + s) * Ge + kf_now())) : hl, hh);
                }
            }
            if ((MODE == MODE_HW || MODE == MODE_OV) && !SKEW && k == kLast && !hasNext) {
                // last query row = last register of the last lane = this column's outH
#pragma unroll
                for (int hh = 0; hh < NH; hh++) {
                    const int v = V::half(SKEW ? V::subs(outH, V::adds(sRv, SR1v)) : outH, hh);
                    if (c < len[hh]) {
                        if (c == len[hh] - 1) prevBest[hh] = best[hh];
                        if (v > best[hh]) { best[hh] = v; bestCol[hh] = c; }  // strict >: first column wins (opal.cpp:824-834)
                    }
                }
            }
            if (MODE == MODE_OV) {
                bool lastc = false;
#pragma unroll
                for (int hh = 0; hh < NH; hh++) lastc |= (c == len[hh] - 1);
                if constexpr (!SKEW) {
                if (lastc) {
#pragma unroll
                    for (int hh = 0; hh < NH; hh++) {
                        if (c == len[hh] - 1) {
                            int bv = lcMax[hh], bj = -1;
#pragma unroll
                            for (int j = S - 1; j >= 0; j--) {
                                const int v = V::half(Hp[j], hh);
                                if (j >= jlo && j < jhi && v >= bv) { bv = v; bj = j; }
                                asm volatile("" : "+v"(bv), "+v"(bj));  // keep the scan a short serial chain (register pressure)
                            }
                            if (bj >= 0) { lcMax[hh] = bv; lcRow[hh] = r0 + bj; }
                        }
                    }
                }
                }
            }
        }
        if constexpr ((MODE == MODE_HW || MODE == MODE_OV) && SKEW) {
            // Last-row maximum and its first column (last query row = last register of the last lane = this column's outH),
            // tracked by EVERY lane in the skewed frame with full-rate 16-bit instructions -- no branch, no unskewing per
            // step: the running best moves into the next step's frame with one addition (only the lane holding row Q-1 is
            // read at the end). HW: over all columns; OV: over columns 0 .. L-2 -- the last column is covered by the
            // last-column maximum below, which includes row Q-1. (Columns before my first one and beyond the sequence fail
            // the one unsigned comparison.)
            const unsigned limit = (unsigned)(MODE == MODE_OV ? len[0] - 1 : len[0]);
            Vec<17>::track_best(outH, bestF, bestCol[0], c, limit, Rs);
        }
        if constexpr (MODE == MODE_OV && SKEW) {
            // The last column of a sequence (single launch, one sequence per lane): lane k reaches it one step after lane
            // k-1, so the maximum over the rows above travels down the lanes with it (one DPP move per such step, all lanes
            // active: the branch is wave-uniform) as ONE key: (value + 32768) << 16 | (65535 - row) -- the larger key is the
            // larger value, then the smaller row. The lane holding query row Q-1 ends up with the column's maximum and the
            // smallest row attaining it.
            const bool lastc = c >= 0 && c == len[0] - 1;
            if (s >= firstLastStep && __ballot(lastc) != 0ull) {  // (marking this and the other rare blocks unlikely, so that they are laid out behind the loop, measured no faster)
                unsigned inKey = shr1((unsigned)lcMax[0]);
                if (k == 0) inKey = 0u;
                if (lastc) {
                    unsigned myKey = 0u;
                    if (jlo == 0 && jhi == S) {  // all my rows are query rows: the value first, in the skewed frame
                        reg bf = Hp[S - 1];
                        if constexpr (S > 1) OvColumnMax<V, S, S - 2>::run(Hp, Rv, bf);
                        const unsigned bv = (unsigned)(V::half(V::subs(bf, V::splat(s * Ge + kf_now())), 0) + 32768);
                        if (bv > (inKey >> 16)) {  // (rare) ... then the smallest row holding it
                            reg t = V::adds(bf, SR1v);
                            int bj = 0;
                            OvColumnRow<V, S, S - 1>::run(Hp, Rv, t, bj);
                            myKey = (bv << 16) | (unsigned)(65535 - (r0 + bj));
                        }
                    } else {
                        int bv = INT_MIN, bj = -1;
                        const int kfl = kf_now();
#pragma unroll
                        for (int j = S - 1; j >= 0; j--) {
                            const int v = V::half(V::subs(Hp[j], V::splat((j + s) * Ge + kfl)), 0);
                            if (j >= jlo && j < jhi && v >= bv) { bv = v; bj = j; }
                            asm volatile("" : "+v"(bv), "+v"(bj));
                        }
                        if (bj >= 0) myKey = ((unsigned)(bv + 32768) << 16) | (unsigned)(65535 - (r0 + bj));
                    }
                    lcMax[0] = (int)(myKey > inKey ? myKey : inKey);
                }
            }
        }
        diagIn = inH;
        if constexpr (SKEW) sRv = Vec<17>::adds_s(sRv, Rs);
    };
    if constexpr (SPLIT) {
        const int nfill = nsteps < G - 1 ? nsteps : G - 1;
        for (int s = 0; s < nfill; s++) column_step(s, std::true_type{});
        if (nfill < nsteps) {
            if constexpr (PF_LATE) { advance(nfill); pfirst = lds_ld16(offv[0]); }
            for (int s = nfill; s < nsteps; s++) column_step(s, std::false_type{});
        }
    } else {
        for (int s = 0; s < nsteps; s++) column_step(s, std::integral_constant<bool, MODE != MODE_SW>{});
    }

    // ---- results
    if (MODE == MODE_SW) {
        if (WRAP) {
            for (int d = 1; d < G; d <<= 1) wrapped |= __shfl_xor(wrapped, d, 64);
        }
        // Every lane evaluates its slot's result (lane 0 of the slot is the writer). Saturated scores are
        // collected with one __ballot per wave: the wave reserves popcount(ballot) entries of the re-run
        // list with a single atomic, each flagged lane takes the entry given by its rank in the ballot.
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            // best cell over the slot's lanes: highest score, then smallest column, then smallest lane (= rows)
            int sc = V::half(mx, hh), col = endCol[hh], row = endRow[hh], src = k;
            if constexpr (CKPT) {  // the slot's best cell is known in every lane; its row: the smallest one of the kept column holding the score
                sc = bestV; col = bestV > 0 ? bestC : INT_MAX;
                int kw = 0;
                const int r = swend_kept_row<S>(ckpt_addr(), sc, kw);
                row = q0 + kw * S - padTop + r;
            } else
            for (int d = 1; d < G; d <<= 1) {
                const int osc = __shfl_xor(sc, d, 64);
                if (SWEND) {
                    const int ocol = __shfl_xor(col, d, 64), orow = __shfl_xor(row, d, 64), osrc = __shfl_xor(src, d, 64);
                    if (osc > sc || (osc == sc && (ocol < col || (ocol == col && osrc < src)))) { sc = osc; col = ocol; row = orow; src = osrc; }
                } else {
                    sc = osc > sc ? osc : sc;
                }
            }
            const bool writer = (k == 0) && seq[hh] >= 0;
            int prev = 0;
            if (writer && hasPrev) {  // chunks above hold smaller rows: they keep the cell unless ours is better
                prev = p.outScore[seq[hh]];
                bool keepPrev = prev > sc;
                if (SWEND && prev == sc) {
                    const int pc = p.outEndT[seq[hh]];
                    keepPrev = (pc < 0 ? INT_MAX : pc) <= col;
                }
                if (!SWEND && prev == sc) keepPrev = true;
                if (keepPrev) { sc = prev; if (SWEND) { col = p.outEndT[seq[hh]]; row = p.outEndQ[seq[hh]]; if (col < 0) col = INT_MAX; } }
            }
            const bool sat = writer && WIDTH != 32 && sc >= satLimit;             // (opal.cpp:364-370): hand over to width 32
            const bool append = sat && !(hasPrev && prev >= satLimit);            // not yet listed by an upper chunk
            const unsigned long long ballot = __ballot(append);
            if (ballot) {
                int base = 0;
                if (lane == __builtin_ctzll(ballot)) base = atomicAdd(p.ovfCount, (int)__builtin_popcountll(ballot));
                base = __shfl(base, __builtin_ctzll(ballot), 64);
                if (append) p.ovfItems[base + (int)__builtin_popcountll(ballot & ((1ull << lane) - 1ull))] = pairIdx * 2 + half0 + hh;
            }
            if (sat) {
                if (MULTI) p.outScore[seq[hh]] = satLimit;  // marker for the chunks below
            } else if (writer && WIDTH == 32 && WRAP && wrapped) {
                atomicAdd(p.ovfCount, 1);  // 32-bit overflow: the host returns OPAL_ERR_OVERFLOW
            } else if (writer) {
                p.outScore[seq[hh]] = sc;
                if (p.outEndQ) {
                    const bool none = !SWEND || sc <= 0 || col == INT_MAX;  // score 0: no cell (test.cpp:242-243)
                    p.outEndQ[seq[hh]] = none ? -1 : row;
                    p.outEndT[seq[hh]] = none ? -1 : col;
                }
            }
        }
    }
    if (MODE != MODE_SW) {
        if constexpr ((MODE == MODE_HW || MODE == MODE_OV) && SKEW) {  // out of the frame of the step after the last one
            best[0] = bestCol[0] >= 0 ? V::half(V::subs(bestF, V::adds(V::adds(sRv, SR1v), V::splat(kf_now()))), 0) : INT_MIN;
            prevBest[0] = best[0];  // (OV: the maximum over columns 0 .. L-2, see the loop)
        }
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            int sc = best[hh], eq = Q - 1, et = (MODE == MODE_NW) ? len[hh] - 1 : bestCol[hh];
            if (MODE == MODE_OV) {
                // last-column maximum over all lanes of the slot (value, then smallest row) (opal.cpp:881-904)
                int v = lcMax[hh], r = lcRow[hh];
                if constexpr (SKEW) {  // the lane holding row Q-1 has the whole column's key (see the loop)
                    const unsigned key = (unsigned)lcMax[hh];
                    v = key ? (int)(key >> 16) - 32768 : INT_MIN;
                    r = key ? 65535 - (int)(key & 0xffffu) : -1;
                } else
                for (int d = 1; d < G; d <<= 1) {
                    const int ov = __shfl_xor(v, d, 64), orow = __shfl_xor(r, d, 64);
                    if (ov > v || (ov == v && orow >= 0 && (r < 0 || orow < r))) { v = ov; r = orow; }
                }
                if (MULTI && seq[hh] >= 0) {  // carry the last-column maximum across chunks (upper rows win ties)
                    if (hasPrev) {
                        const int pv = p.scratch[2 * seq[hh]], pr = p.scratch[2 * seq[hh] + 1];
                        if (pv >= v) { v = pv; r = pr; }
                    }
                    if (hasNext && k == 0) { p.scratch[2 * seq[hh]] = v; p.scratch[2 * seq[hh] + 1] = r; }
                }
                if (k == kLast && v > prevBest[hh]) { sc = v; et = len[hh] - 1; eq = r; }
            }
            if (k == kLast && seq[hh] >= 0 && !hasNext) {
                p.outScore[seq[hh]] = sc;
                if (p.outEndQ) { p.outEndQ[seq[hh]] = eq; p.outEndT[seq[hh]] = et; }
            }
        }
    }
    clk_end(clkSlot);
}

template <int WIDTH, int MODE, int FLAGS, int S>
__global__ void __launch_bounds__(opal_max_waves_per_block(WIDTH) * 64, (WIDTH == 17 && !(FLAGS & FLAG_MULTI)) ? 4 : 1) opal_dp_kernel(const KParams p) {
    count_started(p.started);
    opal_dp_body<WIDTH, MODE, FLAGS, S>(p, KClass{p.log2G, p.firstItem, p.numItems, p.prio}, (int)blockIdx.x);
}

// ------------------------------------------------------------------ one launch for all length classes of a search
// The host splits a database into up to three length classes: the bulk on G0 lanes x S0 rows, long sequences on 2*G0 x
// ceil(S0/2) and the longest on 4*G0 x ceil(S0/4) (a 5000-residue sequence must not pin one wave for the whole launch).
// As separate launches they need a stream each, fork/join events (~35 us of host calls before the bulk launch is even
// submitted) and hardware queues that nothing else is using. Here they are ONE grid: the workgroups of the longest
// class come first in the grid (dispatch order = longest first), every workgroup runs the body of its class, and the
// register budget is the bulk body's. Plain-16-bit kernels, single-launch variants only.
struct UberParams {
    KParams base;
    KClass cls[3];      // [0] longest sequences (4*G0 lanes), [1] long (2*G0), [2] the bulk (G0); empty classes have numItems 0
    int firstBlock[3];  // first workgroup of each class in the grid
};
typedef void (*UberFn)(const UberParams);

template <int MODE, int FLAGS, int S0>
__global__ void __launch_bounds__(WAVES_PER_BLOCK * 64, 4) opal_dp_uber(const UberParams u) {
    const int b = (int)blockIdx.x;
    count_started(u.base.started);
    // (SW with end locations at 40 rows: the slot-exact end cell is two registers over inside this three-body kernel; row scans there)
    constexpr int FLAGS0 = (S0 >= 40 && MODE == MODE_SW && (FLAGS & FLAG_SWEND)) ? (FLAGS | FLAG_NOCKPT) : FLAGS;
    if (b >= u.firstBlock[2]) opal_dp_body<17, MODE, FLAGS0, S0>(u.base, u.cls[2], b - u.firstBlock[2]);
    else if (b >= u.firstBlock[1]) opal_dp_body<17, MODE, FLAGS, (S0 + 1) / 2>(u.base, u.cls[1], b - u.firstBlock[1]);
    else opal_dp_body<17, MODE, FLAGS, (S0 + 3) / 4>(u.base, u.cls[0], b - u.firstBlock[0]);
}

// Kernel tables, one per translation unit (inst_*.hip); nullptr = shape not instantiated.
KernelFn opal_kernel_sw16(int rows);                      // width 16, SW, score only: rows 1..40
KernelFn opal_kernel_w16(int mode, int flags, int rows);  // width 16, the other variants: rows 4,8,..,40
KernelFn opal_kernel_w32(int mode, int flags, int rows);  // width 32: rows 8,16,24,32,40
KernelFn opal_kernel_s16sw(int rows);                     // width 17, SW, score only: rows 1..40
KernelFn opal_kernel_s16swend(int rows);                  // width 17, SW, score + end location: rows 1..40
UberFn opal_uber_sw(int flags, int rows);                // one-launch kernels: SW score / SW end, bulk rows 1..40; FLAG_ZC (score): 1..38
UberFn opal_uber_skew(int mode, int rows);               // NW/HW/OV, FLAG_SKEW
KernelFn opal_kernel_s18(int mode, int flags, int rows);   // width 18 (2-byte profile entries): rows 8,16,..,40
KernelFn opal_kernel_s16skew(int mode, int rows);           // width 17, NW/HW/OV, FLAG_SKEW: rows 1..40
KernelFn opal_kernel_s16(int mode, int flags, int rows);  // width 17, other variants incl. FLAG_MULTI: rows 4,8,..,40
KernelFn opal_kernel_w16_multi(int mode, int flags, int rows);  // FLAG_MULTI variants
KernelFn opal_kernel_s18_multi(int mode, int flags, int rows);  // width 18, FLAG_MULTI: rows 8,16,24,32
KernelFn opal_kernel_w32_multi(int mode, int flags, int rows);
