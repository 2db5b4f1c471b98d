// planner.cpp -- launch planning for the database search: which wave geometry (lanes per sequence x rows per lane,
// waves per workgroup) a kernel shape gets, how a length-sorted item range is cut into length classes, and when the
// streaming kernels take over. Plays the part of the reference's loadNextSequence bookkeeping (reference
// src/opal.cpp:472-490): deciding which lanes work on which sequence, done ahead of the launch instead of per column.
#include "opal_internal.hpp"

namespace opal_host {

// Measured on MI355X (profiles/r01_ubench_*): packed-int16 and 32-bit max/min issue one wave
// instruction per ~4 clocks per SIMD. A lane's column step costs about (12*rows + 25) instructions.
static double step_clocks(int width, int rows) {
    if (width == 17) return 2.15 * 9.5 * rows + 90.0;  // full-rate scalar 16-bit ops, fixed per-step overhead
    if (width == 18) return 2.15 * 10.0 * rows + 90.0; // + one shift per two rows (2-byte profile entries)
    return 4.3 * (11.0 * rows + 12.0);
}

// A wave that is alone on its critical path (the longest sequences of a database) issues one instruction per ~4.8
// clocks whatever the pipe could take (profiles/r01_ubench_rowblock.txt: 42.9 clk per 9-instruction row on one wave):
// clocks per column step of such a wave.
static double lone_step_clocks(int width, int rows) {
    if (width == 17 || width == 18) return 4.8 * (9.5 * rows + 16.0);
    return 4.8 * (11.0 * rows + 14.0);
}

// Registers per lane of a kernel variant, asked once per variant from the code object.
int kernel_vgprs(int width, int mode, int swEndPass, int multi, int skew, int rows) {
    static std::mutex m;
    static std::map<long long, int> cache;
    const long long key = ((((long long)width * 8 + mode) * 4 + (swEndPass & 1) * 2 + (multi ? 1 : 0)) * 2 + (skew ? 1 : 0)) * 64 + rows;
    std::lock_guard<std::mutex> lk(m);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    OpalHipLaunch a;
    memset(&a, 0, sizeof a);
    a.width = width; a.mode = mode; a.want_end = swEndPass & 1; a.chunk_flags = multi ? 1 : 0; a.skew = skew; a.rows_per_lane = rows;
    const int v = opal_hip_kernel_vgprs(&a);
    cache[key] = v;
    return v;
}

// Measured: one wave per SIMD issues at half rate; two fill the VALU pipe of a pure-arithmetic loop (r01 ubench) but the
// real kernels (LDS reads with bank conflicts, DPP hand-offs) run 10-16 % slower on 2 or 3 waves than on 4
// (round 2: the OV kernel at 38 rows per lane, 129 VGPRs -> 2 waves, against 40 rows, 125 VGPRs -> 4 waves).
static double issue_factor(int wavesPerSimd) {
    return wavesPerSimd <= 1 ? 0.5 : wavesPerSimd == 2 ? 0.86 : wavesPerSimd == 3 ? 0.8 : 1.0;
}

/*
 * Occupancy of one kernel shape (profiles/r01_ubench_rowblock.txt: one wave per SIMD issues at half
 * rate, odd wave counts lose, two or four fill the pipe). Registers: 512 per lane per SIMD in granules
 * of 8 (MI355X_MICROARCH.md, register files); LDS: 160 KiB per CU, the query profile is per workgroup;
 * at most 32 waves per CU. Picks the waves per workgroup (and, when the register count alone would
 * give an odd number of waves, an LDS request padded so that fewer workgroups fit) giving every SIMD
 * an even number of waves. Returns waves per SIMD; `ldsPad` is the LDS bytes to request (0: the profile's own size).
 */
int pick_occupancy(int vgprs, size_t ldsBytes, int* wpbOut, int* ldsPadOut, int maxWpb) {
    const int alloc = vgprs > 0 ? (vgprs + 7) / 8 * 8 : 128;
    const int regWaves = std::max(1, std::min(8, 512 / alloc));
    int bestW = 0, bestWpb = 4, bestPad = 0;
    double bestScore = -1;
    for (int wpb : {4, 8, 16}) {
        if (wpb > maxWpb) continue;
        const int perSimdPerBlock = wpb / 4;
        const int blocksByLds = (int)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
        int blocks = std::min({blocksByLds, 32 / wpb, regWaves / perSimdPerBlock});
        if (blocks < 1) {
            if (perSimdPerBlock > regWaves) continue;  // a workgroup of this size does not fit the register file
            blocks = 1;
        }
        int w = blocks * perSimdPerBlock;
        int pad = 0;
        if (w > 2 && (w & 1)) {  // odd: drop to the even count below by padding the LDS request
            const int wantBlocks = (w - 1) / perSimdPerBlock;
            if (wantBlocks >= 1 && wantBlocks * perSimdPerBlock == w - 1) {
                pad = (160 * 1024) / (wantBlocks + 1) + 1024;  // wantBlocks fit, wantBlocks+1 do not
                if ((size_t)pad < ldsBytes) pad = 0;
                if ((size_t)pad * wantBlocks > 160 * 1024) pad = 0;
                if (pad) w = w - 1;
            }
        }
        const double score = issue_factor(w) + 0.001 * w + (wpb == 4 ? 0.0005 : 0.0);
        if (score > bestScore) { bestScore = score; bestW = w; bestWpb = wpb; bestPad = pad; }
    }
    *wpbOut = bestWpb;
    *ldsPadOut = bestPad;
    return bestW;
}
/*
 * Wave geometry of a query that runs in row chunks (launch_chunked): every geometry G x rows of the width is priced the way
 * plan_classes prices a class -- per-step overhead, fill and drain of the G-lane pipeline against the mean sequence
 * length, waves per SIMD that the chunk's profile leaves room for, and the longest sequence alone on its wave, once per
 * pass -- with the rows per lane lowered to the smallest value that still covers the query in the same number of passes
 * (equal chunks). Measured, 3000-aa query x 100 000 sequences, ms per search: 32 x 32 19.0, 16 x 32 19.7, 64 x 24 21.0,
 * 8 x 32 41.1 (width 18; the model: 18.1 : 19.3 : 20.4 : 38.5).
 */
bool plan_chunks(int width, int mode, int swEndPass, int Q, int A, long long nItems, double meanLen, int maxLen, int* Gout, int* rowsOut, int* wpbOut) {
    double bestClocks = 0;
    for (int G = 1; G <= 64; G *= 2)
        for (int S = OPAL_HIP_MAX_ROWS; S >= 1; S--) {
            if (opal_hip_rows_supported(width, mode, swEndPass | 2, S) != S) continue;
            if (opal_hip_lds_bytes(width, G, S, A) > 160 * 1024) continue;
            const int passes = (Q + G * S - 1) / (G * S);
            const int need = ((Q + passes - 1) / passes + G - 1) / G;  // rows per lane of a balanced chunk
            const int rows = opal_hip_rows_supported(width, mode, swEndPass | 2, need);
            if (!rows || rows > S) continue;
            int wpb = 4, ldsPad = 0;
            const int waves = pick_occupancy(kernel_vgprs(width, mode, swEndPass, 1, 0, rows), opal_hip_lds_bytes(width, G, rows, A), &wpb, &ldsPad,
                                             opal_hip_max_waves_per_block(width));
            // one pass: every item's len+G-1 steps on G of a wave's 64 lanes over 1024 SIMDs -- or the longest item's steps at the
            // lone-wave rate if that takes longer (passes run one after the other: each needs the boundary row of the one above)
            const double bulk = (double)nItems * (meanLen + G - 1) * G / 64.0 * step_clocks(width, rows) / 1024.0 / issue_factor(waves);
            const double tail = ((double)maxLen + G - 1) * lone_step_clocks(width, rows);
            // (the two overlap, but not for free: measured passes take the larger of the two plus about a third of the smaller)
            const double clocks = passes * (std::max(bulk, tail) + 0.35 * std::min(bulk, tail) + 24000.0);
            if (debug_on())
                fprintf(stderr, "[opal] chunk plan width=%d G=%d rows=%d: %d passes, %d waves/SIMD (waves/block %d), bulk %.0f tail %.0f -> %.0f clocks\n",
                        width, G, rows, passes, waves, wpb, bulk, tail, clocks);
            if (bestClocks == 0 || clocks < bestClocks) { bestClocks = clocks; *Gout = G; *rowsOut = rows; *wpbOut = wpb; }
        }
    return bestClocks > 0;
}

/*
 * Split the item range [first, first+count) (sorted by length, longest first) into classes of
 * equal wave geometry. Each sequence (pair) is spread over G lanes; a wave therefore needs
 * len+G-1 column steps of `rows` rows each. Short sequences want a small G (little fill/drain and
 * per-step overhead), the few very long ones a large G so that no single wave outlives the rest of
 * the launch. `lenOf(i)` is the length of item i.
 */
std::vector<LaunchClass> plan_classes(int width, int mode, int swEndPass, int skew, int Q, int A, LenView lenOf, int first,
                                      int count, int maxGFactor) {
    std::vector<LaunchClass> out;
    if (count <= 0) return out;
    struct Cand { int G, rows, wpb, ldsPad; double eff; };
    std::vector<Cand> cands;
    const int forceG = opts().lanesPerSeq;
    double lenSum = 0;
    const int sample = std::max(1, count / 4096);
    int nsamp = 0;
    for (int i = 0; i < count; i += sample) { lenSum += lenOf(first + i); nsamp++; }
    const double lavg = lenSum / nsamp;
    for (int G = 1; G <= 64; G *= 2) {
        if (forceG && forceG != G) continue;
        const int need = (Q + G - 1) / G;
        if (need > OPAL_HIP_MAX_ROWS) continue;
        // rows per lane: the smallest instantiated value, or up to 3 more if that one compiles to a register count
        // that leaves fewer waves per SIMD (a few padding rows cost less than an odd or halved occupancy)
        Cand best{0, 0, 4, 0, -1.0};
        int prevRows = 0;
        for (int want = need; want <= need + 3 && want <= OPAL_HIP_MAX_ROWS; want++) {
            const int rows = opal_hip_rows_supported(width, mode, swEndPass | (skew ? 4 : 0), want);
            if (!rows || rows == prevRows) continue;
            prevRows = rows;
            // (SW with end locations, single launch: + the slots' column buffers, priced at four waves per workgroup)
            const size_t lds = opal_hip_lds_bytes(width, G, rows, A) + ((swEndPass && mode == OPAL_MODE_SW && width == 17) ? opal_hip_swend_ckpt_bytes(G, rows, 4) : 0);
            if (lds > 160 * 1024) continue;
            if (G > 1 && (G / 2) * rows >= Q && !forceG) continue;  // half the lanes would hold padding only
            int wpb = 4, ldsPad = 0;
            const int waves = pick_occupancy(kernel_vgprs(width, mode, swEndPass, 0, skew, rows), lds, &wpb, &ldsPad, opal_hip_max_waves_per_block(width));
            const double eff = ((double)Q / (G * rows)) * (rows / step_clocks(width, rows)) * (lavg / (lavg + G - 1)) * issue_factor(waves);
            if (debug_on())
                fprintf(stderr, "[opal] plan width=%d mode=%d G=%d rows=%d: %d VGPRs, %zu B LDS -> %d waves/SIMD (waves/block %d, ldsPad %d), eff %.5f\n",
                        width, mode, G, rows, kernel_vgprs(width, mode, swEndPass, 0, skew, rows), lds, waves, wpb, ldsPad, eff);
            if (eff > best.eff) best = Cand{G, rows, wpb, ldsPad, eff};
        }
        if (best.eff > 0) cands.push_back(best);
    }
    if (cands.empty()) return out;
    size_t base = 0;
    for (size_t i = 1; i < cands.size(); i++)
        if (cands[i].eff > cands[base].eff * 1.02) base = i;
    // time of the whole range at the base geometry if the machine were perfectly filled
    const double simds = 1024.0;
    const int slotsBase = 64 / cands[base].G;
    const double totalClocks = lenSum * sample / slotsBase * step_clocks(width, cands[base].rows) / simds;
    const double budget = std::max(totalClocks * (opts().tailFraction > 0 ? opts().tailFraction : 0.7), 20000.0);
    // walk from the longest item: the smallest geometry (>= base) whose lone-wave time fits the budget
    size_t ci = 0;
    int start = first;
    // A class of long sequences is a handful of waves, each the oldest wave of its SIMD and therefore running at the
    // lone-wave rate: that is the time a boundary candidate is judged by. (Round 2, C2 database, 100 000 / 50 000
    // sequences: 1.594 -> 1.567 ms and 1.013 -> 0.967 ms against the pipe-rate model; OPAL_HIP_TAIL_MODEL=pipe restores it.)
    const bool pipeModel = opts().tailPipeModel;
    auto fits = [&](size_t c, int len) {
        const double per = pipeModel ? step_clocks(width, cands[c].rows) : lone_step_clocks(width, cands[c].rows);
        return (len + cands[c].G - 1) * per <= budget;
    };
    // One-launch kernels hold the bodies for G0, 2*G0 and 4*G0 lanes per sequence: keep the classes inside that chain --
    // unless the longest sequence would then outlast the whole search. A database with a real tail (Swiss-Prot: 35 213
    // residues) searched with a short query (G0 = 4: at most 16 lanes) is such a case: 110 aa x 541 762 sequences took 6.9 ms
    // with the chain against ~3 ms of DP work; the classes then go wider and are launched separately.
    if (maxGFactor > 0) {
        size_t keep = cands.size();
        while (keep > base + 1 && cands[keep - 1].G > maxGFactor * cands[base].G) keep--;
        if (keep == cands.size() || fits(keep - 1, lenOf(first)) || opts().uberOnly) cands.resize(keep);
    }
    auto cls = [&](int s, int n, size_t c) { return LaunchClass{s, n, cands[c].G, cands[c].rows, cands[c].wpb, cands[c].ldsPad}; };
    int i = first;
    while (i < first + count) {
        const int len = lenOf(i);
        size_t want = base;
        while (want + 1 < cands.size() && !fits(want, len)) want++;
        if (i == first) ci = want;
        if (want != ci) {
            // close the current class at a multiple of its slots-per-wave
            out.push_back(cls(start, i - start, ci));
            start = i;
            ci = want;
        }
        if (ci == base) break;  // everything from here on uses the base geometry
        i += 64 / cands[ci].G;
    }
    out.push_back(cls(start, first + count - start, ci));
    // Two side streams by default (three classes). HIP multiplexes normal-priority streams onto 4 hardware queues
    // shared with the caller's own streams: with normal-priority side streams two launches of one search could land on
    // the same queue and run one after the other (2.3 ms instead of 1.6 ms on C2, depending on how many streams the
    // process had created: scripts/gpu_queue_probe.py). The side streams are therefore HIGH-priority streams, which
    // get queues of their own; with them three side streams work too but measure 1-3 % slower than two.
    const size_t maxClasses = opts().sideStreams == 3 ? 4 : 3;  // (tuning hook: 3 side streams)
    while (out.size() > maxClasses) {  // fold the two longest classes together
        out[1].count += out[0].count;
        out[1].first = out[0].first;
        out[1].G = out[0].G; out[1].rows = out[0].rows; out[1].wpb = out[0].wpb; out[1].ldsPad = out[0].ldsPad;
        out.erase(out.begin());
    }
    return out;
}

/*
 * The role-swapped kernel (swap_kernel.hpp): how many of the longest items of [first, first + count) -- sorted longest first
 * -- leave the class kernels for it. A search with a short query keeps its length classes inside the one-launch chain (G0,
 * 2 G0, 4 G0 lanes per sequence); a target that would outlast the whole search even on the widest of them, alone on its
 * wave, is the critical path of the search (Swiss-Prot shape: 35 213 residues against a 110-aa query, 16 lanes x 7 rows:
 * 14 M clocks for a search of 7 M). Those targets go to the swapped kernel -- one workgroup each, the target on the rows,
 * (Q + 63) steps per chunk of 64 x rows residues -- provided it is the faster place for the longest of them and there are
 * no more of them than CUs (one workgroup fills most of a CU's LDS: the kernel is for a handful of outliers, not a class).
 */
int plan_swap(int Q, int A, int mode, LenView lenOf, int first, int count, long long residues, int* rowsOut, bool fits16, bool quiet) {
    *rowsOut = 0;
    const int forced = opts().swap;  // 0: never; N > 0 (test hook): the N longest items whatever the model says
    if (forced == 0) return 0;
    const bool gl = mode != OPAL_MODE_SW;  // NW / HW / OV: the int32 swapped kernels (swap_kernel.hpp, opal_swap_gl)
    int rows = gl ? opal_hip_swap_gl_rows(Q, A, fits16 ? 1 : 0) : opal_hip_swap_rows(Q, A);
    if (opts().swapRows) {  // test / tuning hook
        const size_t need = gl ? opal_hip_swap_gl_lds_bytes(opts().swapRows, Q, A) : opal_hip_swap_lds_bytes(opts().swapRows, Q, A);
        if (need <= 160 * 1024 && !(gl && opts().swapRows == 16 && !fits16)) rows = opts().swapRows;
    }
    if (!rows || count <= 0) return 0;
    *rowsOut = rows;
    if (forced > 0) return std::min(forced, count);
    int G0 = 1;
    while (G0 < 64 && (Q + G0 - 1) / G0 > OPAL_HIP_MAX_ROWS) G0 *= 2;
    const int S0 = (Q + G0 - 1) / G0;
    const int Gw = std::min(64, 4 * G0);
    const int Sw = (Q + Gw - 1) / Gw;
    // (global modes: 6-instruction rows in the bulk; the long targets mostly on the unskewed 8-instruction rows; the swapped
    // int32 kernel at 9 per row)
    const double totalClocks = (double)residues / (64 / G0) * (gl ? 2.15 * 6.5 * S0 + 90.0 : step_clocks(17, S0)) / 1024.0;
    // (global modes, round 6: the gate is the longest target itself -- alone on the class body it must outlast the WHOLE search by the
    // uncorrected model before anything is swapped. That keeps C3 shut (100 000 synthetic sequences to 5 000 residues, 300-aa query:
    // 0.95 of the search; forty swapped there cost 3-4 %, tests/test_gpu_fullsize.py holds it) and opens every search the forced-count
    // runs gained on: published shape 180-320 aa 1.2-1.3, its 20 000- / 100 000-sequence heads and 20 000 synthetic sequences 2-7:
    // profiles/r06_planner_sweep_swap_first.txt)
    const double budget = std::max(totalClocks * (opts().tailFraction > 0 ? opts().tailFraction : (gl ? 1.05 * 1.35 : 0.7)), 20000.0);
    // (x 1.35 in the global modes, round 6: a 400-aa search of the first 100 000 sequences of the published shape takes 8.1 ms with its
    // long targets on the class body -- 19 M clocks where this model says 14 -- and 5.8 with 40 of them swapped)
    auto classClocks = [&](int len) { return (len + Gw - 1) * (gl ? 1.35 * 4.8 * (8.0 * Sw + 28.0) : lone_step_clocks(17, Sw)); };
    // A swapped step is a latency chain with a large fixed part (address -> LDS -> first row, lane hand-off): measured on the
    // published shape, clocks per step: the 16-bit SW kernel 600 at 8 rows per lane, 1 900 at 40; the int32 global kernels 730
    // at 8 rows, 1 060 at 16. (With the first, optimistic model -- 4.8 clocks per instruction -- the planner sent the 5 000-residue
    // targets of the C2 database to the int32 kernel: C3 NW / HW / OV 1.15 -> 1.67 ms.)
    const double swapStep = gl ? 5.0 * (8.3 * rows + 80.0) : 275.0 + 40.6 * rows;
    auto swapClocks = [&](int len) { return (double)((len + 64 * rows - 1) / (64 * rows)) * (Q + 63) * swapStep; };
    const int longest = lenOf(first);
    // (a clear win only: at 64 lanes on the chain the two are within 10 % of each other -- 390-aa query, 35 213 residues)
    // (round 6, global modes: any win. The 30 % margin kept 160-400-aa queries off the swapped kernel on the published shape --
    // measured with 40 targets forced there: NW 180 aa 5.9 -> 4.5 ms, 256 aa 7.9 -> 5.8, 320 aa 8.0 -> 6.7: the class body runs
    // these targets UNSKEWED, one wave each, and they end last)
    if (classClocks(longest) <= budget || swapClocks(longest) >= (gl ? 1.0 : 0.7) * classClocks(longest)) return 0;
    // the search cannot end before its longest target has, wherever that runs: whatever fits THAT time stays on the chain
    // (how many: global modes by the round-5 figures -- 0.7 of the search, the uncorrected model; with the new ones a 150-aa search of
    // 100 000 synthetic sequences counted more than 240 and swapped none, 0.75 ms against 0.67 with its 80)
    const double limit = std::max(gl ? std::max(totalClocks * (opts().tailFraction > 0 ? opts().tailFraction : 0.7), 20000.0) : budget, swapClocks(longest));
    int n = 0;
    while (n < count && n <= 240 && classClocks(lenOf(first + n)) > (gl ? 1.35 : 1.0) * limit) n++;
    if (n > 240) {  // (not a handful of outliers: a class. Global modes, short queries: the forty longest all the same -- 40 aa x 100 000
        // synthetic sequences 0.89 ms with none swapped, 0.55 with forty)
        if (!(gl && Q <= 600)) return 0;
        n = 40;
    }
    // (global modes, queries to 600 residues: forty at least. A swapped workgroup has a CU's LDS to itself and there are 256 CUs; what
    // the limit above leaves behind -- targets whose modelled time on the class body just fits -- shares CUs eight waves to a workgroup
    // and ends last: with the model's 10-19 targets 180-400-aa searches of the published shape took 4.7-10.5 ms, with 40: 4.1-8.5)
    if (gl && Q <= 600) n = std::max(n, std::min(40, count));
    if (debug_on() && !quiet)
        fprintf(stderr, "[opal] swap plan: Q=%d, chain up to %d lanes x %d rows, budget %.0f clocks: the %d longest targets (%d .. %d residues) on the swapped kernel, %d rows per lane (longest: %.0f clocks there, %.0f on the chain)\n",
                Q, Gw, Sw, budget, n, longest, lenOf(first + n - 1), rows, swapClocks(longest), classClocks(longest));
    return n;
}

/*
 * One pair over several waves (pair_kernel.hpp, NW / HW / OV): how many of the longest items of [first, first + count) go there.
 * The need32 items the a-priori bounds send to the 32-bit rung always do (if the kernel has a shape for the query) -- on the
 * 32-bit class kernels each is one wave's 32-row steps, 15-22 ms for a 35 213-residue target --, and behind them the targets
 * that, alone on a wave of the 16-bit class body, would outlast half the search. Where the role-swapped kernels exist (queries
 * to 1024 residues) and are the faster place for the longest target, nothing goes here.
 */
int plan_pair(int Q, int A, int mode, LenView lenOf, int first, int count, long long residues, int need32, bool fits16, int* rowsOut, int* wavesOut) {
    *rowsOut = 0; *wavesOut = 0;
    const int forced = opts().pair;
    if (forced == 0 || mode == OPAL_MODE_SW || !fits16 || count <= 0) return 0;
    int S = 0, W = 0;
    if (!opal_hip_pair_shape(Q, A, &S, &W)) return 0;
    *rowsOut = S; *wavesOut = W;
    if (forced > 0) return std::min(forced, count);
    if (opts().swap > 0) return 0;  // (test hook: the N longest targets are FORCED onto the swapped kernels)
    if (need32 > 128) return 0;
    // (per step: ~9 instructions per row + ~40 of lane hand-off, queues, result tracking and the barrier share, at the rate of
    // one or two waves per SIMD. Measured, 100 000 sequences of the published shape: 513-aa HW 7.7 ms with four targets here against
    // 9.9 without, 2000-aa HW 11.1 against 26; all 541 762, 300 aa: 8.4 ms with 34 targets here, 7.9 with 10 on the swapped kernel)
    auto pairClocks = [&](int len) { return (double)(len + 63 + (W - 1) * 160) * (9.0 * S + 40.0) * (W <= 8 ? 5.2 : 8.6); };
    int G0 = 1;
    while (G0 < 64 && (Q + G0 - 1) / G0 > OPAL_HIP_MAX_ROWS) G0 *= 2;
    const int S0 = (Q + G0 - 1) / G0;
    const int Gw = std::min(64, 4 * G0);
    const int Sw = (Q + Gw - 1) / Gw;
    const int S64 = ((Q + 63) / 64 + 7) / 8 * 8;  // rows per lane of the 32-bit class kernels (64 lanes per sequence)
    const double totalClocks = (double)residues / (64 / G0) * (2.15 * 6.5 * S0 + 90.0) / 1024.0;
    // (a sixth of the search: what stays behind shares its CU with other long targets -- 16 waves per side workgroup of a stream,
    // 8.6 clocks per instruction instead of a lone wave's 4.8 -- and a 30 000-residue target left on the class body of a
    // 2000-aa search was 34 ms of a 66 ms search)
    const double budget = std::max(0.15 * totalClocks, 20000.0);
    auto classClocks = [&](int len, bool wide) { return wide ? (len + 63.0) * 4.8 * (11.0 * S64 + 14.0) : (len + Gw - 1.0) * 4.8 * (8.0 * Sw + 28.0); };
    const int longest = lenOf(first);
    // (the swapped kernels where they are the faster place: short queries)
    const int swapRows = opal_hip_swap_gl_rows(Q, A, 1);
    if (swapRows && opts().swap != 0) {
        const double swapClocks = (double)((longest + 64 * swapRows - 1) / (64 * swapRows)) * (Q + 63) * 5.0 * (8.3 * swapRows + 80.0);
        if (swapClocks <= pairClocks(longest)) return 0;
        // (round 6: ... and to 600 residues wherever plan_swap takes anything at all: at 2-3 waves x 2 rows per lane a pair step is all overhead --
        // 256 aa on the published shape 7.7 ms with 19 targets here, 5.3 with 40 on the swapped kernel; from 700 aa on the pair kernel wins)
        if (Q <= 600 && need32 <= 240) {
            int r = 0;
            if (plan_swap(Q, A, mode, lenOf, first, count, residues, &r, fits16, /*quiet=*/true) > 0) return 0;
        }
    }
    const double limit = std::max(budget, pairClocks(longest));  // the search cannot end before its longest target has
    int n = 0;
    while (n < count && n < 128) {
        const int len = lenOf(first + n);
        const bool wide = n < need32;
        if (!wide && !(classClocks(len, false) > limit && pairClocks(len) < classClocks(len, false))) break;  // (round 6: any win, as plan_swap)
        n++;
    }
    if (n < need32) return 0;  // (all of the 32-bit ones or none: what is left of them would need the 32-bit class kernels anyway)
    if (n > 0 && debug_on())
        fprintf(stderr, "[opal] pair plan: Q=%d, %d waves x 64 lanes x %d rows: the %d longest targets (%d .. %d residues, %d of them beyond the 16-bit bounds); longest: %.0f clocks there, %.0f on the class kernels; budget %.0f\n",
                Q, W, S, n, longest, lenOf(first + n - 1), need32, pairClocks(longest), classClocks(longest, need32 > 0), budget);
    return n;
}

// Streaming SW score search (stream_kernel.hpp): which geometry, if any. Long queries need 32 or 64 lanes per sequence,
// where the class kernels lose (G-1)/(L+G-1) of their steps to pipeline fill and drain; the streaming kernel has no
// such loss, so it is chosen by the cost of a column step per useful query row alone.
StreamPlan plan_stream(int Q, int A, int mode, int wantEnd, long long residues, int longest, int numSeqs) {
    StreamPlan best;
    if (opts().stream == 0) return best;     // OPAL_HIP_STREAM: 0 never, 1 whenever a geometry exists (tests), unset: long queries
    const bool force = opts().stream == 1;
    const int forceG = opts().lanesPerSeq;
    const int forceW = opts().narrow;
    double bestCost = 1e30;
    for (int width : {17, 18}) {
        if (forceW && forceW != width && (forceW == 17 || forceW == 18)) continue;
        for (int G = 1; G <= 64; G *= 2) {
            if (forceG && forceG != G) continue;
            const int need = (Q + G - 1) / G;
            if (need > OPAL_HIP_MAX_ROWS) continue;
            const int rows = opal_hip_stream_rows_supported(width, mode, wantEnd, need);
            if (!rows || opal_hip_stream_lds_bytes(width, mode, G, rows, A) > 160 * 1024) continue;
            const double perRow = (mode == OPAL_MODE_SW ? 9.0 + (wantEnd ? 0.6 : 0.0) : 6.0) + (width == 18 ? 0.5 : 0.0);
            const double cost = (perRow * rows + 30.0) * G / Q;
            if (cost < bestCost) { bestCost = cost; best.width = width; best.G = G; best.rows = rows; }
        }
    }
    if (!best.G && !forceW && !opts().noStreamChunks) {
        // No single launch holds the query (more than 64 lanes x 40 rows, or a profile beyond LDS): row chunks, one streaming
        // launch each (stream_kernel.hpp, CHUNK), equal chunks at the fewest passes each geometry allows.
        for (int G = 16; G <= 64; G *= 2) {
            if (forceG && forceG != G) continue;
            if (!force && (wantEnd || mode != OPAL_MODE_SW) && G < 32) continue;  // (as below: those run at 32 lanes or more)
            int cap = 0;
            for (int S = OPAL_HIP_MAX_ROWS; S >= 1 && !cap; S--)
                if (opal_hip_stream_rows_supported(17, mode, wantEnd | 2, S) == S && opal_hip_stream_lds_bytes(17, mode, G, S, A) <= 160 * 1024) cap = S;
            if (!cap) continue;
            const int passes = (Q + G * cap - 1) / (G * cap);
            const int need = ((Q + passes - 1) / passes + G - 1) / G;
            const int rows = opal_hip_stream_rows_supported(17, mode, wantEnd | 2, need);
            if (!rows || rows > cap) continue;
            const double perRowC = mode == OPAL_MODE_SW ? (wantEnd ? 10.5 : 9.0) : 6.0;
            const double cost = passes * (perRowC * rows + 36.0) * G / Q;
            // the longest sequence alone on its lanes against the whole pass, as below (the alternative here are the class
            // kernels' chunks, which no geometry brings beyond 0.60-0.65 of the issue bound: 3000-aa query, 100 000 sequences)
            const double stepInstr = perRowC * rows + 36.0;
            const double total = (double)residues * G / 64.0 * stepInstr * 2.15 / 1024.0;
            const double crit = ((double)longest + G) * stepInstr * 3.5;
            if (!force && crit > (G >= 32 ? 1.5 : 0.5) * total) continue;
            // (a persistent launch per row chunk: with ~2 sequences per slot it loses to the class kernels' chunks -- planner sweep,
            // 20 000 sequences, 2600- / 4100-aa queries: 8.6 ms against 6.9; with 12 per slot, 100 000 sequences, it wins)
            if (!force && (double)numSeqs < 6.0 * 4096.0 * (64 / G)) continue;
            if (cost < bestCost) { bestCost = cost; best.width = 17; best.G = G; best.rows = rows; best.passes = passes; }
        }
    }
    if (opts().streamChunkG > 0) {  // OPAL_HIP_STREAM_CHUNKS "G,S": test hook -- row chunks of G x S rows whatever the query length
        const int g = opts().streamChunkG, r = opts().streamChunkS;
        if (opal_hip_stream_rows_supported(17, mode, wantEnd | 2, r) == r && opal_hip_stream_lds_bytes(17, mode, g, r, A) <= 160 * 1024 && Q > g * r) {
            best.width = 17; best.G = g; best.rows = r; best.passes = (Q + g * r - 1) / (g * r);
            return best;
        }
    }
    if (best.G && !force && mode != OPAL_MODE_SW && best.G < 32) best = StreamPlan();  // (NW/HW/OV at 16 lanes or fewer: class kernels, unmeasured otherwise)
    if (best.G && !force && best.passes == 1) {
        // A sequence stays on its G lanes from its first column to its last: the longest one (handed out first) sets a
        // floor of (longest + G) steps at the rate of a wave that ends up alone on its SIMD (~3.5 clk per instruction).
        // On a small database that floor exceeds the time of the whole search and the class kernels (4*G lanes for
        // the longest sequences) win: measured with 20 000 sequences, 1000-aa query 2.6 ms against 1.7 ms; with 100 000
        // sequences 4.9 ms against 7.0 ms.
        const double stepInstr = (best.width == 18 ? 9.5 : 9.0) * best.rows + 30.0;
        const double total = (double)residues * best.G / 64.0 * stepInstr * 2.15 / 1024.0;  // (of one pass; so is crit)
        const double crit = ((double)longest + best.G) * stepInstr * 3.5;
        // (do the class kernels still hold the query at full rate -- 4-byte profile entries, 64 lanes?)
        bool class17 = false;
        {
            const int need = (Q + 63) / 64;
            const int rows = need <= OPAL_HIP_MAX_ROWS ? opal_hip_rows_supported(17, mode, 0, need) : 0;
            class17 = rows && opal_hip_lds_bytes(17, 64, rows, A) <= 160 * 1024;
        }
        if (best.G >= 64 && best.width == 18 && !class17 && numSeqs >= 1000) {
            // every lane of a wave on one sequence and a query whose 4-byte profile exceeds LDS: the class kernels have nothing wider
            // to give the longest sequence, run on packed pairs or 2-byte entries themselves, and pay fill and drain per sequence on
            // top (planner sweep, 2 000 sequences, 1700- / 2000-aa queries: 1.13-1.45 ms streamed against 1.54-1.95; Swiss-Prot-like
            // tail, 20 000 sequences: 12.1 against 14.9). A 1400-aa query, whose 4-byte profile fits the class kernels' workgroups
            // but not a persistent one of 16 waves, stays with the rule below: 0.96 ms streamed against 0.69 on the class kernels
            // for 2 000 sequences.
        } else if (best.G >= 32) {
            if (crit > 1.5 * total) best = StreamPlan();
            // (round 6) ... and the rule of the branch below for 32 lanes: with end locations the streams stop at 36 rows per lane, so a
            // query of 16 x 37..40 rows streams at 32 x 19..20 -- 608 aa on the published shape 25.5 ms against 20.9 on the class
            // kernels at 16 lanes, 640 aa 25.6 against 22.8, synthetic 541 762 sequences (gaps 11/1) 19.2 against 17.1. Not one step
            // up: 1216-1280 aa stream at 64 x 19..20 in 47 ms where the class kernels at 32 lanes take 60-66.
            if (best.G == 32 && wantEnd && mode == OPAL_MODE_SW) {
                const int needHalf = (Q + 15) / 16;
                const int rowsHalf = needHalf <= OPAL_HIP_MAX_ROWS ? opal_hip_rows_supported(17, mode, 1, needHalf) : 0;
                if (rowsHalf && opal_hip_lds_bytes(17, 16, rowsHalf, A) <= 160 * 1024) best = StreamPlan();
            }
        } else {
            // 16 lanes per sequence or fewer: the class kernels lose little to fill and drain, and streaming only wins
            // (2-4 %: 513-aa query x 625 000 sequences 6 790 -> 7 050 GCUPS, 300-aa x 1 000 000 7 230 -> 7 390) when every
            // slot works through many sequences; with ~3 per slot (300-aa x 100 000) it is 14 % slower.
            const double perSlot = (double)numSeqs / (4096.0 * (64 / best.G));
            if (crit > 0.5 * total || perSlot < 20.0) best = StreamPlan();  // (300-aa query: break-even at ~18 sequences per slot, 600 000 sequences)
            // With end locations the streaming kernels stop at 36 rows per lane, the class kernels at 40: a query that streams at
            // twice the lanes and half the rows only because of that (300 aa: 16 x 19 instead of 8 x 38) pays the per-step work
            // twice and loses to the class kernels (Swiss-Prot shape, gaps 3/1: 13.2 ms against 10.0).
            if (best.G && wantEnd && mode == OPAL_MODE_SW) {
                const int needHalf = (Q + best.G / 2 - 1) / (best.G / 2);
                if (best.G >= 2 && needHalf <= OPAL_HIP_MAX_ROWS && opal_hip_rows_supported(17, mode, 1, needHalf)) best = StreamPlan();
            }
        }
    }
    return best;
}

}  // namespace opal_host
