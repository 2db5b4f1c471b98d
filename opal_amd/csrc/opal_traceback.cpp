// opal_traceback.cpp -- host traceback for OPAL_SEARCH_ALIGNMENT (SURVEY.md 8f "next #1").
//
// What it restates: reference src/opal.cpp:1475-1507 (the reverse-prefix driver) and :1236-1431
// (findAlignment). Given a score and an end cell (both from the GPU search), the query prefix
// [0..endQ] and the target prefix [0..endT] are reversed and aligned from the corner with the mode's
// stop rule until the first column that reaches the score; the cell where that happens is the start
// of the alignment, the traceback from it gives the operations.
//
// Differences in HOW (not in results):
//   * one byte of traceback flags per cell instead of three ints, no band (cells outside the
//     reference's band cannot lie on a path that reaches the score, and every tie the traceback
//     tests is between paths that do, so the band never changes a decision);
//   * sequences are independent, so they are spread over host threads.
// The decisions themselves follow the reference exactly, so the operation strings are identical:
//   H prefers E, then F, then the diagonal (opal.cpp:1378-1386); a gap state prefers to close
//   (opal.cpp:1389, 1394); the first column reaching the score wins (:1275), then the first row
//   (:1355).
//
// With start cells already known (the device's reverse pass, rev_kernel.hip) only the box between start and end
// is traced: the DP is causal from the corner, so the box of the reversed prefixes holds exactly the cells the
// full pass would have computed there, and the traceback from the box's far corner is the same walk.
// Memory: one byte per cell of whatever is traced; concurrent tracebacks share a byte budget
// (OPAL_TRACEBACK_BYTES, default 8 GiB) and an allocation failure is reported as OPALX_ERR_NO_MEMORY
// instead of escaping through the C ABI.
#include <algorithm>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "host_pool.hpp"
#include "opal.h"
#include "opal_hip.h"

namespace {

constexpr int NEG = INT_MIN / 2;  // "minus infinity" that survives one subtraction of a gap penalty

// flags per cell
constexpr uint8_t H_FROM_E = 0, H_FROM_F = 1, H_FROM_DIAG = 2, H_MASK = 3;
constexpr uint8_t E_OPENS = 4;  // E == H[left] - gapOpen : the gap starts here
constexpr uint8_t F_OPENS = 8;  // F == H[up]   - gapOpen

struct Aligned {
    int score = 0, endQ = -1, endT = -1;  // in the frame of the two sequences given
    std::vector<unsigned char> ops;       // from (0,0) to (endQ,endT)
};

/*
 * Align q[0..Q) with t[0..T) starting in the top-left corner (both leading gaps are charged like in
 * NW), stop at the first column where a cell the mode allows as an end reaches `scoreLimit`
 * (reference opal.cpp:1236-1431).
 */
// Buffers a host thread keeps across the alignments it traces (no allocation per alignment in the steady state).
struct Workspace {
    std::vector<int> Hp, Ep, Hc, Ec;
    std::vector<uint8_t> dirs;
    std::vector<unsigned char> rTarget;
    // The workspace belongs to a pool thread that never exits: after a call it keeps only modest buffers (one search
    // with 35 000-residue targets must not leave gigabytes of flag matrices resident for the life of the process).
    void trim() {
        constexpr size_t KEEP = 32u << 20;  // bytes of flags kept per thread
        if (dirs.capacity() > KEEP) std::vector<uint8_t>().swap(dirs);
        if (rTarget.capacity() > (1u << 20)) std::vector<unsigned char>().swap(rTarget);
        for (std::vector<int>* v : {&Hp, &Ep, &Hc, &Ec})
            if (v->capacity() > (1u << 18)) std::vector<int>().swap(*v);
    }
};

// `prof` = the query profile of q: prof[letter * profStride + r] = M[q[r]][letter] (built once per call for the whole
// reversed query; an alignment's rows are a suffix of it), so the inner loop reads its scores sequentially.
static void find_alignment(const unsigned char* q, int Q, const unsigned char* t, int T, int Go, int Ge, const int* prof,
                           int profStride, int scoreLimit, int mode, Aligned& out, Workspace& ws) {
    std::vector<int>&Hv = ws.Hp, &Ev = ws.Ep;  // one column of H and E, updated in place
    Hv.resize((size_t)Q); Ev.resize((size_t)Q);
    for (int r = 0; r < Q; r++) { Hv[r] = -Go - r * Ge; Ev[r] = NEG; }  // initial column (opal.cpp:1266-1269)
    std::vector<uint8_t>& dirs = ws.dirs;
    if (dirs.size() < (size_t)Q * (size_t)std::min(T, 64)) dirs.resize((size_t)Q * (size_t)std::min(T, 64));
    int maxScore = INT_MIN, H = INT_MIN, c = 0;
    int* const Hc = Hv.data();
    int* const Ec = Ev.data();
    for (c = 0; c < T && maxScore < scoreLimit; c++) {
        if (dirs.size() < (size_t)(c + 1) * Q) dirs.resize(std::max((size_t)(c + 1) * Q, dirs.size() * 2));  // grows, never shrinks
        uint8_t* d = dirs.data() + (size_t)c * Q;
        const int* pc = prof + (size_t)t[c] * profStride;
        int uF = NEG, uH = -Go - c * Ge, ulH = c == 0 ? 0 : uH + Ge;
        int colMax = INT_MIN;
        for (int r = 0; r < Q; r++) {
            const int hPrev = Hc[r];  // H[r][c-1]
            const int eo = hPrev - Go, ee = Ec[r] - Ge;
            const int E = eo > ee ? eo : ee;
            const int fo = uH - Go, fe = uF - Ge;
            const int F = fo > fe ? fo : fe;
            const int diag = ulH + pc[r];
            const int ef = E > F ? E : F;
            H = ef > diag ? ef : diag;
            // H prefers E, then F, then the diagonal (opal.cpp:1378-1386); E_OPENS / F_OPENS: the gap starts here
            const unsigned f = (H == E ? H_FROM_E : (H == F ? H_FROM_F : H_FROM_DIAG)) | (E == eo ? E_OPENS : 0u) | (F == fo ? F_OPENS : 0u);
            d[r] = (uint8_t)f;
            colMax = H > colMax ? H : colMax;
            uF = F; uH = H; ulH = hPrev;
            Hc[r] = H; Ec[r] = E;
        }
        if (mode == OPAL_MODE_SW || (mode == OPAL_MODE_OV && c == T - 1)) maxScore = std::max(maxScore, colMax);
        if (mode == OPAL_MODE_HW || mode == OPAL_MODE_OV) maxScore = std::max(maxScore, H);  // last row
    }
    std::vector<int>& Hp = Hv;  // the last computed column
    const int lastCol = c - 1;
    // score and end cell (opal.cpp:1340-1362); Hp now holds the last computed column
    if (mode == OPAL_MODE_NW) {
        out.score = H; out.endT = T - 1; out.endQ = Q - 1;
    } else if (mode == OPAL_MODE_HW) {
        out.score = maxScore; out.endT = lastCol; out.endQ = Q - 1;
    } else {
        out.score = maxScore; out.endT = lastCol;
        int r = 0;
        while (r < Q && Hp[r] != maxScore) r++;
        out.endQ = r < Q ? r : Q - 1;
    }
    // traceback (opal.cpp:1369-1409)
    std::vector<unsigned char>& ops = out.ops;
    ops.clear();
    int ri = out.endQ, ci = out.endT;
    enum { SH, SE, SF } state = SH;
    while (ri >= 0 && ci >= 0) {
        const uint8_t f = dirs[(size_t)ci * Q + ri];
        if (state == SH) {
            if ((f & H_MASK) == H_FROM_E) state = SE;
            else if ((f & H_MASK) == H_FROM_F) state = SF;
            else { ops.push_back(q[ri] == t[ci] ? OPAL_ALIGN_MATCH : OPAL_ALIGN_MISMATCH); ri--; ci--; }
        } else if (state == SE) {
            state = (f & E_OPENS) ? SH : SE;
            ops.push_back(OPAL_ALIGN_INS);
            ci--;
        } else {
            state = (f & F_OPENS) ? SH : SF;
            ops.push_back(OPAL_ALIGN_DEL);
            ri--;
        }
    }
    while (ri >= 0) { ops.push_back(OPAL_ALIGN_DEL); ri--; }  // leading gap: only one of the two loops runs
    while (ci >= 0) { ops.push_back(OPAL_ALIGN_INS); ci--; }
    std::reverse(ops.begin(), ops.end());
}

// Bytes of traceback flags in flight across threads: a thread waits until its matrix fits the budget
// (a matrix larger than the whole budget runs alone).
class ByteBudget {
public:
    explicit ByteBudget(uint64_t total) : total_(total) {}
    // Requests below 32 MiB (a 300 x 350 alignment is 0.1 MB) bypass the budget: a lock and a wake-up per alignment
    // cost more than the alignment itself (20 000 alignments: 390 ms -> see profiles/r02_align_demo.txt).
    static bool small(uint64_t n) { return n < (32ull << 20); }
    void acquire(uint64_t n) {
        if (small(n)) return;
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return used_ == 0 || used_ + n <= total_; });
        used_ += n;
    }
    void release(uint64_t n) {
        if (small(n)) return;
        { std::lock_guard<std::mutex> lk(m_); used_ -= n; }
        cv_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    uint64_t total_, used_ = 0;
};

static void emit(OpalSearchResult* res, const Aligned& a, int aq, int at) {
    res->startLocationQuery = aq - a.endQ - 1;
    res->startLocationTarget = at - a.endT - 1;
    res->alignmentLength = (int)a.ops.size();
    res->alignment = (unsigned char*)malloc(a.ops.size() ? a.ops.size() : 1);  // caller free()s (opal.h)
    for (size_t i = 0; i < a.ops.size(); i++) res->alignment[i] = a.ops[a.ops.size() - 1 - i];  // back to forward order
}

// knownStartQ/T: start cell if already known (>= 0), else INT_MIN -> the full score-limited reverse pass finds it.
static void align_one(const unsigned char* rQuery, int queryLength, const unsigned char* target, int Go, int Ge,
                      const int* profR, OpalSearchResult* res, int mode, Workspace& ws,
                      int knownStartQ, int knownStartT, ByteBudget& budget) {
    std::vector<unsigned char>& rTarget = ws.rTarget;
    if (mode == OPAL_MODE_SW && res->score == 0) {  // no alignment at all (opal.cpp:1479-1483)
        res->alignment = NULL; res->alignmentLength = 0;
        res->startLocationQuery = res->startLocationTarget = -1;
        res->endLocationQuery = res->endLocationTarget = -1;
        return;
    }
    const int aq = res->endLocationQuery + 1, at = res->endLocationTarget + 1;
    if (aq <= 0 || at <= 0) {  // degenerate shapes (empty query or target): nothing to trace
        res->alignment = NULL; res->alignmentLength = 0;
        res->startLocationQuery = res->startLocationTarget = -1;
        return;
    }
    const unsigned char* alignQuery = rQuery + queryLength - aq;  // reversed prefix of the query
    rTarget.resize((size_t)at);
    for (int i = 0; i < at; i++) rTarget[i] = target[at - 1 - i];
    Aligned a;
    if (knownStartQ >= 0 && knownStartT >= 0 && knownStartQ < aq && knownStartT < at) {
        // box between start and end: the first bq x bt cells of the reversed prefixes, traced from its far corner
        const int bq = aq - knownStartQ, bt = at - knownStartT;
        const uint64_t bytes = (uint64_t)bq * (uint64_t)bt;
        budget.acquire(bytes);
        struct Rel { ByteBudget& b; uint64_t n; ~Rel() { b.release(n); } } rel{budget, bytes};
        find_alignment(alignQuery, bq, rTarget.data(), bt, Go, Ge, profR + (queryLength - aq), queryLength, INT_MAX, OPAL_MODE_NW, a, ws);
        if (a.score == res->score) { emit(res, a, aq, at); return; }
        // the given start does not carry the score: fall through to the full pass (never seen; kept for safety)
        if (opal_host::opts().debug) fprintf(stderr, "[opal] traceback: start (%d,%d) gives %d, expected %d; full pass\n", knownStartQ, knownStartT, a.score, res->score);
    }
    const uint64_t bytes = (uint64_t)aq * (uint64_t)at;  // upper bound (the pass stops at the start column)
    budget.acquire(bytes);
    struct Rel { ByteBudget& b; uint64_t n; ~Rel() { b.release(n); } } rel{budget, bytes};
    find_alignment(alignQuery, aq, rTarget.data(), at, Go, Ge, profR + (queryLength - aq), queryLength, res->score, mode, a, ws);
    emit(res, a, aq, at);
}

}  // namespace

extern "C" int opalx_find_alignments_masked(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                            int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                            OpalSearchResult* results[], int mode, const int* startQuery, const int* startTarget,
                                            const unsigned char* skip) {
    (void)dbSeqLengths;
    std::vector<unsigned char> rQuery((size_t)std::max(queryLength, 1));
    for (int i = 0; i < queryLength; i++) rQuery[i] = query[queryLength - 1 - i];
    // profile of the reversed query: profR[letter * queryLength + r] = M[rQuery[r]][letter]
    std::vector<int> profR((size_t)std::max(queryLength, 1) * (size_t)std::max(alphabetLength, 1));
    for (int l = 0; l < alphabetLength; l++)
        for (int r = 0; r < queryLength; r++) profR[(size_t)l * queryLength + r] = scoreMatrix[(int)rQuery[r] * alphabetLength + l];
    uint64_t total = 8ull << 30;
    if (opal_host::opts().tracebackBytes > 0) total = (uint64_t)opal_host::opts().tracebackBytes;
    ByteBudget budget(total);
    std::atomic<int> failed{0};
    // (the wide pool did not help: 20 000 alignments took 480 ms on 256 threads against 380 ms on 64; OPAL_HOST_WIDE=1 selects it)
    HostPool& pool = (opal_host::opts().hostWide == 1 && dbLength >= 512) ? HostPool::get_wide() : HostPool::get();
    const int tasks = std::max(1, std::min(dbLength, pool.width() * 4));
    pool.parallel_for(tasks, [&](int t) {
        // one workspace per host thread for the life of the process: its buffers grow once (fresh pages and mmap calls
        // from many threads at a time serialise on the process's memory map)
        static thread_local Workspace ws;
        for (int i = t; i < dbLength; i += tasks) {
            if (failed.load(std::memory_order_relaxed)) return;
            if (skip && skip[i]) continue;
            try {
                align_one(rQuery.data(), queryLength, db[i], gapOpen, gapExt, profR.data(), results[i], mode, ws,
                          startQuery ? startQuery[i] : INT_MIN, startTarget ? startTarget[i] : INT_MIN, budget);
            } catch (const std::bad_alloc&) {
                failed.store(1);
            }
        }
        ws.trim();
    });
    if (failed.load()) {
        fprintf(stderr, "[opal] traceback: out of host memory (a %d-row alignment matrix did not fit)\n", queryLength);
        return OPALX_ERR_NO_MEMORY;
    }
    return 0;
}

extern "C" int opalx_find_alignments_from(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                          int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                          OpalSearchResult* results[], int mode, const int* startQuery, const int* startTarget) {
    return opalx_find_alignments_masked(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix, alphabetLength,
                                        results, mode, startQuery, startTarget, nullptr);
}

extern "C" int opalx_find_alignments(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                     int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                     OpalSearchResult* results[], int mode) {
    return opalx_find_alignments_from(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix, alphabetLength,
                                      results, mode, nullptr, nullptr);
}
