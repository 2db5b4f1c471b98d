// opal_api.cpp -- host side of the MI355X-native aligner: the opal.h C API, the database packer
// and the launch scheduler (which plays the role of the reference's precision ladder).
//
// Plain C++ (no HIP headers): everything that touches the device goes through the extern "C"
// shim of include/opal_hip.h.
//
//   opalSearchDatabase        reference src/opal.cpp:1435-1519
//   opalSearchDatabaseCharSW  reference src/opal.cpp:1522-1546
//   result helpers            reference src/opal.cpp:1549-1564
//   precision ladder          reference src/opal.cpp:496-535 (SW), :983-1021 (NW/HW/OV)
//
// There is NO CPU fallback: without a usable gfx950 device every search returns
// OPAL_ERR_NO_SIMD_SUPPORT and says why on stderr.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "opal.h"
#include "opal_hip.h"

namespace {

// ------------------------------------------------------------------------------------ packed db
// Sequences sorted by length (descending), paired up neighbour with neighbour. A pair is the unit a
// wave slot works on at width 16 (two sequences packed in one register); its residues are stored
// column by column as 16-bit tokens (low byte A, high byte B) so one 2-byte load feeds both.
// The host only decides the order and gathers the residues into one buffer; the interleaving into
// tokens is done on the device (opal_hip_pack_tokens) at HBM speed.
struct HostPack {
    std::vector<uint64_t> pairOff;  // [P] start of each pair's token stream
    std::vector<int> pairLen;       // [2P]
    std::vector<int> pairSeq;       // [2P] caller index, -1 = absent
    std::vector<uint64_t> rawOff;   // [2P] where each sequence's residues start in the raw buffer
    std::vector<int> emptySeqs;     // caller indices of zero-length sequences (handled on the host)
    std::unique_ptr<unsigned char[]> gathered;  // raw residues in pair order (pointer-array inputs only)
    const unsigned char* raw = nullptr;   // the buffer to upload: `gathered` or the caller's flat buffer
    uint64_t rawBytes = 0;
    uint64_t numTokens = 0;
    int64_t residues = 0;
    int numSeqs = 0;
};

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool debug_on() {
    const char* e = getenv("OPAL_HIP_DEBUG");
    return e && *e && *e != '0';
}
static unsigned host_threads() {
    unsigned nt = std::thread::hardware_concurrency();
    if (const char* e = getenv("OPAL_HOST_THREADS")) nt = (unsigned)atoi(e);
    return std::max(1u, std::min(nt, 32u));
}

// Order and pairing (no residue is touched): stable sort by length, longest first.
static void pack_order(int n, const int* lens, HostPack& out) {
    out.numSeqs = n;
    int maxLen = 0;
    for (int i = 0; i < n; i++) maxLen = std::max(maxLen, lens[i]);
    std::vector<int> order;
    if (maxLen <= (1 << 22)) {  // counting sort
        std::vector<int> start((size_t)maxLen + 2, 0);
        for (int i = 0; i < n; i++) if (lens[i] > 0) start[(size_t)maxLen - lens[i] + 1]++;
        for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
        order.resize((size_t)start[(size_t)maxLen]);
        for (int i = 0; i < n; i++) {
            if (lens[i] > 0) order[(size_t)start[(size_t)maxLen - lens[i]]++] = i;
            else out.emptySeqs.push_back(i);
        }
    } else {
        for (int i = 0; i < n; i++) {
            if (lens[i] > 0) order.push_back(i);
            else out.emptySeqs.push_back(i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    }
    const int m = (int)order.size();
    const int P = (m + 1) / 2;
    out.pairOff.resize((size_t)P);
    out.pairLen.assign(2 * (size_t)P, 0);
    out.pairSeq.assign(2 * (size_t)P, -1);
    uint64_t total = 0;
    for (int p = 0; p < P; p++) {
        const int a = order[2 * (size_t)p];
        const int b = (2 * p + 1 < m) ? order[2 * (size_t)p + 1] : -1;
        out.pairSeq[2 * (size_t)p] = a; out.pairLen[2 * (size_t)p] = lens[a];
        out.pairSeq[2 * (size_t)p + 1] = b; out.pairLen[2 * (size_t)p + 1] = b >= 0 ? lens[b] : 0;
        out.pairOff[p] = total;
        total += (uint64_t)lens[a];  // A is the longer one
        out.residues += lens[a] + (b >= 0 ? lens[b] : 0);
    }
    out.numTokens = total;
}

// Residues given as separate host allocations: gather them (pair order) into one buffer, on host threads.
template <typename GetSeq>
static void gather_residues(GetSeq getSeq, HostPack& out) {
    const size_t n2 = out.pairSeq.size();
    out.rawOff.resize(n2);
    uint64_t total = 0;
    for (size_t i = 0; i < n2; i++) { out.rawOff[i] = total; total += (uint64_t)out.pairLen[i]; }
    out.gathered.reset(new unsigned char[total + 16]);  // uninitialised on purpose: every byte below `total` is copied
    auto copy = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++)
            if (out.pairSeq[i] >= 0) memcpy(out.gathered.get() + out.rawOff[i], getSeq(out.pairSeq[i]), (size_t)out.pairLen[i]);
    };
    const unsigned nt = total > (1u << 20) ? host_threads() : 1u;
    if (nt <= 1) copy(0, n2);
    else {
        std::vector<std::thread> pool;
        size_t i0 = 0;
        for (unsigned t = 0; t < nt; t++) {
            const uint64_t goal = total * (t + 1) / nt;
            size_t i1 = i0;
            while (i1 < n2 && (t + 1 == nt || out.rawOff[i1] < goal)) i1++;
            pool.emplace_back(copy, i0, i1);
            i0 = i1;
        }
        for (auto& th : pool) th.join();
    }
    out.raw = out.gathered.get();
    out.rawBytes = total;
}

}  // namespace

struct OpalxDb {
    int numSeqs = 0;
    int numPairs = 0;
    int64_t residues = 0;
    // host copies the scheduler needs
    std::vector<int> pairLen;  // [2P], pairs sorted by length of A descending
    std::vector<int> pairSeq;
    std::vector<int> emptySeqs;
    // device
    uint16_t* dTokens = nullptr;
    uint64_t* dPairOff = nullptr;
    int* dPairLen = nullptr;
    int* dPairSeq = nullptr;
    int* dOvfCount = nullptr;  // [2] counters: width-16 overflow list length, width-32 wrap count
    int* dOvfItems = nullptr;  // [2P]
    size_t numTokens = 0;
    void* dBoundary = nullptr; // 16 B per token: chunk-boundary rows of long queries (allocated on first use)
    int* dScratch = nullptr;   // [2N] per-sequence values carried across query chunks
    int* dResults = nullptr;   // [3N] device staging of host-array searches (opalx_search_host)
    unsigned char* dQuery = nullptr;
    int* dMatrix = nullptr;
    size_t queryCap = 0, matrixCap = 0;
    // side streams for launches that may overlap (created on first use; blocking streams: they order
    // themselves against the null stream, which carries the copies, memsets and timing events)
    void* streams[3] = {nullptr, nullptr, nullptr};
    void* events[4] = {nullptr, nullptr, nullptr, nullptr};
    void* stream(int i) {
        if (!streams[i]) streams[i] = opal_hip_stream_create();
        return streams[i];
    }
    void* event(int i) {
        if (!events[i]) events[i] = opal_hip_event_create();
        return events[i];
    }
};

namespace {

// The HIP "current device" is a per-thread setting, so the check (and the optional OPAL_HIP_DEVICE
// selection) is done once per calling thread. Without OPAL_HIP_DEVICE the thread's current device is
// used as it is (bench.py sets it through torch before calling in).
static bool device_ok() {
    static thread_local int checked = 0;  // 0 unknown, 1 ok, -1 none
    if (checked == 0) {
        if (opal_hip_device_count() > 0) {
            checked = 1;
            const char* env = getenv("OPAL_HIP_DEVICE");
            if (env && *env && opal_hip_set_device(atoi(env)) != 0) checked = -1;
        } else {
            checked = -1;
        }
        if (checked < 0)
            fprintf(stderr, "[opal] no usable HIP device (gfx950 required); there is no CPU fallback\n");
    }
    return checked > 0;
}

static OpalxDb* upload(HostPack& hp) {
    OpalxDb* db = new OpalxDb;
    db->numSeqs = hp.numSeqs;
    db->numPairs = (int)hp.pairOff.size();
    db->residues = hp.residues;
    db->pairLen = hp.pairLen;
    db->pairSeq = hp.pairSeq;
    db->emptySeqs = hp.emptySeqs;
    db->numTokens = hp.numTokens + 8;
    const size_t P = hp.pairOff.size();
    db->dTokens = (uint16_t*)opal_hip_malloc(db->numTokens * sizeof(uint16_t));
    db->dPairOff = (uint64_t*)opal_hip_malloc(P * sizeof(uint64_t));
    db->dPairLen = (int*)opal_hip_malloc(2 * P * sizeof(int));
    db->dPairSeq = (int*)opal_hip_malloc(2 * P * sizeof(int));
    db->dOvfCount = (int*)opal_hip_malloc(2 * sizeof(int));
    db->dOvfItems = (int*)opal_hip_malloc(2 * P * sizeof(int));
    unsigned char* dRaw = (unsigned char*)opal_hip_malloc(hp.rawBytes + 16);
    uint64_t* dRawOff = (uint64_t*)opal_hip_malloc(2 * P * sizeof(uint64_t));
    bool ok = db->dTokens && db->dPairOff && db->dPairLen && db->dPairSeq && db->dOvfCount && db->dOvfItems && dRaw && dRawOff;
    if (ok && P) {
        ok = !opal_hip_memcpy_h2d(dRaw, hp.raw, hp.rawBytes, nullptr) &&
             !opal_hip_memcpy_h2d(dRawOff, hp.rawOff.data(), 2 * P * sizeof(uint64_t), nullptr) &&
             !opal_hip_memcpy_h2d(db->dPairOff, hp.pairOff.data(), P * sizeof(uint64_t), nullptr) &&
             !opal_hip_memcpy_h2d(db->dPairLen, hp.pairLen.data(), 2 * P * sizeof(int), nullptr) &&
             !opal_hip_memcpy_h2d(db->dPairSeq, hp.pairSeq.data(), 2 * P * sizeof(int), nullptr) &&
             !opal_hip_pack_tokens(dRaw, dRawOff, db->dPairOff, db->dPairLen, (int)P, db->dTokens, db->numTokens, nullptr) &&
             !opal_hip_stream_sync(nullptr);
    }
    opal_hip_free(dRaw);
    opal_hip_free(dRawOff);
    if (!ok) { opalx_db_destroy(db); return nullptr; }
    return db;
}

// ------------------------------------------------------------------------------------ scheduling
struct LaunchClass { int first, count, G, rows; };

struct SearchCtx {
    OpalxDb* db;
    int Q, A, Go, Ge, mode, wantEnd;
    int* dScore; int* dEndQ; int* dEndT;
    int satLimit, negPad;  // width 17 (plain 16-bit) parameters
    long long maxScore;    // largest matrix entry (>= 0)
};

// Measured on MI355X (profiles/r01_ubench_*): packed-int16 and 32-bit max/min issue one wave
// instruction per ~4 clocks per SIMD. A lane's column step costs about (12*rows + 25) instructions.
static double step_clocks(int width, int rows) {
    if (width == 17) return 2.15 * 9.5 * rows + 90.0;  // full-rate scalar 16-bit ops, fixed per-step overhead
    return 4.3 * (11.0 * rows + 12.0);
}

/*
 * Split the item range [first, first+count) (sorted by length, longest first) into classes of
 * equal wave geometry. Each sequence (pair) is spread over G lanes; a wave therefore needs
 * len+G-1 column steps of `rows` rows each. Short sequences want a small G (little fill/drain and
 * per-step overhead), the few very long ones a large G so that no single wave outlives the rest of
 * the launch. `lenOf(i)` is the length of item i.
 */
template <typename LenOf>
static std::vector<LaunchClass> plan_classes(int width, int mode, int swEndPass, int Q, int A, LenOf lenOf, int first,
                                             int count) {
    std::vector<LaunchClass> out;
    if (count <= 0) return out;
    struct Cand { int G, rows; double eff; };
    std::vector<Cand> cands;
    const char* forceG = getenv("OPAL_HIP_LANES_PER_SEQ");
    double lenSum = 0;
    const int sample = std::max(1, count / 4096);
    int nsamp = 0;
    for (int i = 0; i < count; i += sample) { lenSum += lenOf(first + i); nsamp++; }
    const double lavg = lenSum / nsamp;
    for (int G = 1; G <= 64; G *= 2) {
        if (forceG && *forceG && atoi(forceG) != G) continue;
        const int need = (Q + G - 1) / G;
        if (need > OPAL_HIP_MAX_ROWS) continue;
        const int rows = opal_hip_rows_supported(width, mode, swEndPass, need);
        if (!rows) continue;
        if (opal_hip_lds_bytes(width, G, rows, A) > 160 * 1024) continue;
        if (G > 1 && (G / 2) * rows >= Q && !(forceG && *forceG)) continue;  // half the lanes would hold padding only
        const double eff = ((double)Q / (G * rows)) * (rows / step_clocks(width, rows)) * (lavg / (lavg + G - 1));
        cands.push_back({G, rows, eff});
    }
    if (cands.empty()) return out;
    size_t base = 0;
    for (size_t i = 1; i < cands.size(); i++)
        if (cands[i].eff > cands[base].eff * 1.02) base = i;
    // time of the whole range at the base geometry if the machine were perfectly filled
    const double simds = 1024.0;
    const int slotsBase = 64 / cands[base].G;
    const double totalClocks = lenSum * sample / slotsBase * step_clocks(width, cands[base].rows) / simds;
    const char* budEnv = getenv("OPAL_HIP_TAIL_FRACTION");
    const double budget = std::max(totalClocks * (budEnv && *budEnv ? atof(budEnv) : 0.35), 20000.0);
    // walk from the longest item: the smallest geometry (>= base) whose lone-wave time fits the budget
    size_t ci = cands.size() - 1;
    int start = first;
    auto fits = [&](size_t c, int len) { return (len + cands[c].G - 1) * step_clocks(width, cands[c].rows) <= budget; };
    int i = first;
    while (i < first + count) {
        const int len = lenOf(i);
        size_t want = base;
        while (want + 1 < cands.size() && !fits(want, len)) want++;
        if (i == first) ci = want;
        if (want != ci) {
            // close the current class at a multiple of its slots-per-wave
            out.push_back({start, i - start, cands[ci].G, cands[ci].rows});
            start = i;
            ci = want;
        }
        if (ci == base) break;  // everything from here on uses the base geometry
        i += 64 / cands[ci].G;
    }
    out.push_back({start, first + count - start, cands[ci].G, cands[ci].rows});
    while (out.size() > 4) {  // at most three side streams: fold the two longest classes together
        out[1].count += out[0].count;
        out[1].first = out[0].first;
        out[1].G = out[0].G; out[1].rows = out[0].rows;
        out.erase(out.begin());
    }
    return out;
}

// Queries that no single wave geometry can hold (more than 64 lanes x 40 rows, or a query profile
// beyond the 160 KiB of LDS): the query is cut into row chunks processed top to bottom by one launch each;
// the bottom row of a chunk (H and the F input of the next row, per column) travels through HBM.
static int launch_chunked(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, int firstItem,
                          int numItems, int* ovfCounter) {
    int bestG = 0, bestS = 0;
    for (int G = 64; G >= 1; G /= 2)
        for (int S = OPAL_HIP_MAX_ROWS; S >= 1; S--) {
            if (opal_hip_rows_supported(width, c.mode, swEndPass | 2, S) != S) continue;
            if (opal_hip_lds_bytes(width, G, S, c.A) > 160 * 1024) continue;
            if ((long long)G * S > (long long)bestG * bestS) { bestG = G; bestS = S; }
            break;  // largest S for this G
        }
    if (const char* geo = getenv("OPAL_HIP_CHUNK_GEOMETRY")) {  // "G,S": test hook to force small chunks
        int g = 0, r = 0;
        if (sscanf(geo, "%d,%d", &g, &r) == 2 && opal_hip_rows_supported(width, c.mode, swEndPass | 2, r) == r &&
            opal_hip_lds_bytes(width, g, r, c.A) <= 160 * 1024) { bestG = g; bestS = r; }
    }
    if (!bestG) {
        fprintf(stderr, "[opal] alphabet of %d letters does not fit the on-chip query profile at width %d\n", c.A, width);
        return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    OpalxDb* db = c.db;
    if (!db->dBoundary) db->dBoundary = opal_hip_malloc(16 * (db->numTokens + 8));
    if (!db->dScratch) db->dScratch = (int*)opal_hip_malloc(2 * sizeof(int) * (size_t)std::max(db->numSeqs, 1));
    if (!db->dBoundary || !db->dScratch) return OPAL_ERR_NO_SIMD_SUPPORT;
    const int cap = bestG * bestS;
    const int nchunks = (c.Q + cap - 1) / cap;
    const int firstRows = c.Q - (nchunks - 1) * cap;  // SW/HW/OV: the partial chunk is the first one
    for (int ch = 0; ch < nchunks; ch++) {
        int row0, rows;
        if (c.mode == OPAL_MODE_NW) { row0 = ch * cap; rows = std::min(cap, c.Q - row0); }
        else { row0 = ch == 0 ? 0 : firstRows + (ch - 1) * cap; rows = ch == 0 ? firstRows : cap; }
        OpalHipLaunch a;
        memset(&a, 0, sizeof a);
        a.width = width; a.mode = c.mode; a.want_end = swEndPass; a.detect_wrap = detectWrap;
        a.lanes_per_seq = bestG; a.rows_per_lane = bestS;
        a.query = db->dQuery; a.query_length = c.Q; a.score_matrix = db->dMatrix; a.alphabet_length = c.A;
        a.gap_open = c.Go; a.gap_ext = c.Ge;
        a.tokens = db->dTokens; a.pair_token_offset = db->dPairOff; a.pair_len = db->dPairLen; a.pair_seq = db->dPairSeq;
        a.work = work; a.first_item = firstItem; a.num_items = numItems;
        a.out_score = c.dScore; a.out_end_query = c.wantEnd ? c.dEndQ : nullptr; a.out_end_target = c.wantEnd ? c.dEndT : nullptr;
        a.overflow_count = ovfCounter; a.overflow_items = db->dOvfItems;
        a.sat_limit = c.satLimit; a.neg_pad = c.negPad;
        a.chunk_flags = (ch > 0 ? 1 : 0) | (ch < nchunks - 1 ? 2 : 0);
        a.chunk_row0 = row0; a.chunk_rows = rows;
        a.boundary = db->dBoundary; a.scratch = db->dScratch;
        if (debug_on())
            fprintf(stderr, "[opal] chunk %d/%d rows [%d,+%d) width=%d mode=%d end=%d G=%d rows/lane=%d\n", ch + 1, nchunks, row0,
                    rows, width, c.mode, swEndPass, bestG, bestS);
        if (opal_hip_launch(&a)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    return 0;
}

template <typename LenOf>
static int launch_range(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, int firstItem,
                        int numItems, LenOf lenOf, int* ovfCounter) {
    if (numItems <= 0) return 0;
    std::vector<LaunchClass> classes;
    if (work) {
        // re-run lists are unordered: one class sized for the longest sequence of the database
        auto l0 = [&](int) { return c.db->pairLen.empty() ? 1 : c.db->pairLen[0]; };
        classes = plan_classes(width, c.mode, swEndPass, c.Q, c.A, l0, firstItem, numItems);
    } else {
        classes = plan_classes(width, c.mode, swEndPass, c.Q, c.A, lenOf, firstItem, numItems);
    }
    if (classes.empty() || getenv("OPAL_HIP_FORCE_CHUNKS"))
        return launch_chunked(c, width, swEndPass, detectWrap, work, firstItem, numItems, ovfCounter);
    // The last class (the bulk of the database) runs on the null stream; the classes of long
    // sequences run beside it on up to three side streams (one hardware queue each), forked from and
    // joined back into the null stream with events, longest sequences enqueued first.
    const int nside = std::min<int>((int)classes.size() - 1, 3);
    void* fork = nullptr;
    if (nside > 0) {
        fork = c.db->event(0);
        if (!fork || opal_hip_event_record(fork, nullptr)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    for (size_t ki = 0; ki < classes.size(); ki++) {
        const LaunchClass& k = classes[ki];
        OpalHipLaunch a;
        memset(&a, 0, sizeof a);
        a.width = width; a.mode = c.mode; a.want_end = swEndPass; a.detect_wrap = detectWrap;
        a.lanes_per_seq = k.G; a.rows_per_lane = k.rows;
        a.query = c.db->dQuery; a.query_length = c.Q; a.score_matrix = c.db->dMatrix; a.alphabet_length = c.A;
        a.gap_open = c.Go; a.gap_ext = c.Ge;
        a.tokens = c.db->dTokens; a.pair_token_offset = c.db->dPairOff; a.pair_len = c.db->dPairLen; a.pair_seq = c.db->dPairSeq;
        a.work = work; a.first_item = k.first; a.num_items = k.count;
        a.out_score = c.dScore; a.out_end_query = c.wantEnd ? c.dEndQ : nullptr; a.out_end_target = c.wantEnd ? c.dEndT : nullptr;
        a.overflow_count = ovfCounter; a.overflow_items = c.db->dOvfItems;
        a.sat_limit = c.satLimit; a.neg_pad = c.negPad;
        if (width == 17 && c.mode != OPAL_MODE_SW && !work && !getenv("OPAL_HIP_NO_SKEW")) {
            // skewed kernels add up to (rows + steps) * Ge to every value: allowed when even the longest
            // sequence of the class stays inside the 16-bit range
            const long long L0 = lenOf(k.first);
            const long long top = std::min<long long>(c.Q, L0) * c.maxScore + (OPAL_HIP_MAX_ROWS + L0 + 64 + 4) * (long long)c.Ge;
            a.skew = top < 32000;
        }
        const bool side = (int)ki < nside;
        a.stream = side ? c.db->stream((int)ki) : nullptr;
        a.priority = side ? 1 : 0;
        if (debug_on())
            fprintf(stderr, "[opal] launch width=%d mode=%d end=%d items=[%d,+%d) G=%d rows=%d%s\n", width, c.mode, swEndPass,
                    k.first, k.count, k.G, k.rows, side ? " (side stream)" : "");
        if (side && opal_hip_stream_wait_event(a.stream, fork)) return OPAL_ERR_NO_SIMD_SUPPORT;
        if (opal_hip_launch(&a)) return OPAL_ERR_NO_SIMD_SUPPORT;
        if (side) {
            void* done = c.db->event(1 + (int)ki);
            if (!done || opal_hip_event_record(done, a.stream)) return OPAL_ERR_NO_SIMD_SUPPORT;
        }
    }
    for (int i = 0; i < nside; i++)
        if (opal_hip_stream_wait_event(nullptr, c.db->event(1 + i))) return OPAL_ERR_NO_SIMD_SUPPORT;
    return 0;
}

// reference opal.cpp:178-198 / :610-630 at 32 bits
static bool args_fit_int32(int Go, int Ge, const int* M, int A) {
    const int lo = INT_MIN / 2, hi = INT_MAX / 2;
    if (Go <= lo || hi <= Go || Ge <= lo || hi <= Ge) return false;
    for (int i = 0; i < A * A; i++)
        if (M[i] <= lo || hi <= M[i]) return false;
    return true;
}

static int ensure_query_buffers(OpalxDb* db, const unsigned char* query, int Q, const int* M, int A) {
    const size_t qb = (size_t)std::max(Q, 1), mb = (size_t)A * A * sizeof(int);
    if (db->queryCap < qb) {
        opal_hip_free(db->dQuery);
        db->dQuery = (unsigned char*)opal_hip_malloc(qb);
        db->queryCap = db->dQuery ? qb : 0;
    }
    if (db->matrixCap < mb) {
        opal_hip_free(db->dMatrix);
        db->dMatrix = (int*)opal_hip_malloc(mb);
        db->matrixCap = db->dMatrix ? mb : 0;
    }
    if (!db->dQuery || !db->dMatrix) return 1;
    if (Q && opal_hip_memcpy_h2d(db->dQuery, query, (size_t)Q, nullptr)) return 1;
    if (opal_hip_memcpy_h2d(db->dMatrix, M, mb, nullptr)) return 1;
    return 0;
}

// Values for the degenerate shapes the kernels never see (empty query or empty target): the
// boundary column/row of the DP matrix (same definitions as oracle/opal_oracle.c).
static void degenerate_result(int mode, int Q, int L, int Go, int Ge, int* score, int* endQ, int* endT) {
    if (mode == OPAL_MODE_SW) { *score = 0; *endQ = -1; *endT = -1; return; }
    if (mode == OPAL_MODE_NW) {
        const int n = Q + L;
        *score = n ? (int)(-(long long)Go - (long long)(n - 1) * Ge) : 0;
    } else if (mode == OPAL_MODE_HW) {
        *score = (Q && !L) ? (int)(-(long long)Go - (long long)(Q - 1) * Ge) : 0;
    } else {
        *score = 0;
    }
    *endQ = Q - 1; *endT = L - 1;
}

static int write_one(int* dArr, int idx, int value) {
    if (!dArr) return 0;
    return opal_hip_memcpy_h2d(dArr + idx, &value, sizeof(int), nullptr) || opal_hip_stream_sync(nullptr);
}

/*
 * The scheduler. Plays the role of the reference's precision ladder (opal.cpp:496-535, :983-1021)
 * with the rungs gfx950 has: packed 16-bit first, 32-bit for whatever saturates or cannot be
 * proven safe. OPAL_OVERFLOW_SIMPLE and OPAL_OVERFLOW_BUCKETS give identical results in the
 * reference (only the re-run granularity differs); here both map to the same schedule.
 */
static int search_core(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A,
                       int searchType, int mode, int* dScore, int* dEndQ, int* dEndT, float* kernelMs) {
    if (kernelMs) *kernelMs = 0.f;
    if (mode != OPAL_MODE_NW && mode != OPAL_MODE_HW && mode != OPAL_MODE_OV && mode != OPAL_MODE_SW)
        return OPAL_ERR_INVALID_MODE;
    if (!args_fit_int32(Go, Ge, M, A)) return OPAL_ERR_OVERFLOW;
    if (Go < 0 || Ge < 0) {
        fprintf(stderr, "[opal] negative gap penalties are not supported (opal.h: non-negative)\n");
        return OPAL_ERR_OVERFLOW;
    }
    if (!device_ok()) return OPAL_ERR_NO_SIMD_SUPPORT;
    const int wantEnd = searchType != OPAL_SEARCH_SCORE;
    const int P = db->numPairs;
    int rc = 0;

    // ---- degenerate shapes: answered from the matrix boundary
    for (int idx : db->emptySeqs) {
        int s, eq, et;
        degenerate_result(mode, Q, 0, Go, Ge, &s, &eq, &et);
        if (!wantEnd) { eq = et = -1; }
        if (write_one(dScore, idx, s) || write_one(dEndQ, idx, eq) || write_one(dEndT, idx, et)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    if (Q == 0) {
        for (int i = 0; i < 2 * P; i++) {
            if (db->pairSeq[i] < 0) continue;
            int s, eq, et;
            degenerate_result(mode, 0, db->pairLen[i], Go, Ge, &s, &eq, &et);
            if (!wantEnd) { eq = et = -1; }
            if (write_one(dScore, db->pairSeq[i], s) || write_one(dEndQ, db->pairSeq[i], eq) || write_one(dEndT, db->pairSeq[i], et))
                return OPAL_ERR_NO_SIMD_SUPPORT;
        }
        return opal_hip_stream_sync(nullptr) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
    }
    if (P == 0) return opal_hip_stream_sync(nullptr) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;

    if (ensure_query_buffers(db, query, Q, M, A)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (opal_hip_memset(db->dOvfCount, 0, 2 * sizeof(int), nullptr)) return OPAL_ERR_NO_SIMD_SUPPORT;

    long long maxS = 0, minS = 0;
    for (int i = 0; i < A * A; i++) { maxS = std::max<long long>(maxS, M[i]); minS = std::min<long long>(minS, M[i]); }
    const bool fits16 = Go <= 32767 && Ge <= 32767 && maxS <= 32767 && minS >= -32768;

    // Narrow rung: 17 = plain 16-bit, one sequence per lane (full-rate VALU, 4-byte LDS profile entries);
    // 16 = packed pairs (half-rate VALU, 2-byte entries: used when the wider profile does not fit LDS).
    int narrow = 17;
    if (const char* e = getenv("OPAL_HIP_NARROW")) narrow = atoi(e) == 16 ? 16 : 17;
    {
        bool fits17 = false;  // does some single-launch geometry hold the query with 4-byte profile entries?
        for (int G = 1; G <= 64 && !fits17; G *= 2) {
            const int need = (Q + G - 1) / G;
            if (need > OPAL_HIP_MAX_ROWS) continue;
            const int rows = opal_hip_rows_supported(17, mode, 0, need);
            fits17 = rows && opal_hip_lds_bytes(17, G, rows, A) <= 160 * 1024;
        }
        const char* force = getenv("OPAL_HIP_NARROW_FORCE");  // "17": run width 17 in row chunks rather than packed pairs
        if (!fits17 && !(force && atoi(force) == 17)) narrow = 16;
    }
    const int satLimit = narrow == 17 ? (int)(32767 - maxS) : 32767;
    SearchCtx c{db, Q, A, Go, Ge, mode, wantEnd, dScore, dEndQ, dEndT, satLimit, mode == OPAL_MODE_SW ? -satLimit : 0, maxS};
    const int nNarrow = narrow == 16 ? P : 2 * P;  // items of the narrow rung: pairs or sequences
    auto pairLenOf = [db](int p) { return db->pairLen[2 * (size_t)p]; };   // width-16 items are pairs
    auto seqLenOf = [db](int it) { return db->pairLen[(size_t)it]; };      // width-32 items are pair*2+half
    struct Events {  // timing events around every DP launch of this call, released on every return path
        void* start = nullptr; void* stop = nullptr;
        ~Events() { opal_hip_event_destroy(start); opal_hip_event_destroy(stop); }
    } ev;
    if (kernelMs) { ev.start = opal_hip_event_create(); ev.stop = opal_hip_event_create(); }
    void* const evStart = ev.start; void* const evStop = ev.stop;
    if (evStart) opal_hip_event_record(evStart, nullptr);

    if (mode == OPAL_MODE_SW) {
        // rung 1: everything at 16 bits (if the inputs fit), saturated scores go to the re-run list.
        // End locations are found in the same pass (the kernels record where a lane's running maximum
        // was first reached).
        int n32 = 0;
        const int* work32 = nullptr;
        int first32 = 0;
        if (fits16) {
            if (narrow == 16) rc = launch_range(c, 16, wantEnd, 0, nullptr, 0, nNarrow, pairLenOf, db->dOvfCount);
            else rc = launch_range(c, 17, wantEnd, 0, nullptr, 0, nNarrow, seqLenOf, db->dOvfCount);
            if (rc) return rc;
            int cnt = 0;
            if (opal_hip_memcpy_d2h(&cnt, db->dOvfCount, sizeof(int), nullptr) || opal_hip_stream_sync(nullptr))
                return OPAL_ERR_NO_SIMD_SUPPORT;
            n32 = cnt; work32 = db->dOvfItems;
        } else {
            n32 = 2 * P; first32 = 0; work32 = nullptr;  // items pair*2+half enumerate every sequence
        }
        // rung 2: width 32, with the reference's wrap test when a wrap is arithmetically possible
        if (n32 > 0) {
            const long long maxLen = db->pairLen[0];
            const int detect = (double)std::min<long long>(Q, maxLen) * (double)maxS > (double)INT_MAX;
            if ((rc = launch_range(c, 32, wantEnd, detect, work32, first32, n32, seqLenOf, db->dOvfCount + 1))) return rc;
            if (detect) {
                int wraps = 0;
                if (opal_hip_memcpy_d2h(&wraps, db->dOvfCount + 1, sizeof(int), nullptr) || opal_hip_stream_sync(nullptr))
                    return OPAL_ERR_NO_SIMD_SUPPORT;
                if (wraps) rc = OPAL_ERR_OVERFLOW;
            }
        }
    } else {
        // NW/HW/OV: width 16 where an a-priori bound proves that no value leaves the 16-bit range,
        // width 32 otherwise. Pairs are sorted by length (descending) and both bounds grow with the
        // length, so the width-32 pairs form a prefix of the pair list.
        int p32 = 0;
        if (!fits16) p32 = P;
        else {
            while (p32 < P) {
                const long long L = db->pairLen[2 * (size_t)p32];
                const long long lo = 3LL * Go + (Q + L + 2) * (long long)Ge - minS;  // -minS: diag + P must not wrap either
                const long long hi = std::min<long long>(Q, L) * maxS;
                if (lo < 32768 - 1 && hi < 32767) break;
                p32++;
            }
        }
        // 32-bit safety (our definition where the reference is undefined, see oracle/opal_oracle.c):
        // sequences whose bound leaves [-2^30, 2^30] are not computed and the call returns OVERFLOW.
        // Width-32 items (pair*2+half) enumerate the sequences in descending length, so the unsafe
        // ones are a prefix of the item list.
        int nBad = 0;
        while (nBad < 2 * p32) {
            const long long L = db->pairLen[nBad];
            const long long lo = 3LL * Go + (Q + L + 2) * (long long)Ge;
            const long long hi = std::min<long long>(Q, L) * maxS;
            if (lo < (1LL << 30) && hi < (1LL << 30)) break;
            nBad++;
        }
        if (nBad > 0) rc = OPAL_ERR_OVERFLOW;
        if (2 * p32 > nBad &&
            launch_range(c, 32, 0, 0, nullptr, nBad, 2 * p32 - nBad, seqLenOf, db->dOvfCount + 1))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        if (P > p32) {
            const int e = narrow == 16 ? launch_range(c, 16, 0, 0, nullptr, p32, P - p32, pairLenOf, db->dOvfCount)
                                       : launch_range(c, 17, 0, 0, nullptr, 2 * p32, 2 * (P - p32), seqLenOf, db->dOvfCount);
            if (e) return OPAL_ERR_NO_SIMD_SUPPORT;
        }
    }
    if (evStop) opal_hip_event_record(evStop, nullptr);
    if (opal_hip_stream_sync(nullptr)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (kernelMs && evStart && evStop) *kernelMs = opal_hip_event_elapsed_ms(evStart, evStop);
    return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------ opalx_* API
extern "C" OpalxDb* opalx_db_create(unsigned char* const db[], int db_length, const int db_seq_lengths[]) {
    if (!device_ok()) return nullptr;
    HostPack hp;
    pack_order(db_length, db_seq_lengths, hp);
    gather_residues([&](int i) { return (const unsigned char*)db[i]; }, hp);
    return upload(hp);
}

extern "C" OpalxDb* opalx_db_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length) {
    if (!device_ok()) return nullptr;
    std::vector<int> lens((size_t)db_length);
    for (int i = 0; i < db_length; i++) lens[i] = (int)(offsets[i + 1] - offsets[i]);
    HostPack hp;
    pack_order(db_length, lens.data(), hp);
    // already contiguous: upload as it is, the device packer reads each sequence at its offset
    hp.rawOff.resize(hp.pairSeq.size());
    for (size_t i = 0; i < hp.pairSeq.size(); i++) hp.rawOff[i] = hp.pairSeq[i] >= 0 ? (uint64_t)offsets[hp.pairSeq[i]] : 0;
    hp.raw = residues;
    hp.rawBytes = (uint64_t)offsets[db_length];
    return upload(hp);
}

extern "C" void opalx_db_destroy(OpalxDb* db) {
    if (!db) return;
    opal_hip_free(db->dTokens); opal_hip_free(db->dPairOff); opal_hip_free(db->dPairLen); opal_hip_free(db->dPairSeq);
    opal_hip_free(db->dOvfCount); opal_hip_free(db->dOvfItems);
    opal_hip_free(db->dQuery); opal_hip_free(db->dMatrix);
    opal_hip_free(db->dBoundary); opal_hip_free(db->dScratch); opal_hip_free(db->dResults);
    for (void* st : db->streams) opal_hip_stream_destroy(st);
    for (void* ev : db->events) opal_hip_event_destroy(ev);
    delete db;
}

extern "C" int opalx_db_num_sequences(const OpalxDb* db) { return db ? db->numSeqs : 0; }
extern "C" int64_t opalx_db_num_residues(const OpalxDb* db) { return db ? db->residues : 0; }

extern "C" int opalx_search_device(OpalxDb* db, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                   const int* score_matrix, int alphabet_length, int search_type, int mode,
                                   int overflow_method, int* d_score, int* d_end_query, int* d_end_target,
                                   float* kernel_ms) {
    (void)overflow_method;  // both methods give identical results (see search_core)
    if (!db) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (search_type != OPAL_SEARCH_SCORE && (!d_end_query || !d_end_target)) return OPAL_ERR_NO_SIMD_SUPPORT;
    return search_core(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode,
                       d_score, d_end_query, d_end_target, kernel_ms);
}

extern "C" int opalx_search_host(OpalxDb* db, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                 const int* score_matrix, int alphabet_length, int search_type, int mode,
                                 int overflow_method, int* score, int* end_query, int* end_target) {
    (void)overflow_method;
    if (!db) return OPAL_ERR_NO_SIMD_SUPPORT;
    const int n = db->numSeqs;
    const int wantEnd = search_type != OPAL_SEARCH_SCORE;
    const size_t bytes = (size_t)std::max(n, 1) * sizeof(int);
    if (!db->dResults) db->dResults = (int*)opal_hip_malloc(3 * bytes);  // kept for the life of the handle
    int* d = db->dResults;
    if (!d) return OPAL_ERR_NO_SIMD_SUPPORT;
    int* dS = d; int* dQ = d + std::max(n, 1); int* dT = d + 2 * (size_t)std::max(n, 1);
    // pre-load the caller's values so that sequences the search does not write keep them
    int rc = 0;
    if (opal_hip_memcpy_h2d(dS, score, (size_t)n * sizeof(int), nullptr)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc && end_query && opal_hip_memcpy_h2d(dQ, end_query, (size_t)n * sizeof(int), nullptr)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc && end_target && opal_hip_memcpy_h2d(dT, end_target, (size_t)n * sizeof(int), nullptr)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc)
        rc = search_core(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode, dS,
                         wantEnd ? dQ : nullptr, wantEnd ? dT : nullptr, nullptr);
    if (rc != OPAL_ERR_NO_SIMD_SUPPORT && rc != OPAL_ERR_INVALID_MODE) {
        int e = opal_hip_memcpy_d2h(score, dS, (size_t)n * sizeof(int), nullptr);
        if (wantEnd && end_query) e |= opal_hip_memcpy_d2h(end_query, dQ, (size_t)n * sizeof(int), nullptr);
        if (wantEnd && end_target) e |= opal_hip_memcpy_d2h(end_target, dT, (size_t)n * sizeof(int), nullptr);
        e |= opal_hip_stream_sync(nullptr);
        if (e) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    }
    return rc;
}

// ------------------------------------------------------------------------------------ opal.h API
extern "C" void opalInitSearchResult(OpalSearchResult* result) {
    result->scoreSet = 0;
    result->endLocationTarget = result->endLocationQuery = -1;
    result->startLocationTarget = result->startLocationQuery = -1;
    result->alignment = NULL;
    result->alignmentLength = 0;  // `score` is deliberately left alone, like the reference
}

extern "C" int opalSearchResultIsEmpty(const OpalSearchResult result) { return !result.scoreSet; }

extern "C" void opalSearchResultSetScore(OpalSearchResult* result, int score) {
    result->scoreSet = 1;
    result->score = score;
}

extern "C" int opalSearchDatabase(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                  int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                  OpalSearchResult* results[], const int searchType, int mode, int overflowMethod) {
    if (mode != OPAL_MODE_NW && mode != OPAL_MODE_HW && mode != OPAL_MODE_OV && mode != OPAL_MODE_SW)
        return OPAL_ERR_INVALID_MODE;
    // Results that already hold what is asked for are kept (opal.h:118-122, opal.cpp:1446-1451);
    // implemented with the documented, whole-database indexing (the reference indexes `skip`
    // bucket-locally, opal.cpp:510,998).
    std::vector<int> todo;
    todo.reserve((size_t)dbLength);
    for (int i = 0; i < dbLength; i++) {
        const bool have = results[i]->scoreSet &&
                          (searchType == OPAL_SEARCH_SCORE ||
                           (results[i]->endLocationQuery >= 0 && results[i]->endLocationTarget >= 0));
        if (!have) todo.push_back(i);
    }
    int rc = 0;
    if (!todo.empty()) {
        if (!device_ok()) return OPAL_ERR_NO_SIMD_SUPPORT;
        const int n = (int)todo.size();
        std::vector<int> lens((size_t)n);
        for (int i = 0; i < n; i++) lens[i] = dbSeqLengths[todo[i]];
        const double t0 = now_ms();
        HostPack hp;
        pack_order(n, lens.data(), hp);
        gather_residues([&](int i) { return (const unsigned char*)db[todo[i]]; }, hp);
        const double t1 = now_ms();
        OpalxDb* pdb = upload(hp);
        if (!pdb) return OPAL_ERR_NO_SIMD_SUPPORT;
        const double t2 = now_ms();
        const int st = searchType == OPAL_SEARCH_SCORE ? OPAL_SEARCH_SCORE : OPAL_SEARCH_SCORE_END;
        // INT_MIN marks "not written by the search" (a sequence the search had to give up on)
        std::vector<int> score((size_t)n, INT_MIN), endQ((size_t)n, -1), endT((size_t)n, -1);
        rc = opalx_search_host(pdb, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, st, mode,
                               overflowMethod, score.data(), endQ.data(), endT.data());
        const double t3 = now_ms();
        opalx_db_destroy(pdb);
        if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) return rc;
        for (int i = 0; i < n; i++) {
            if (score[i] == INT_MIN) continue;
            OpalSearchResult* r = results[todo[i]];
            opalSearchResultSetScore(r, score[i]);
            r->endLocationQuery = endQ[i];
            r->endLocationTarget = endT[i];
        }
        if (debug_on())
            fprintf(stderr, "[opal] opalSearchDatabase: pack %.2f ms, upload %.2f ms, search+download %.2f ms, free+scatter %.2f ms\n",
                    t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
    } else if (!args_fit_int32(gapOpen, gapExt, scoreMatrix, alphabetLength)) {
        rc = OPAL_ERR_OVERFLOW;
    }
    if (rc) return rc;  // alignment fields untouched on error (opal.cpp:1473)

    if (searchType == OPAL_SEARCH_ALIGNMENT) {
        return opalx_find_alignments(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix,
                                    alphabetLength, results, mode);
    }
    for (int i = 0; i < dbLength; i++) {  // opal.cpp:1508-1515
        results[i]->alignment = NULL;
        results[i]->alignmentLength = -1;
        results[i]->startLocationQuery = -1;
        results[i]->startLocationTarget = -1;
    }
    return 0;
}

extern "C" int opalSearchDatabaseCharSW(unsigned char query[], int queryLength, unsigned char** db, int dbLength,
                                        int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                        OpalSearchResult* results[]) {
    // The reference runs its 8-bit kernel only (opal.cpp:1533-1536): arguments that do not fit a
    // signed char fail the pass before any result is written (opal.cpp:178-198).
    bool fits = gapOpen >= SCHAR_MIN && gapOpen <= SCHAR_MAX && gapExt >= SCHAR_MIN && gapExt <= SCHAR_MAX;
    for (int i = 0; fits && i < alphabetLength * alphabetLength; i++)
        fits = scoreMatrix[i] >= SCHAR_MIN && scoreMatrix[i] <= SCHAR_MAX;
    if (!fits) {
        for (int i = 0; i < dbLength; i++) { results[i]->score = -1; results[i]->scoreSet = 0; }
        return OPAL_ERR_OVERFLOW;
    }
    if (!device_ok()) return OPAL_ERR_NO_SIMD_SUPPORT;
    OpalxDb* pdb = opalx_db_create(db, dbLength, dbSeqLengths);
    if (!pdb) return OPAL_ERR_NO_SIMD_SUPPORT;
    std::vector<int> score((size_t)dbLength, INT_MIN);
    int rc = opalx_search_host(pdb, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, OPAL_SEARCH_SCORE,
                               OPAL_MODE_SW, OPAL_OVERFLOW_SIMPLE, score.data(), nullptr, nullptr);
    opalx_db_destroy(pdb);
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT) return rc;
    rc = 0;
    for (int i = 0; i < dbLength; i++) {
        // the 8-bit pass represents 0..127: a sequence overflows iff its exact score is >= 128
        if (score[i] == INT_MIN || score[i] >= 128) {
            results[i]->score = -1; results[i]->scoreSet = 0; rc = OPAL_ERR_OVERFLOW;
        } else {
            opalSearchResultSetScore(results[i], score[i]);
            results[i]->endLocationQuery = -1; results[i]->endLocationTarget = -1;
        }
    }
    return rc;
}
