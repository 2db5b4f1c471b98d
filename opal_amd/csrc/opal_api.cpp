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
//
// What one opalSearchDatabase call costs on an MI355X is host work, not DP (the kernels need under two
// milliseconds for 10^10 cells), so the boundary is built for that:
//   * device buffers, pinned staging memory, streams and events live in cached CONTEXTS that calls
//     borrow and give back (no hipMalloc/hipFree, stream or event creation in the steady state);
//   * the N separate host allocations are gathered into pinned memory by a persistent thread pool,
//     each chunk going to the device as soon as it is complete; the interleaving into tokens is a kernel;
//   * results come back through pinned memory and are scattered into the N result structs by the pool;
//   * with OPAL_HIP_DEVICES set, the call shards the database over several GPUs (sequences dealt in
//     64-sequence groups balancing the residue count), one host thread per GPU.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host_pool.hpp"
#include "opal.h"
#include "opal_hip.h"

namespace {

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static std::atomic<long long> g_streamLaunches{0};  // searches that took the streaming SW kernel (tests ask)
static bool debug_on() {
    static const int on = [] { const char* e = getenv("OPAL_HIP_DEBUG"); return (e && *e && *e != '0') ? 1 : 0; }();
    return on != 0;
}

// ------------------------------------------------------------------------------------ buffers
// Device / pinned-host buffers that only ever grow: a cached context keeps them across calls.
// (A cached context that once held a multi-gigabyte database does not keep those buffers for a later small one:
// a buffer more than 4x too large and above 256 MiB is given back first.)
template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    bool ensure(size_t n, double slack = 1.0) {
        if (p && cap > 4 * std::max<size_t>(n, 1) && cap * sizeof(T) > (256u << 20)) release();
        if (n <= cap && p) return true;
        opal_hip_free(p);
        const size_t want = std::max<size_t>((size_t)((double)n * slack), 1);
        p = (T*)opal_hip_malloc(want * sizeof(T));
        cap = p ? want : 0;
        return p != nullptr;
    }
    void release() { opal_hip_free(p); p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n, double slack = 1.0) {
        if (p && cap > 4 * std::max<size_t>(n, 1) && cap * sizeof(T) > (256u << 20)) release();
        if (n <= cap && p) return true;
        opal_hip_host_free(p);
        const size_t want = std::max<size_t>((size_t)((double)n * slack), 1);
        p = (T*)opal_hip_host_malloc(want * sizeof(T));
        cap = p ? want : 0;
        return p != nullptr;
    }
    void release() { opal_hip_host_free(p); p = nullptr; cap = 0; }
};

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
    std::vector<int> emptySeqs;     // caller indices of zero-length sequences (answered on the host)
    uint64_t rawBytes = 0;
    uint64_t numTokens = 0;
    int64_t residues = 0;
    int numSeqs = 0;
};

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

}  // namespace

struct OpalxDb {
    int device = 0;
    bool cached = false;  // a context of the opalSearchDatabase cache (returned to the free list, not destroyed)
    int numSeqs = 0;
    int numPairs = 0;
    int64_t residues = 0;
    size_t numTokens = 0;
    // host copies the scheduler needs
    std::vector<int> pairLen;  // [2P], pairs sorted by length of A descending
    std::vector<int> pairSeq;
    std::vector<int> emptySeqs;
    // device: the packed database
    DevBuf<uint16_t> tokens;
    DevBuf<uint64_t> pairOff;
    DevBuf<int> dPairLen, dPairSeq;
    // One "search header" per search, uploaded with a single copy: {counters[4], score matrix, query}. ovfCount[0] =
    // length of the 16-bit rung's re-run list, [1] = width-32 wrap count.
    DevBuf<unsigned char> header;
    PinBuf<unsigned char> hHeader;
    struct View32 { int* p = nullptr; } ovfCount, matrix;
    struct View8 { unsigned char* p = nullptr; } query;
    DevBuf<int> ovfItems;  // [2P]
    // device: per-search scratch
    DevBuf<unsigned char> boundary;  // 16 B per token: chunk-boundary rows of long queries (allocated on first use)
    DevBuf<int> scratch;             // [2N] per-sequence values carried across query chunks
    DevBuf<int> results;             // [3N] device staging of host-array searches
    DevBuf<int> quads;               // degenerate results on their way to the scatter kernel
    // start-location pass (rev_kernel.hip): caller index -> pair*2+half, flags and the host-redo list
    DevBuf<int> seqItem;
    bool seqItemValid = false;
    DevBuf<int> revFlags;
    DevBuf<int> redo;
    // alignment operations on the device (trace_kernel.hip): flag matrices, operation strings, per-batch lists and outputs
    DevBuf<unsigned char> trDirs, trOps;
    DevBuf<uint64_t> trOff;
    DevBuf<int> trItems, trOut;
    // launch plans of this database, keyed by what they depend on (cleared when the database is refilled): a search
    // with the same query length, alphabet and mode re-uses the class split instead of recomputing it with the GPU idle
    std::map<std::string, std::vector<int>> planCache;
    // staging used while (re)filling the database
    DevBuf<unsigned char> raw;
    DevBuf<uint64_t> rawOff;
    PinBuf<unsigned char> hRaw;
    PinBuf<int> hResults;
    // Streams: everything of one search runs on `main` (a non-blocking stream: no implicit ordering
    // against the legacy null stream or against other contexts); long-sequence classes run on up to three
    // HIGH-priority side streams (hardware queues of their own), forked from and joined back into `main` with events.
    void* mainStream = nullptr;
    void* streams[3] = {nullptr, nullptr, nullptr};
    void* events[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // 0 fork, 1-3 joins, 4-5 timing
    void* main() {
        if (!mainStream) mainStream = opal_hip_stream_create();
        return mainStream;
    }
    void* stream(int i) {
        if (!streams[i]) { const char* e = getenv("OPAL_HIP_SIDE_STREAM_PRIO"); streams[i] = (e && *e == '0') ? opal_hip_stream_create() : opal_hip_stream_create_high(); }
        return streams[i];
    }
    void* event(int i) {
        if (!events[i]) events[i] = opal_hip_event_create();
        return events[i];
    }
    // Wait for everything this handle may have in flight (used on error paths before buffers are reused).
    void quiesce() {
        for (void* st : streams) if (st) opal_hip_stream_sync(st);
        if (mainStream) opal_hip_stream_sync(mainStream);
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

static void db_release(OpalxDb* db) {
    db->tokens.release(); db->pairOff.release(); db->dPairLen.release(); db->dPairSeq.release();
    db->header.release(); db->hHeader.release(); db->ovfItems.release(); db->boundary.release(); db->scratch.release();
    db->results.release(); db->quads.release();
    db->ovfCount.p = nullptr; db->matrix.p = nullptr; db->query.p = nullptr;
    db->seqItem.release(); db->revFlags.release(); db->redo.release();
    db->trDirs.release(); db->trOps.release(); db->trOff.release(); db->trItems.release(); db->trOut.release();
    db->raw.release(); db->rawOff.release(); db->hRaw.release(); db->hResults.release();
    for (void*& st : db->streams) { opal_hip_stream_destroy(st); st = nullptr; }
    opal_hip_stream_destroy(db->mainStream); db->mainStream = nullptr;
    for (void*& ev : db->events) { opal_hip_event_destroy(ev); ev = nullptr; }
}

enum RawKind { RAW_POINTERS, RAW_HOST_FLAT, RAW_DEVICE_FLAT };

/*
 * (Re)fill `db` with the packed form of `hp`. Residues come from
 *   RAW_POINTERS     getSeq(i): N separate host allocations -> gathered into pinned memory by the thread pool,
 *                    every ~2 MB chunk uploaded as soon as it is complete;
 *   RAW_HOST_FLAT    one contiguous host buffer (uploaded as it is, the device packer reads at hp.rawOff);
 *   RAW_DEVICE_FLAT  the same already in device memory (nothing is copied).
 * Buffers of a cached context are reused when large enough.
 */
static bool db_fill(OpalxDb* db, HostPack& hp, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
                    const unsigned char* flat, const int* orderLens = nullptr, int orderN = 0) {
    db->device = opal_hip_get_device();
    const double slack0 = db->cached ? 1.15 : 1.0;
    const unsigned char* dRawEarly = nullptr;
    if (kind == RAW_POINTERS && orderLens) {
        // `hp` is not ordered yet: the residues are gathered in CALLER order (that needs the lengths only) while one pool
        // task sorts and pairs the sequences (pack_order, ~0.3 ms for 100 000 sequences, used to run before the gather);
        // the device packer finds every sequence through rawOff, whatever the order of the raw buffer.
        void* st0 = db->main();
        if (!st0) return false;
        std::vector<uint64_t> callerOff((size_t)orderN + 1, 0);
        for (int i = 0; i < orderN; i++) callerOff[(size_t)i + 1] = callerOff[(size_t)i] + (uint64_t)std::max(orderLens[i], 0);
        const uint64_t total = callerOff[(size_t)orderN];
        if (!db->raw.ensure(total + 16, slack0) || !db->hRaw.ensure(total + 16, slack0)) return false;
        const uint64_t chunkBytes = 2u << 20;
        std::vector<int> cut{0};
        for (int i = 0; i < orderN; i++)
            if (callerOff[(size_t)i] - callerOff[(size_t)cut.back()] >= chunkBytes) cut.push_back(i);
        cut.push_back(orderN);
        std::atomic<int> bad{0};
        unsigned char* h = db->hRaw.p;
        unsigned char* d = db->raw.p;
        const int dev = db->device;
        HostPool::get().parallel_for((int)cut.size(), [&](int t) {
            if (t == 0) { pack_order(orderN, orderLens, hp); return; }
            static thread_local int myDevice = -1;  // pool threads start on device 0: the copy below names a stream of `dev`
            if (myDevice != dev) { opal_hip_set_device(dev); myDevice = dev; }
            const int i0 = cut[(size_t)t - 1], i1 = cut[(size_t)t];
            for (int i = i0; i < i1; i++)
                if (orderLens[i] > 0) memcpy(h + callerOff[(size_t)i], getSeq(i), (size_t)orderLens[i]);
            const uint64_t b0 = callerOff[(size_t)i0], b1 = callerOff[(size_t)i1];
            if (b1 > b0 && opal_hip_memcpy_h2d(d + b0, h + b0, b1 - b0, st0)) bad.fetch_add(1);
        });
        if (bad.load()) return false;
        hp.rawOff.resize(hp.pairSeq.size());
        for (size_t i = 0; i < hp.pairSeq.size(); i++) hp.rawOff[i] = hp.pairSeq[i] >= 0 ? callerOff[(size_t)hp.pairSeq[i]] : 0;
        hp.rawBytes = total;
        dRawEarly = d;
    }
    const size_t P = hp.pairOff.size();
    db->numSeqs = hp.numSeqs;
    db->numPairs = (int)P;
    db->residues = hp.residues;
    db->pairLen.swap(hp.pairLen);
    db->pairSeq.swap(hp.pairSeq);
    db->emptySeqs.swap(hp.emptySeqs);
    db->numTokens = hp.numTokens + 8;
    db->seqItemValid = false;
    db->planCache.clear();
    const double slack = db->cached ? 1.15 : 1.0;
    void* st = db->main();
    bool ok = st && db->tokens.ensure(db->numTokens, slack) && db->pairOff.ensure(P, slack) && db->dPairLen.ensure(2 * P, slack) &&
              db->dPairSeq.ensure(2 * P, slack) && db->ovfItems.ensure(2 * P, slack) &&
              db->rawOff.ensure(2 * P, slack);
    if (!ok) return false;
    if (!P) return true;
    const unsigned char* dRaw = nullptr;
    if (dRawEarly) {
        dRaw = dRawEarly;
    } else if (kind == RAW_POINTERS) {
        // rawOff = prefix sums of the lengths in pair order
        hp.rawOff.resize(2 * P);
        uint64_t total = 0;
        for (size_t i = 0; i < 2 * P; i++) { hp.rawOff[i] = total; total += (uint64_t)db->pairLen[i]; }
        hp.rawBytes = total;
        if (!db->raw.ensure(total + 16, slack) || !db->hRaw.ensure(total + 16, slack)) return false;
        // chunk boundaries: ~2 MB of residues each
        const uint64_t chunkBytes = 2u << 20;
        std::vector<size_t> cut{0};
        for (size_t i = 0; i < 2 * P; i++)
            if (hp.rawOff[i] - hp.rawOff[cut.back()] >= chunkBytes) cut.push_back(i);
        cut.push_back(2 * P);
        std::atomic<int> bad{0};
        unsigned char* h = db->hRaw.p;
        unsigned char* d = db->raw.p;
        const std::vector<int>& seq = db->pairSeq;
        const std::vector<int>& len = db->pairLen;
        const int dev = db->device;
        HostPool::get().parallel_for((int)cut.size() - 1, [&](int t) {
            static thread_local int myDevice = -1;  // pool threads start on device 0: the copy below names a stream of `dev`
            if (myDevice != dev) { opal_hip_set_device(dev); myDevice = dev; }
            const size_t i0 = cut[(size_t)t], i1 = cut[(size_t)t + 1];
            for (size_t i = i0; i < i1; i++)
                if (seq[i] >= 0 && len[i] > 0) memcpy(h + hp.rawOff[i], getSeq(seq[i]), (size_t)len[i]);
            const uint64_t b0 = hp.rawOff[i0], b1 = i1 < 2 * P ? hp.rawOff[i1] : total;
            if (b1 > b0 && opal_hip_memcpy_h2d(d + b0, h + b0, b1 - b0, st)) bad.fetch_add(1);
        });
        if (bad.load()) return false;
        dRaw = d;
    } else if (kind == RAW_HOST_FLAT) {
        if (!db->raw.ensure(hp.rawBytes + 16, slack)) return false;
        if (hp.rawBytes && opal_hip_memcpy_h2d(db->raw.p, flat, hp.rawBytes, st)) return false;
        dRaw = db->raw.p;
    } else {
        dRaw = flat;
    }
    ok = !opal_hip_memcpy_h2d(db->rawOff.p, hp.rawOff.data(), 2 * P * sizeof(uint64_t), st) &&
         !opal_hip_memcpy_h2d(db->pairOff.p, hp.pairOff.data(), P * sizeof(uint64_t), st) &&
         !opal_hip_memcpy_h2d(db->dPairLen.p, db->pairLen.data(), 2 * P * sizeof(int), st) &&
         !opal_hip_memcpy_h2d(db->dPairSeq.p, db->pairSeq.data(), 2 * P * sizeof(int), st) &&
         !opal_hip_pack_tokens(dRaw, db->rawOff.p, db->pairOff.p, db->dPairLen.p, (int)P, db->tokens.p, db->numTokens, st);
    // hp's vectors are pageable host memory read by the async copies above: wait before they go away
    if (opal_hip_stream_sync(st)) ok = false;
    if (ok && !db->cached) { db->raw.release(); db->rawOff.release(); db->hRaw.release(); }
    return ok;
}

// ------------------------------------------------------------------------------------ context cache
// Contexts borrowed by opalSearchDatabase / opalSearchDatabaseCharSW: per device, a free list. A call takes one
// (or creates it), fills it with its database, searches, and gives it back with all buffers kept. They are
// never destroyed implicitly (opalx_release_cached() frees them on request): freeing device memory from a
// static destructor races with the HIP runtime's own shutdown.
struct ContextCache {
    std::mutex m;
    std::map<int, std::vector<OpalxDb*>> freeList;
    OpalxDb* acquire(int device) {
        {
            std::lock_guard<std::mutex> lk(m);
            auto& v = freeList[device];
            if (!v.empty()) { OpalxDb* db = v.back(); v.pop_back(); return db; }
        }
        OpalxDb* db = new (std::nothrow) OpalxDb;
        if (db) { db->cached = true; db->device = device; }
        return db;
    }
    void give_back(OpalxDb* db) {
        std::lock_guard<std::mutex> lk(m);
        freeList[db->device].push_back(db);
    }
    static ContextCache& get() {
        static ContextCache* c = new ContextCache;
        return *c;
    }
};

struct BorrowedContext {
    OpalxDb* db;
    explicit BorrowedContext(int device) : db(ContextCache::get().acquire(device)) {}
    ~BorrowedContext() {
        if (db) { db->quiesce(); ContextCache::get().give_back(db); }
    }
};

// ------------------------------------------------------------------------------------ scheduling
struct LaunchClass { int first, count, G, rows, wpb, ldsPad; };

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
static int kernel_vgprs(int width, int mode, int swEndPass, int multi, int skew, int rows) {
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
static int pick_occupancy(int vgprs, size_t ldsBytes, int* wpbOut, int* ldsPadOut, int maxWpb = 8) {
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
 * Split the item range [first, first+count) (sorted by length, longest first) into classes of
 * equal wave geometry. Each sequence (pair) is spread over G lanes; a wave therefore needs
 * len+G-1 column steps of `rows` rows each. Short sequences want a small G (little fill/drain and
 * per-step overhead), the few very long ones a large G so that no single wave outlives the rest of
 * the launch. `lenOf(i)` is the length of item i.
 */
template <typename LenOf>
static std::vector<LaunchClass> plan_classes(int width, int mode, int swEndPass, int skew, int Q, int A, LenOf lenOf, int first,
                                             int count, int maxGFactor = 0) {
    std::vector<LaunchClass> out;
    if (count <= 0) return out;
    struct Cand { int G, rows, wpb, ldsPad; double eff; };
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
        // rows per lane: the smallest instantiated value, or up to 3 more if that one compiles to a register count
        // that leaves fewer waves per SIMD (a few padding rows cost less than an odd or halved occupancy)
        Cand best{0, 0, 4, 0, -1.0};
        int prevRows = 0;
        for (int want = need; want <= need + 3 && want <= OPAL_HIP_MAX_ROWS; want++) {
            const int rows = opal_hip_rows_supported(width, mode, swEndPass | (skew ? 4 : 0), want);
            if (!rows || rows == prevRows) continue;
            prevRows = rows;
            const size_t lds = opal_hip_lds_bytes(width, G, rows, A);
            if (lds > 160 * 1024) continue;
            if (G > 1 && (G / 2) * rows >= Q && !(forceG && *forceG)) continue;  // half the lanes would hold padding only
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
    // one-launch kernels hold the bodies for G0, 2*G0 and 4*G0 lanes per sequence: keep the classes inside that chain
    if (maxGFactor > 0)
        while (cands.size() > base + 1 && cands.back().G > maxGFactor * cands[base].G) cands.pop_back();
    // time of the whole range at the base geometry if the machine were perfectly filled
    const double simds = 1024.0;
    const int slotsBase = 64 / cands[base].G;
    const double totalClocks = lenSum * sample / slotsBase * step_clocks(width, cands[base].rows) / simds;
    const char* budEnv = getenv("OPAL_HIP_TAIL_FRACTION");
    const double budget = std::max(totalClocks * (budEnv && *budEnv ? atof(budEnv) : 0.7), 20000.0);
    // walk from the longest item: the smallest geometry (>= base) whose lone-wave time fits the budget
    size_t ci = cands.size() - 1;
    int start = first;
    // A class of long sequences is a handful of waves, each the oldest wave of its SIMD and therefore running at the
    // lone-wave rate: that is the time a boundary candidate is judged by. (Round 2, C2 database, 100 000 / 50 000
    // sequences: 1.594 -> 1.567 ms and 1.013 -> 0.967 ms against the pipe-rate model; OPAL_HIP_TAIL_MODEL=pipe restores it.)
    const char* tm = getenv("OPAL_HIP_TAIL_MODEL");
    const bool pipeModel = tm && !strcmp(tm, "pipe");
    auto fits = [&](size_t c, int len) {
        const double per = pipeModel ? step_clocks(width, cands[c].rows) : lone_step_clocks(width, cands[c].rows);
        return (len + cands[c].G - 1) * per <= budget;
    };
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
    // process had created: scripts/gpu_queue_test.py). The side streams are therefore HIGH-priority streams, which
    // get queues of their own; with them three side streams work too but measure 1-3 % slower than two.
    const char* nsEnv = getenv("OPAL_HIP_SIDE_STREAMS");  // tuning hook: 3 side streams
    const size_t maxClasses = (nsEnv && atoi(nsEnv) == 3) ? 4 : 3;
    while (out.size() > maxClasses) {  // fold the two longest classes together
        out[1].count += out[0].count;
        out[1].first = out[0].first;
        out[1].G = out[0].G; out[1].rows = out[0].rows; out[1].wpb = out[0].wpb; out[1].ldsPad = out[0].ldsPad;
        out.erase(out.begin());
    }
    return out;
}

static void fill_launch(OpalHipLaunch& a, const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work,
                        int* ovfCounter) {
    memset(&a, 0, sizeof a);
    OpalxDb* db = c.db;
    a.width = width; a.mode = c.mode; a.want_end = swEndPass; a.detect_wrap = detectWrap;
    a.query = db->query.p; a.query_length = c.Q; a.score_matrix = db->matrix.p; a.alphabet_length = c.A;
    a.gap_open = c.Go; a.gap_ext = c.Ge;
    a.tokens = db->tokens.p; a.pair_token_offset = db->pairOff.p; a.pair_len = db->dPairLen.p; a.pair_seq = db->dPairSeq.p;
    a.work = work;
    a.out_score = c.dScore; a.out_end_query = c.wantEnd ? c.dEndQ : nullptr; a.out_end_target = c.wantEnd ? c.dEndT : nullptr;
    a.overflow_count = ovfCounter; a.overflow_items = db->ovfItems.p;
    a.sat_limit = c.satLimit; a.neg_pad = c.negPad;
    a.stream = db->main();
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
    if (!db->boundary.ensure(16 * (db->numTokens + 8)) || !db->scratch.ensure(2 * (size_t)std::max(db->numSeqs, 1)))
        return OPAL_ERR_NO_SIMD_SUPPORT;
    int wpb = 4, ldsPad = 0;
    pick_occupancy(kernel_vgprs(width, c.mode, swEndPass, 1, 0, bestS), opal_hip_lds_bytes(width, bestG, bestS, c.A), &wpb, &ldsPad);
    // Equal chunks: with nchunks passes needed anyway, the smallest rows-per-lane value that still covers Q in that many
    // passes (a 3000-aa query ran as 440 + 2560 rows at 40 rows per lane, 2 900 GCUPS; as 2 x 1500 at 24 rows per lane ...)
    if (!getenv("OPAL_HIP_CHUNK_GEOMETRY")) {
        const int passes = (c.Q + bestG * bestS - 1) / (bestG * bestS);
        const int need = ((c.Q + passes - 1) / passes + bestG - 1) / bestG;  // rows per lane of a balanced chunk
        const int rows = opal_hip_rows_supported(width, c.mode, swEndPass | 2, need);
        if (rows && rows < bestS) bestS = rows;
    }
    const int cap = bestG * bestS;
    const int nchunks = (c.Q + cap - 1) / cap;
    const int firstRows = c.Q - (nchunks - 1) * cap;  // SW/HW/OV: the partial chunk is the first one
    for (int ch = 0; ch < nchunks; ch++) {
        int row0, rows;
        if (c.mode == OPAL_MODE_NW) { row0 = ch * cap; rows = std::min(cap, c.Q - row0); }
        else { row0 = ch == 0 ? 0 : firstRows + (ch - 1) * cap; rows = ch == 0 ? firstRows : cap; }
        OpalHipLaunch a;
        fill_launch(a, c, width, swEndPass, detectWrap, work, ovfCounter);
        a.lanes_per_seq = bestG; a.rows_per_lane = bestS; a.waves_per_block = wpb;
        a.first_item = firstItem; a.num_items = numItems;
        a.chunk_flags = (ch > 0 ? 1 : 0) | (ch < nchunks - 1 ? 2 : 0);
        a.chunk_row0 = row0; a.chunk_rows = rows;
        a.boundary = db->boundary.p; a.scratch = db->scratch.p;
        if (debug_on())
            fprintf(stderr, "[opal] chunk %d/%d rows [%d,+%d) width=%d mode=%d end=%d G=%d rows/lane=%d waves/block=%d\n", ch + 1,
                    nchunks, row0, rows, width, c.mode, swEndPass, bestG, bestS, wpb);
        if (opal_hip_launch(&a)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    return 0;
}

template <typename LenOf>
static int launch_range(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, int firstItem,
                        int numItems, LenOf lenOf, int* ovfCounter) {
    if (numItems <= 0) return 0;
    OpalxDb* db = c.db;
    // skewed kernels (width 17, NW/HW/OV) add up to (rows + steps) * Ge to every value: allowed for a class when
    // even its longest sequence stays inside the 16-bit range
    auto skew_ok = [&](int item) {
        const long long L0 = lenOf(item);
        return std::min<long long>(c.Q, L0) * c.maxScore + (OPAL_HIP_MAX_ROWS + L0 + 64 + 4) * (long long)c.Ge < 32000;
    };
    const bool maySkew = width == 17 && c.mode != OPAL_MODE_SW && !work && !getenv("OPAL_HIP_NO_SKEW");
    // One launch for all classes where a one-launch kernel exists (plain 16-bit, no re-run list): see dp_kernel.hpp.
    const bool uberEligible = width == 17 && !work && !getenv("OPAL_HIP_NO_UBER") && !getenv("OPAL_HIP_LANES_PER_SEQ");
    std::vector<LaunchClass> classes;
    std::string key;
    if (!work) {  // (re-run lists differ from search to search: planned every time)
        char buf[160];
        const char* e1 = getenv("OPAL_HIP_LANES_PER_SEQ"); const char* e2 = getenv("OPAL_HIP_TAIL_FRACTION"); const char* e3 = getenv("OPAL_HIP_TAIL_MODEL");
        snprintf(buf, sizeof buf, "%d/%d/%d/%d/%d/%d/%d/%lld/%d/%d/%d|%s|%s|%s", width, c.mode, swEndPass, c.Q, c.A, c.Go, c.Ge, c.maxScore, firstItem, numItems,
                 (int)uberEligible, e1 ? e1 : "", e2 ? e2 : "", e3 ? e3 : "");
        key = buf;
        auto it = db->planCache.find(key);
        if (it != db->planCache.end()) {
            const std::vector<int>& v = it->second;
            for (size_t i = 0; i + 6 <= v.size(); i += 6) classes.push_back({v[i], v[i + 1], v[i + 2], v[i + 3], v[i + 4], v[i + 5]});
        }
    }
    if (classes.empty()) {
        classes = plan_classes(width, c.mode, swEndPass, maySkew && skew_ok(firstItem + numItems - 1), c.Q, c.A, lenOf, firstItem, numItems,
                               uberEligible ? 4 : 0);
        if (!work && !classes.empty()) {
            std::vector<int>& v = db->planCache[key];
            v.clear();
            for (const LaunchClass& k : classes) v.insert(v.end(), {k.first, k.count, k.G, k.rows, k.wpb, k.ldsPad});
        }
    }
    if (classes.empty() || getenv("OPAL_HIP_FORCE_CHUNKS"))
        return launch_chunked(c, width, swEndPass, detectWrap, work, firstItem, numItems, ovfCounter);
    if (uberEligible && classes.size() >= 2 && classes.size() <= 3) {
        const LaunchClass& bulk = classes.back();
        const int skewBulk = maySkew && skew_ok(bulk.first);
        bool ok = bulk.wpb == 4 && bulk.ldsPad == 0 && opal_hip_uber_supported(c.mode, swEndPass, skewBulk, bulk.rows);
        OpalHipUberLaunch u;
        memset(&u, 0, sizeof u);
        for (size_t ki = 0; ok && ki + 1 < classes.size(); ki++) {
            const LaunchClass& k = classes[ki];
            if ((maySkew && skew_ok(k.first)) != skewBulk) ok = false;  // a class that cannot use the skewed kernels: separate launches
            else if (k.G == 2 * bulk.G) { u.long_first_item = k.first; u.long_num_items = k.count; u.long_priority = 1; }
            else if (k.G == 4 * bulk.G) { u.longest_first_item = k.first; u.longest_num_items = k.count; u.longest_priority = 1; }
            else ok = false;
        }
        if (ok) {
            fill_launch(u.base, c, width, swEndPass, detectWrap, nullptr, ovfCounter);
            u.base.lanes_per_seq = bulk.G; u.base.rows_per_lane = bulk.rows; u.base.waves_per_block = 4;
            u.base.first_item = bulk.first; u.base.num_items = bulk.count; u.base.skew = skewBulk;
            if (debug_on())
                fprintf(stderr, "[opal] one launch width=%d mode=%d end=%d G0=%d rows=%d%s: bulk [%d,+%d) long [%d,+%d) longest [%d,+%d)\n", width,
                        c.mode, swEndPass, bulk.G, bulk.rows, skewBulk ? " skew" : "", bulk.first, bulk.count, u.long_first_item,
                        u.long_num_items, u.longest_first_item, u.longest_num_items);
            if (opal_hip_launch_uber(&u)) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
            return 0;
        }
    }
    // The last class (the bulk of the database) runs on the main stream; the classes of long
    // sequences run beside it on up to three side streams (one hardware queue each), forked from and
    // joined back into the main stream with events, longest sequences enqueued first.
    const int nside = std::min<int>((int)classes.size() - 1, 3);
    void* fork = nullptr;
    if (nside > 0) {
        fork = db->event(0);
        if (!fork || opal_hip_event_record(fork, db->main())) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    int err = 0;
    bool recorded[3] = {false, false, false};
    // Submission order matters: the side launches (long sequences) go first. Submitting the bulk launch first was
    // measured at 2.2 ms instead of 1.55 ms on C2: its workgroups fill every CU and the long sequences start late.
    for (size_t ki = 0; ki < classes.size() && !err; ki++) {
        LaunchClass k = classes[ki];
        const int skew = maySkew && skew_ok(k.first);
        {   // the plan assumed the skewed kernels (every rows-per-lane value exists); a class that cannot use them
            // takes the plain kernel's next instantiated value
            const int rows = opal_hip_rows_supported(width, c.mode, swEndPass | (skew ? 4 : 0), k.rows);  // k.rows if it exists
            if (rows && rows != k.rows && opal_hip_lds_bytes(width, k.G, rows, c.A) <= 160 * 1024) {
                k.rows = rows;
                pick_occupancy(kernel_vgprs(width, c.mode, swEndPass, 0, skew, rows), opal_hip_lds_bytes(width, k.G, rows, c.A), &k.wpb, &k.ldsPad,
                               opal_hip_max_waves_per_block(width));
            }
        }
        OpalHipLaunch a;
        fill_launch(a, c, width, swEndPass, detectWrap, work, ovfCounter);
        a.lanes_per_seq = k.G; a.rows_per_lane = k.rows; a.waves_per_block = k.wpb; a.lds_bytes_min = k.ldsPad;
        a.first_item = k.first; a.num_items = k.count;
        a.skew = skew;
        const bool side = (int)ki < nside;
        if (side) a.stream = db->stream((int)ki);
        a.priority = side ? 1 : 0;
        if (const char* pe = getenv("OPAL_HIP_SIDE_PRIO")) a.priority = side ? atoi(pe) : 0;  // tuning hook
        if (debug_on())
            fprintf(stderr, "[opal] launch width=%d mode=%d end=%d items=[%d,+%d) G=%d rows=%d waves/block=%d ldsPad=%d%s%s\n", width,
                    c.mode, swEndPass, k.first, k.count, k.G, k.rows, k.wpb, k.ldsPad, skew ? " skew" : "", side ? " (side stream)" : "");
        if (side && (!a.stream || opal_hip_stream_wait_event(a.stream, fork))) { err = 1; break; }
        if (opal_hip_launch(&a)) { err = 1; break; }
        if (side) {
            void* done = db->event(1 + (int)ki);
            if (!done || opal_hip_event_record(done, a.stream)) { err = 1; break; }
            recorded[ki] = true;
        }
    }
    for (int i = 0; i < 3; i++)
        if (recorded[i] && opal_hip_stream_wait_event(db->main(), db->event(1 + i))) err = 1;
    if (err) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }  // nothing of this call stays in flight
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

// Query, score matrix and zeroed counters reach the device as ONE copy from pinned memory (three separate small
// copies and a memset cost ~10 us of stream time each, on a search whose kernels take 1.6 ms).
static int ensure_query_buffers(OpalxDb* db, const unsigned char* query, int Q, const int* M, int A) {
    const size_t mb = (size_t)A * A * sizeof(int);
    const size_t bytes = (16 + mb + (size_t)std::max(Q, 1) + 15) / 16 * 16;
    if (!db->header.ensure(bytes, 1.5) || !db->hHeader.ensure(bytes, 1.5)) return 1;
    unsigned char* h = db->hHeader.p;
    memset(h, 0, 16);
    memcpy(h + 16, M, mb);
    if (Q) memcpy(h + 16 + mb, query, (size_t)Q);
    if (opal_hip_memcpy_h2d(db->header.p, h, 16 + mb + (size_t)Q, db->main())) return 1;
    db->ovfCount.p = (int*)db->header.p;
    db->matrix.p = (int*)(db->header.p + 16);
    db->query.p = db->header.p + 16 + mb;
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

// {caller index, score, endQ, endT} quads -> one upload, one scatter kernel
static int write_degenerate(OpalxDb* db, const std::vector<int>& quads, int* dScore, int* dEndQ, int* dEndT) {
    const int count = (int)(quads.size() / 4);
    if (!count) return 0;
    if (!db->quads.ensure(quads.size(), 1.5)) return 1;
    if (opal_hip_memcpy_h2d(db->quads.p, quads.data(), quads.size() * sizeof(int), db->main())) return 1;
    if (opal_hip_scatter_results(db->quads.p, count, dScore, dEndQ, dEndT, db->main())) return 1;
    return opal_hip_stream_sync(db->main());  // `quads` is pageable host memory of the caller's frame
}

// Streaming SW score search (stream_kernel.hpp): which geometry, if any. Long queries need 32 or 64 lanes per sequence,
// where the class kernels lose (G-1)/(L+G-1) of their steps to pipeline fill and drain; the streaming kernel has no
// such loss, so it is chosen by the cost of a column step per useful query row alone.
struct StreamPlan { int width = 0, G = 0, rows = 0; };
static StreamPlan plan_stream(int Q, int A, int mode, int wantEnd, long long residues, int longest, int numSeqs) {
    StreamPlan best;
    const char* en = getenv("OPAL_HIP_STREAM");          // "0": never, "1": whenever a geometry exists (tests), unset: long queries
    if (en && *en == '0') return best;
    const bool force = en && *en == '1';
    const char* forceG = getenv("OPAL_HIP_LANES_PER_SEQ");
    const char* forceW = getenv("OPAL_HIP_NARROW");
    double bestCost = 1e30;
    for (int width : {17, 18}) {
        if (forceW && *forceW && atoi(forceW) != width && (atoi(forceW) == 17 || atoi(forceW) == 18)) continue;
        for (int G = 1; G <= 64; G *= 2) {
            if (forceG && *forceG && atoi(forceG) != G) continue;
            const int need = (Q + G - 1) / G;
            if (need > OPAL_HIP_MAX_ROWS) continue;
            const int rows = opal_hip_stream_rows_supported(width, mode, wantEnd, need);
            if (!rows || opal_hip_stream_lds_bytes(width, mode, G, rows, A) > 160 * 1024) continue;
            const double perRow = (mode == OPAL_MODE_SW ? 9.0 + (wantEnd ? 1.5 : 0.0) : 6.0) + (width == 18 ? 0.5 : 0.0);
            const double cost = (perRow * rows + 30.0) * G / Q;
            if (cost < bestCost) { bestCost = cost; best.width = width; best.G = G; best.rows = rows; }
        }
    }
    if (best.G && !force && (wantEnd || mode != OPAL_MODE_SW) && best.G < 32) best = StreamPlan();  // (end locations, NW/HW/OV at 16 lanes or fewer: class kernels, unmeasured otherwise)
    if (best.G && !force) {
        // A sequence stays on its G lanes from its first column to its last: the longest one (handed out first) sets a
        // floor of (longest + G) steps at the rate of a wave that ends up alone on its SIMD (~3.5 clk per instruction).
        // On a small database that floor exceeds the time of the whole search and the class kernels (4*G lanes for
        // the longest sequences) win: measured with 20 000 sequences, 1000-aa query 2.6 ms against 1.7 ms; with 100 000
        // sequences 4.9 ms against 7.0 ms.
        const double stepInstr = (best.width == 18 ? 9.5 : 9.0) * best.rows + 30.0;
        const double total = (double)residues * best.G / 64.0 * stepInstr * 2.15 / 1024.0;
        const double crit = ((double)longest + best.G) * stepInstr * 3.5;
        if (best.G >= 32) {
            if (crit > 1.5 * total) best = StreamPlan();
        } else {
            // 16 lanes per sequence or fewer: the class kernels lose little to fill and drain, and streaming only wins
            // (2-4 %: 513-aa query x 625 000 sequences 6 790 -> 7 050 GCUPS, 300-aa x 1 000 000 7 230 -> 7 390) when every
            // slot works through many sequences; with ~3 per slot (300-aa x 100 000) it is 14 % slower.
            const double perSlot = (double)numSeqs / (4096.0 * (64 / best.G));
            if (crit > 0.5 * total || perSlot < 20.0) best = StreamPlan();  // (300-aa query: break-even at ~18 sequences per slot, 600 000 sequences)
        }
    }
    return best;
}

/*
 * The scheduler. Plays the role of the reference's precision ladder (opal.cpp:496-535, :983-1021)
 * with the rungs gfx950 has: 16-bit first, 32-bit for whatever saturates or cannot be
 * proven safe. OPAL_OVERFLOW_SIMPLE and OPAL_OVERFLOW_BUCKETS give identical results in the
 * reference (only the re-run granularity differs); here both map to the same schedule.
 */
static int search_core(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A,
                       int searchType, int mode, int* dScore, int* dEndQ, int* dEndT, float* kernelMs, bool finalSync = true) {
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
    void* const st = db->main();
    if (!st) return OPAL_ERR_NO_SIMD_SUPPORT;
    int rc = 0;

    // ---- degenerate shapes: answered from the matrix boundary
    {
        std::vector<int> quads;
        auto add = [&](int idx, int L) {
            int s, eq, et;
            degenerate_result(mode, Q, L, Go, Ge, &s, &eq, &et);
            if (!wantEnd) { eq = et = -1; }
            quads.insert(quads.end(), {idx, s, eq, et});
        };
        for (int idx : db->emptySeqs) add(idx, 0);
        if (Q == 0)
            for (int i = 0; i < 2 * P; i++)
                if (db->pairSeq[i] >= 0) add(db->pairSeq[i], db->pairLen[i]);
        if (write_degenerate(db, quads, dScore, dEndQ, dEndT)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    if (Q == 0 || P == 0) return opal_hip_stream_sync(st) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;

    if (ensure_query_buffers(db, query, Q, M, A)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (also zeroes the counters)

    long long maxS = 0, minS = 0;
    for (int i = 0; i < A * A; i++) { maxS = std::max<long long>(maxS, M[i]); minS = std::min<long long>(minS, M[i]); }
    bool fits16 = Go <= 32767 && Ge <= 32767 && maxS <= 32767 && minS >= -32768;
    if (const char* e = getenv("OPAL_HIP_FORCE_WIDTH")) { if (atoi(e) == 32) fits16 = false; }  // test / profiling hook

    // Narrow rung: 17 = plain 16-bit, one sequence per lane (full-rate VALU, 4-byte LDS profile entries); 16 = packed
    // pairs (half-rate VALU, 2-byte entries: queries whose 4-byte profile exceeds LDS); 18 = plain with 2-byte entries
    // (+0.5 instructions per cell; experimental, on request only).
    int narrow = 17;
    if (const char* e = getenv("OPAL_HIP_NARROW")) { const int v = atoi(e); narrow = (v == 16 || v == 18) ? v : 17; }
    {
        auto fits = [&](int width) {  // does some single-launch geometry hold the query at this width?
            for (int G = 1; G <= 64; G *= 2) {
                const int need = (Q + G - 1) / G;
                if (need > OPAL_HIP_MAX_ROWS) continue;
                const int rows = opal_hip_rows_supported(width, mode, 0, need);
                if (rows && opal_hip_lds_bytes(width, G, rows, A) <= 160 * 1024) return true;
            }
            return false;
        };
        const char* force = getenv("OPAL_HIP_NARROW_FORCE");  // "17": run width 17 in row chunks rather than switching width
        // (width 18 measured SLOWER than the packed pairs on the 2000-aa workload, 15.1 vs 14.8 ms, so it is never
        // chosen automatically: OPAL_HIP_NARROW=18 selects it for experiments)
        if (narrow == 17 && !fits(17) && !(force && atoi(force) == 17)) {
            narrow = 16;
            // ... except where the plain kernels with 2-byte profile entries measure faster (2000-aa query, class kernels,
            // 100 000 / 20 000 sequences: SW + end 4 150 against 3 350 / 3 720 against 2 970 GCUPS, OV 4 270 : 4 130 / 3 810 : 3 660,
            // NW and HW 4 970 : 5 100 / 4 400 : 4 040)
            if (!getenv("OPAL_HIP_NARROW") && fits(18) &&
                ((mode == OPAL_MODE_SW && wantEnd) || mode == OPAL_MODE_OV || (mode != OPAL_MODE_SW && db->numSeqs < 50000)))
                narrow = 18;
        }
        if (narrow == 18 && !fits(18)) narrow = 16;
    }
    // SW score-only searches with long queries stream their sequences through persistent waves (plain 16-bit arithmetic)
    StreamPlan splan;
    if (mode == OPAL_MODE_SW && fits16) splan = plan_stream(Q, A, mode, wantEnd, db->residues, db->pairLen.empty() ? 0 : db->pairLen[0], db->numSeqs);
    const bool plain = narrow != 16 || splan.G;
    const int satLimit = plain ? (int)(32767 - maxS) : 32767;
    SearchCtx c{db, Q, A, Go, Ge, mode, wantEnd, dScore, dEndQ, dEndT, satLimit, mode == OPAL_MODE_SW ? -satLimit : 0, maxS};
    const int nNarrow = narrow == 16 ? P : 2 * P;  // items of the narrow rung: pairs or sequences
    auto pairLenOf = [db](int p) { return db->pairLen[2 * (size_t)p]; };   // width-16 items are pairs
    auto seqLenOf = [db](int it) { return db->pairLen[(size_t)it]; };      // width-17/32 items are pair*2+half
    void* evStart = nullptr; void* evStop = nullptr;
    if (kernelMs) { evStart = db->event(4); evStop = db->event(5); }
    if (evStart) opal_hip_event_record(evStart, st);

    if (mode == OPAL_MODE_SW) {
        // rung 1: everything at 16 bits (if the inputs fit), saturated scores go to the re-run list.
        // End locations are found in the same pass (the kernels record where a lane's running maximum
        // was first reached).
        int n32 = 0;
        const int* work32 = nullptr;
        std::vector<int> rerun;  // the re-run list, longest sequence first
        if (fits16) {
            if (splan.G) {
                OpalHipLaunch a;
                fill_launch(a, c, splan.width, wantEnd, 0, nullptr, db->ovfCount.p);
                a.lanes_per_seq = splan.G; a.rows_per_lane = splan.rows;
                a.first_item = 0; a.num_items = 2 * P;
                if (debug_on())
                    fprintf(stderr, "[opal] streaming launch width=%d G=%d rows=%d end=%d items=%d\n", splan.width, splan.G, splan.rows, wantEnd, 2 * P);
                rc = opal_hip_launch_stream(&a, db->ovfCount.p + 2) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
                g_streamLaunches.fetch_add(1);
            }
            else if (narrow == 16) rc = launch_range(c, 16, wantEnd, 0, nullptr, 0, nNarrow, pairLenOf, db->ovfCount.p);
            else rc = launch_range(c, narrow, wantEnd, 0, nullptr, 0, nNarrow, seqLenOf, db->ovfCount.p);
            if (rc) return rc;
            int cnt = 0;
            // The re-run list can only be non-empty if some score can reach the rung's limit at all: with
            // min(Q, longest sequence) * max(matrix) below it (a 300-aa query on BLOSUM50: 4 500 against 32 752) the
            // count need not be read back, and the search has no host round trip between its launches and its end.
            const bool canSaturate = (long long)std::min<long long>(Q, db->pairLen[0]) * maxS >= (long long)satLimit;
            if (canSaturate && (opal_hip_memcpy_d2h(&cnt, db->ovfCount.p, sizeof(int), st) || opal_hip_stream_sync(st)))
                return OPAL_ERR_NO_SIMD_SUPPORT;
            if (cnt > 0) {
                // The kernels append in completion order: sort the list by length so that the width-32 launch
                // classes get the geometry of the sequences they really hold (not the database's longest).
                rerun.resize((size_t)cnt);
                if (opal_hip_memcpy_d2h(rerun.data(), db->ovfItems.p, (size_t)cnt * sizeof(int), st) || opal_hip_stream_sync(st))
                    return OPAL_ERR_NO_SIMD_SUPPORT;
                std::sort(rerun.begin(), rerun.end(), [&](int a, int b) {
                    const int la = db->pairLen[(size_t)a], lb = db->pairLen[(size_t)b];
                    return la != lb ? la > lb : a < b;
                });
                if (opal_hip_memcpy_h2d(db->ovfItems.p, rerun.data(), (size_t)cnt * sizeof(int), st) || opal_hip_stream_sync(st))
                    return OPAL_ERR_NO_SIMD_SUPPORT;
            }
            n32 = cnt; work32 = db->ovfItems.p;
        } else {
            n32 = 2 * P; work32 = nullptr;  // items pair*2+half enumerate every sequence
        }
        // rung 2: width 32, with the reference's wrap test when a wrap is arithmetically possible
        if (n32 > 0) {
            const long long maxLen = db->pairLen[0];
            const int detect = (double)std::min<long long>(Q, maxLen) * (double)maxS > (double)INT_MAX;
            if (work32) {
                auto rerunLenOf = [&](int i) { return db->pairLen[(size_t)rerun[(size_t)i]]; };
                rc = launch_range(c, 32, wantEnd, detect, work32, 0, n32, rerunLenOf, db->ovfCount.p + 1);
            } else {
                rc = launch_range(c, 32, wantEnd, detect, nullptr, 0, n32, seqLenOf, db->ovfCount.p + 1);
            }
            if (rc) return rc;
            if (detect) {
                int wraps = 0;
                if (opal_hip_memcpy_d2h(&wraps, db->ovfCount.p + 1, sizeof(int), st) || opal_hip_stream_sync(st))
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
        // Long queries on large databases: the 16-bit part as a stream of sequences through persistent waves
        // (stream_kernel.hpp, diagonal-skewed kernels). The few sequences whose skewed values could leave 16 bits (the longest
        // ones) run on the unskewed class body in the first workgroups of the same launch; if they hold more than 2 % of the
        // residues the class kernels run everything.
        StreamPlan gplan;
        int q32 = p32;  // pairs [p32, q32): 16-bit, but not safe for the skewed stream kernels -> class kernels beside the stream
        if (fits16 && p32 < P) {
            long long moved = 0;
            auto skew_safe = [&](long long L) {
                return std::min<long long>(Q, L) * maxS + (OPAL_HIP_MAX_ROWS + L + 64 + 4) * (long long)Ge < 32000;
            };
            while (q32 < P && !skew_safe(db->pairLen[2 * (size_t)q32])) { moved += db->pairLen[2 * (size_t)q32] + db->pairLen[2 * (size_t)q32 + 1]; q32++; }
            const char* sen = getenv("OPAL_HIP_STREAM");
            if (q32 < P && (moved * 50 <= db->residues || (sen && *sen == '1')) && !getenv("OPAL_HIP_NO_SKEW")) {
                gplan = plan_stream(Q, A, mode, wantEnd, db->residues - moved, db->pairLen[2 * (size_t)q32], db->numSeqs - 2 * q32);
            }
            if (!gplan.G) q32 = p32;
        }
        auto launch_global_stream = [&]() -> int {
            OpalHipLaunch a;
            fill_launch(a, c, gplan.width, 0, 0, nullptr, db->ovfCount.p);
            a.lanes_per_seq = gplan.G; a.rows_per_lane = gplan.rows; a.skew = 1;
            a.first_item = 2 * q32; a.num_items = 2 * (P - q32);
            // the skew-unsafe sequences (the longest ones) ride along in the same launch on the unskewed class body
            a.side_first_item = 2 * p32; a.side_num_items = 2 * (q32 - p32);
            if (debug_on())
                fprintf(stderr, "[opal] streaming launch mode=%d width=%d G=%d rows=%d items=[%d,+%d)\n", mode, gplan.width, gplan.G, gplan.rows,
                        a.first_item, a.num_items);
            g_streamLaunches.fetch_add(1);
            return opal_hip_launch_stream(&a, db->ovfCount.p + 2);
        };
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
        if (2 * p32 > nBad && launch_range(c, 32, 0, 0, nullptr, nBad, 2 * p32 - nBad, seqLenOf, db->ovfCount.p + 1))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        if (P > q32) {
            const int e = gplan.G ? launch_global_stream()
                        : narrow == 16 ? launch_range(c, 16, 0, 0, nullptr, p32, P - p32, pairLenOf, db->ovfCount.p)
                                       : launch_range(c, narrow, 0, 0, nullptr, 2 * p32, 2 * (P - p32), seqLenOf, db->ovfCount.p);
            if (e) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        }
    }
    if (evStop) opal_hip_event_record(evStop, st);
    // (a caller that copies the results back on the same stream and waits for that copy needs no wait of its own here)
    if ((finalSync || kernelMs) && opal_hip_stream_sync(st)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (kernelMs && evStart && evStop) *kernelMs = opal_hip_event_elapsed_ms(evStart, evStop);
    return rc;
}

// Start cells on the device for every sequence of `db` (see include/opal_hip.h, opalx_search_starts).
static int search_starts(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A, int mode,
                         const int* dScore, const int* dEndQ, const int* dEndT, int* dStartQ, int* dStartT, int* redoHost,
                         int* redoCount) {
    if (redoCount) *redoCount = 0;
    const int n = db->numSeqs;
    if (n <= 0) return 0;
    if (Q < 1 || !opal_hip_rev_rows_supported(Q) || (size_t)(A + 1) * A * sizeof(int) > 64 * 1024) return OPAL_ERR_NO_SIMD_SUPPORT;
    void* st = db->main();
    if (!st) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (!db->seqItemValid) {
        std::vector<int> inv((size_t)n, 0);
        for (size_t i = 0; i < db->pairSeq.size(); i++)
            if (db->pairSeq[i] >= 0) inv[(size_t)db->pairSeq[i]] = (int)i;
        if (!db->seqItem.ensure((size_t)n, db->cached ? 1.15 : 1.0)) return OPAL_ERR_NO_SIMD_SUPPORT;
        if (opal_hip_memcpy_h2d(db->seqItem.p, inv.data(), (size_t)n * sizeof(int), st) || opal_hip_stream_sync(st))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        db->seqItemValid = true;
    }
    if (!db->revFlags.ensure(4) || !db->redo.ensure(2 * (size_t)n, db->cached ? 1.15 : 1.0)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (ensure_query_buffers(db, query, Q, M, A)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (opal_hip_memset(db->revFlags.p, 0, 4 * sizeof(int), st)) return OPAL_ERR_NO_SIMD_SUPPORT;
    OpalHipRevLaunch a;
    memset(&a, 0, sizeof a);
    a.mode = mode; a.query = db->query.p; a.query_length = Q; a.score_matrix = db->matrix.p; a.alphabet_length = A;
    a.gap_open = Go; a.gap_ext = Ge;
    a.tokens = db->tokens.p; a.pair_token_offset = db->pairOff.p; a.seq_item = db->seqItem.p;
    a.items = nullptr; a.num_items = n;
    a.score = dScore; a.end_query = dEndQ; a.end_target = dEndT;
    a.start_query = dStartQ; a.start_target = dStartT; a.flags = db->revFlags.p; a.redo = db->redo.p;
    a.stream = st;
    if (opal_hip_rev_launch(&a)) return OPAL_ERR_NO_SIMD_SUPPORT;
    int flags[2] = {0, 0};
    if (opal_hip_memcpy_d2h(flags, db->revFlags.p, sizeof flags, st) || opal_hip_stream_sync(st)) return OPAL_ERR_NO_SIMD_SUPPORT;
    // empty sequences have no token stream and no end cell: nothing to start from (the kernel saw endT = -1)
    const int doubtful = flags[0];
    int left = doubtful;
    const int* leftList = db->redo.p;
    if (doubtful > 0 && !getenv("OPAL_HIP_NO_EXACT_ROW")) {
        // Second pass over the doubtful ones (OV: a cell that is not an allowed end held the score in the stop column): their
        // stop column is final, the reference's row scan of that column is repeated exactly on the device. Only sequences
        // without a stop column at all stay for the host.
        a.items = db->redo.p; a.num_items = doubtful; a.exact_row = 1;
        a.flags = db->revFlags.p + 2; a.redo = db->redo.p + n;
        int flags2[2] = {0, 0};
        if (opal_hip_rev_launch(&a) || opal_hip_memcpy_d2h(flags2, db->revFlags.p + 2, sizeof flags2, st) || opal_hip_stream_sync(st))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        left = flags2[0];
        leftList = db->redo.p + n;
    }
    if (left > 0 && redoHost) {
        if (opal_hip_memcpy_d2h(redoHost, leftList, (size_t)left * sizeof(int), st) || opal_hip_stream_sync(st))
            return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    if (redoCount) *redoCount = left;
    if (debug_on())
        fprintf(stderr, "[opal] start locations on the device: %d sequences, %d with a second (exact-row) pass, %d to redo on the host, %d not found\n", n,
                doubtful, left, flags[1]);
    return 0;
}

// Operations traced on the device, by position in the shard: at(i)[0 .. len[i]), or len[i] < 0 (not traced: the host does it).
struct TraceOut {
    std::vector<std::vector<unsigned char>> chunks;  // one buffer of operation strings per device batch, as downloaded
    std::vector<int> chunk;                          // [n] which one
    std::vector<uint64_t> off;
    std::vector<int> len;
    const unsigned char* at(size_t i) const { return chunks[(size_t)chunk[i]].data() + off[i]; }
};

/*
 * Alignment operations on the device (trace_kernel.hip) for every sequence of the shard whose score, end cell and start
 * cell are known: the box between start and end is computed with one byte of flags per cell in HBM and walked back there;
 * only the operation strings come to the host. Batches are bounded by OPAL_TRACEBACK_DEVICE_BYTES of flags (default
 * 8 GiB). Whatever is left out (no start cell, SW score 0, boxes of more than 2560 rows, a box larger than the budget, any
 * failure) keeps len < 0 and is traced by the host as before. `dEndQ` .. `dStartT` are the device arrays in caller
 * (= shard) order the search and the start-location pass have filled; the host arrays hold the same values.
 */
static void device_traceback(OpalxDb* db, int Q, int Go, int Ge, int A, int mode, int n, const int* score, const int* endQ,
                             const int* endT, const int* startQ, const int* startT, const int* dEndQ, const int* dEndT,
                             const int* dStartQ, const int* dStartT, TraceOut& out) {
    out.len.assign((size_t)n, -1);
    out.off.assign((size_t)n, 0);
    out.chunk.assign((size_t)n, 0);
    out.chunks.clear();
    if (getenv("OPAL_HIP_HOST_TRACEBACK") || !db->seqItemValid || n <= 0) return;
    uint64_t budget = 8ull << 30;
    if (const char* e = getenv("OPAL_TRACEBACK_DEVICE_BYTES")) { const long long v = atoll(e); if (v > 0) budget = (uint64_t)v; }
    void* st = db->main();
    std::vector<int> items;
    std::vector<uint64_t> offs;  // [2 * batch]: flag-matrix offsets, then operation offsets
    const double t0 = now_ms();
    double tDev = 0, tPack = 0;
    int traced = 0, batches = 0;
    int i = 0;
    while (i < n) {
        items.clear(); offs.clear();
        std::vector<uint64_t> dirOff, opsOff;
        uint64_t dirBytes = 0, opsBytes = 0;
        int maxRows = 0;
        for (; i < n; i++) {
            if (score[i] == INT_MIN || (mode == OPAL_MODE_SW && score[i] == 0)) continue;
            if (endQ[i] < 0 || endT[i] < 0 || startQ[i] < 0 || startT[i] < 0 || startQ[i] > endQ[i] || startT[i] > endT[i]) continue;
            const int bq = endQ[i] - startQ[i] + 1, bt = endT[i] - startT[i] + 1;
            if (bq > Q || !opal_hip_trace_rows_supported(bq)) continue;
            const uint64_t bytes = (uint64_t)((bq + 3) & ~3) * (uint64_t)bt;
            if (bytes > budget) continue;                          // a box of its own: the host's job
            if (dirBytes + bytes > budget && !items.empty()) break;  // next batch
            items.push_back(i);
            dirOff.push_back(dirBytes); opsOff.push_back(opsBytes);
            dirBytes += (bytes + 15) & ~15ull;
            opsBytes += (uint64_t)(bq + bt);
            maxRows = std::max(maxRows, bq);
        }
        const size_t m = items.size();
        if (!m) break;
        offs = dirOff;
        offs.insert(offs.end(), opsOff.begin(), opsOff.end());
        bool ok = db->trDirs.ensure((size_t)dirBytes + 16) && db->trOps.ensure((size_t)opsBytes + 16) && db->trOff.ensure(2 * m, 1.25) &&
                  db->trItems.ensure(m, 1.25) && db->trOut.ensure(2 * m, 1.25);
        ok = ok && !opal_hip_memcpy_h2d(db->trItems.p, items.data(), m * sizeof(int), st) &&
             !opal_hip_memcpy_h2d(db->trOff.p, offs.data(), 2 * m * sizeof(uint64_t), st);
        std::vector<int> res(2 * m);
        std::vector<unsigned char> ops((size_t)opsBytes);
        const double tb0 = now_ms();
        if (ok) {
            OpalHipTraceLaunch a;
            memset(&a, 0, sizeof a);
            a.query = db->query.p; a.alphabet_length = A; a.score_matrix = db->matrix.p; a.gap_open = Go; a.gap_ext = Ge;
            a.tokens = db->tokens.p; a.pair_token_offset = db->pairOff.p; a.seq_item = db->seqItem.p;
            a.items = db->trItems.p; a.num_items = (int)m; a.max_rows = maxRows;
            a.end_query = dEndQ; a.end_target = dEndT; a.start_query = dStartQ; a.start_target = dStartT;
            a.dir_offset = db->trOff.p; a.dirs = db->trDirs.p; a.ops_offset = db->trOff.p + m; a.ops = db->trOps.p;
            a.ops_length = db->trOut.p; a.box_score = db->trOut.p + m;
            a.stream = st;
            ok = !opal_hip_memset32(db->trOut.p, INT_MIN, 2 * m, st) && !opal_hip_trace_launch(&a) &&
                 !opal_hip_memcpy_d2h(res.data(), db->trOut.p, 2 * m * sizeof(int), st) &&
                 !opal_hip_memcpy_d2h(ops.data(), db->trOps.p, (size_t)opsBytes, st) && !opal_hip_stream_sync(st);
        }
        if (!ok) {  // nothing of this batch is trusted; the host traces it
            db->quiesce();
            if (debug_on()) fprintf(stderr, "[opal] device traceback: batch of %zu failed (%s); host traceback instead\n", m, opal_hip_last_error());
            continue;
        }
        batches++;
        const double tb1 = now_ms();
        tDev += tb1 - tb0;
        for (size_t b = 0; b < m; b++) {
            const int idx = items[b];
            const int len = res[b], bs = res[m + b];
            const int cap = (endQ[idx] - startQ[idx] + 1) + (endT[idx] - startT[idx] + 1);
            if (bs != score[idx] || len <= 0 || len > cap) continue;  // the box does not carry the score: the host's full pass decides
            out.chunk[(size_t)idx] = (int)out.chunks.size();
            out.off[(size_t)idx] = opsOff[b];
            out.len[(size_t)idx] = len;
            traced++;
        }
        out.chunks.push_back(std::move(ops));
        tPack += now_ms() - tb1;
    }
    if (db->trDirs.cap > (2ull << 30)) db->trDirs.release();  // a cached context does not sit on gigabytes of flags between calls
    if (debug_on())
        fprintf(stderr, "[opal] device traceback: %d of %d alignments in %d batches, %.2f ms (kernels + copies %.2f, bookkeeping %.2f)\n", traced, n,
                batches, now_ms() - t0, tDev, tPack);
}

// ------------------------------------------------------------------------------------ devices of a call
// OPAL_HIP_DEVICES = "all" | "0,1,2,..." makes opalSearchDatabase shard every call over those GPUs. A device
// may be listed more than once (two contexts on one GPU): that is how the sharding is tested on a one-GPU box.
static std::vector<int> call_devices() {
    std::vector<int> out;
    const char* env = getenv("OPAL_HIP_DEVICES");
    if (!env || !*env) return out;
    const int n = opal_hip_device_count();
    if (!strcmp(env, "all")) {
        for (int i = 0; i < n; i++) out.push_back(i);
        return out;
    }
    const char* p = env;
    while (*p) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        if (v >= 0 && v < n) out.push_back((int)v);
        p = *end ? end + 1 : end;
    }
    return out;
}

/*
 * Deal sequences to `parts` shards: indices sorted by length (descending), cut into groups of
 * 64 consecutive ones (sequences of one wave stay together), every group to the shard with the fewest residues
 * so far. Returns shard[k] = list of positions into `lens`.
 */
static std::vector<std::vector<int>> deal_groups(const int* lens, int n, int parts) {
    // order by length, longest first (counting sort, stable; a comparison sort of 100 000 lengths costs milliseconds)
    int maxLen = 0;
    for (int i = 0; i < n; i++) maxLen = std::max(maxLen, lens[i]);
    std::vector<int> order((size_t)n);
    if (maxLen <= (1 << 22)) {
        std::vector<int> start((size_t)maxLen + 2, 0);
        for (int i = 0; i < n; i++) start[(size_t)maxLen - std::max(lens[i], 0) + 1]++;
        for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
        for (int i = 0; i < n; i++) order[(size_t)start[(size_t)maxLen - std::max(lens[i], 0)]++] = i;
    } else {
        for (int i = 0; i < n; i++) order[(size_t)i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    }
    std::vector<std::vector<int>> shard((size_t)parts);
    for (auto& v : shard) v.reserve((size_t)n / parts + 64);
    std::vector<long long> load((size_t)parts, 0);
    for (int g = 0; g < n; g += 64) {
        const int e = std::min(n, g + 64);
        long long sum = 0;
        for (int i = g; i < e; i++) sum += lens[order[(size_t)i]];
        size_t best = 0;
        for (size_t k = 1; k < shard.size(); k++)
            if (load[k] < load[best]) best = k;
        load[best] += sum;
        shard[best].insert(shard[best].end(), order.begin() + g, order.begin() + e);
    }
    return shard;
}

/*
 * One shard of an opalSearchDatabase call on the calling thread's current device: pack, upload, search,
 * download. `ids` (may be null = identity) maps shard position -> caller index; results land in
 * score/endQ/endT at SHARD positions (INT_MIN = not written by the search).
 */
static int run_shard(int device, const unsigned char* query, int Q, unsigned char* const* db, const int* dbLens, const int* ids,
                     int n, int Go, int Ge, const int* M, int A, int searchType, int mode, int* score, int* endQ, int* endT,
                     double* tms, int* startQ = nullptr, int* startT = nullptr, TraceOut* trace = nullptr) {
    BorrowedContext ctx(device);
    OpalxDb* pdb = ctx.db;
    if (!pdb) return OPAL_ERR_NO_SIMD_SUPPORT;
    const double t0 = now_ms();
    HostPack hp;
    std::vector<int> lensBuf;
    const int* lens = dbLens;
    if (ids) {
        lensBuf.resize((size_t)n);
        for (int i = 0; i < n; i++) lensBuf[(size_t)i] = dbLens[ids[i]];
        lens = lensBuf.data();
    }
    const double t1 = now_ms();
    auto getSeq = [&](int i) { return (const unsigned char*)db[ids ? ids[i] : i]; };
    // (ordering and pairing run as one task beside the gather of the residues: see db_fill)
    if (!db_fill(pdb, hp, RAW_POINTERS, getSeq, nullptr, lens, n)) return OPAL_ERR_NO_SIMD_SUPPORT;
    const double t2 = now_ms();
    const size_t nn = (size_t)std::max(n, 1);
    const int wantEnd = searchType != OPAL_SEARCH_SCORE;
    if (!pdb->results.ensure(5 * nn, 1.15) || !pdb->hResults.ensure(5 * nn, 1.15)) return OPAL_ERR_NO_SIMD_SUPPORT;
    int* dS = pdb->results.p; int* dQ = dS + nn; int* dT = dS + 2 * nn;
    void* st = pdb->main();
    // INT_MIN marks "not written by the search" (a sequence the search had to give up on)
    if (opal_hip_memset32(dS, INT_MIN, nn, st) || (wantEnd && opal_hip_memset32(dQ, -1, 2 * nn, st))) return OPAL_ERR_NO_SIMD_SUPPORT;
    int rc = search_core(pdb, query, Q, Go, Ge, M, A, searchType, mode, dS, wantEnd ? dQ : nullptr, wantEnd ? dT : nullptr, nullptr,
                         /*finalSync=*/false);
    const double t3 = now_ms();
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) return rc;
    int* h = pdb->hResults.p;
    // start cells for the alignment search, on the device while the end cells are still there (n4). Any failure
    // just leaves them unknown (INT_MIN): the host traceback then finds them with its own full pass.
    bool haveStarts = false;
    std::vector<int> redo;
    int nredo = 0;
    if (startQ && startT && rc == 0 && wantEnd && n > 0 && !getenv("OPAL_HIP_HOST_STARTS")) {
        redo.resize((size_t)n);
        haveStarts = search_starts(pdb, query, Q, Go, Ge, M, A, mode, dS, dQ, dT, dS + 3 * nn, dS + 4 * nn, redo.data(), &nredo) == 0;
    }
    if (opal_hip_memcpy_d2h(h, dS, (haveStarts ? 5 : wantEnd ? 3 : 1) * nn * sizeof(int), st) || opal_hip_stream_sync(st))
        return OPAL_ERR_NO_SIMD_SUPPORT;
    memcpy(score, h, (size_t)n * sizeof(int));
    if (wantEnd) { memcpy(endQ, h + nn, (size_t)n * sizeof(int)); memcpy(endT, h + 2 * nn, (size_t)n * sizeof(int)); }
    if (startQ && startT) {
        if (haveStarts) {
            memcpy(startQ, h + 3 * nn, (size_t)n * sizeof(int)); memcpy(startT, h + 4 * nn, (size_t)n * sizeof(int));
            for (int i = 0; i < nredo; i++) { startQ[redo[(size_t)i]] = INT_MIN; startT[redo[(size_t)i]] = INT_MIN; }
            for (int i = 0; i < n; i++) if (score[i] == INT_MIN) { startQ[i] = INT_MIN; startT[i] = INT_MIN; }
        } else {
            for (int i = 0; i < n; i++) { startQ[i] = INT_MIN; startT[i] = INT_MIN; }
        }
        // the operations themselves, while database, query and end/start cells are still on the device
        if (trace && haveStarts) device_traceback(pdb, Q, Go, Ge, A, mode, n, score, endQ, endT, startQ, startT, dQ, dT, dS + 3 * nn, dS + 4 * nn, *trace);
    }
    if (tms) { tms[0] += t1 - t0; tms[1] += t2 - t1; tms[2] += t3 - t2; tms[3] += now_ms() - t3; }
    return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------ opalx_* API
static OpalxDb* create_from(HostPack& hp, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
                            const unsigned char* flat) {
    OpalxDb* db = new (std::nothrow) OpalxDb;
    if (!db) return nullptr;
    if (!db_fill(db, hp, kind, getSeq, flat)) { opalx_db_destroy(db); return nullptr; }
    return db;
}

extern "C" OpalxDb* opalx_db_create(unsigned char* const db[], int db_length, const int db_seq_lengths[]) {
    if (!device_ok()) return nullptr;
    HostPack hp;
    pack_order(db_length, db_seq_lengths, hp);
    return create_from(hp, RAW_POINTERS, [&](int i) { return (const unsigned char*)db[i]; }, nullptr);
}

static void flat_offsets(const int64_t* offsets, int db_length, HostPack& hp) {
    std::vector<int> lens((size_t)db_length);
    for (int i = 0; i < db_length; i++) lens[(size_t)i] = (int)(offsets[i + 1] - offsets[i]);
    pack_order(db_length, lens.data(), hp);
    // already contiguous: the device packer reads each sequence at its offset
    hp.rawOff.resize(hp.pairSeq.size());
    for (size_t i = 0; i < hp.pairSeq.size(); i++) hp.rawOff[i] = hp.pairSeq[i] >= 0 ? (uint64_t)offsets[hp.pairSeq[i]] : 0;
    hp.rawBytes = (uint64_t)offsets[db_length];
}

extern "C" OpalxDb* opalx_db_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length) {
    if (!device_ok()) return nullptr;
    HostPack hp;
    flat_offsets(offsets, db_length, hp);
    return create_from(hp, RAW_HOST_FLAT, nullptr, residues);
}

extern "C" OpalxDb* opalx_db_create_flat_device(const unsigned char* d_residues, const int64_t* offsets, int db_length) {
    if (!device_ok()) return nullptr;
    HostPack hp;
    flat_offsets(offsets, db_length, hp);
    return create_from(hp, RAW_DEVICE_FLAT, nullptr, d_residues);
}

extern "C" void opalx_db_destroy(OpalxDb* db) {
    if (!db) return;
    db->quiesce();
    db_release(db);
    delete db;
}

extern "C" long long opalx_stream_launch_count(void) { return g_streamLaunches.load(); }

extern "C" void opalx_release_cached(void) {
    ContextCache& c = ContextCache::get();
    std::lock_guard<std::mutex> lk(c.m);
    const int cur = opal_hip_get_device();
    for (auto& kv : c.freeList) {
        if (kv.second.empty()) continue;
        opal_hip_set_device(kv.first);
        for (OpalxDb* db : kv.second) { db->quiesce(); db_release(db); delete db; }
        kv.second.clear();
    }
    if (cur >= 0) opal_hip_set_device(cur);
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
    const size_t nn = (size_t)std::max(n, 1);
    if (!db->results.ensure(3 * nn)) return OPAL_ERR_NO_SIMD_SUPPORT;  // kept for the life of the handle
    int* dS = db->results.p; int* dQ = dS + nn; int* dT = dS + 2 * nn;
    void* st = db->main();
    if (!st) return OPAL_ERR_NO_SIMD_SUPPORT;
    // pre-load the caller's values so that sequences the search does not write keep them
    int rc = 0;
    if (opal_hip_memcpy_h2d(dS, score, (size_t)n * sizeof(int), st)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc && end_query && opal_hip_memcpy_h2d(dQ, end_query, (size_t)n * sizeof(int), st)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc && end_target && opal_hip_memcpy_h2d(dT, end_target, (size_t)n * sizeof(int), st)) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    if (!rc)
        rc = search_core(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode, dS,
                         wantEnd ? dQ : nullptr, wantEnd ? dT : nullptr, nullptr);
    if (rc != OPAL_ERR_NO_SIMD_SUPPORT && rc != OPAL_ERR_INVALID_MODE) {
        int e = opal_hip_memcpy_d2h(score, dS, (size_t)n * sizeof(int), st);
        if (wantEnd && end_query) e |= opal_hip_memcpy_d2h(end_query, dQ, (size_t)n * sizeof(int), st);
        if (wantEnd && end_target) e |= opal_hip_memcpy_d2h(end_target, dT, (size_t)n * sizeof(int), st);
        e |= opal_hip_stream_sync(st);
        if (e) rc = OPAL_ERR_NO_SIMD_SUPPORT;
    }
    return rc;
}

extern "C" int opalx_search_starts(OpalxDb* db, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                   const int* score_matrix, int alphabet_length, int mode, const int* d_score,
                                   const int* d_end_query, const int* d_end_target, int* d_start_query, int* d_start_target,
                                   int* redo, int* redo_count) {
    if (!db || !d_score || !d_end_query || !d_end_target || !d_start_query || !d_start_target) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (mode != OPAL_MODE_NW && mode != OPAL_MODE_HW && mode != OPAL_MODE_OV && mode != OPAL_MODE_SW) return OPAL_ERR_INVALID_MODE;
    if (!device_ok()) return OPAL_ERR_NO_SIMD_SUPPORT;
    return search_starts(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, mode, d_score, d_end_query,
                         d_end_target, d_start_query, d_start_target, redo, redo_count);
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

namespace {

// Which results still need computing (opal.h:118-122, opal.cpp:1446-1451), with the documented,
// whole-database indexing (the reference indexes `skip` bucket-locally, opal.cpp:510,998).
// Returns true if EVERY result needs computing (then `todo` stays empty = identity).
static bool collect_todo(OpalSearchResult* results[], int n, int searchType, std::vector<int>& todo) {
    HostPool& pool = HostPool::get();
    const int tasks = n >= (1 << 16) ? std::min(pool.width(), 16) : 1;
    std::vector<std::vector<int>> have((size_t)tasks);  // positions that are already computed, per range
    auto scan = [&](int t) {
        const int i0 = (int)((long long)n * t / tasks), i1 = (int)((long long)n * (t + 1) / tasks);
        for (int i = i0; i < i1; i++) {
            const OpalSearchResult* r = results[i];
            if (r->scoreSet && (searchType == OPAL_SEARCH_SCORE || (r->endLocationQuery >= 0 && r->endLocationTarget >= 0)))
                have[(size_t)t].push_back(i);
        }
    };
    pool.parallel_for(tasks, scan);
    size_t nh = 0;
    for (auto& v : have) nh += v.size();
    if (nh == 0) return true;
    todo.reserve((size_t)n - nh);
    int next = 0;
    for (auto& v : have)
        for (int i : v) { while (next < i) todo.push_back(next++); next = i + 1; }
    while (next < n) todo.push_back(next++);
    return false;
}

// `starts`: NULL, or 2*dbLength ints receiving the start cells found on the device for the sequences this call
// computed ({startQ[dbLength], startT[dbLength]}, INT_MIN = unknown).
static int search_into_results(unsigned char query[], int Q, unsigned char* db[], int dbLength, int dbSeqLengths[], int Go, int Ge,
                               int* M, int A, OpalSearchResult* results[], int searchType, int mode, bool clearAlignment,
                               int* starts, unsigned char* traced = nullptr) {
    const double tA = now_ms();
    std::vector<int> todo;
    const bool all = collect_todo(results, dbLength, searchType, todo);
    const int n = all ? dbLength : (int)todo.size();
    const int* ids = all ? nullptr : todo.data();
    int rc = 0;
    double tms[4] = {0, 0, 0, 0};
    const double tB = now_ms();
    double tC = tB;
    if (n > 0) {
        if (!device_ok()) return OPAL_ERR_NO_SIMD_SUPPORT;
        const int st = searchType == OPAL_SEARCH_SCORE ? OPAL_SEARCH_SCORE : OPAL_SEARCH_SCORE_END;
        const int wantEnd = st != OPAL_SEARCH_SCORE;
        std::unique_ptr<int[]> buf(new (std::nothrow) int[5 * (size_t)n]);
        if (!buf) return OPAL_ERR_NO_SIMD_SUPPORT;
        int* score = buf.get(); int* endQ = score + n; int* endT = score + 2 * (size_t)n;
        int* startQ = starts ? score + 3 * (size_t)n : nullptr; int* startT = starts ? score + 4 * (size_t)n : nullptr;
        std::vector<int> devices = call_devices();
        // alignment search: operation strings traced on the device, per shard (positions of a shard in the to-do list: tracePos)
        std::vector<TraceOut> traces(std::max<size_t>(devices.size(), 1));
        std::vector<std::vector<int>> tracePos;
        if (devices.size() <= 1 || n < 2 * 64) {
            if (devices.size() == 1 && opal_hip_set_device(devices[0])) return OPAL_ERR_NO_SIMD_SUPPORT;
            rc = run_shard(opal_hip_get_device(), query, Q, db, dbSeqLengths, ids, n, Go, Ge, M, A, st, mode, score, endQ, endT, tms,
                           startQ, startT, traced ? &traces[0] : nullptr);
            tracePos.resize(1);  // (identity: shard position = position in the to-do list)
        } else {
            // several GPUs: deal 64-sequence groups balancing the residue count, one host thread per GPU, each
            // running the single-GPU pipeline on its shard; the host is the gather root (results live in host structs)
            std::vector<int> lens((size_t)n);
            for (int i = 0; i < n; i++) lens[(size_t)i] = dbSeqLengths[ids ? ids[i] : i];
            std::vector<std::vector<int>> shard = deal_groups(lens.data(), n, (int)devices.size());
            std::vector<int> rcs(devices.size(), 0);
            std::vector<std::vector<int>> out(devices.size());
            std::vector<std::thread> threads;
            const int callerDevice = opal_hip_get_device();
            for (size_t k = 0; k < devices.size(); k++) {
                threads.emplace_back([&, k] {
                    std::vector<int>& pos = shard[k];
                    const int m = (int)pos.size();
                    if (!m) return;
                    // device_ok() first: its once-per-thread OPAL_HIP_DEVICE selection must not override the shard's GPU
                    if (!device_ok() || opal_hip_set_device(devices[k])) { rcs[k] = OPAL_ERR_NO_SIMD_SUPPORT; return; }
                    std::vector<int> sid((size_t)m);  // shard position -> caller index
                    for (int i = 0; i < m; i++) sid[(size_t)i] = ids ? ids[pos[(size_t)i]] : pos[(size_t)i];
                    out[k].assign(5 * (size_t)m, INT_MIN);
                    int* s = out[k].data();
                    rcs[k] = run_shard(devices[k], query, Q, db, dbSeqLengths, sid.data(), m, Go, Ge, M, A, st, mode, s, s + m,
                                       s + 2 * (size_t)m, nullptr, starts ? s + 3 * (size_t)m : nullptr,
                                       starts ? s + 4 * (size_t)m : nullptr, traced ? &traces[k] : nullptr);
                });
            }
            for (auto& th : threads) th.join();
            if (callerDevice >= 0) opal_hip_set_device(callerDevice);
            for (size_t k = 0; k < devices.size(); k++) {
                if (rcs[k] == OPAL_ERR_NO_SIMD_SUPPORT || rcs[k] == OPAL_ERR_INVALID_MODE) return rcs[k];
                if (rcs[k]) rc = rcs[k];
                const int m = (int)shard[k].size();
                for (int i = 0; i < m; i++) {
                    const int p = shard[k][(size_t)i];
                    score[p] = out[k][(size_t)i];
                    endQ[p] = wantEnd ? out[k][(size_t)m + i] : -1;
                    endT[p] = wantEnd ? out[k][2 * (size_t)m + i] : -1;
                    if (starts) { startQ[p] = out[k][3 * (size_t)m + i]; startT[p] = out[k][4 * (size_t)m + i]; }
                }
            }
            tracePos = shard;
        }
        tC = now_ms();
        if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) return rc;
        // scatter into the caller's result structs (and, for non-alignment searches, the post-fill of
        // opal.cpp:1508-1515 in the same pass when the call succeeded)
        const bool post = clearAlignment && rc == 0;
        HostPool& pool = HostPool::get();
        const int tasks = n >= (1 << 14) ? std::min(pool.width(), 32) : 1;
        pool.parallel_for(tasks, [&](int t) {
            const int i0 = (int)((long long)n * t / tasks), i1 = (int)((long long)n * (t + 1) / tasks);
            for (int i = i0; i < i1; i++) {
                OpalSearchResult* r = results[ids ? ids[i] : i];
                if (score[i] != INT_MIN) {
                    r->scoreSet = 1;
                    r->score = score[i];
                    r->endLocationQuery = wantEnd ? endQ[i] : -1;
                    r->endLocationTarget = wantEnd ? endT[i] : -1;
                }
                if (post && all) { r->alignment = NULL; r->alignmentLength = -1; r->startLocationQuery = -1; r->startLocationTarget = -1; }
                if (starts) { starts[ids ? ids[i] : i] = startQ[i]; starts[(size_t)dbLength + (ids ? ids[i] : i)] = startT[i]; }
            }
        });
        // alignments traced on the device: start cell and operations into the result structs (`alignment` is malloc()ed: the
        // caller free()s it, opal.h); whatever is not marked in `traced` is left to the host traceback
        if (traced && rc == 0) {
            for (size_t k = 0; k < traces.size() && k < tracePos.size(); k++) {
                const TraceOut& tr = traces[k];
                const int m = (int)tr.len.size();
                const std::vector<int>& pos = tracePos[k];
                pool.parallel_for(std::max(1, std::min(pool.width(), m / 256)), [&](int t) {
                    const int parts = std::max(1, std::min(pool.width(), m / 256));
                    const int i0 = (int)((long long)m * t / parts), i1 = (int)((long long)m * (t + 1) / parts);
                    for (int i = i0; i < i1; i++) {
                        if (tr.len[(size_t)i] <= 0) continue;
                        const int p = pos.empty() ? i : pos[(size_t)i];  // position in the to-do list
                        const int caller = ids ? ids[p] : p;
                        unsigned char* a = (unsigned char*)malloc((size_t)tr.len[(size_t)i]);
                        if (!a) continue;  // the host traceback will try (and report) instead
                        memcpy(a, tr.at((size_t)i), (size_t)tr.len[(size_t)i]);
                        OpalSearchResult* r = results[caller];
                        r->startLocationQuery = startQ[p];
                        r->startLocationTarget = startT[p];
                        r->alignment = a;
                        r->alignmentLength = tr.len[(size_t)i];
                        traced[caller] = 1;
                    }
                });
            }
        }
        if (debug_on())
            fprintf(stderr, "[opal] opalSearchDatabase: todo %.2f ms | order %.2f, gather+upload+pack %.2f, search %.2f, download %.2f | "
                    "scatter %.2f ms\n", tB - tA, tms[0], tms[1], tms[2], tms[3], now_ms() - tC);
    } else if (!args_fit_int32(Go, Ge, M, A)) {
        rc = OPAL_ERR_OVERFLOW;
    }
    if (rc) return rc;  // alignment fields untouched on error (opal.cpp:1473)
    if (clearAlignment && !(all && n > 0)) {  // opal.cpp:1508-1515 (already done in the scatter pass when everything was computed)
        for (int i = 0; i < dbLength; i++) {
            results[i]->alignment = NULL;
            results[i]->alignmentLength = -1;
            results[i]->startLocationQuery = -1;
            results[i]->startLocationTarget = -1;
        }
    }
    return 0;
}

}  // namespace

extern "C" int opalSearchDatabase(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                  int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                  OpalSearchResult* results[], const int searchType, int mode, int overflowMethod) {
    (void)overflowMethod;  // OPAL_OVERFLOW_SIMPLE and _BUCKETS give identical results (see search_core)
    if (mode != OPAL_MODE_NW && mode != OPAL_MODE_HW && mode != OPAL_MODE_OV && mode != OPAL_MODE_SW)
        return OPAL_ERR_INVALID_MODE;
    // Alignment search: score + end cells on the device, start cells on the device (rev_kernel.hip), then the host
    // traces the operations inside each (start..end) box. Sequences whose score/end the caller supplied (not
    // recomputed here, opal.h:118-122) have no device start: the host pass finds theirs.
    std::vector<int> starts;
    std::vector<unsigned char> traced;  // alignments the device has already traced (trace_kernel.hip)
    if (searchType == OPAL_SEARCH_ALIGNMENT) {
        starts.assign(2 * (size_t)std::max(dbLength, 1), INT_MIN);
        traced.assign((size_t)std::max(dbLength, 1), 0);
    }
    const int rc = search_into_results(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix,
                                       alphabetLength, results, searchType, mode, searchType != OPAL_SEARCH_ALIGNMENT,
                                       searchType == OPAL_SEARCH_ALIGNMENT ? starts.data() : nullptr,
                                       searchType == OPAL_SEARCH_ALIGNMENT ? traced.data() : nullptr);
    if (rc) return rc;
    if (searchType == OPAL_SEARCH_ALIGNMENT) {
        const double t0 = now_ms();
        int left = 0;
        for (int i = 0; i < dbLength; i++) left += traced[(size_t)i] ? 0 : 1;
        const int arc = left ? opalx_find_alignments_masked(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix,
                                                            alphabetLength, results, mode, starts.data(), starts.data() + dbLength, traced.data())
                             : 0;
        if (debug_on()) fprintf(stderr, "[opal] host traceback of %d of %d alignments: %.2f ms\n", left, dbLength, now_ms() - t0);
        return arc;
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
    std::vector<int> score((size_t)std::max(dbLength, 1), INT_MIN);
    int rc = run_shard(opal_hip_get_device(), query, queryLength, db, dbSeqLengths, nullptr, dbLength, gapOpen, gapExt, scoreMatrix,
                       alphabetLength, OPAL_SEARCH_SCORE, OPAL_MODE_SW, score.data(), nullptr, nullptr, nullptr);
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT) return rc;
    rc = 0;
    for (int i = 0; i < dbLength; i++) {
        // the 8-bit pass represents 0..127: a sequence overflows iff its exact score is >= 128
        if (score[(size_t)i] == INT_MIN || score[(size_t)i] >= 128) {
            results[i]->score = -1; results[i]->scoreSet = 0; rc = OPAL_ERR_OVERFLOW;
        } else {
            opalSearchResultSetScore(results[i], score[(size_t)i]);
            results[i]->endLocationQuery = -1; results[i]->endLocationTarget = -1;
        }
    }
    return rc;
}
