// opal_internal.hpp -- declarations shared by the host translation units of libopal.so (not installed; the public
// interface is include/opal.h + include/opal_hip.h). Plain C++: everything that touches the device goes through the
// extern "C" shim of include/opal_hip.h.
//
//   db_pack.cpp       database packer, (re)fill of a device-resident database, cached contexts, opalx_db_* handles
//   planner.cpp       launch planning: occupancy, length classes, streaming geometry
//   ladder.cpp        the precision ladder (reference src/opal.cpp:496-535, :983-1021): launchers + search_core
//   trace_driver.cpp  start cells and alignment operations on the device (drivers of rev_kernel.hip / trace_kernel.hip)
//   multi_gpu.cpp     devices of a call, residue-balanced deal of sequences
//   opal_api.cpp      the opal.h C API and the opalx_search_* entries on top of the above
#pragma once
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

#include "options.hpp"
#include "host_pool.hpp"
#include "opal.h"
#include "opal_hip.h"

namespace opal_host {

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline bool debug_on() { return opts().debug; }

// What the searches of this process ran on (diagnostics and tests: opalx_counters).
enum Counter {
    CNT_STREAM_SW = 0,   // searches on the streaming SW kernels
    CNT_STREAM_GLOBAL,   // searches on the streaming NW/HW/OV kernels
    CNT_ONE_LAUNCH,      // class kernels, all length classes as one grid
    CNT_CLASS_LAUNCHES,  // class kernels launched one class at a time
    CNT_CHUNK_LAUNCHES,  // row-chunk launches of queries beyond one wave
    CNT_WIDTH32,         // launches of the 32-bit rung
    CNT_DEVICE_TRACED,   // alignments traced on the device
    CNT_HOST_TRACED,     // alignments traced (or started) by the host
    CNT_REUSED_FILLS,
    CNT_SWAPPED,         // targets searched on the role-swapped kernel (the longest ones of a search with a short query)    // opalSearchDatabase calls that found their database already on the device (same lengths; changed chunks re-uploaded)
    CNT_PAIRED,          // targets searched with one pair spread over several waves (pair_kernel.hpp: the longest ones of a global-mode search)
    CNT_COUNT
};
extern std::atomic<long long> g_counters[CNT_COUNT];
inline void count(Counter c, long long n = 1) { g_counters[c].fetch_add(n, std::memory_order_relaxed); }

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
    bool would_realloc(size_t n) const {  // ensure(n) would free the buffer (nothing on the device may still be using it)
        return !p || n > cap || (cap > 4 * std::max<size_t>(n, 1) && cap * sizeof(T) > (256u << 20));
    }
    void release() { opal_hip_free(p); p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool deviceCached = false;  // coarse-grained: kernels read it directly over PCIe (see opal_hip_host_malloc_device_cached)
    bool ensure(size_t n, double slack = 1.0) {
        if (p && cap > 4 * std::max<size_t>(n, 1) && cap * sizeof(T) > (256u << 20)) release();
        if (n <= cap && p) return true;
        opal_hip_host_free(p);
        const size_t want = std::max<size_t>((size_t)((double)n * slack), 1);
        p = (T*)(deviceCached ? opal_hip_host_malloc_device_cached(want * sizeof(T)) : opal_hip_host_malloc(want * sizeof(T)));
        cap = p ? want : 0;
        return p != nullptr;
    }
    void release() { opal_hip_host_free(p); p = nullptr; cap = 0; }
};

// Pinned host memory with the little of std::vector the scheduler uses: what it holds can be handed to an asynchronous
// copy and read by the host at the same time (pageable vectors would have to outlive copies that may still be staging them).
template <typename T> struct PinVec {
    PinBuf<T> buf;
    size_t n = 0;
    bool resize(size_t k, double slack = 1.0) { if (!buf.ensure(std::max<size_t>(k, 1), slack)) { n = 0; return false; } n = k; return true; }
    T& operator[](size_t i) { return buf.p[i]; }
    const T& operator[](size_t i) const { return buf.p[i]; }
    T* data() { return buf.p; }
    const T* data() const { return buf.p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void release() { buf.release(); n = 0; }
};

// Scratch of the ordering (db_pack.cpp, order_into): per-thread histograms, bucket starts (item position, byte offset).
struct OrderScratch { std::vector<int> hist, pos, seq; std::vector<uint64_t> byte; };

// Residues on their way to the device, chunk by chunk, while the search of the sequences that have landed is being
// launched (opalSearchDatabase: the gather of N host allocations and their upload overlap the search instead of preceding
// it). Chunk c holds the items [cutItem[c], cutItem[c+1]); a pool thread gathers it into pinned memory, enqueues its copy on
// the context's copy stream, records events[c] behind it and sets state[c] (1 enqueued, -1 failed). (The chunk is gathered
// by several threads, piece by piece; the one that finishes its last piece does the copy.)
struct UploadGate {
    bool active = false;
    bool timed = false;  // (debug) the chunk events carry timestamps
    std::vector<int> cutItem;
    std::vector<int> pieceCut, pieceChunk;  // the pool's tasks: piece p = items [pieceCut[p], pieceCut[p+1]) of chunk pieceChunk[p]
    std::vector<uint64_t> cutByte, pieceByte;  // byte offset (in item order) at which chunk c / piece p begins; one more entry: the total
    uint64_t cutsTotal = 0, cutsChunkBytes = 0, cutsPieceBytes = 0;  // what the cuts above were made for
    size_t cutsItems = 0;
    std::vector<void*> events;  // (owned by the context: kept across calls)
    std::unique_ptr<std::atomic<int>[]> state;
    std::unique_ptr<std::atomic<int>[]> left;  // pieces of chunk c still being gathered
    std::unique_ptr<std::atomic<int>[]> dirty; // chunk c must be uploaded (always, unless the device already holds the same bytes)
    std::atomic<int> uploaded{0};              // chunks copied by this fill
    bool reuse = false;                        // this fill found the database of the previous call (same lengths)
    bool zeroCopy = false;                     // the sliced launches of this fill's search read the pinned staging buffer directly:
                                               // the pool only gathers, a chunk is copied when something asks for its device copy (gate_items)
    bool kernelFilled = false;                 // ... and they covered every item: the kernels have written the whole device copy
    int kernelFilledFrom = INT_MAX;            // ... or every item from this one on (the items before it: chunks copied on demand)
    std::unique_ptr<std::atomic<int>[]> copied; // chunk c's copy has been enqueued
    bool speculated = false;                   // ... and its search was launched without waiting for the comparison
    size_t stateCap = 0;
    int chunks() const { return active ? (int)cutItem.size() - 1 : 0; }
};

}  // namespace opal_host

// ------------------------------------------------------------------------------------ packed db
// The sequences sorted by length (descending; stable), residues back to back in that order, one byte each: ITEM i is the
// i-th of them. A wave slot works on one item (widths 17, 18, 32) or, at width 16, on a PAIR of neighbours (items 2p and
// 2p+1 packed in one register), whose residues the packed kernels read as 16-bit tokens (low byte A, high byte B): those
// token streams are built on the device, on first use (ensure_pair_tokens).
struct OpalxDb {
    int device = 0;
    bool cached = false;  // a context of the opalSearchDatabase cache (returned to the free list, not destroyed)
    int numSeqs = 0;
    int numPairs = 0;     // items = 2 * numPairs (the last one absent when the number of non-empty sequences is odd)
    int64_t residues = 0;
    // host copies the scheduler needs (pinned: they are also the source of the uploads)
    opal_host::PinVec<int> pairLen;       // [items] lengths
    opal_host::PinVec<int> pairSeq;       // [items] caller indices, -1 = absent
    opal_host::PinVec<uint64_t> seqOff;   // [items] start of each sequence in `res`
    std::vector<int> emptySeqs;           // caller indices of zero-length sequences (answered on the host)
    std::vector<int> lastLens;            // cached contexts: the lengths of the database the context holds (caller order) ...
    bool lastValid = false;               // ... if its fill completed: the next call may find the same database again
    bool speculate = true;                // search a re-seen database before its residues have been compared (db_fill)
    opal_host::OrderScratch ordScratch;   // scratch of the ordering, kept across calls (no fresh pages per call)
    // the ordering on the device (order_kernel.hip): lengths in caller order (pinned staging + device), its scratch, the
    // zero-length sequences it found
    opal_host::PinVec<int> hLens, hEmpty;
    opal_host::DevBuf<int> dLens, dEmpty;
    opal_host::PinVec<uint64_t> hCuts;    // answers of opal_hip_order_cuts: {item, byte offset} per target of cutTargets
    opal_host::DevBuf<uint64_t> dCuts;
    std::vector<uint64_t> cutTargets;     // ... the byte offsets asked about, in the order plan_cuts asks (empty: ask the host's seqOff)
    uint64_t cutTargetsChunk = 0, cutTargetsPiece = 0;  // ... for this chunk / piece size
    bool hostSeqOffValid = true;          // db->seqOff (host) holds the item offsets (a device ordering does not bring them back)
    opal_host::DevBuf<unsigned char> dOrderScratch;
    bool fillAborted = false;             // db_fill gave up because its `meanwhile` hook said so (not a failure of the device)
    bool deviceOrdered = false;           // the item arrays of the current database were written on the device (nothing to upload)
    // device: the packed database
    opal_host::DevBuf<unsigned char> res;
    opal_host::DevBuf<uint64_t> dSeqOff;
    opal_host::DevBuf<int> dPairLen, dPairSeq;
    // width 16 only, built on first use: pair token streams
    opal_host::DevBuf<uint16_t> tokens;
    opal_host::DevBuf<uint64_t> pairOff;
    opal_host::PinVec<uint64_t> hPairOff;
    size_t numTokens = 0;
    bool tokensValid = false;
    bool asyncTimed = false;   // search_core is running for opalx_search_device_async: it times itself with the events of slot asyncHead
    bool asyncEvents = false;  // ... and recorded them
    // searches queued by opalx_search_device_async and not yet waited for: [asyncTail, asyncHead), slot = index % OPALX_ASYNC_DEPTH
    unsigned asyncHead = 0, asyncTail = 0;
    unsigned asyncWaited = 0;  // searches waited for so far (opalx_db_wait2: the one before has a stop event to measure from)
    void* asyncEv[2 * OPALX_ASYNC_DEPTH] = {};   // their start / stop events (timed)
    // the pinned staging copies of the search header (query, matrix, zeroed counters): one slot per search in flight, reused
    // only after the upload recorded behind it has run
    unsigned hdrSeq = 0;
    void* hdrEv[OPALX_ASYNC_DEPTH] = {};
    bool hdrEvSet[OPALX_ASYNC_DEPTH] = {};
    // Queued searches take turns on TWO streams (mainStream / pipeStream, ladder.cpp: SearchPipe): a search that is one grid counts
    // its workgroups in as they start (the "started" word of its own header slot), and the grid of the search queued behind it -- on the
    // other stream, behind a one-wave wait kernel -- reaches the dispatcher when the last of them has: it fills the CUs the first one
    // leaves as it drains (C2: a fixed 0.15-0.2 ms of fill and drain per 1.5 ms search on one stream). Whatever is not one grid
    // (side launches, row chunks, a re-run list) waits for the whole search before it and is waited for as a whole.
    void* pipeStream = nullptr;
    void* curStream = nullptr;          // not null while search_core runs a queued search: the stream main() hands out
    unsigned char* hdrDev = nullptr;    // the device header slot of the search enqueued last (header.p + slot * slot bytes)
    struct Pipe {
        bool active = false;      // search_core is running a queued search on curStream
        bool resolved = true;     // ... its dependency on the search before it has been enqueued
        bool untracked = false;   // ... it has launched something beside one tracked grid
        bool launched = false;    // ... it has launched its tracked grid
        void* startEv = nullptr;  // ... its start event, recorded behind the wait in front of its grid (nullptr: at the start of the search)
        int lastIdx = 0;          // stream of the search enqueued last: 0 mainStream, 1 pipeStream
        int blocks = 0;           // ... workgroups it has launched that count themselves in on its header's "started" word
        int prevKind = 0;         // the search enqueued last: 0 nothing in flight, 1 one tracked grid, 2 anything else
        int prevTarget = 0;       // kind 1: the "started" word of its header once all its workgroups have started ...
        int* prevCtr = nullptr;   // ... and where that word is
        void* done[2] = {nullptr, nullptr};  // event behind the last search on mainStream / pipeStream
        bool unjoined = false;    // pipeStream carries work that mainStream has not been made to wait for
        bool foreign = true;      // mainStream has been handed out outside a queued search since the last one: whatever was enqueued
                                  // on it (a search outside the queue, the multi-GPU handle's peer copies behind a shard's search) the
                                  // next queued search waits for as a whole
    } pipe;
    bool clockProbe = false;   // opalx_db_clock_probe
    bool timelineProbe = false;
    opal_host::DevBuf<unsigned long long> timeline;  // [2 x 65536 workgroups] ticks, then [2 x 65536] shader cycles; zeroed before a probed search
    double lastClockGhz = 0;
    // One "search header" per search, uploaded with a single copy: {counters[4], clock probe[2 x u64], score matrix, query}. ovfCount[0] =
    // length of the 16-bit rung's re-run list, [1] = width-32 wrap count.
    opal_host::DevBuf<unsigned char> header;
    opal_host::PinBuf<unsigned char> hHeader;
    struct View32 { int* p = nullptr; } ovfCount, matrix;
    struct View8 { unsigned char* p = nullptr; } query;
    opal_host::DevBuf<int> ovfItems;  // [items]
    // device: per-search scratch
    opal_host::DevBuf<unsigned char> boundary;  // 8 B per residue: chunk-boundary rows of long queries (allocated on first use)
    opal_host::DevBuf<unsigned long long> pairBnd;  // pair_kernel.hpp: the bottom rows the waves of a pair hand each other (allocated on first use)
    opal_host::DevBuf<int> scratch;             // [2N] per-sequence values carried across query chunks
    opal_host::DevBuf<int> results;             // [3N] device staging of host-array searches
    opal_host::DevBuf<int> quads;               // degenerate results on their way to the scatter kernel
    // start-location pass (rev_kernel.hip): caller index -> item, flags and the host-redo list
    opal_host::DevBuf<int> seqItem;
    bool seqItemValid = false;
    opal_host::DevBuf<int> revFlags;
    opal_host::DevBuf<int> redo;
    // alignment operations on the device (trace_kernel.hip): flag matrices, operation strings, per-batch lists and outputs
    opal_host::DevBuf<unsigned char> trDirs, trOps;
    opal_host::DevBuf<uint64_t> trOff;
    opal_host::DevBuf<int> trItems, trOut;
    // launch plans of this database, keyed by what they depend on (cleared when the database is refilled): a search
    // with the same query length, alphabet and mode re-uses the class split instead of recomputing it with the GPU idle
    std::map<std::string, std::vector<int>> planCache;
    // staging used while (re)filling the database
    opal_host::DevBuf<unsigned char> raw;    // a flat device database in caller order is reordered from here / from the caller's buffer
    opal_host::DevBuf<uint64_t> rawOff;
    opal_host::PinVec<uint64_t> hRawOff;
    opal_host::PinBuf<unsigned char> hRaw;   // the gathered residues in item order, chunk by chunk on their way to `res`
    opal_host::PinBuf<int> hResults;
    opal_host::UploadGate gate;
    std::function<void(int)> gatherFn;  // the pool job of a (re)fill in flight (alive until db_fill_finish)
    bool gatherPending = false;
    double fillT0 = 0;  // (debug timing)
    // Streams: everything of one search runs on `main` (a non-blocking stream: no implicit ordering
    // against the legacy null stream or against other contexts); long-sequence classes run on up to three
    // HIGH-priority side streams (hardware queues of their own), forked from and joined back into `main` with events.
    // `copy` carries the residue chunks of a (re)fill, `aux` every second slice of a search launched while they arrive.
    void* mainStream = nullptr;
    void* copyStream = nullptr;
    void* copyStream2 = nullptr;  // odd chunks (the per-copy set-up of one overlaps the transfer of the other)
    void* auxStream = nullptr;
    void* swapStream = nullptr;   // the role-swapped launch of a search's longest targets (high priority, beside everything else)
    void* streams[3] = {nullptr, nullptr, nullptr};
    void* events[13] = {};  // 0 fork, 1-3 joins, 4-5 timing, 6 aux join, 7 fill, 8-10 fork / joins of the 32-bit and unskewable prefixes beside row chunks, 11-12 fork / join of the swapped launch
    void* main() {
        if (curStream) return curStream;
        if (!mainStream) mainStream = opal_hip_stream_create();
        // (outside a queued search: whatever the queued searches left on the second stream comes first)
        if (pipe.unjoined && mainStream) { pipe.unjoined = false; if (pipe.done[1]) opal_hip_stream_wait_event(mainStream, pipe.done[1]); }
        pipe.foreign = true;
        return mainStream;
    }
    void* copy() {
        if (!copyStream) copyStream = opal_hip_stream_create();
        if (!copyStream2) copyStream2 = opal_hip_stream_create();
        return copyStream && copyStream2 ? copyStream : nullptr;
    }
    void* aux() {
        if (!auxStream) auxStream = opal_host::opts().auxHighPrio ? opal_hip_stream_create_high() : opal_hip_stream_create();
        return auxStream;
    }
    void* swap() {
        if (!swapStream) swapStream = opal_hip_stream_create_high();
        return swapStream;
    }
    void* stream(int i) {
        if (!streams[i]) streams[i] = opal_host::opts().sideStreamsNormalPrio ? opal_hip_stream_create() : opal_hip_stream_create_high();
        return streams[i];
    }
    void* event(int i) {
        if (!events[i]) events[i] = (i == 4 || i == 5 || i == 7) ? opal_hip_event_create() : opal_hip_event_create_untimed();  // (4, 5: kernel timing)
        return events[i];
    }
    // Wait for everything this handle may have in flight (used on error paths before buffers are reused).
    void quiesce() {
        for (void* st : streams) if (st) opal_hip_stream_sync(st);
        if (copyStream) opal_hip_stream_sync(copyStream);
        if (copyStream2) opal_hip_stream_sync(copyStream2);
        if (auxStream) opal_hip_stream_sync(auxStream);
        if (swapStream) opal_hip_stream_sync(swapStream);
        if (pipeStream) opal_hip_stream_sync(pipeStream);
        if (mainStream) opal_hip_stream_sync(mainStream);
        pipe.prevKind = 0; pipe.unjoined = false;  // (nothing in flight any more)
    }
};

namespace opal_host {

bool device_ok();
void db_release(OpalxDb* db);

/*
 * (Re)fill `db` with `n` sequences of lengths `lens`. Residues come from
 *   RAW_POINTERS     getSeq(i): separate host allocations (or slices of one host buffer), gathered in ITEM order into
 *                    pinned memory by the thread pool, every chunk uploaded as soon as it is complete;
 *   RAW_DEVICE_FLAT  one flat device buffer, sequence i at flatOff(i): reordered on the device.
 * With `async` (RAW_POINTERS only) the call returns while the pool is still gathering: launches must pass through
 * gate_items() for the items they read, and the caller must call db_fill_finish() before it uses the pool again or
 * gives the context back. Buffers of a cached context are reused when large enough.
 */
enum RawKind { RAW_POINTERS, RAW_DEVICE_FLAT };
bool db_fill(OpalxDb* db, int n, const int* lens, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
             const unsigned char* flatDevice, const std::function<uint64_t(int)>& flatOff, bool async = false,
             const std::function<bool()>* meanwhile = nullptr);
// (`meanwhile`, if given, is called exactly once, at the point of the fill where the calling thread would otherwise wait --
// while the device orders the lengths --; if it returns false the fill is dropped: db_fill returns false with db->fillAborted set)
bool db_fill_finish(OpalxDb* db);  // waits for the pool; false if a chunk failed
// After db_fill_finish: the fill had recognised the previous call's database, searched it without waiting for the
// comparison, and some chunk turned out to differ (it has been uploaded since): the search must be repeated.
bool fill_changed(const OpalxDb* db);
// Make `stream` wait for the residues of items [firstItem, firstItem + count) (host-waits until their chunks have been
// enqueued, then one hipStreamWaitEvent per chunk). No-op without an active gate. False if a chunk failed.
bool gate_items(OpalxDb* db, void* stream, int firstItem, int count, bool needDevice = true);
// (needDevice = false: the launch reads the residues from the pinned staging buffer itself -- zero copy --, so it only has to
// wait until its chunks have been GATHERED)
// Make `stream` wait for every chunk copy of the last fill (after a zero-copy search: what follows reads the device copy).
bool wait_uploads(OpalxDb* db, void* stream);
// The 16-bit pair token streams of the width-16 kernels (built on the device on first use). 0 on success.
int ensure_pair_tokens(OpalxDb* db);

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
    long long qBound;      // no cell of this query's DP matrices holds more: every query residue against its best letter (<= Q * maxScore)
    bool zeroCopy = false; // launches read the residues from the pinned staging buffer (db->hRaw) instead of the device copy
    void* stream = nullptr; // launch on this stream instead of the handle's main one (row chunks of a 32-bit prefix beside the 16-bit ones)
    int* sideBlocks = nullptr;  // (launches on `stream`) counts their workgroups: a persistent streaming launch beside them leaves that many CUs free
};

// Lengths of the items of a launch range: item i has length lens[(idx ? idx[i] : i) * stride]
// (width-16 items are pairs: stride 2 over pairLen; width-17/18/32 items are pair*2+half: stride 1; a re-run list
// goes through idx).
struct LenView {
    const int* lens; int stride; const int* idx;
    int operator()(int i) const { return lens[(size_t)(idx ? idx[i] : i) * (size_t)stride]; }
};

int kernel_vgprs(int width, int mode, int swEndPass, int multi, int skew, int rows);
int pick_occupancy(int vgprs, size_t ldsBytes, int* wpbOut, int* ldsPadOut, int maxWpb = 8);
std::vector<LaunchClass> plan_classes(int width, int mode, int swEndPass, int skew, int Q, int A, LenView lenOf, int first,
                                      int count, int maxGFactor = 0);
bool plan_chunks(int width, int mode, int swEndPass, int Q, int A, long long nItems, double meanLen, int maxLen, int* G, int* rows, int* wpb);
struct StreamPlan { int width = 0, G = 0, rows = 0, passes = 1; };  // passes > 1: the query runs in that many row chunks
StreamPlan plan_stream(int Q, int A, int mode, int wantEnd, long long residues, int longest, int numSeqs);
int plan_swap(int Q, int A, int mode, LenView lenOf, int first, int count, long long residues, int* rowsOut, bool fits16 = true, bool quiet = false);  // items for swap_kernel.hpp (0: none)
// items for pair_kernel.hpp (0: none): the targets that need 32 bits (need32 of them, a prefix) and the ones behind them that would
// outlast the search alone on a wave; rows / waves: the kernel's shape for this query
int plan_pair(int Q, int A, int mode, LenView lenOf, int first, int count, long long residues, int need32, bool fits16, int* rowsOut, int* wavesOut);

long long query_score_bound(const unsigned char* query, int Q, const int* M, int A);  // ladder.cpp: the a-priori bound of the 16-bit rung
bool args_fit_int32(int Go, int Ge, const int* M, int A);
int ensure_query_buffers(OpalxDb* db, const unsigned char* query, int Q, const int* M, int A);
int search_core(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A, int searchType, int mode,
                int* dScore, int* dEndQ, int* dEndT, float* kernelMs, bool finalSync = true);

int search_starts(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A, int mode,
                  const int* dScore, const int* dEndQ, const int* dEndT, int* dStartQ, int* dStartT, int* redoHost,
                  int* redoCount);
// Operations traced on the device, by position in the shard: at(i)[0 .. len[i]), or len[i] < 0 (not traced: the host does it).
struct TraceOut {
    std::vector<std::vector<unsigned char>> chunks;  // one buffer of operation strings per device batch, as downloaded
    std::vector<int> chunk;                          // [n] which one
    std::vector<uint64_t> off;
    std::vector<int> len;
    const unsigned char* at(size_t i) const { return chunks[(size_t)chunk[i]].data() + off[i]; }
};
void device_traceback(OpalxDb* db, int Q, int Go, int Ge, int A, int mode, int n, const int* score, const int* endQ,
                      const int* endT, const int* startQ, const int* startT, const int* dEndQ, const int* dEndT,
                      const int* dStartQ, const int* dStartT, TraceOut& out);

std::vector<int> call_devices();
// Persistent host workers of the multi-GPU paths, one per SLOT of a device list (a device listed twice has two): fn(k) runs on
// the worker of slot k, which stays bound to its GPU between calls (no thread creation, no first-call device set-up per
// search: ~50-100 us of a 1.5 ms search at 8 GPUs). Returns when every fn(k) has returned. Calls from several host threads at
// once: one uses the workers, the others fall back to threads of their own for that call.
void run_on_device_workers(size_t slots, const std::function<void(size_t)>& fn);
std::vector<std::vector<int>> deal_groups(const int* lens, int n, int parts);

}  // namespace opal_host
