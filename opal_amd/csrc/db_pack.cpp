// db_pack.cpp -- the packed database: ordering and pairing of the sequences, (re)fill of a device-resident database
// from N host allocations / one flat host buffer / one flat device buffer, the cached contexts opalSearchDatabase
// borrows, and the opalx_db_* handle entries of include/opal_hip.h. Takes the place of the reference's per-call
// pointer walk (reference src/opal.cpp:245-268: the reference reads db[i] in place; a GPU needs the residues in HBM).
#include "opal_internal.hpp"

namespace opal_host {

std::atomic<long long> g_counters[CNT_COUNT];

// Order and pairing (no residue is touched): stable sort by length, longest first.
void pack_order(int n, const int* lens, HostPack& out) {
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

// The HIP "current device" is a per-thread setting, so the check (and the optional OPAL_HIP_DEVICE
// selection) is done once per calling thread. Without OPAL_HIP_DEVICE the thread's current device is
// used as it is (bench.py sets it through torch before calling in).
bool device_ok() {
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

void db_release(OpalxDb* db) {
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

/*
 * (Re)fill `db` with the packed form of `hp`. Residues come from
 *   RAW_POINTERS     getSeq(i): N separate host allocations -> gathered into pinned memory by the thread pool,
 *                    every ~2 MB chunk uploaded as soon as it is complete;
 *   RAW_HOST_FLAT    one contiguous host buffer (uploaded as it is, the device packer reads at hp.rawOff);
 *   RAW_DEVICE_FLAT  the same already in device memory (nothing is copied).
 * Buffers of a cached context are reused when large enough.
 */
bool db_fill(OpalxDb* db, HostPack& hp, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
                    const unsigned char* flat, const int* orderLens, int orderN) {
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

}  // namespace opal_host

using namespace opal_host;

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

extern "C" long long opalx_stream_launch_count(void) { return g_counters[CNT_STREAM_SW].load() + g_counters[CNT_STREAM_GLOBAL].load(); }

extern "C" int opalx_counters(long long* out, int capacity) {
    for (int i = 0; i < capacity && i < (int)CNT_COUNT; i++) out[i] = g_counters[i].load();
    return (int)CNT_COUNT;
}

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

extern "C" void opalx_db_clock_probe(OpalxDb* db, int enable) { if (db) db->clockProbe = enable != 0; }
extern "C" double opalx_db_last_clock_ghz(const OpalxDb* db) { return db ? db->lastClockGhz : 0.0; }
extern "C" int opalx_db_num_sequences(const OpalxDb* db) { return db ? db->numSeqs : 0; }
extern "C" int64_t opalx_db_num_residues(const OpalxDb* db) { return db ? db->residues : 0; }
