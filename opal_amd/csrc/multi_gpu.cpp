// multi_gpu.cpp -- the devices of a call, the deal of sequences to them, and the multi-GPU database handle with its
// device-side gather (the reference's natural shard point is its per-chunk call, reference
// src/opal_aligner.cpp:125-160: sequences are independent, query and matrix are broadcast constants).
#include "opal_internal.hpp"

namespace opal_host {

// ------------------------------------------------------------------------------------ devices of a call
// OPAL_HIP_DEVICES = "all" | "0,1,2,..." makes opalSearchDatabase shard every call over those GPUs. A device
// may be listed more than once (two contexts on one GPU): that is how the sharding is tested on a one-GPU box.
std::vector<int> call_devices() {
    std::vector<int> out;
    const std::string devs = opts().devices;  // (a copy: opalx_set_option may replace the string)
    const char* env = devs.c_str();
    if (!*env) return out;
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

// ------------------------------------------------------------------------------------ persistent per-slot workers
namespace {
struct SlotWorker {
    std::mutex m;
    std::condition_variable cv;
    const std::function<void(size_t)>* fn = nullptr;  // the job at hand (nullptr: none)
    std::atomic<int> posted{0};                        // (polled for a while after a job before the worker blocks)
    size_t slot = 0;
    std::atomic<bool> spin{true};
    std::atomic<int>* left = nullptr;
    std::mutex* doneM = nullptr;
    std::condition_variable* doneCv = nullptr;
    void loop() {
        for (;;) {
            const std::function<void(size_t)>* job = nullptr;
            // a caller that sends search after search finds the worker still polling: no wake-up latency between two searches
            // (only while the workers of a device list fit the CPU quota: eight spinning workers on a quota of four cores take
            // the time the gather needs)
            if (spin) {
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
                while (!posted.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
            }
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return fn != nullptr; });
                job = fn;
                fn = nullptr;
                posted.store(0, std::memory_order_relaxed);
            }
            (*job)(slot);
            if (left->fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(*doneM);
                doneCv->notify_all();
            }
        }
    }
};
struct SlotWorkers {
    std::mutex busy;                                  // one job at a time on the shared workers
    std::vector<std::unique_ptr<SlotWorker>> w;       // never destroyed (like the host pool: see opalx_release_cached)
    std::mutex doneM;
    std::condition_variable doneCv;
    static SlotWorkers& get() { static SlotWorkers* s = new SlotWorkers; return *s; }
};
}  // namespace

void run_on_device_workers(size_t slots, const std::function<void(size_t)>& fn) {
    if (slots == 0) return;
    if (slots == 1) { fn(0); return; }
    SlotWorkers& sw = SlotWorkers::get();
    std::unique_lock<std::mutex> own(sw.busy, std::try_to_lock);
    if (!own.owns_lock()) {  // another host thread is using the workers: threads of this call's own
        std::vector<std::thread> threads;
        for (size_t k = 0; k < slots; k++) threads.emplace_back([&fn, k] { fn(k); });
        for (auto& th : threads) th.join();
        return;
    }
    while (sw.w.size() < slots) {
        sw.w.emplace_back(new SlotWorker);
        SlotWorker* me = sw.w.back().get();
        me->slot = sw.w.size() - 1;
        me->doneM = &sw.doneM; me->doneCv = &sw.doneCv;
        std::thread([me] { me->loop(); }).detach();
    }
    std::atomic<int> left{(int)slots};
    // (the workers poll for their next job only if all of them together stay within half the CPU quota)
    static const double quota = HostPool::cpu_quota_cores();
    const bool spin = quota <= 0 || (double)slots <= 0.5 * quota;
    for (size_t k = 0; k < slots; k++) {
        SlotWorker* me = sw.w[k].get();
        me->spin.store(spin, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(me->m);
            me->left = &left;
            me->fn = &fn;
            me->posted.store(1, std::memory_order_release);
        }
        me->cv.notify_one();
    }
    std::unique_lock<std::mutex> lk(sw.doneM);
    sw.doneCv.wait(lk, [&] { return left.load(std::memory_order_acquire) == 0; });
}

/*
 * Deal sequences to `parts` shards: indices sorted by length (descending), cut into groups of
 * 64 consecutive ones (sequences of one wave stay together), every group to the shard with the fewest residues
 * so far. Returns shard[k] = list of positions into `lens`.
 */
std::vector<std::vector<int>> deal_groups(const int* lens, int n, int parts) {
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

}  // namespace opal_host

// ------------------------------------------------------------------------------------ one database over several GPUs
struct OpalxMultiDb {
    std::vector<int> devices;
    std::vector<OpalxDb*> shard;             // one packed database per listed device
    std::vector<int> count, first;           // sequences of each shard, its offset in the gathered block
    std::vector<opal_host::DevBuf<int>> res; // per shard, on its device: {score, endQ, endT} x count
    opal_host::DevBuf<int> gathered;         // root: [3][total] -- the shards' blocks side by side, per result vector
    opal_host::DevBuf<int> perm;             // root: gathered position -> database index
    std::vector<void*> done;                 // per shard: recorded behind its peer copy
    void* rootStream = nullptr;
    int total = 0;
    // searches queued by opalx_multi_search_device_async and not yet waited for: [tail, head), slot = index % OPALX_ASYNC_DEPTH
    unsigned head = 0, tail = 0;
    void* rootDone[OPALX_ASYNC_DEPTH] = {};  // recorded on the root stream behind a search's permutation
    int queuedRc[OPALX_ASYNC_DEPTH] = {};
    // OPAL_HIP_MULTI_GATHER=rccl: the gather is ONE ncclGather per search instead of peer copies. Every shard's
    // block is padded to the largest shard (a gather moves equal counts): res[k] = [3][maxCount], the root receives
    // [shards][3][maxCount] and permutes through permPad (-1 = padding).
    void* rccl = nullptr;
    int maxCount = 0;
    opal_host::DevBuf<int> gatheredR, permPad;
};

using namespace opal_host;

extern "C" void opalx_multi_destroy(OpalxMultiDb* m) {
    if (!m) return;
    const int cur = opal_hip_get_device();
    for (size_t k = 0; k < m->shard.size(); k++) {
        opal_hip_set_device(m->devices[k]);
        if (m->shard[k]) opalx_db_destroy(m->shard[k]);
        if (k < m->res.size()) m->res[k].release();
        if (k < m->done.size()) opal_hip_event_destroy(m->done[k]);
    }
    if (m->rccl) opal_hip_rccl_destroy(m->rccl);
    if (!m->devices.empty()) {
        opal_hip_set_device(m->devices[0]);
        m->gathered.release(); m->perm.release(); m->gatheredR.release(); m->permPad.release();
        if (m->rootStream) opal_hip_stream_sync(m->rootStream);
        for (void*& ev : m->rootDone) { opal_hip_event_destroy(ev); ev = nullptr; }
        opal_hip_stream_destroy(m->rootStream);
    }
    if (cur >= 0) opal_hip_set_device(cur);
    delete m;
}

extern "C" OpalxMultiDb* opalx_multi_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length,
                                                 const int* devices, int num_devices) {
    if (!device_ok() || num_devices < 1 || db_length < 0) return nullptr;
    const int ndev = opal_hip_device_count();
    for (int k = 0; k < num_devices; k++)
        if (devices[k] < 0 || devices[k] >= ndev) return nullptr;
    OpalxMultiDb* m = new (std::nothrow) OpalxMultiDb;
    if (!m) return nullptr;
    const int cur = opal_hip_get_device();
    m->devices.assign(devices, devices + num_devices);
    m->total = db_length;
    std::vector<int> lens((size_t)db_length);
    for (int i = 0; i < db_length; i++) lens[(size_t)i] = (int)(offsets[i + 1] - offsets[i]);
    const std::vector<std::vector<int>> deal = deal_groups(lens.data(), db_length, num_devices);
    m->shard.assign((size_t)num_devices, nullptr);
    m->res.resize((size_t)num_devices);
    m->done.assign((size_t)num_devices, nullptr);
    std::vector<int> permHost;
    permHost.reserve((size_t)db_length);
    bool ok = true;
    for (const auto& ids : deal) m->maxCount = std::max(m->maxCount, (int)ids.size());
    if (opts().multiGatherRccl) {
        m->rccl = opal_hip_rccl_init(devices, num_devices);
        if (!m->rccl) fprintf(stderr, "[opal] OPAL_HIP_MULTI_GATHER=rccl: %s -- this handle gathers by peer copies\n", opal_hip_last_error());
    }
    for (int k = 0; k < num_devices && ok; k++) {
        const std::vector<int>& ids = deal[(size_t)k];
        m->first.push_back((int)permHost.size());
        m->count.push_back((int)ids.size());
        permHost.insert(permHost.end(), ids.begin(), ids.end());
        if (opal_hip_set_device(devices[k])) { ok = false; break; }
        std::vector<int> slens(ids.size());
        for (size_t i = 0; i < ids.size(); i++) slens[i] = lens[(size_t)ids[i]];
        OpalxDb* db = new (std::nothrow) OpalxDb;
        if (!db) { ok = false; break; }
        m->shard[(size_t)k] = db;
        ok = db_fill(db, (int)ids.size(), slens.data(), RAW_POINTERS,
                     [residues, offsets, &ids](int i) { return residues + offsets[ids[(size_t)i]]; }, nullptr, nullptr) &&
             m->res[(size_t)k].ensure(3 * std::max<size_t>(m->rccl ? (size_t)m->maxCount : ids.size(), 1)) &&
             (m->done[(size_t)k] = opal_hip_event_create_untimed()) != nullptr;
        if (ok && devices[k] != devices[0]) opal_hip_enable_peer(devices[0]);  // (without it the copy is staged through the host)
    }
    if (ok) {
        ok = !opal_hip_set_device(devices[0]) && m->gathered.ensure(3 * std::max<size_t>((size_t)db_length, 1)) &&
             m->perm.ensure(std::max<size_t>((size_t)db_length, 1)) && (m->rootStream = opal_hip_stream_create()) != nullptr;
        if (ok && db_length)
            ok = !opal_hip_memcpy_h2d(m->perm.p, permHost.data(), (size_t)db_length * sizeof(int), m->rootStream) && !opal_hip_stream_sync(m->rootStream);
        if (ok && m->rccl && db_length) {
            const size_t K = (size_t)num_devices, per = (size_t)m->maxCount;
            std::vector<int> pad(K * per, -1);
            for (size_t k = 0; k < K; k++)
                for (int i = 0; i < m->count[k]; i++) pad[k * per + (size_t)i] = permHost[(size_t)m->first[k] + (size_t)i];
            ok = m->gatheredR.ensure(K * 3 * per) && m->permPad.ensure(K * per) &&
                 !opal_hip_memcpy_h2d(m->permPad.p, pad.data(), pad.size() * sizeof(int), m->rootStream) && !opal_hip_stream_sync(m->rootStream);
        }
    }
    if (cur >= 0) opal_hip_set_device(cur);
    if (!ok) { opalx_multi_destroy(m); return nullptr; }
    return m;
}

/*
 * Enqueue one search on every shard and the gather behind it; nothing is waited for. Every shard's search goes through its
 * handle's own queue (opalx_search_device_async: header slot, timing events), its block then travels to the root GPU by peer
 * DMA on the shard's stream -- behind the root's permutation of the PREVIOUS queued search, which is still reading the gathered
 * block -- and the root stream permutes the three vectors into database order and records the search's `rootDone` event.
 */
extern "C" int opalx_multi_search_device_async(OpalxMultiDb* m, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                               const int* score_matrix, int alphabet_length, int search_type, int mode,
                                               int* d_score, int* d_end_query, int* d_end_target) {
    if (!m || !d_score) return OPAL_ERR_NO_SIMD_SUPPORT;
    const int wantEnd = search_type != OPAL_SEARCH_SCORE;
    if (wantEnd && (!d_end_query || !d_end_target)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (mode != OPAL_MODE_NW && mode != OPAL_MODE_HW && mode != OPAL_MODE_OV && mode != OPAL_MODE_SW) return OPAL_ERR_INVALID_MODE;
    if (m->head - m->tail >= OPALX_ASYNC_DEPTH - 1) return OPAL_ERR_NO_SIMD_SUPPORT;  // the queue of this handle is full
    const size_t K = m->shard.size();
    const int nres = wantEnd ? 3 : 1;
    const size_t total = (size_t)m->total;
    const int slot = (int)(m->head % OPALX_ASYNC_DEPTH);
    void* prevDone = m->head > m->tail ? m->rootDone[(m->head - 1) % OPALX_ASYNC_DEPTH] : nullptr;
    std::vector<int> rcs(K, 0);
    std::vector<char> queued(K, 1);  // the shard's search took a place in its handle's queue
    const int cur = opal_hip_get_device();
    run_on_device_workers(K, [&](size_t k) {
        const int n = m->count[k];
        if (!n && !m->rccl) return;
        // device_ok() first: its once-per-thread OPAL_HIP_DEVICE selection must not override the shard's GPU
        if (!device_ok() || opal_hip_set_device(m->devices[k])) { rcs[k] = OPAL_ERR_NO_SIMD_SUPPORT; return; }
        OpalxDb* db = m->shard[k];
        if (!n) {  // (RCCL: a shard without sequences still takes part in the gather, behind the root's previous permutation)
            if (prevDone && db->main() && opal_hip_stream_wait_event(db->main(), prevDone)) rcs[k] = OPAL_ERR_NO_SIMD_SUPPORT;
            return;
        }
        int* r = m->res[k].p;
        const size_t vs = m->rccl ? (size_t)m->maxCount : (size_t)n;  // stride between the three result vectors of the shard's block
        const unsigned headBefore = db->asyncHead;
        rcs[k] = opalx_search_device_async(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode, r,
                                           wantEnd ? r + vs : nullptr, wantEnd ? r + 2 * vs : nullptr);
        if (rcs[k] == OPAL_ERR_NO_SIMD_SUPPORT || rcs[k] == OPAL_ERR_INVALID_MODE) return;
        // (a search refused before anything was launched -- arguments beyond 32 bits: OPAL_ERR_OVERFLOW -- took no place in
        // the shard's queue and wrote no block: there is nothing to gather and nothing to wait for)
        if (db->asyncHead == headBefore) { queued[k] = 0; return; }
        // the path's only exchange: this shard's block to the root GPU (peer DMA over xGMI), vector by vector into its
        // place in the gathered block, on the shard's own stream behind its kernels
        void* st = db->main();
        int e = prevDone ? opal_hip_stream_wait_event(st, prevDone) : 0;
        if (m->rccl) { if (e) rcs[k] = OPAL_ERR_NO_SIMD_SUPPORT; return; }  // (the gather is one group call, below)
        for (int v = 0; v < nres; v++)
            e |= opal_hip_memcpy_peer(m->gathered.p + (size_t)v * total + (size_t)m->first[k], m->devices[0], r + (size_t)v * n, m->devices[k],
                                      (size_t)n * sizeof(int), st);
        e |= opal_hip_event_record(m->done[k], st);
        if (e) rcs[k] = OPAL_ERR_NO_SIMD_SUPPORT;
    });
    int rc = 0;
    for (size_t k = 0; k < K; k++) {
        if (rcs[k] == OPAL_ERR_NO_SIMD_SUPPORT || rcs[k] == OPAL_ERR_INVALID_MODE) rc = rcs[k];
        else if (rcs[k] && !rc) rc = rcs[k];
    }
    // Nothing was enqueued anywhere (every shard refused the same arguments): the multi-GPU queue keeps its length too, so
    // that opalx_multi_wait never pops a shard entry that belongs to another search. (Some shards queued, others not, cannot
    // come from the arguments -- they are the same everywhere --: treated as a failure, the queues are drained below.)
    size_t nQueued = 0, nAsked = 0;
    for (size_t k = 0; k < K; k++) if (m->count[k]) { nAsked++; nQueued += queued[k] ? 1 : 0; }
    if (rc != OPAL_ERR_NO_SIMD_SUPPORT && rc != OPAL_ERR_INVALID_MODE && nAsked > 0 && nQueued == 0) {
        if (cur >= 0) opal_hip_set_device(cur);
        return rc ? rc : OPAL_ERR_NO_SIMD_SUPPORT;
    }
    int e = opal_hip_set_device(m->devices[0]);
    if (nQueued != nAsked) e = 1;
    if (!e && m->rccl && rc != OPAL_ERR_NO_SIMD_SUPPORT && rc != OPAL_ERR_INVALID_MODE) {
        // the path's one collective: every shard's padded block to the root in ONE ncclGather, each rank on its shard's stream
        // behind its kernels (a shard without sequences still takes part: a collective needs every rank)
        std::vector<const void*> send(K);
        std::vector<void*> streams(K);
        for (size_t k = 0; k < K; k++) { send[k] = m->res[k].p; streams[k] = m->shard[k]->main(); }
        e |= opal_hip_rccl_gather(m->rccl, send.data(), m->gatheredR.p, (size_t)nres * (size_t)m->maxCount, streams.data());
        for (size_t k = 0; k < K && !e; k++) {
            e |= opal_hip_set_device(m->devices[k]);
            e |= opal_hip_event_record(m->done[k], streams[k]);
        }
        e |= opal_hip_set_device(m->devices[0]);
    }
    if (!e && rc != OPAL_ERR_NO_SIMD_SUPPORT && rc != OPAL_ERR_INVALID_MODE) {
        // root: behind every shard's copy, the three vectors into database order
        for (size_t k = 0; k < K; k++)
            if (m->count[k] || m->rccl) e |= opal_hip_stream_wait_event(m->rootStream, m->done[k]);
        int* outs[3] = {d_score, d_end_query, d_end_target};
        if (m->rccl)
            for (int v = 0; v < nres; v++)
                e |= opal_hip_permute_blocks(m->gatheredR.p + (size_t)v * (size_t)m->maxCount, m->permPad.p, (int)K, m->maxCount, nres * m->maxCount, outs[v], m->rootStream);
        else
        for (int v = 0; v < nres; v++) e |= opal_hip_permute(m->gathered.p + (size_t)v * total, m->perm.p, m->total, outs[v], m->rootStream);
        if (!m->rootDone[slot]) m->rootDone[slot] = opal_hip_event_create_untimed();
        e |= !m->rootDone[slot] || opal_hip_event_record(m->rootDone[slot], m->rootStream);
    }
    if (e || rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) {
        // (an error on one shard: nothing of this handle stays in flight, the queue is emptied)
        for (size_t k = 0; k < K; k++) {
            opal_hip_set_device(m->devices[k]);
            m->shard[k]->quiesce();
            float ms;
            while (m->shard[k]->asyncTail != m->shard[k]->asyncHead) opalx_db_wait(m->shard[k], &ms);
        }
        opal_hip_set_device(m->devices[0]);
        opal_hip_stream_sync(m->rootStream);
        m->tail = m->head;
        if (cur >= 0) opal_hip_set_device(cur);
        return e ? OPAL_ERR_NO_SIMD_SUPPORT : rc;
    }
    m->queuedRc[slot] = rc;
    m->head++;
    if (cur >= 0) opal_hip_set_device(cur);
    return rc;
}

/* Wait for the OLDEST queued search of the handle (first in, first out); `kernel_ms`: the longest DP time over its shards. */
extern "C" int opalx_multi_wait(OpalxMultiDb* m, float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.f;
    if (!m) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (m->tail == m->head) return 0;
    const int slot = (int)(m->tail % OPALX_ASYNC_DEPTH);
    m->tail++;
    const int cur = opal_hip_get_device();
    int e = 0;
    for (size_t k = 0; k < m->shard.size(); k++) {
        if (!m->count[k]) continue;
        float ms = 0.f;
        e |= opal_hip_set_device(m->devices[k]);
        if (opalx_db_wait(m->shard[k], &ms)) e = 1;
        if (kernel_ms && ms > *kernel_ms) *kernel_ms = ms;
    }
    e |= opal_hip_set_device(m->devices[0]);
    if (m->rootDone[slot]) e |= opal_hip_event_sync(m->rootDone[slot]);
    if (cur >= 0) opal_hip_set_device(cur);
    return e ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
}

extern "C" int opalx_multi_gather_kind(const OpalxMultiDb* m) { return m && m->rccl ? 1 : 0; }

extern "C" int opalx_multi_search_device(OpalxMultiDb* m, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                         const int* score_matrix, int alphabet_length, int search_type, int mode,
                                         int* d_score, int* d_end_query, int* d_end_target, float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.f;
    if (!m) return OPAL_ERR_NO_SIMD_SUPPORT;
    // (searches still queued on the handle come first: this call returns with ITS results complete)
    while (m->tail != m->head)
        if (opalx_multi_wait(m, nullptr)) return OPAL_ERR_NO_SIMD_SUPPORT;
    const int rc = opalx_multi_search_device_async(m, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode,
                                                   d_score, d_end_query, d_end_target);
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE || m->tail == m->head) return rc;
    return opalx_multi_wait(m, kernel_ms) ? OPAL_ERR_NO_SIMD_SUPPORT : rc;
}
