// db_pack.cpp -- the packed database: ordering and pairing of the sequences, (re)fill of a device-resident database
// from N host allocations / one flat host buffer / one flat device buffer, the cached contexts opalSearchDatabase
// borrows, and the opalx_db_* handle entries of include/opal_hip.h. Takes the place of the reference's per-call
// pointer walk (reference src/opal.cpp:245-268: the reference reads db[i] in place; a GPU needs the residues in HBM).
#include "opal_internal.hpp"

namespace opal_host {

std::atomic<long long> g_counters[CNT_COUNT];

// Order and pairing (no residue is touched): stable counting sort by length, longest first, spread over the pool's
// threads (per-thread histograms; a thread places the sequences of its own caller range). Fills db->pairLen / pairSeq /
// seqOff (item order) and db->emptySeqs. The byte offset of an item follows from its bucket alone: all sequences of a
// bucket have the same length.
namespace {
struct OrderOut { int* len; int* seq; uint64_t* off; };
}
// `alloc(items, out)` provides the three output arrays once the number of items is known.
static bool order_into(int n, const int* lens, OrderScratch& scratch, std::vector<int>& emptySeqs, int64_t* residues, size_t* itemsOut,
                       const std::function<bool(size_t, OrderOut&)>& alloc) {
    emptySeqs.clear();
    OrderOut out{nullptr, nullptr, nullptr};
    const double to0 = now_ms();
    int maxLen = 0;
    for (int i = 0; i < n; i++) maxLen = std::max(maxLen, lens[i]);
    if (maxLen > (1 << 22)) {  // (absurd lengths: comparison sort)
        scratch.pos.clear(); scratch.byte.clear();
        std::vector<int> order;
        for (int i = 0; i < n; i++) {
            if (lens[i] > 0) order.push_back(i);
            else emptySeqs.push_back(i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
        const size_t m = order.size(), items = (m + 1) / 2 * 2;
        if (!alloc(items, out)) return false;
        uint64_t total = 0;
        for (size_t k = 0; k < items; k++) {
            const bool have = k < m;
            out.len[k] = have ? lens[order[k]] : 0;
            out.seq[k] = have ? order[k] : -1;
            out.off[k] = total;
            total += (uint64_t)out.len[k];
        }
        *residues = (int64_t)total;
        *itemsOut = items;
        return true;
    }
    HostPool& pool = HostPool::get();
    const size_t B = (size_t)maxLen + 1;  // bucket b holds the sequences of length maxLen - b; length 0 is not placed
    // (measured on the GPU box's host, 100 000 / 625 000 sequences: 1 thread 0.235 / 1.46 ms, 4 threads 0.26 / 0.83 median --
    // a job of the pool costs ~0.1 ms in wake-ups, so only large databases are worth spreading)
    int T = n >= 400000 ? std::min(pool.width(), 4) : 1;
    if (opts().orderThreads >= 1) T = std::min(opts().orderThreads, pool.width());  // tuning hook
    while (T > 1 && (size_t)T * B > (4u << 20)) T /= 2;
    std::vector<int>& hist = scratch.hist;  // (kept in the context: no fresh pages per call)
    hist.assign((size_t)T * B, 0);
    auto range = [&](int t, int& i0, int& i1) { i0 = (int)((long long)n * t / T); i1 = (int)((long long)n * (t + 1) / T); };
    pool.parallel_for(T, [&](int t) {
        int i0, i1; range(t, i0, i1);
        int* h = hist.data() + (size_t)t * B;
        for (int i = i0; i < i1; i++) if (lens[i] > 0) h[(size_t)maxLen - lens[i]]++;
    });
    const double to1 = now_ms();
    // bucket starts (item position and byte offset), and where every thread starts inside every bucket
    std::vector<int>& bucketPos = scratch.pos;
    std::vector<uint64_t>& bucketByte = scratch.byte;
    bucketPos.assign(B + 1, 0);
    bucketByte.assign(B + 1, 0);
    int pos = 0;
    uint64_t byte = 0;
    for (size_t b = 0; b < B; b++) {
        bucketPos[b] = pos; bucketByte[b] = byte;
        const int before = pos;
        for (int t = 0; t < T; t++) { int& h = hist[(size_t)t * B + b]; const int c = h; h = pos; pos += c; }
        byte += (uint64_t)(pos - before) * (uint64_t)(maxLen - (int)b);
        if ((int)b == maxLen - 1) break;  // (bucket maxLen = length 0: not placed)
    }
    const int m = pos;
    bucketPos[(size_t)maxLen] = m; bucketByte[(size_t)maxLen] = byte;  // (end of the last real bucket)
    const size_t items = ((size_t)m + 1) / 2 * 2;
    if (!alloc(items, out)) return false;
    std::vector<std::vector<int>> empties((size_t)T);
    const double to2 = now_ms();
    int* const pl = out.len; uint64_t* const so = out.off;
    // (the scattered writes go to ordinary memory and are copied to the caller's -- pinned -- array in one sweep: scattered
    // 4-byte stores to pinned memory measured 15 ns each)
    scratch.seq.resize(items);
    int* const ps = scratch.seq.data();
    // One job, two kinds of tasks: tasks [0, T) place the caller indices (ONE scattered write per sequence); tasks [T, 2T)
    // fill lengths and byte offsets bucket range by bucket range (sequential writes: every sequence of a bucket has the
    // same length, so both follow from the bucket table alone).
    pool.parallel_for(2 * T, [&](int task) {
        if (task < T) {
            int i0, i1; range(task, i0, i1);
            int* h = hist.data() + (size_t)task * B;
            for (int i = i0; i < i1; i++) {
                const int len = lens[i];
                if (len <= 0) { empties[(size_t)task].push_back(i); continue; }
                ps[h[(size_t)maxLen - len]++] = i;
            }
        } else {
            const int t = task - T;
            const size_t b0 = (size_t)((long long)maxLen * t / T), b1 = (size_t)((long long)maxLen * (t + 1) / T);  // my buckets
            for (size_t b = b0; b < b1; b++) {
                const int len = maxLen - (int)b;
                uint64_t off = bucketByte[b];
                for (int at = bucketPos[b], e = bucketPos[b + 1]; at < e; at++) { pl[at] = len; so[at] = off; off += (uint64_t)len; }
            }
        }
    });
    if ((size_t)m < items) ps[m] = -1;
    if (items) memcpy(out.seq, ps, items * sizeof(int));
    for (auto& v : empties) emptySeqs.insert(emptySeqs.end(), v.begin(), v.end());
    if (debug_on()) fprintf(stderr, "[opal] order: max + histogram %.3f ms (%d threads), prefix %.3f, placement %.3f\n", to1 - to0, T, to2 - to1, now_ms() - to2);
    if ((size_t)m < items) { pl[m] = 0; out.seq[m] = -1; so[m] = byte; }
    *residues = (int64_t)byte;
    *itemsOut = items;
    return true;
}

// Upload chunks and gather pieces of a fill (db_fill): a CHUNK (~2 MB, fewer and larger ones for gigabyte databases) is what one
// copy moves, a PIECE (~128 KB) what one pool task gathers.
static void chunk_geometry(uint64_t total, uint64_t* chunkBytes, uint64_t* pieceBytes) {
    *chunkBytes = opts().chunkBytes > 0 ? (uint64_t)opts().chunkBytes : std::max<uint64_t>(2u << 20, total / 96);
    *pieceBytes = opts().chunkBytes > 0 ? std::max<uint64_t>(1024, *chunkBytes / 4) : std::max<uint64_t>(128u << 10, *chunkBytes / 16);
}

// Where they begin. item_at(b) -> {first item at or behind byte offset b, that item's own offset}; it is asked about the same
// offsets in the same order whatever it answers, so the questions can be recorded first and answered by the device
// (opal_hip_order_cuts) before the answers are replayed.
template <typename F>
static void plan_cuts(UploadGate& g, uint64_t total, size_t items, uint64_t chunkBytes, uint64_t pieceBytes, F&& item_at) {
    g.cutItem.assign(1, 0); g.cutByte.assign(1, 0);
    g.pieceCut.assign(1, 0); g.pieceByte.assign(1, 0);
    g.pieceChunk.clear();
    for (uint64_t cb = 0; cb < total; cb += chunkBytes) {
        const int c = (int)g.cutItem.size() - 1;
        const uint64_t ce = std::min<uint64_t>(cb + chunkBytes, total);
        for (uint64_t pb = cb + pieceBytes; pb < ce; pb += pieceBytes) {
            const std::pair<int, uint64_t> at = item_at(pb);
            if (at.first > g.pieceCut.back() && at.first < (int)items) { g.pieceChunk.push_back(c); g.pieceCut.push_back(at.first); g.pieceByte.push_back(at.second); }
        }
        const std::pair<int, uint64_t> at = ce >= total ? std::make_pair((int)items, total) : item_at(ce);
        if (at.first > g.pieceCut.back()) { g.pieceChunk.push_back(c); g.pieceCut.push_back(at.first); g.pieceByte.push_back(at.second); }
        if (at.first > g.cutItem.back()) { g.cutItem.push_back(at.first); g.cutByte.push_back(at.second); }
        if (at.first >= (int)items) break;
    }
    if (g.cutItem.back() != (int)items) {  // (rounding left a tail: it joins the last chunk)
        if (g.cutItem.size() > 1) { g.cutItem.back() = (int)items; g.cutByte.back() = total; }
        else { g.cutItem.push_back((int)items); g.cutByte.push_back(total); }
        if (g.pieceCut.back() != (int)items) { g.pieceChunk.push_back((int)g.cutItem.size() - 2); g.pieceCut.push_back((int)items); g.pieceByte.push_back(total); }
    } else {
        g.cutByte.back() = total; g.pieceByte.back() = total;  // (the absent last item of an odd count sits AT the total)
    }
}

// The same ordering on the DEVICE (order_kernel.hip): the lengths go up in one copy, four small launches sort them and
// write the item arrays where the DP kernels read them, and the host's copies (the scheduler reads lengths, the gather
// reads caller indices and offsets) come back in three copies. One pass of one host thread over the lengths (copy into pinned
// memory, maximum, count of the empty ones, sum) instead of the histogram + scattered placement: on a cold
// opalSearchDatabase call the ordering is what stands between the call and its first gathered residue.
// Returns 0 done, 1 not applicable (small database, lengths beyond the kernel's table: the host orders), -1 failed.
static int order_sequences_device(OpalxDb* db, int n, const int* lens, bool force = false, const std::function<bool()>* meanwhile = nullptr,
                                  bool wantCuts = false, std::vector<int>* keepLens = nullptr) {
    const int mode = force ? 1 : opts().deviceOrder;
    if (mode == 0 || n <= 0 || (mode < 0 && n < 16384)) return 1;
    const double t0 = now_ms();
    const double slack = db->cached ? 1.15 : 1.0;
    if (!db->hLens.resize((size_t)n, slack)) return -1;
    if (keepLens) keepLens->resize((size_t)n);
    int* h = db->hLens.data();
    int maxLen = 0, nonEmpty = 0;
    int64_t total = 0;
    int* const keep = keepLens ? keepLens->data() : nullptr;  // (cached contexts: the lengths as handed over, for the next call's "same database?")
    auto stage = [&](int i0, int i1, int* mx, int* ne, int64_t* tot) {
        int m = 0, e = 0;
        int64_t t = 0;
        for (int i = i0; i < i1; i++) {
            const int len = lens[i] > 0 ? lens[i] : 0;
            h[i] = len;
            if (keep) keep[i] = lens[i];
            m = std::max(m, len);
            e += len > 0;
            t += len;
        }
        *mx = m; *ne = e; *tot = t;
    };
    if (n >= 400000) {  // (a pool job costs ~50 us: worth it from a few hundred thousand lengths on -- 625 000: 0.51 ms on one thread)
        constexpr int PARTS = 8;
        int mx[PARTS], ne[PARTS];
        int64_t tot[PARTS];
        HostPool::get().parallel_for(PARTS, [&](int t) { stage((int)((long long)n * t / PARTS), (int)((long long)n * (t + 1) / PARTS), &mx[t], &ne[t], &tot[t]); });
        for (int t = 0; t < PARTS; t++) { maxLen = std::max(maxLen, mx[t]); nonEmpty += ne[t]; total += tot[t]; }
    } else {
        stage(0, n, &maxLen, &nonEmpty, &total);
    }
    if (maxLen > opal_hip_order_max_length()) return 1;
    const size_t items = ((size_t)nonEmpty + 1) / 2 * 2;
    const size_t empties = (size_t)n - (size_t)nonEmpty;
    // the byte offsets at which the fill will want to cut its upload chunks and gather pieces (plan_cuts, recorded): the device
    // answers them right behind the sort, so the item offsets themselves never come back to the host
    db->cutTargets.clear();
    if (wantCuts && total > 0) {
        uint64_t chunkBytes, pieceBytes;
        chunk_geometry((uint64_t)total, &chunkBytes, &pieceBytes);
        db->cutTargetsChunk = chunkBytes; db->cutTargetsPiece = pieceBytes;
        UploadGate scratchGate;
        plan_cuts(scratchGate, (uint64_t)total, items, chunkBytes, pieceBytes, [&](uint64_t b) { db->cutTargets.push_back(b); return std::make_pair(0, (uint64_t)0); });
    }
    const size_t T = db->cutTargets.size();
    void* st = db->main();
    if (!st || !db->dLens.ensure((size_t)n, slack) || !db->dOrderScratch.ensure(opal_hip_order_scratch_bytes(n, maxLen), slack) ||
        !db->dSeqOff.ensure(items + 1, slack) || !db->dPairLen.ensure(items + 1, slack) || !db->dPairSeq.ensure(items + 1, slack) ||
        !db->pairLen.resize(items, slack) || !db->pairSeq.resize(items, slack) ||
        (!wantCuts && !db->seqOff.resize(items, slack)) ||
        (T && (!db->dCuts.ensure(3 * T, 1.5) || !db->hCuts.resize(3 * T, 1.5))) ||
        (empties && (!db->dEmpty.ensure(empties, slack) || !db->hEmpty.resize(empties, slack))))
        return -1;
    for (size_t t = 0; t < T; t++) db->hCuts[t] = db->cutTargets[t];
    const double t1 = now_ms();
    bool ok = !opal_hip_memcpy_h2d(db->dLens.p, h, (size_t)n * sizeof(int), st) &&
              (!T || !opal_hip_memcpy_h2d(db->dCuts.p, db->hCuts.data(), T * sizeof(uint64_t), st)) &&
              !opal_hip_order_by_length(db->dLens.p, n, maxLen, nonEmpty, db->dOrderScratch.p, db->dPairSeq.p, db->dPairLen.p, db->dSeqOff.p,
                                        db->dEmpty.p, st) &&
              (!T || !opal_hip_order_cuts(db->dCuts.p, (int)T, db->dSeqOff.p, nonEmpty, (uint64_t)total, db->dCuts.p + T, st));
    if (ok && items)
        ok = !opal_hip_memcpy_d2h(db->pairSeq.data(), db->dPairSeq.p, items * sizeof(int), st) &&
             !opal_hip_memcpy_d2h(db->pairLen.data(), db->dPairLen.p, items * sizeof(int), st) &&
             (wantCuts || !opal_hip_memcpy_d2h(db->seqOff.data(), db->dSeqOff.p, items * sizeof(uint64_t), st));
    if (ok && T) ok = !opal_hip_memcpy_d2h(db->hCuts.data() + T, db->dCuts.p + T, 2 * T * sizeof(uint64_t), st);
    if (ok && empties) ok = !opal_hip_memcpy_d2h(db->hEmpty.data(), db->dEmpty.p, empties * sizeof(int), st);
    // (while the device sorts: whatever else the caller has to do before the gather can begin -- opalSearchDatabase looks
    // through its N result structs for the ones that need computing; if they are not all of them, this ordering is dropped)
    if (ok && meanwhile && !(*meanwhile)()) { opal_hip_stream_sync(st); return -2; }
    if (!ok || opal_hip_stream_sync(st)) return -1;
    db->emptySeqs.assign(db->hEmpty.data(), db->hEmpty.data() + empties);
    db->numSeqs = n;
    db->numPairs = (int)(items / 2);
    db->residues = total;
    db->ordScratch.pos.clear(); db->ordScratch.byte.clear();
    db->deviceOrdered = true;
    db->hostSeqOffValid = !wantCuts;
    if (debug_on()) fprintf(stderr, "[opal] order on the device: lengths staged %.3f ms, upload + sort + item arrays back %.3f ms\n", t1 - t0, now_ms() - t1);
    return 0;
}

static bool order_sequences(OpalxDb* db, int n, const int* lens, bool mayUseDevice, const std::function<bool()>* meanwhile, bool* keptLens) {
    db->deviceOrdered = false;
    db->hostSeqOffValid = true;
    db->cutTargets.clear();
    *keptLens = false;
    if (mayUseDevice) {
        const bool keep = db->cached;
        const int r = order_sequences_device(db, n, lens, false, meanwhile, /*wantCuts=*/true, keep ? &db->lastLens : nullptr);
        *keptLens = keep && r == 0;
        if (r == -2) db->fillAborted = true;
        if (r <= 0) return r == 0;
    }
    if (meanwhile && !(*meanwhile)()) { db->fillAborted = true; return false; }
    db->numSeqs = n;
    const double slack = db->cached ? 1.15 : 1.0;
    OrderScratch& scratch = db->ordScratch;
    size_t items = 0;
    const bool ok = order_into(n, lens, scratch, db->emptySeqs, &db->residues, &items, [&](size_t k, OrderOut& o) {
        if (!db->pairLen.resize(k, slack) || !db->pairSeq.resize(k, slack) || !db->seqOff.resize(k, slack)) return false;
        o = OrderOut{db->pairLen.data(), db->pairSeq.data(), db->seqOff.data()};
        return true;
    });
    db->numPairs = (int)(items / 2);
    return ok;
}

// The HIP "current device" is a per-thread setting, so the check (and the optional OPAL_HIP_DEVICE
// selection) is done once per calling thread. Without OPAL_HIP_DEVICE the thread's current device is
// used as it is (bench.py sets it through torch before calling in).
bool device_ok() {
    static thread_local int checked = 0;  // 0 unknown, 1 ok, -1 none
    if (checked == 0) {
        if (opal_hip_device_count() > 0) {
            checked = 1;
            if (opts().device >= 0 && opal_hip_set_device(opts().device) != 0) checked = -1;
        } else {
            checked = -1;
        }
        if (checked < 0)
            fprintf(stderr, "[opal] no usable HIP device (gfx950 required); there is no CPU fallback\n");
    }
    return checked > 0;
}

void db_release(OpalxDb* db) {
    db->quiesce();  // (searches queued by opalx_search_device_async and never waited for: nothing may still read these buffers)
    db->res.release(); db->dSeqOff.release(); db->dPairLen.release(); db->dPairSeq.release();
    db->pairLen.release(); db->pairSeq.release(); db->seqOff.release();
    db->hLens.release(); db->hEmpty.release(); db->dLens.release(); db->dEmpty.release(); db->dOrderScratch.release();
    db->tokens.release(); db->pairOff.release(); db->hPairOff.release();
    db->timeline.release();
    db->header.release(); db->hHeader.release(); db->ovfItems.release(); db->boundary.release(); db->pairBnd.release(); db->scratch.release();
    db->results.release(); db->quads.release();
    db->ovfCount.p = nullptr; db->matrix.p = nullptr; db->query.p = nullptr;
    db->seqItem.release(); db->revFlags.release(); db->redo.release();
    db->trDirs.release(); db->trOps.release(); db->trOff.release(); db->trItems.release(); db->trOut.release();
    db->raw.release(); db->rawOff.release(); db->hRawOff.release(); db->hRaw.release(); db->hResults.release();
    for (void*& st : db->streams) { opal_hip_stream_destroy(st); st = nullptr; }
    opal_hip_stream_destroy(db->mainStream); db->mainStream = nullptr;
    opal_hip_stream_destroy(db->pipeStream); db->pipeStream = nullptr;
    for (void*& ev : db->pipe.done) { opal_hip_event_destroy(ev); ev = nullptr; }
    db->pipe = OpalxDb::Pipe(); db->hdrDev = nullptr; db->curStream = nullptr;
    opal_hip_stream_destroy(db->copyStream); db->copyStream = nullptr;
    opal_hip_stream_destroy(db->copyStream2); db->copyStream2 = nullptr;
    opal_hip_stream_destroy(db->auxStream); db->auxStream = nullptr;
    opal_hip_stream_destroy(db->swapStream); db->swapStream = nullptr;
    for (void*& ev : db->events) { opal_hip_event_destroy(ev); ev = nullptr; }
    for (void*& ev : db->asyncEv) { opal_hip_event_destroy(ev); ev = nullptr; }
    for (int i = 0; i < OPALX_ASYNC_DEPTH; i++) { opal_hip_event_destroy(db->hdrEv[i]); db->hdrEv[i] = nullptr; db->hdrEvSet[i] = false; }
    db->asyncHead = db->asyncTail = 0; db->asyncWaited = 0;
    for (void*& ev : db->gate.events) { opal_hip_event_destroy(ev); ev = nullptr; }
    db->gate.events.clear();
}

bool db_fill(OpalxDb* db, int n, const int* lens, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
             const unsigned char* flatDevice, const std::function<uint64_t(int)>& flatOff, bool async, const std::function<bool()>* meanwhile) {
    db->device = opal_hip_get_device();
    db->fillAborted = false;
    db->gate.active = false;
    db->gate.speculated = false;
    db->gate.reuse = false;
    db->gate.zeroCopy = false;
    db->gate.kernelFilled = false;
    db->gate.kernelFilledFrom = INT_MAX;
    db->tokensValid = false;
    const double tf0 = now_ms();
    // The SAME database as last time? (A caller that sends many queries against one database through the plain
    // opalSearchDatabase.) A cached context still holds the previous call's lengths, its item order, the gathered residues
    // (pinned, item order) and their device copy. Equal lengths (one memcmp) mean the same order and metadata; the residues
    // are then COMPARED while they are gathered, and only the chunks that differ are uploaded again -- reading the
    // caller's bytes is unavoidable, sending 35 MB over PCIe (0.8-1.0 ms on C2) is not.
    const bool reuse = db->cached && kind == RAW_POINTERS && db->lastValid && db->lastLens.size() == (size_t)n && n > 0 &&
                       !opts().noReuse && memcmp(db->lastLens.data(), lens, (size_t)n * sizeof(int)) == 0;
    db->lastValid = false;
    if (reuse && meanwhile && !(*meanwhile)()) { db->fillAborted = true; return false; }
    if (!reuse) {
        db->seqItemValid = false;
        db->planCache.clear();
        bool keptLens = false;
        if (!order_sequences(db, n, lens, kind == RAW_POINTERS, meanwhile, &keptLens)) return false;
        if (db->cached && kind == RAW_POINTERS && !keptLens) db->lastLens.assign(lens, lens + n);
    }
    const double tf1 = now_ms();
    const size_t items = db->pairLen.size();
    const double slack = db->cached ? 1.15 : 1.0;
    void* st = db->main();
    const uint64_t total = (uint64_t)db->residues;
    bool ok = st && db->res.ensure(total + 16, slack) && db->dSeqOff.ensure(items, slack) && db->dPairLen.ensure(items, slack) &&
              db->dPairSeq.ensure(items, slack) && db->ovfItems.ensure(items, slack);
    if (!ok) return false;
    if (!items) return true;
    // (pinned sources: the copies need no wait on the host side)
    if (!reuse && !db->deviceOrdered)
        ok = !opal_hip_memcpy_h2d(db->dSeqOff.p, db->seqOff.data(), items * sizeof(uint64_t), st) &&
             !opal_hip_memcpy_h2d(db->dPairLen.p, db->pairLen.data(), items * sizeof(int), st) &&
             !opal_hip_memcpy_h2d(db->dPairSeq.p, db->pairSeq.data(), items * sizeof(int), st);
    if (!ok) return false;
    if (kind == RAW_DEVICE_FLAT) {
        // one flat device buffer in caller order -> item order, on the device
        if (!db->hRawOff.resize(items, slack) || !db->rawOff.ensure(items, slack)) return false;
        for (size_t i = 0; i < items; i++) db->hRawOff[i] = db->pairSeq[i] >= 0 ? flatOff(db->pairSeq[i]) : 0;
        ok = !opal_hip_memcpy_h2d(db->rawOff.p, db->hRawOff.data(), items * sizeof(uint64_t), st) &&
             !opal_hip_reorder_residues(flatDevice, db->rawOff.p, db->dSeqOff.p, db->dPairLen.p, (int)items, db->res.p, st);
        if (opal_hip_stream_sync(st)) ok = false;  // the caller's buffer is only read during the call
        if (ok && !db->cached) { db->rawOff.release(); db->hRawOff.release(); }
        return ok;
    }
    // Host residues: gathered in ITEM order into pinned memory. A CHUNK (~2 MB, fewer and larger ones for gigabyte
    // databases) is what one copy moves; it is gathered by many threads at a time -- the pool's tasks are PIECES of ~128 KB,
    // handed out in item order, and the thread that completes the last piece of a chunk enqueues its copy on the copy
    // stream -- so the chunks reach the device in the order the kernels work through them (longest sequences first), the
    // first one a few tens of microseconds after the gather began.
    db->hRaw.deviceCached = true;
    if (!db->hRaw.ensure(total + 16, slack) || !db->copy()) return false;
    UploadGate& g = db->gate;
    uint64_t chunkBytes, pieceBytes;
    chunk_geometry(total, &chunkBytes, &pieceBytes);
    const double tf2 = now_ms();
    // (the same database as last time: the same cuts -- unless the chunk size was changed in between)
    const bool keepCuts = reuse && g.cutsTotal == total && g.cutsItems == items && g.cutsChunkBytes == chunkBytes && g.cutsPieceBytes == pieceBytes &&
                          g.cutItem.size() >= 2;
    if (!keepCuts) {
        auto host_seq_off = [&]() -> bool {  // the item offsets on the host (a device ordering leaves them on the device)
            if (db->hostSeqOffValid) return true;
            if (!db->seqOff.resize(items, slack) || opal_hip_memcpy_d2h(db->seqOff.data(), db->dSeqOff.p, items * sizeof(uint64_t), st) || opal_hip_stream_sync(st)) return false;
            db->hostSeqOffValid = true;
            return true;
        };
        if (db->deviceOrdered && !db->cutTargets.empty() && db->cutTargetsChunk == chunkBytes && db->cutTargetsPiece == pieceBytes) {
            // the device has answered plan_cuts' questions already (order_sequences_device): replayed in the order they were asked
            const size_t T = db->cutTargets.size();
            size_t next = 0;
            plan_cuts(g, total, items, chunkBytes, pieceBytes, [&](uint64_t) {
                const size_t t = std::min(next++, T - 1);
                return std::make_pair((int)db->hCuts[T + 2 * t], (uint64_t)db->hCuts[T + 2 * t + 1]);
            });
        } else {
            if (!host_seq_off()) return false;
            const uint64_t* so = db->seqOff.data();
            plan_cuts(g, total, items, chunkBytes, pieceBytes, [&](uint64_t b) {
                const size_t it = (size_t)(std::lower_bound(so, so + items, b) - so);
                return std::make_pair((int)it, it < items ? so[it] : total);
            });
        }
        // (strictly increasing cuts, every piece inside one chunk, offsets in step; otherwise: the plain walk over the items)
        bool cutsOk = g.cutItem.size() >= 2 && g.cutItem.back() == (int)items && g.pieceCut.size() >= 2 && g.pieceCut.back() == (int)items &&
                      g.pieceChunk.size() + 1 == g.pieceCut.size() && g.cutByte.size() == g.cutItem.size() && g.pieceByte.size() == g.pieceCut.size() &&
                      g.cutByte.back() == total && g.pieceByte.back() == total;
        for (size_t c = 0; cutsOk && c + 1 < g.cutItem.size(); c++) cutsOk = g.cutItem[c] < g.cutItem[c + 1] && g.cutByte[c] <= g.cutByte[c + 1];
        for (size_t q = 0; cutsOk && q + 1 < g.pieceCut.size(); q++) {
            const int c = g.pieceChunk[q];
            cutsOk = g.pieceCut[q] < g.pieceCut[q + 1] && g.pieceByte[q] <= g.pieceByte[q + 1] && c >= 0 && c + 1 < (int)g.cutItem.size() &&
                     g.cutItem[(size_t)c] <= g.pieceCut[q] && g.pieceCut[q + 1] <= g.cutItem[(size_t)c + 1];
        }
        if (!cutsOk) {
            if (debug_on()) fprintf(stderr, "[opal] fill: chunk cuts rejected, walking the items\n");
            if (!host_seq_off()) return false;
            g.cutItem.assign(1, 0); g.cutByte.assign(1, 0);
            g.pieceCut.assign(1, 0); g.pieceByte.assign(1, 0);
            g.pieceChunk.clear();
            for (size_t i = 0; i < items; i++) {
                if (db->seqOff[i] - db->seqOff[(size_t)g.cutItem.back()] >= chunkBytes) {
                    g.cutItem.push_back((int)i); g.cutByte.push_back(db->seqOff[i]);
                    if (g.pieceCut.back() != (int)i) { g.pieceChunk.push_back((int)g.cutItem.size() - 2); g.pieceCut.push_back((int)i); g.pieceByte.push_back(db->seqOff[i]); }
                } else if (db->seqOff[i] - db->seqOff[(size_t)g.pieceCut.back()] >= pieceBytes) {
                    g.pieceChunk.push_back((int)g.cutItem.size() - 1); g.pieceCut.push_back((int)i); g.pieceByte.push_back(db->seqOff[i]);
                }
            }
            g.cutItem.push_back((int)items); g.cutByte.push_back(total);
            if (g.pieceCut.back() != (int)items) { g.pieceChunk.push_back((int)g.cutItem.size() - 2); g.pieceCut.push_back((int)items); g.pieceByte.push_back(total); }
        }
        g.cutsTotal = total; g.cutsItems = items; g.cutsChunkBytes = chunkBytes; g.cutsPieceBytes = pieceBytes;
    }
    const int nchunks = (int)g.cutItem.size() - 1;
    const int npieces = (int)g.pieceCut.size() - 1;
    const double tf3 = now_ms();
    if (debug_on() && !g.timed) {  // (debug: timed chunk events, to print when each chunk's copy completed)
        for (void*& ev : g.events) opal_hip_event_destroy(ev);
        g.events.clear();
        g.timed = true;
    }
    while ((int)g.events.size() < nchunks) {
        void* ev = g.timed ? opal_hip_event_create() : opal_hip_event_create_untimed();
        if (!ev) return false;
        g.events.push_back(ev);
    }
    if (g.stateCap < (size_t)nchunks) {
        g.state.reset(new std::atomic<int>[(size_t)nchunks]);
        g.left.reset(new std::atomic<int>[(size_t)nchunks]);
        g.dirty.reset(new std::atomic<int>[(size_t)nchunks]);
        g.copied.reset(new std::atomic<int>[(size_t)nchunks]);
        g.stateCap = (size_t)nchunks;
    }
    for (int c = 0; c < nchunks; c++) { g.state[(size_t)c].store(0, std::memory_order_relaxed); g.left[(size_t)c].store(0, std::memory_order_relaxed); }
    for (int p = 0; p < npieces; p++) g.left[(size_t)g.pieceChunk[(size_t)p]].fetch_add(1, std::memory_order_relaxed);
    const int dev = db->device;
    for (int c = 0; c < nchunks; c++) { g.dirty[(size_t)c].store(reuse ? 0 : 1, std::memory_order_relaxed); g.copied[(size_t)c].store(0, std::memory_order_relaxed); }
    db->gatherFn = [db, getSeq, dev, reuse](int p) {
        static thread_local int myDevice = -1;  // pool threads start on device 0: the copy below names a stream of `dev`
        if (myDevice != dev) { opal_hip_set_device(dev); myDevice = dev; }
        UploadGate& gg = db->gate;
        unsigned char* h = db->hRaw.p;
        const int* len = db->pairLen.data(); const int* seq = db->pairSeq.data();
        const int c = gg.pieceChunk[(size_t)p];
        bool changed = false;
        uint64_t off = gg.pieceByte[(size_t)p];  // items lie back to back in item order: the offset runs along
        for (size_t i = (size_t)gg.pieceCut[(size_t)p], i1 = (size_t)gg.pieceCut[(size_t)p + 1]; i < i1; i++) {
            if (seq[i] < 0 || len[i] <= 0) continue;
            const unsigned char* src = getSeq(seq[i]);
            unsigned char* dst = h + off;
            off += (uint64_t)len[i];
            if (reuse && memcmp(dst, src, (size_t)len[i]) == 0) continue;  // what the device already holds
            memcpy(dst, src, (size_t)len[i]);
            changed = true;
        }
        if (reuse && changed) gg.dirty[(size_t)c].store(1, std::memory_order_relaxed);
        if (gg.left[(size_t)c].fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        // the last piece of chunk c: its copy, and the event the launches wait for (an unchanged chunk of a database the
        // device already holds needs neither: its event of the earlier call has long completed)
        bool good = true;
        if (gg.dirty[(size_t)c].load(std::memory_order_relaxed) && !gg.zeroCopy) {
            gg.copied[(size_t)c].store(1, std::memory_order_relaxed);
            const uint64_t b0 = gg.cutByte[(size_t)c], b1 = gg.cutByte[(size_t)c + 1];
            const bool two = !opts().oneCopyStream;
            void* cs = (two && (c & 1)) ? db->copyStream2 : db->copyStream;
            if (b1 > b0 && opal_hip_memcpy_h2d(db->res.p + b0, h + b0, b1 - b0, cs)) good = false;
            if (good && opal_hip_event_record(gg.events[(size_t)c], cs)) good = false;
            gg.uploaded.fetch_add(1, std::memory_order_relaxed);
        }
        gg.state[(size_t)c].store(good ? 1 : -1, std::memory_order_release);
        if (debug_on() && (c == 0 || c + 1 == gg.chunks() || c == gg.chunks() / 2))
            fprintf(stderr, "[opal] fill: chunk %d of %d %s %.3f ms after the gather began\n", c, gg.chunks(),
                    gg.dirty[(size_t)c].load() ? "enqueued" : "found unchanged", now_ms() - db->fillT0);
    };
    g.uploaded.store(0);
    g.reuse = reuse;
    // A database seen for the first time is searched ZERO-COPY: the sliced launches read the residues from this pinned buffer
    // over PCIe as soon as they have been gathered (35 MB in 1.5 ms of DP is 23 GB/s; tokens are fetched a batch of columns
    // ahead), instead of waiting for the copies -- which still run, beside the search, for whatever reads the device copy
    // afterwards (start cells, traceback, the next call on the same database).
    g.zeroCopy = async && !reuse && !opts().noZeroCopy;
    g.active = true;
    HostPool& pool = HostPool::get();
    if (g.timed) opal_hip_event_record(db->event(7), db->copyStream);  // (debug reference: the copy stream when the gather begins)
    db->fillT0 = now_ms();
    if (debug_on()) fprintf(stderr, "[opal] fill: %s %.3f ms, metadata + chunk plan %.3f ms (buffers %.3f, cuts %.3f), %d chunks of ~%.1f MB in %d pieces\n", reuse ? "same lengths as the last call (order kept):" : "order", tf1 - tf0, db->fillT0 - tf1, tf2 - tf1, tf3 - tf2, nchunks, (double)chunkBytes / 1048576.0, npieces);
    if (async) {
        // A database found again is searched SPECULATIVELY: the launches do not wait for the comparison (no gate), the
        // caller repeats the search if db_fill_finish reports that a chunk had changed after all (fill_changed()).
        // (Not right after a call that was caught out: a caller that edits its database in place pays the repeated search once.)
        if (reuse && db->speculate && !opts().noSpeculation) g.active = false;
        g.speculated = !g.active;
        pool.begin(npieces, db->gatherFn);
        db->gatherPending = true;
        return true;
    }
    pool.parallel_for(npieces, db->gatherFn);
    ok = gate_items(db, st, 0, (int)items);  // everything on `main` from here on comes after the last chunk
    g.active = false;
    db->gatherFn = nullptr;
    db->lastValid = ok && db->cached;
    if (ok && !db->cached) {
        if (opal_hip_stream_sync(st)) ok = false;
        db->hRaw.release();
    }
    return ok;
}

bool db_fill_finish(OpalxDb* db) {
    bool ok = true;
    if (db->gatherPending) {
        HostPool::get().end();
        db->gatherPending = false;
        for (int c = 0; c < db->gate.chunks(); c++)
            if (db->gate.state[(size_t)c].load(std::memory_order_acquire) < 0) ok = false;
        // (a zero-copy fill leaves a complete device copy only if its kernels wrote every item, or every chunk was copied on demand)
        bool deviceComplete = !db->gate.zeroCopy || db->gate.kernelFilled;
        if (!deviceComplete) {  // (a chunk the kernels wrote only from some item on must have been copied as well)
            deviceComplete = true;
            for (int c = 0; c < db->gate.chunks(); c++)
                deviceComplete &= db->gate.copied[(size_t)c].load() != 0 || db->gate.cutItem[(size_t)c] >= db->gate.kernelFilledFrom;
        }
        db->lastValid = ok && db->cached && deviceComplete;
        db->speculate = !(db->gate.reuse && db->gate.uploaded.load() > 0);  // (the database was edited under the same lengths: the next call waits for its comparison)
        if (debug_on()) fprintf(stderr, "[opal] fill: %d of %d chunks uploaded%s\n", db->gate.uploaded.load(), db->gate.chunks(), db->gate.reuse ? " (the others unchanged since the last call)" : "");
        if (db->gate.reuse) count(CNT_REUSED_FILLS);
        if (ok && db->gate.timed && debug_on() && db->gate.chunks() > 0 && !db->gate.reuse && !db->gate.zeroCopy) {  // (debug only: waits for the copies)
            opal_hip_stream_sync(db->copyStream); opal_hip_stream_sync(db->copyStream2);
            const int nc = db->gate.chunks();
            for (int c : {0, nc / 4, nc / 2, 3 * nc / 4, nc - 1})
                fprintf(stderr, "[opal] fill: chunk %d of %d on the device %.3f ms after the gather began\n", c, nc,
                        opal_hip_event_elapsed_ms(db->events[7], db->gate.events[(size_t)c]));
        }
        db->gatherFn = nullptr;
    }
    db->gate.active = false;
    return ok;
}

bool fill_changed(const OpalxDb* db) { return db->gate.reuse && db->gate.speculated && db->gate.uploaded.load() > 0; }

bool wait_uploads(OpalxDb* db, void* stream) {
    for (int c = 0; c < (int)db->gate.cutItem.size() - 1 && c < (int)db->gate.events.size(); c++)
        if (opal_hip_stream_wait_event(stream, db->gate.events[(size_t)c])) return false;
    return true;
}

bool gate_items(OpalxDb* db, void* stream, int firstItem, int count, bool needDevice) {
    UploadGate& g = db->gate;
    if (!g.active || count <= 0) return true;
    const int last = firstItem + count - 1;
    const int c0 = (int)(std::upper_bound(g.cutItem.begin(), g.cutItem.end(), firstItem) - g.cutItem.begin()) - 1;
    const int c1 = (int)(std::upper_bound(g.cutItem.begin(), g.cutItem.end(), last) - g.cutItem.begin()) - 1;
    for (int c = std::max(c0, 0); c <= c1 && c < g.chunks(); c++) {
        int s, spins = 0;
        while ((s = g.state[(size_t)c].load(std::memory_order_acquire)) == 0) {
            if (++spins < 2000) __builtin_ia32_pause();
            else std::this_thread::yield();
        }
        if (s < 0) return false;
        if (!needDevice) continue;
        if (g.zeroCopy && g.copied[(size_t)c].exchange(1, std::memory_order_acq_rel) == 0) {
            // a zero-copy fill copies a chunk only when something asks for its device copy: now
            const uint64_t b0 = g.cutByte[(size_t)c], b1 = g.cutByte[(size_t)c + 1];
            void* cs = (c & 1) ? db->copyStream2 : db->copyStream;
            if ((b1 > b0 && opal_hip_memcpy_h2d(db->res.p + b0, db->hRaw.p + b0, b1 - b0, cs)) || opal_hip_event_record(g.events[(size_t)c], cs)) return false;
            g.uploaded.fetch_add(1, std::memory_order_relaxed);
        }
        if (opal_hip_stream_wait_event(stream, g.events[(size_t)c])) return false;
    }
    return true;
}

int ensure_pair_tokens(OpalxDb* db) {
    if (db->tokensValid) return 0;
    const size_t P = (size_t)db->numPairs;
    const double slack = db->cached ? 1.15 : 1.0;
    if (!db->hPairOff.resize(P, slack)) return 1;
    uint64_t total = 0;
    for (size_t p = 0; p < P; p++) { db->hPairOff[p] = total; total += (uint64_t)db->pairLen[2 * p]; }  // A is the longer one
    db->numTokens = (size_t)total + 8;
    void* st = db->main();
    if (!db->tokens.ensure(db->numTokens, slack) || !db->pairOff.ensure(std::max<size_t>(P, 1), slack)) return 1;
    if (!gate_items(db, st, 0, (int)(2 * P))) return 1;
    if (P && (opal_hip_memcpy_h2d(db->pairOff.p, db->hPairOff.data(), P * sizeof(uint64_t), st) ||
              opal_hip_pack_tokens(db->res.p, db->dSeqOff.p, db->pairOff.p, db->dPairLen.p, (int)P, db->tokens.p, db->numTokens, st)))
        return 1;
    db->tokensValid = true;
    return 0;
}

}  // namespace opal_host

using namespace opal_host;

// ------------------------------------------------------------------------------------ opalx_* API
static OpalxDb* create_from(int n, const int* lens, RawKind kind, const std::function<const unsigned char*(int)>& getSeq,
                            const unsigned char* flatDevice, const std::function<uint64_t(int)>& flatOff) {
    OpalxDb* db = new (std::nothrow) OpalxDb;
    if (!db) return nullptr;
    if (!db_fill(db, n, lens, kind, getSeq, flatDevice, flatOff)) { opalx_db_destroy(db); return nullptr; }
    return db;
}

extern "C" OpalxDb* opalx_db_create(unsigned char* const db[], int db_length, const int db_seq_lengths[]) {
    if (!device_ok()) return nullptr;
    return create_from(db_length, db_seq_lengths, RAW_POINTERS, [db](int i) { return (const unsigned char*)db[i]; }, nullptr, nullptr);
}

static std::vector<int> flat_lengths(const int64_t* offsets, int db_length) {
    std::vector<int> lens((size_t)std::max(db_length, 0));
    for (int i = 0; i < db_length; i++) lens[(size_t)i] = (int)(offsets[i + 1] - offsets[i]);
    return lens;
}

extern "C" OpalxDb* opalx_db_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length) {
    if (!device_ok()) return nullptr;
    const std::vector<int> lens = flat_lengths(offsets, db_length);
    return create_from(db_length, lens.data(), RAW_POINTERS, [residues, offsets](int i) { return residues + offsets[i]; }, nullptr, nullptr);
}

extern "C" OpalxDb* opalx_db_create_flat_device(const unsigned char* d_residues, const int64_t* offsets, int db_length) {
    if (!device_ok()) return nullptr;
    const std::vector<int> lens = flat_lengths(offsets, db_length);
    return create_from(db_length, lens.data(), RAW_DEVICE_FLAT, nullptr, d_residues, [offsets](int i) { return (uint64_t)offsets[i]; });
}

extern "C" void opalx_db_destroy(OpalxDb* db) {
    if (!db) return;
    db->quiesce();
    db_release(db);
    delete db;
}

// Test hook (no device needed): the streaming plan of a search, see include/opal_hip.h.
extern "C" int opalx_debug_plan_stream(int query_length, int alphabet_length, int mode, int want_end, long long residues, int longest,
                                       int num_seqs, int* out) {
    const opal_host::StreamPlan sp = opal_host::plan_stream(query_length, alphabet_length, mode, want_end, residues, longest, num_seqs);
    out[0] = sp.G ? sp.width : 0; out[1] = sp.G; out[2] = sp.rows; out[3] = sp.G ? sp.passes : 0;
    return 0;
}

// Test hook (no device needed): the score bound the 16-bit rung is planned with, see include/opal_hip.h.
extern "C" long long opalx_debug_score_bound(const unsigned char* query, int query_length, const int* score_matrix, int alphabet_length) {
    return opal_host::query_score_bound(query, query_length, score_matrix, alphabet_length);
}

// Test hook (no device needed): where the longest sequences of a search would go, see include/opal_hip.h.
extern "C" int opalx_debug_plan_long_targets(int query_length, int alphabet_length, int mode, const int* lengths_desc, int count,
                                             long long residues, int need32, int* out) {
    const opal_host::LenView lenOf{lengths_desc, 1, nullptr};
    int rowsP = 0, wavesP = 0, rowsS = 0;
    const int pairN = opal_host::plan_pair(query_length, alphabet_length, mode, lenOf, 0, count, residues, need32, true, &rowsP, &wavesP);
    const int swapN = pairN > 0 ? 0 : opal_host::plan_swap(query_length, alphabet_length, mode, lenOf, 0, count, residues, &rowsS, true);
    out[0] = swapN; out[1] = swapN ? rowsS : 0; out[2] = pairN; out[3] = pairN ? rowsP : 0; out[4] = pairN ? wavesP : 0;
    return 0;
}

// Test hook (no device needed): the ordering of opalx_db_create* / opalSearchDatabase on plain host arrays. Fills
// seq / len / off (capacity entries each) in item order; returns the number of items (sequences of length > 0, rounded up
// to even), or -1 if capacity is too small.
extern "C" int opalx_debug_order(const int* lens, int n, int* seq, int* len, unsigned long long* off, int capacity) {
    OrderScratch scratch;
    std::vector<int> empties;
    int64_t residues = 0;
    size_t items = 0;
    bool fits = true;
    order_into(n, lens, scratch, empties, &residues, &items, [&](size_t k, OrderOut& o) {
        if (k > (size_t)capacity) { fits = false; return false; }
        o = OrderOut{len, seq, reinterpret_cast<uint64_t*>(off)};
        return true;
    });
    return fits ? (int)items : -1;
}

// Test hook (needs the device): the same ordering by order_kernel.hip, same outputs; `empties` (capacity entries) receives the
// caller indices of the zero-length sequences, *num_empties their count. -1 if it does not apply or fails.
extern "C" int opalx_debug_order_device(const int* lens, int n, int* seq, int* len, unsigned long long* off, int capacity, int* empties,
                                        int* num_empties) {
    if (!device_ok()) return -1;
    OpalxDb* db = new (std::nothrow) OpalxDb;
    if (!db) return -1;
    int items = -1;
    if (order_sequences_device(db, n, lens, /*force=*/true) == 0 && db->pairLen.size() <= (size_t)capacity) {  // (without cuts: the offsets come back)
        items = (int)db->pairLen.size();
        for (int i = 0; i < items; i++) { seq[i] = db->pairSeq[(size_t)i]; len[i] = db->pairLen[(size_t)i]; off[i] = db->seqOff[(size_t)i]; }
        *num_empties = (int)db->emptySeqs.size();
        for (size_t i = 0; i < db->emptySeqs.size() && i < (size_t)capacity; i++) empties[i] = db->emptySeqs[i];
    }
    opalx_db_destroy(db);
    return items;
}

// Test hook (no device needed): a stress of the host thread pool -- jobs of 1..300 tasks, synchronous and begin/end in
// turn, every task adds its index+1 to a sum; from `callers` threads at once when callers > 1 (submissions queue up).
// Returns 0 if every task of every job ran exactly once.
extern "C" int opalx_debug_pool_selftest(int rounds, int callers) {
    std::atomic<long long> sum{0};
    long long want = 0;
    for (int rep = 0; rep < rounds; rep++) { const long long k = 1 + rep % 300; want += k * (k + 1) / 2; }
    auto work = [&] {
        HostPool& p = HostPool::get();
        for (int rep = 0; rep < rounds; rep++) {
            const int k = 1 + rep % 300;
            std::vector<std::atomic<int>> hit((size_t)k);
            for (auto& h : hit) h.store(0);
            std::function<void(int)> fn = [&](int t) { hit[(size_t)t].fetch_add(1); sum.fetch_add(t + 1); };
            if (rep % 3 == 0) { p.begin(k, fn); p.end(); } else p.parallel_for(k, fn);
            for (auto& h : hit) if (h.load() != 1) sum.fetch_add(1LL << 40);  // a task ran twice or not at all
        }
    };
    if (callers <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int c = 0; c < callers; c++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    return sum.load() == want * std::max(callers, 1) ? 0 : 1;
}

extern "C" long long opalx_stream_launch_count(void) { return g_counters[CNT_STREAM_SW].load() + g_counters[CNT_STREAM_GLOBAL].load(); }

extern "C" int opalx_set_option(const char* name, const char* value) { return set_option(name, value); }

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

extern "C" void opalx_db_clock_probe(OpalxDb* db, int enable) { if (db) { db->clockProbe = enable != 0; db->timelineProbe = enable == 2; } }
extern "C" int opalx_db_last_timeline(OpalxDb* db, unsigned long long* out, int capacity) {
    if (!db || !db->timeline.p || capacity < 2) return 0;
    const int blocks = std::min(capacity / 2, 65536);
    void* st = db->main();
    if (opal_hip_memcpy_d2h(out, db->timeline.p, (size_t)blocks * 2 * sizeof(unsigned long long), st) || opal_hip_stream_sync(st)) return 0;
    int n = blocks;
    while (n > 0 && out[2 * (size_t)n - 1] == 0) n--;
    return n;
}
extern "C" int opalx_db_last_timeline_cycles(OpalxDb* db, unsigned long long* out, int capacity) {
    if (!db || !db->timeline.p || db->timeline.cap < 4 * 65536 || capacity < 2) return 0;
    const int blocks = std::min(capacity / 2, 65536);
    void* st = db->main();
    if (opal_hip_memcpy_d2h(out, db->timeline.p + 2 * 65536, (size_t)blocks * 2 * sizeof(unsigned long long), st) || opal_hip_stream_sync(st)) return 0;
    int n = blocks;
    while (n > 0 && out[2 * (size_t)n - 1] == 0) n--;
    return n;
}
extern "C" double opalx_db_last_clock_ghz(const OpalxDb* dbc) {
    OpalxDb* db = const_cast<OpalxDb*>(dbc);
    if (!db) return 0.0;
    if (db->clockProbe && db->hdrDev) {  // (the counters of the last search enqueued on the handle: waits for it)
        unsigned long long v[2] = {0, 0};
        void* st = db->main();
        if (st && !opal_hip_memcpy_d2h(v, db->hdrDev + 16, sizeof v, st) && !opal_hip_stream_sync(st) && v[1])
            db->lastClockGhz = (double)v[0] / (double)v[1] * opal_hip_wall_clock_hz() * 1e-9;
    }
    return db->lastClockGhz;
}
extern "C" int opalx_db_num_sequences(const OpalxDb* db) { return db ? db->numSeqs : 0; }
extern "C" int64_t opalx_db_num_residues(const OpalxDb* db) { return db ? db->residues : 0; }
