// opal_api.cpp -- the opal.h C API of the MI355X-native aligner and the opalx_search_* entries of include/opal_hip.h.
//
//   opalSearchDatabase        reference src/opal.cpp:1435-1519
//   opalSearchDatabaseCharSW  reference src/opal.cpp:1522-1546
//   result helpers            reference src/opal.cpp:1549-1564
//
// There is NO CPU fallback: without a usable gfx950 device every search returns OPAL_ERR_NO_SIMD_SUPPORT and says why
// on stderr.
//
// What one opalSearchDatabase call costs on an MI355X is host work, not DP (the kernels need under two
// milliseconds for 10^10 cells), so the boundary is built for that:
//   * device buffers, pinned staging memory, streams and events live in cached CONTEXTS that calls
//     borrow and give back (db_pack.cpp; no hipMalloc/hipFree, stream or event creation in the steady state);
//   * the N separate host allocations are gathered into pinned memory by a persistent thread pool, in the order the
//     kernels work through them (longest first), each chunk going to the device as soon as it is complete -- and the
//     search of a slice of the database is launched as soon as ITS chunks are on their way (run_shard, gate_items);
//   * results come back through pinned memory and are scattered into the N result structs by the pool;
//   * with OPAL_HIP_DEVICES set, the call shards the database over several GPUs (multi_gpu.cpp), one host thread per GPU.
#include "opal_internal.hpp"

using namespace opal_host;

namespace {

constexpr int RUN_SHARD_DROPPED = -1000;  // run_shard: the `meanwhile` hook ended the fill (not an error of the device)

/*
 * One shard of an opalSearchDatabase call on the calling thread's current device: order, gather + upload, search,
 * download. `ids` (may be null = identity) maps shard position -> caller index; results land in
 * score/endQ/endT at SHARD positions (INT_MIN = not written by the search).
 * The gather of the residues (pool threads, item order = longest first, chunk by chunk onto the copy stream) runs BESIDE
 * the search: search_core launches every slice of the database as soon as its chunks have been enqueued (gate_items).
 */
static int run_shard(int device, const unsigned char* query, int Q, unsigned char* const* db, const int* dbLens, const int* ids,
                     int n, int Go, int Ge, const int* M, int A, int searchType, int mode, int* score, int* endQ, int* endT,
                     double* tms, int* startQ = nullptr, int* startT = nullptr, TraceOut* trace = nullptr,
                     const std::function<bool()>* meanwhile = nullptr) {
    BorrowedContext ctx(device);
    OpalxDb* pdb = ctx.db;
    if (!pdb) return OPAL_ERR_NO_SIMD_SUPPORT;
    const double t0 = now_ms();
    std::vector<int> lensBuf;
    const int* lens = dbLens;
    if (ids) {
        lensBuf.resize((size_t)n);
        for (int i = 0; i < n; i++) lensBuf[(size_t)i] = dbLens[ids[i]];
        lens = lensBuf.data();
    }
    const double t1 = now_ms();
    struct FillGuard {  // the pool job of the fill must be over before anything else uses the pool or the context goes back
        OpalxDb* db;
        ~FillGuard() { db_fill_finish(db); }
    } guard{pdb};
    const bool async = !opts().syncFill;
    if (!db_fill(pdb, n, lens, RAW_POINTERS, [db, ids](int i) { return (const unsigned char*)db[ids ? ids[i] : i]; }, nullptr, nullptr, async, meanwhile))
        return pdb->fillAborted ? RUN_SHARD_DROPPED : OPAL_ERR_NO_SIMD_SUPPORT;
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
    const double t2b = now_ms();
    const bool zeroCopied = pdb->gate.zeroCopy;
    if (!db_fill_finish(pdb)) return OPAL_ERR_NO_SIMD_SUPPORT;
    // (a zero-copy search did not wait for the chunk copies: whatever follows on `main` -- start cells, traceback -- reads the
    // device copy of the residues)
    (void)zeroCopied;  // (its kernels wrote the device copy themselves, on streams joined into `main`; chunks copied on demand were waited for)
    if (fill_changed(pdb)) {
        // the database looked like the previous call's (same lengths) and was searched before its residues had been
        // compared; some differ: they are on the device by now (after the copy streams), the search is done again
        pdb->quiesce();
        // (width 16: the speculative search may have built its pair tokens from residues the copy streams were still replacing)
        pdb->tokensValid = false;
        if (debug_on()) fprintf(stderr, "[opal] shard: the database had changed under the same lengths: searching again\n");
        if (opal_hip_memset32(dS, INT_MIN, nn, st) || (wantEnd && opal_hip_memset32(dQ, -1, 2 * nn, st))) return OPAL_ERR_NO_SIMD_SUPPORT;
        rc = search_core(pdb, query, Q, Go, Ge, M, A, searchType, mode, dS, wantEnd ? dQ : nullptr, wantEnd ? dT : nullptr, nullptr,
                         /*finalSync=*/false);
    }
    const double t3 = now_ms();
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) pdb->lastValid = false;  // (the device copy may be incomplete)
    if (debug_on()) fprintf(stderr, "[opal] shard: launches issued %.3f ms after the fill began, pool done at %.3f ms\n", t2b - t1, t3 - t1);
    if (rc == OPAL_ERR_NO_SIMD_SUPPORT || rc == OPAL_ERR_INVALID_MODE) return rc;
    int* h = pdb->hResults.p;
    // start cells for the alignment search, on the device while the end cells are still there (n4). Any failure
    // just leaves them unknown (INT_MIN): the host traceback then finds them with its own full pass.
    bool haveStarts = false;
    std::vector<int> redo;
    int nredo = 0;
    if (startQ && startT && rc == 0 && wantEnd && n > 0 && !opts().hostStarts) {
        redo.resize((size_t)n);
        haveStarts = search_starts(pdb, query, Q, Go, Ge, M, A, mode, dS, dQ, dT, dS + 3 * nn, dS + 4 * nn, redo.data(), &nredo) == 0;
    }
    if (opal_hip_memcpy_d2h(h, dS, (haveStarts ? 5 : wantEnd ? 3 : 1) * nn * sizeof(int), st) || opal_hip_stream_sync(st))
        return OPAL_ERR_NO_SIMD_SUPPORT;
    if (debug_on()) fprintf(stderr, "[opal] shard: results on the host %.3f ms after the fill began\n", now_ms() - t1);
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

extern "C" int opalx_search_device_async(OpalxDb* db, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                         const int* score_matrix, int alphabet_length, int search_type, int mode,
                                         int* d_score, int* d_end_query, int* d_end_target) {
    if (!db) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (search_type != OPAL_SEARCH_SCORE && (!d_end_query || !d_end_target)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (db->asyncHead - db->asyncTail >= OPALX_ASYNC_DEPTH - 1) return OPAL_ERR_NO_SIMD_SUPPORT;  // the queue of this handle is full
    db->asyncTimed = true;
    const int rc = search_core(db, query, query_length, gap_open, gap_ext, score_matrix, alphabet_length, search_type, mode,
                               d_score, d_end_query, d_end_target, nullptr, /*finalSync=*/false);
    db->asyncTimed = false;
    if (db->asyncEvents) db->asyncHead++;  // (its events were recorded: opalx_db_wait will wait for them)
    else if (rc == 0) {
        // a search that launched nothing (empty query or database): it still owns its place in the queue -- the searches
        // queued before it keep theirs, and its own opalx_db_wait reports 0 ms
        const int slot = (int)(db->asyncHead % OPALX_ASYNC_DEPTH);
        void* st = db->main();
        for (int i = 0; i < 2; i++)
            if (!db->asyncEv[2 * slot + i]) db->asyncEv[2 * slot + i] = opal_hip_event_create();
        if (st && db->asyncEv[2 * slot] && db->asyncEv[2 * slot + 1] && !opal_hip_event_record(db->asyncEv[2 * slot], st) &&
            !opal_hip_event_record(db->asyncEv[2 * slot + 1], st))
            db->asyncHead++;
    } else {
        db->quiesce();  // (an error: nothing of THIS call stays in flight; the earlier searches have completed, their entries stay)
    }
    db->asyncEvents = false;
    return rc;
}

extern "C" int opalx_db_wait2(OpalxDb* db, float* kernel_ms, float* period_ms) {
    if (kernel_ms) *kernel_ms = 0.f;
    if (period_ms) *period_ms = 0.f;
    if (!db) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (db->asyncTail == db->asyncHead) {  // nothing queued: the streams
        void* st = db->main();
        if (!st) return OPAL_ERR_NO_SIMD_SUPPORT;
        return opal_hip_stream_sync(st) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
    }
    const int slot = (int)(db->asyncTail % OPALX_ASYNC_DEPTH);
    const int before = (int)((db->asyncTail + OPALX_ASYNC_DEPTH - 1) % OPALX_ASYNC_DEPTH);  // (its events are not reused while fewer than DEPTH - 1 searches are queued)
    const bool havePrev = db->asyncWaited > 0;
    db->asyncTail++;
    if (opal_hip_event_sync(db->asyncEv[2 * slot + 1])) return OPAL_ERR_NO_SIMD_SUPPORT;
    db->asyncWaited++;
    if (kernel_ms) *kernel_ms = opal_hip_event_elapsed_ms(db->asyncEv[2 * slot], db->asyncEv[2 * slot + 1]);
    if (period_ms && havePrev && db->asyncEv[2 * before + 1]) {
        const float p = opal_hip_event_elapsed_ms(db->asyncEv[2 * before + 1], db->asyncEv[2 * slot + 1]);
        *period_ms = p > 0.f ? p : 0.f;
    }
    return 0;
}
extern "C" int opalx_db_wait(OpalxDb* db, float* kernel_ms) { return opalx_db_wait2(db, kernel_ms, nullptr); }

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
    // One GPU and a database large enough to be ordered on the device: the look through the N result structs for the ones
    // that need computing (a pool job) runs WHILE the device orders the lengths, on the assumption -- the usual case, and what
    // both applications of the reference do -- that all of them do. If some hold results already, that fill is dropped and the
    // call goes on with the sequences that are left.
    const std::vector<int> devicesEarly = call_devices();
    bool all = false, looked = false;
    const std::function<bool()> look = [&]() { looked = true; all = collect_todo(results, dbLength, searchType, todo); return all; };
    const bool overlapLook = devicesEarly.size() <= 1 && dbLength >= 16384 && opts().deviceOrder != 0;
    if (!overlapLook) look();
    int n = overlapLook ? dbLength : all ? dbLength : (int)todo.size();
    const int* ids = overlapLook || all ? nullptr : todo.data();
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
        const std::vector<int>& devices = devicesEarly;
        // alignment search: operation strings traced on the device, per shard (positions of a shard in the to-do list: tracePos)
        std::vector<TraceOut> traces(std::max<size_t>(devices.size(), 1));
        std::vector<std::vector<int>> tracePos;
        if (devices.size() <= 1 || n < 2 * 64) {
            if (devices.size() == 1 && opal_hip_set_device(devices[0])) return OPAL_ERR_NO_SIMD_SUPPORT;
            rc = run_shard(opal_hip_get_device(), query, Q, db, dbSeqLengths, ids, n, Go, Ge, M, A, st, mode, score, endQ, endT, tms,
                           startQ, startT, traced ? &traces[0] : nullptr, overlapLook ? &look : nullptr);
            if (overlapLook && !looked) look();  // (cannot happen: db_fill calls the hook exactly once; kept as a guard)
            if (rc == RUN_SHARD_DROPPED || (overlapLook && !all)) {
                // some results were there already: the plain way, with the sequences that are left
                n = (int)todo.size();
                ids = todo.data();
                rc = n > 0 ? run_shard(opal_hip_get_device(), query, Q, db, dbSeqLengths, ids, n, Go, Ge, M, A, st, mode, score, endQ, endT, tms,
                                       startQ, startT, traced ? &traces[0] : nullptr)
                           : (args_fit_int32(Go, Ge, M, A) ? 0 : OPAL_ERR_OVERFLOW);
            }
            tracePos.resize(1);  // (identity: shard position = position in the to-do list)
        } else {
            // several GPUs: deal 64-sequence groups balancing the residue count, one host thread per GPU, each
            // running the single-GPU pipeline on its shard; the host is the gather root (results live in host structs)
            std::vector<int> lens((size_t)n);
            for (int i = 0; i < n; i++) lens[(size_t)i] = dbSeqLengths[ids ? ids[i] : i];
            std::vector<std::vector<int>> shard = deal_groups(lens.data(), n, (int)devices.size());
            std::vector<int> rcs(devices.size(), 0);
            std::vector<std::vector<int>> out(devices.size());
            const int callerDevice = opal_hip_get_device();
            {
                // (persistent workers, one per listed device slot: no thread creation per call)
                run_on_device_workers(devices.size(), [&](size_t k) {
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
    // Alignment search: score + end cells, start cells (rev_kernel.hip) and the operations inside each (start..end) box
    // (trace_kernel.hip) on the device. What the device leaves out -- sequences whose score/end the caller supplied (not
    // recomputed here, opal.h:118-122), boxes beyond its limits -- is traced by the host (opal_traceback.cpp).
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
        count(CNT_HOST_TRACED, left);
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

