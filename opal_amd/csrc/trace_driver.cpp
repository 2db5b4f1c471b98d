// trace_driver.cpp -- host drivers of the alignment search's device passes: start cells (rev_kernel.hip; reference
// src/opal.cpp:1475-1507) and the operations themselves (trace_kernel.hip; reference src/opal.cpp:1369-1409). What the
// device leaves out is traced by the host (opal_traceback.cpp).
#include "opal_internal.hpp"

namespace opal_host {

// Start cells on the device for every sequence of `db` (see include/opal_hip.h, opalx_search_starts).
int search_starts(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A, int mode,
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
    a.residues = db->res.p; a.seq_offset = db->dSeqOff.p; a.seq_item = db->seqItem.p;
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
    if (doubtful > 0 && !opts().noExactRow) {
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

/*
 * Alignment operations on the device (trace_kernel.hip) for every sequence of the shard whose score, end cell and start
 * cell are known: the box between start and end is computed with one byte of flags per cell in HBM and walked back there;
 * only the operation strings come to the host. Batches are bounded by OPAL_TRACEBACK_DEVICE_BYTES of flags (default
 * 8 GiB). Whatever is left out (no start cell, SW score 0, boxes of more than 2560 rows, a box larger than the budget, any
 * failure) keeps len < 0 and is traced by the host as before. `dEndQ` .. `dStartT` are the device arrays in caller
 * (= shard) order the search and the start-location pass have filled; the host arrays hold the same values.
 */
void device_traceback(OpalxDb* db, int Q, int Go, int Ge, int A, int mode, int n, const int* score, const int* endQ,
                             const int* endT, const int* startQ, const int* startT, const int* dEndQ, const int* dEndT,
                             const int* dStartQ, const int* dStartT, TraceOut& out) {
    out.len.assign((size_t)n, -1);
    out.off.assign((size_t)n, 0);
    out.chunk.assign((size_t)n, 0);
    out.chunks.clear();
    if (opts().hostTraceback || !db->seqItemValid || n <= 0) return;
    uint64_t budget = 8ull << 30;
    if (opts().tracebackDeviceBytes > 0) budget = (uint64_t)opts().tracebackDeviceBytes;
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
            a.residues = db->res.p; a.seq_offset = db->dSeqOff.p; a.seq_item = db->seqItem.p;
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
    count(CNT_DEVICE_TRACED, traced);
    if (db->trDirs.cap > (2ull << 30)) db->trDirs.release();  // a cached context does not sit on gigabytes of flags between calls
    if (debug_on())
        fprintf(stderr, "[opal] device traceback: %d of %d alignments in %d batches, %.2f ms (kernels + copies %.2f, bookkeeping %.2f)\n", traced, n,
                batches, now_ms() - t0, tDev, tPack);
}

}  // namespace opal_host
