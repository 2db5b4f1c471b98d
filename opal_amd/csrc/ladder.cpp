// ladder.cpp -- the scheduler of one search: plays the role of the reference's precision ladder
// (reference src/opal.cpp:496-535 searchDatabaseSW, :983-1021 searchDatabase<MODE>) with the rungs gfx950 has -- 16 bit
// first, 32 bit for whatever saturates or cannot be proven safe -- and turns every rung into kernel launches (one-launch
// class kernels, separate class launches, row chunks of long queries, streaming kernels).
#include "opal_internal.hpp"

namespace opal_host {

// ---- queued searches in turn on two streams (OpalxDb::Pipe)
// What the search in hand must wait for before its first launch: if the search before it is ONE tracked grid and this launch is
// one too (`grid`), a one-wave kernel that ends when that grid's last workgroup has started; anything else, the whole search.
// Every launch site of search_core comes through here first (once per search: the first call decides).
static int pipe_resolve(OpalxDb* db, void* stream, bool grid) {
    OpalxDb::Pipe& pp = db->pipe;
    if (!pp.active) return 0;
    if (!grid || pp.launched) pp.untracked = true;  // (a second launch of any kind: the search is no longer one grid)
    if (pp.resolved) return 0;
    pp.resolved = true;
    int e = 0;
    if (pp.prevKind == 1 && grid && opts().pipeline == 1) {
        e = opal_hip_wait_count(pp.prevCtr, pp.prevTarget - opts().pipeSlack, stream);
        if (debug_on()) fprintf(stderr, "[opal] queued search: grid behind a wait for %d started workgroups of the search before it\n", pp.prevTarget);
    } else if (pp.prevKind != 0) {
        void* done = pp.done[pp.lastIdx];
        if (done) e = opal_hip_stream_wait_event(stream, done);
    }
    // the search's own time starts here: behind the wait, in front of its first launch
    if (!e && pp.startEv) e = opal_hip_event_record(pp.startEv, stream);
    pp.startEv = nullptr;
    return e;
}
// ... and after a tracked grid has been launched: its workgroups count themselves in (fill_launch sets started_counter)
static void pipe_launched_grid(OpalxDb* db) {
    OpalxDb::Pipe& pp = db->pipe;
    if (!pp.active) return;
    pp.launched = true;
    pp.blocks += opal_hip_last_launch_blocks();
}

static void fill_launch(OpalHipLaunch& a, const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work,
                        int* ovfCounter) {
    memset(&a, 0, sizeof a);
    OpalxDb* db = c.db;
    a.width = width; a.mode = c.mode; a.want_end = swEndPass; a.detect_wrap = detectWrap;
    a.query = db->query.p; a.query_length = c.Q; a.score_matrix = db->matrix.p; a.alphabet_length = c.A;
    a.gap_open = c.Go; a.gap_ext = c.Ge;
    a.residues = c.zeroCopy ? db->hRaw.p : db->res.p; a.seq_offset = db->dSeqOff.p;  // (zeroCopy: sliced launches of a cold opalSearchDatabase call)
    a.residues_out = c.zeroCopy ? db->res.p : nullptr;
    a.tokens = db->tokens.p; a.pair_token_offset = db->pairOff.p;  // (width 16 only: ensure_pair_tokens)
    a.pair_len = db->dPairLen.p; a.pair_seq = db->dPairSeq.p;
    a.work = work;
    a.out_score = c.dScore; a.out_end_query = c.wantEnd ? c.dEndQ : nullptr; a.out_end_target = c.wantEnd ? c.dEndT : nullptr;
    a.overflow_count = ovfCounter; a.overflow_items = db->ovfItems.p;
    a.sat_limit = c.satLimit; a.neg_pad = c.negPad;
    a.stream = c.stream ? c.stream : db->main();
    a.clock_probe = db->clockProbe ? (unsigned long long*)(db->hdrDev + 16) : nullptr;
    a.timeline = (db->clockProbe && db->timelineProbe) ? db->timeline.p : nullptr;
    a.started_counter = nullptr;  // (the tracked launch sites set it: one-grid searches queued on the handle)
}

// Queries that no single wave geometry can hold (more than 64 lanes x 40 rows, or a query profile
// beyond the 160 KiB of LDS): the query is cut into row chunks processed top to bottom by one launch each;
// the bottom row of a chunk (H and the F input of the next row, per column) travels through HBM.
static int launch_chunked(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, int firstItem,
                          int numItems, LenView lenOf, int* ovfCounter) {
    OpalxDb* db = c.db;
    if (pipe_resolve(db, c.stream ? c.stream : db->main(), false)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (row chunks share db->boundary)
    int bestG = 0, bestS = 0, wpb = 4;
    // the items of THIS range (a re-run list, the 32-bit prefix of a global-mode search: a handful of long sequences, which
    // want every lane on one sequence; the whole database: the bulk decides): longest first, mean from a sample
    double meanLen = 0;
    const int sampleStep = std::max(1, numItems / 256);
    int sampled = 0;
    for (int i = 0; i < numItems; i += sampleStep, sampled++) meanLen += lenOf(firstItem + i);
    meanLen /= std::max(sampled, 1);
    if (!plan_chunks(width, c.mode, swEndPass, c.Q, c.A, numItems, meanLen, numItems > 0 ? lenOf(firstItem) : 0, &bestG, &bestS, &wpb)) {
        fprintf(stderr, "[opal] alphabet of %d letters does not fit the on-chip query profile at width %d\n", c.A, width);
        return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    if (opts().chunkG > 0) {  // OPAL_HIP_CHUNK_GEOMETRY "G,S": test hook to force small chunks
        const int g = opts().chunkG, r = opts().chunkS;
        int ldsPad = 0;
        if (opal_hip_rows_supported(width, c.mode, swEndPass | 2, r) == r &&
            opal_hip_lds_bytes(width, g, r, c.A) <= 160 * 1024) {
            bestG = g; bestS = r;
            pick_occupancy(kernel_vgprs(width, c.mode, swEndPass, 1, 0, bestS), opal_hip_lds_bytes(width, bestG, bestS, c.A), &wpb, &ldsPad,
                           opal_hip_max_waves_per_block(width));
        }
    }
    // (8 bytes per column of every item's -- width 16: every pair's -- residue stream)
    if (!db->boundary.ensure(8 * ((size_t)db->residues + 16)) || !db->scratch.ensure(2 * (size_t)std::max(db->numSeqs, 1)))
        return OPAL_ERR_NO_SIMD_SUPPORT;
    if (c.stream && bestG == 64 && numItems <= 256 && wpb > 4) wpb = 4;  // (as launch_class: the few longest, one wave per SIMD)
    if (c.sideBlocks && c.stream) *c.sideBlocks += (numItems + (64 / bestG) * wpb - 1) / ((64 / bestG) * wpb);
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
        count(CNT_CHUNK_LAUNCHES);
        if (width == 32) count(CNT_WIDTH32);
    }
    return 0;
}

// One class of a launch range on `stream` (a plain class launch). `side`: the launch belongs to a long-sequence class
// beside the bulk (issue priority).
static int launch_class(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, LaunchClass k, int skew,
                        int* ovfCounter, void* stream, bool side, bool grid = false) {  // (grid: the range's only launch -- see pipe_resolve)
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
    // A handful of sequences that each take a whole wave (64 lanes) and run beside the rest of the search are its critical path:
    // one wave per SIMD (the lone-wave rate), not eight waves of one workgroup sharing a CU -- ten 35 000-residue targets of a
    // 1000-aa HW search took 30 ms as two waves per SIMD, 15 as one.
    if ((side || c.stream) && k.G == 64 && k.count <= 256 && a.waves_per_block > 4) a.waves_per_block = 4;
    a.skew = skew;
    a.stream = stream;
    a.priority = side ? 1 : 0;
    if (opts().sidePrio >= 0) a.priority = side ? opts().sidePrio : 0;  // tuning hook
    if (debug_on())
        fprintf(stderr, "[opal] launch width=%d mode=%d end=%d items=[%d,+%d) G=%d rows=%d waves/block=%d ldsPad=%d%s%s\n", width,
                c.mode, swEndPass, k.first, k.count, k.G, k.rows, k.wpb, k.ldsPad, skew ? " skew" : "", side ? " (side)" : "");
    if (pipe_resolve(c.db, stream, grid)) return 1;
    if (grid && c.db->pipe.active) a.started_counter = c.db->ovfCount.p + 3;
    if (opal_hip_launch(&a)) return 1;
    if (grid) pipe_launched_grid(c.db);
    if (c.sideBlocks && c.stream) *c.sideBlocks += (k.count + (64 / k.G) * a.waves_per_block - 1) / ((64 / k.G) * a.waves_per_block);
    count(CNT_CLASS_LAUNCHES);
    if (width == 32) count(CNT_WIDTH32);
    return 0;
}

static int launch_range(const SearchCtx& c, int width, int swEndPass, int detectWrap, const int* work, int firstItem,
                        int numItems, LenView lenOf, int* ovfCounter) {
    if (numItems <= 0) return 0;
    OpalxDb* db = c.db;
    if (width == 16 && ensure_pair_tokens(db)) return OPAL_ERR_NO_SIMD_SUPPORT;
    // skewed kernels (width 17, NW/HW/OV) add up to (rows + steps) * Ge to every value: allowed for a class when
    // even its longest sequence stays inside the 16-bit range
    auto skew_ok = [&](int item) {
        const long long L0 = lenOf(item);
        // (frame: row position among the slot's G * S <= Q + 512 register rows + column, see dp_kernel.hpp)
        return std::min<long long>(c.qBound, L0 * c.maxScore) + (c.Q + 512 + L0 + 64 + 4) * (long long)c.Ge < 32000;
    };
    const bool maySkew = width == 17 && c.mode != OPAL_MODE_SW && !work && !opts().noSkew;
    // One launch for all classes where a one-launch kernel exists (plain 16-bit, no re-run list): see dp_kernel.hpp.
    const bool uberEligible = width == 17 && !work && !opts().noUber && !opts().lanesPerSeq;
    std::vector<LaunchClass> classes;
    std::string key;
    if (!work) {  // (re-run lists differ from search to search: planned every time)
        char buf[256];
        snprintf(buf, sizeof buf, "%d/%d/%d/%d/%d/%d/%d/%lld/%lld/%d/%d/%d|%s", width, c.mode, swEndPass, c.Q, c.A, c.Go, c.Ge, c.maxScore, c.qBound, firstItem, numItems,
                 (int)uberEligible, opts().plan_key().c_str());
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
    if (classes.empty() || opts().forceChunks) {
        if (!gate_items(db, c.stream ? c.stream : db->main(), 0, 2 * db->numPairs)) return OPAL_ERR_NO_SIMD_SUPPORT;
        return launch_chunked(c, width, swEndPass, detectWrap, work, firstItem, numItems, lenOf, ovfCounter);
    }
    // Can the classes run as one grid? (bulk on G0 lanes, the others on 2*G0 and 4*G0, all on the same arithmetic)
    const LaunchClass& bulk = classes.back();
    const int skewBulk = maySkew && skew_ok(bulk.first);
    bool oneGrid = uberEligible && classes.size() <= 3 && bulk.wpb == 4 && (bulk.ldsPad == 0 || opts().uberLdsPad) &&
                   opal_hip_uber_supported(c.mode, swEndPass, skewBulk, bulk.rows);
    for (size_t ki = 0; oneGrid && ki + 1 < classes.size(); ki++) {
        const LaunchClass& k = classes[ki];
        if ((maySkew && skew_ok(k.first)) != skewBulk) oneGrid = false;  // a class that cannot use the skewed kernels: separate launches
        else if (k.G != 2 * bulk.G && k.G != 4 * bulk.G) oneGrid = false;
    }
    // items [i0, i1) of the classes as ONE grid on `stream` (any class may be empty there)
    auto launch_grid = [&](const SearchCtx& cx, int i0, int i1, void* stream, bool tracked = false) -> int {
        OpalHipUberLaunch u;
        memset(&u, 0, sizeof u);
        fill_launch(u.base, cx, width, swEndPass, detectWrap, nullptr, ovfCounter);
        u.base.stream = stream;
        u.base.lanes_per_seq = bulk.G; u.base.rows_per_lane = bulk.rows; u.base.waves_per_block = 4; u.base.skew = skewBulk;
        u.base.lds_bytes_min = bulk.ldsPad;  // (an LDS request padded so that every SIMD holds an even number of waves)
        for (const LaunchClass& k : classes) {
            const int a = std::max(k.first, i0), b = std::min(k.first + k.count, i1);
            const int n = std::max(0, b - a);
            if (k.G == bulk.G) { u.base.first_item = a; u.base.num_items = n; }
            else if (k.G == 2 * bulk.G) { u.long_first_item = a; u.long_num_items = n; u.long_priority = 1; }
            else { u.longest_first_item = a; u.longest_num_items = n; u.longest_priority = 1; }
        }
        if (debug_on())
            fprintf(stderr, "[opal] one launch width=%d mode=%d end=%d G0=%d rows=%d%s: bulk [%d,+%d) long [%d,+%d) longest [%d,+%d)\n", width,
                    c.mode, swEndPass, bulk.G, bulk.rows, skewBulk ? " skew" : "", u.base.first_item, u.base.num_items, u.long_first_item,
                    u.long_num_items, u.longest_first_item, u.longest_num_items);
        if (pipe_resolve(db, stream, tracked)) return 1;
        if (tracked && db->pipe.active) u.base.started_counter = db->ovfCount.p + 3;
        if (opal_hip_launch_uber(&u)) return 1;
        if (tracked) pipe_launched_grid(db);
        count(CNT_ONE_LAUNCH);
        return 0;
    };

    // ---- the residues are still arriving (opalSearchDatabase: gather and upload overlap the search): the range is cut into
    // SLICES of whole upload chunks, every slice launched as soon as its chunks have been enqueued, on two streams in turn
    // (`main` and `aux`) so that a slice starts while the previous one drains. Sequences are sorted longest first, so
    // the first slice holds the long classes.
    if (db->gate.active && !work && !c.stream && (width == 17 || width == 18) && db->gate.chunks() >= 4 && !opts().noSlices) {
        const UploadGate& g = db->gate;
        SearchCtx zc = c;
        // (a database seen for the first time: the launches read the pinned staging buffer over PCIe and write the device copy
        // -- on the kernels instantiated with that store: SW score searches that run as one grid, inst_uber_swzc.hip)
        zc.zeroCopy = g.zeroCopy && oneGrid && width == 17 && c.mode == OPAL_MODE_SW && !swEndPass && opal_hip_uber_supported(OPAL_MODE_SW, 16, 0, bulk.rows);
        const int zcFrom = opts().zcFromSlice;  // slices [0, zcFrom) take the upload path (their chunks copied when asked for), the others read pinned memory
        // chunks per slice, the rest in the last one (C2: 17 chunks of 2 MB). The FIRST slice holds the long classes: its
        // longest sequence is the critical path of the whole search, so it is as small as it can be (it starts as soon as
        // one chunk has landed) and runs on a high-priority side stream of its own; the others take `main` and `aux` in
        // turn, so that a slice starts while the previous one drains.
        std::vector<int> pattern{1, 4};  // (measured on C2, 17 chunks: 1,4 | 1,3,5 | 1,2,4 | 1,3,5,4 all within 0.1 ms; 1,16 and one launch 0.3 ms slower)
        if (!opts().slices.empty()) pattern = opts().slices;
        std::vector<int> cut{firstItem};
        int chunk = 0;
        for (int want : pattern) {
            chunk += want;
            if (chunk >= g.chunks()) break;
            int at = g.cutItem[(size_t)chunk];
            if (cut.size() == 1 && classes.size() > 1) at = std::max(at, bulk.first);  // every long class inside the first slice
            if (at > cut.back() && at < firstItem + numItems) cut.push_back(at);
        }
        // a database of many chunks (hundreds of MB: the copy takes as long as the search) goes on in slices of a sixth of
        // the chunks, none smaller than two resident rounds (65 536 sequences)
        if (opts().slices.empty() && g.chunks() > 24) {
            const int group = std::max(5, g.chunks() / 6);
            while (chunk + group < g.chunks()) {
                chunk += group;
                const int at = g.cutItem[(size_t)chunk];
                if (at - cut.back() >= 65536 && firstItem + numItems - at >= 65536) cut.push_back(at);
            }
        }
        cut.push_back(firstItem + numItems);
        if (zc.zeroCopy) {
            // ... and the kernels that read pinned memory write the device copy: of every item, or of the items from some slice on
            // (the chunks of the items before it are copied when their launches -- or the swapped launch -- ask for them)
            const int filledFrom = zcFrom < (int)cut.size() - 1 ? cut[(size_t)zcFrom] : INT_MAX;
            if (filledFrom == 0 && numItems == 2 * db->numPairs) db->gate.kernelFilled = true;
            else if (filledFrom != INT_MAX && firstItem + numItems == 2 * db->numPairs) db->gate.kernelFilledFrom = filledFrom;
            if (filledFrom == INT_MAX) zc.zeroCopy = false;
        }
        void* fork = db->event(0);
        void* aux = db->aux();
        void* side = db->stream(0);
        const int how = opts().sliceStreams;
        if (!fork || !aux || !side || opal_hip_event_record(fork, db->main()) || opal_hip_stream_wait_event(aux, fork) ||
            opal_hip_stream_wait_event(side, fork))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        if (how == 2 && (!db->stream(1) || !db->stream(2) || opal_hip_stream_wait_event(db->stream(1), fork) || opal_hip_stream_wait_event(db->stream(2), fork)))
            return OPAL_ERR_NO_SIMD_SUPPORT;
        int err = 0;
        bool auxUsed = false;
        const bool dbg = debug_on();
        std::vector<void*> marks;  // (debug: timed events around every slice)
        for (size_t j = 0; j + 1 < cut.size() && !err; j++) {
            // stream of slice j: 0 on the side stream; the others on main / aux in turn ("alternate", the default: a slice
            // starts while the previous one drains), all on `main` ("serial": 0.3 ms slower on C2 -- a slice smaller than one
            // resident round, 32 000 sequences, leaves every SIMD with one or two waves), or main, aux, side streams 1 and 2
            // ("separate": no better than alternate)
            void* stream = j == 0 ? side : db->main();
            if (j > 0 && how == 1) stream = (j & 1) ? db->main() : aux;
            if (j > 0 && how == 2) stream = j == 1 ? db->main() : j == 2 ? aux : db->stream(1 + (int)(j & 1));
            auxUsed |= stream != side && stream != db->main();
            const bool zcj = zc.zeroCopy && (int)j >= zcFrom;
            const SearchCtx& cx = zcj ? zc : c;
            // Slices exist to start before the gather has finished. Once every remaining chunk HAS been gathered (a large database:
            // the gather takes 2.5 ms of a 16 ms search) the rest goes out as one launch -- kernels that read the pinned buffer need
            // nothing else, and one large grid runs better than three that take turns on two streams (C4 shard: slices 4-6 started
            // only when 2 and 3 had ended, 18.7 ms for 15.5 ms of DP).
            if (zcj && j >= 1 && j + 2 < cut.size()) {
                const int c0 = (int)(std::upper_bound(g.cutItem.begin(), g.cutItem.end(), cut[j]) - g.cutItem.begin()) - 1;
                bool gathered = true;
                for (int cc = std::max(c0, 0); cc < g.chunks() && gathered; cc++) gathered = g.state[(size_t)cc].load(std::memory_order_acquire) != 0;
                if (gathered) cut.erase(cut.begin() + (long)j + 1, cut.end() - 1);
            }
            if (!gate_items(db, stream, cut[j], cut[j + 1] - cut[j], /*needDevice=*/!zcj)) { err = 1; break; }
            if (dbg) { void* ev = opal_hip_event_create(); marks.push_back(ev); if (ev) opal_hip_event_record(ev, stream); }
            if (oneGrid) err = launch_grid(cx, cut[j], cut[j + 1], stream);
            else
                for (size_t ki = 0; ki < classes.size() && !err; ki++) {
                    LaunchClass k = classes[ki];
                    const int a = std::max(k.first, cut[j]), b = std::min(k.first + k.count, cut[j + 1]);
                    if (b <= a) continue;
                    k.first = a; k.count = b - a;
                    err = launch_class(cx, width, swEndPass, detectWrap, nullptr, k, maySkew && skew_ok(classes[ki].first), ovfCounter, stream, ki + 1 < classes.size());
                }
            if (dbg) { void* ev = opal_hip_event_create(); marks.push_back(ev); if (ev) opal_hip_event_record(ev, stream); }
        }
        if (!err) {
            void* join = db->event(1);
            if (!join || opal_hip_event_record(join, side) || opal_hip_stream_wait_event(db->main(), join)) err = 1;
        }
        if (!err && auxUsed)  // (whichever of them were used: joining an idle stream costs nothing)
            for (int q = 0; q < (how == 2 ? 3 : 1) && !err; q++) {
                void* other = q == 0 ? aux : db->stream(q);
                void* join = db->event(q == 0 ? 6 : 1 + q);
                if (!other || !join || opal_hip_event_record(join, other) || opal_hip_stream_wait_event(db->main(), join)) err = 1;
            }
        if (err) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        if (dbg && !marks.empty() && marks[0]) {  // (debug only: waits for the search)
            opal_hip_stream_sync(db->main());
            for (size_t j = 0; 2 * j + 1 < marks.size(); j++)
                if (marks[2 * j] && marks[2 * j + 1])
                    fprintf(stderr, "[opal] slice %zu items [%d,%d): starts %.3f ms after slice 0 could, runs %.3f ms\n", j, cut[j], cut[j + 1],
                            opal_hip_event_elapsed_ms(marks[0], marks[2 * j]), opal_hip_event_elapsed_ms(marks[2 * j], marks[2 * j + 1]));
            for (void* ev : marks) opal_hip_event_destroy(ev);
        }
        return 0;
    }
    if (c.stream) {
        // the caller runs this range beside something else (the long sequences of a database whose bulk is streamed): every
        // class on the caller's stream, longest sequences first
        if (!gate_items(db, c.stream, 0, 2 * db->numPairs)) return OPAL_ERR_NO_SIMD_SUPPORT;
        for (size_t ki = 0; ki < classes.size(); ki++)
            if (launch_class(c, width, swEndPass, detectWrap, work, classes[ki], maySkew && skew_ok(classes[ki].first), ovfCounter, c.stream, ki + 1 < classes.size())) {
                db->quiesce();
                return OPAL_ERR_NO_SIMD_SUPPORT;
            }
        return 0;
    }
    // everything this range reads must have landed (no-op unless a fill is still in flight)
    if (!gate_items(db, db->main(), 0, 2 * db->numPairs)) return OPAL_ERR_NO_SIMD_SUPPORT;

    // (one grid, or one class, of a plain 16-bit range that is not a re-run list: what a queued search may overlap with its neighbours)
    const bool trackable = !work && (width == 17 || width == 18);
    if (oneGrid && classes.size() >= 2) {
        if (launch_grid(c, firstItem, firstItem + numItems, db->main(), trackable)) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        return 0;
    }
    // The last class (the bulk of the database) runs on the main stream; the classes of long
    // sequences run beside it on up to three side streams (one hardware queue each), forked from and
    // joined back into the main stream with events, longest sequences enqueued first.
    const int nside = std::min<int>((int)classes.size() - 1, 3);
    void* fork = nullptr;
    if (!(trackable && classes.size() == 1) && pipe_resolve(db, db->main(), false)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (before the fork)
    if (nside > 0) {
        fork = db->event(0);
        if (!fork || opal_hip_event_record(fork, db->main())) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    int err = 0;
    bool recorded[3] = {false, false, false};
    // Submission order matters: the side launches (long sequences) go first. Submitting the bulk launch first was
    // measured at 2.2 ms instead of 1.55 ms on C2: its workgroups fill every CU and the long sequences start late.
    for (size_t ki = 0; ki < classes.size() && !err; ki++) {
        const bool side = (int)ki < nside;
        void* stream = side ? db->stream((int)ki) : db->main();
        if (side && (!stream || opal_hip_stream_wait_event(stream, fork))) { err = 1; break; }
        if (launch_class(c, width, swEndPass, detectWrap, work, classes[ki], maySkew && skew_ok(classes[ki].first), ovfCounter, stream, side,
                         trackable && classes.size() == 1)) { err = 1; break; }
        if (side) {
            void* done = db->event(1 + (int)ki);
            if (!done || opal_hip_event_record(done, stream)) { err = 1; break; }
            recorded[ki] = true;
        }
    }
    for (int i = 0; i < 3; i++)
        if (recorded[i] && opal_hip_stream_wait_event(db->main(), db->event(1 + i))) err = 1;
    if (err) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }  // nothing of this call stays in flight
    return 0;
}

// Streaming launches: how many of the range's longest sequences (handed out first) get their wave a raised issue priority.
// A slot works through residues / slots columns on average; a sequence that alone is a quarter of that or more decides when its
// wave ends, and with it the search. Not more of them than an eighth of the slots: priority among many is none.
static int stream_prio_items(const OpalxDb* db, int firstItem, int numItems, int G) {
    if (opts().streamPrio == 0 || numItems <= 0) return 0;
    const long long slots = 256LL * 16 * (64 / G);
    long long res = 0;
    const int step = std::max(1, numItems / 4096);
    for (int i = 0; i < numItems; i += step) res += db->pairLen[(size_t)firstItem + (size_t)i];
    res *= step;
    const long long share = res / slots;  // columns per slot
    const int* len = db->pairLen.data() + firstItem;
    // lengths are sorted longest first: the number of items longer than share / 4
    int lo = 0, hi = numItems;
    while (lo < hi) { const int mid = (lo + hi) / 2; if ((long long)len[mid] * 4 > share) lo = mid + 1; else hi = mid; }
    return (int)std::min<long long>(lo, slots / 8);
}

// The largest value any DP cell of this query can hold: every query residue against its best letter (never less than 0 per
// residue: a residue may stay unaligned in SW and costs a gap, not a match, elsewhere -- and a bound may be loose, never tight
// from below). A residue outside the alphabet counts with the largest entry of the matrix. See search_core.
long long query_score_bound(const unsigned char* query, int Q, const int* M, int A) {
    long long maxS = 0;
    for (int i = 0; i < A * A; i++) maxS = std::max<long long>(maxS, M[i]);
    long long rowMax[256];
    for (int r = 0; r < 256; r++) {
        rowMax[r] = maxS;
        if (r < A) { rowMax[r] = 0; for (int b = 0; b < A; b++) rowMax[r] = std::max<long long>(rowMax[r], M[r * A + b]); }
    }
    long long bound = 0;
    for (int i = 0; i < Q; i++) bound += rowMax[query[i]];
    return bound;
}

// reference opal.cpp:178-198 / :610-630 at 32 bits
bool args_fit_int32(int Go, int Ge, const int* M, int A) {
    const int lo = INT_MIN / 2, hi = INT_MAX / 2;
    if (Go <= lo || hi <= Go || Ge <= lo || hi <= Ge) return false;
    for (int i = 0; i < A * A; i++)
        if (M[i] <= lo || hi <= M[i]) return false;
    return true;
}

// Query, score matrix and zeroed counters reach the device as ONE copy from pinned memory (three separate small
// copies and a memset cost ~10 us of stream time each, on a search whose kernels take 1.6 ms).
int ensure_query_buffers(OpalxDb* db, const unsigned char* query, int Q, const int* M, int A) {
    const size_t mb = (size_t)A * A * sizeof(int);
    const size_t head = 32;  // {counters[4], clock probe[2 x u64]}
    const size_t bytes = (head + mb + (size_t)std::max(Q, 1) + 255) / 256 * 256;
    // One DEVICE slot per search in flight as well (round 6): queued searches overlap on two streams, and a search must not find its
    // query, matrix or counters overwritten by the upload of the next one.
    // (searches may be queued behind each other, opalx_search_device_async: a buffer that has to grow waits for them first)
    if ((db->header.cap < OPALX_ASYNC_DEPTH * bytes || db->hHeader.cap < OPALX_ASYNC_DEPTH * bytes) && (db->header.p || db->hHeader.p)) {
        if (opal_hip_stream_sync(db->main())) return 1;
        if (db->pipeStream && opal_hip_stream_sync(db->pipeStream)) return 1;
        if (db->mainStream && opal_hip_stream_sync(db->mainStream)) return 1;
    }
    if (!db->header.ensure(OPALX_ASYNC_DEPTH * bytes, 1.5) || !db->hHeader.ensure(OPALX_ASYNC_DEPTH * bytes, 1.5)) return 1;
    // the pinned staging slot of this search: free once the upload recorded behind its last use has run
    const int slot = (int)(db->hdrSeq++ % OPALX_ASYNC_DEPTH);
    if (db->hdrEvSet[slot] && opal_hip_event_sync(db->hdrEv[slot])) return 1;
    unsigned char* h = db->hHeader.p + (size_t)slot * (db->hHeader.cap / OPALX_ASYNC_DEPTH / 256 * 256);
    unsigned char* d = db->header.p + (size_t)slot * (db->header.cap / OPALX_ASYNC_DEPTH / 256 * 256);
    memset(h, 0, head);
    memcpy(h + head, M, mb);
    if (Q) memcpy(h + head + mb, query, (size_t)Q);
    if (opal_hip_memcpy_h2d(d, h, head + mb + (size_t)Q, db->main())) return 1;
    if (!db->hdrEv[slot]) db->hdrEv[slot] = opal_hip_event_create_untimed();
    db->hdrEvSet[slot] = db->hdrEv[slot] && !opal_hip_event_record(db->hdrEv[slot], db->main());
    if (!db->hdrEvSet[slot] && opal_hip_stream_sync(db->main())) return 1;  // (no event: the slot is free once the stream is idle)
    db->hdrDev = d;
    db->ovfCount.p = (int*)d;
    db->matrix.p = (int*)(d + head);
    db->query.p = d + head + mb;
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

/*
 * The scheduler. Plays the role of the reference's precision ladder (opal.cpp:496-535, :983-1021)
 * with the rungs gfx950 has: 16-bit first, 32-bit for whatever saturates or cannot be
 * proven safe. OPAL_OVERFLOW_SIMPLE and OPAL_OVERFLOW_BUCKETS give identical results in the
 * reference (only the re-run granularity differs); here both map to the same schedule.
 */
int search_core(OpalxDb* db, const unsigned char* query, int Q, int Go, int Ge, const int* M, int A,
                       int searchType, int mode, int* dScore, int* dEndQ, int* dEndT, float* kernelMs, bool finalSync) {
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
    // A QUEUED search (opalx_search_device_async) takes the stream the one before it did not: its grid is released when that one
    // has started its last workgroup and fills the CUs it leaves (OpalxDb::Pipe). Everything below asks db->main() for "the"
    // stream and gets this one.
    struct PipeScope {
        OpalxDb* db; bool piped = false; bool ok = false; int idx = 0;
        ~PipeScope() {
            OpalxDb::Pipe& pp = db->pipe;
            if (piped) {
                void* st = db->curStream;
                if (ok && st) {
                    if (!pp.resolved) pipe_resolve(db, st, false);  // (a search that launched nothing still keeps the order)
                    if (!pp.done[idx]) pp.done[idx] = opal_hip_event_create_untimed();
                    if (pp.done[idx] && !opal_hip_event_record(pp.done[idx], st)) {
                        pp.prevKind = (pp.launched && !pp.untracked) ? 1 : 2;
                        pp.prevTarget = pp.blocks;
                        pp.prevCtr = db->ovfCount.p ? db->ovfCount.p + 3 : nullptr;
                        if (!pp.prevCtr) pp.prevKind = 2;
                        pp.lastIdx = idx;
                        if (idx == 1) pp.unjoined = true;
                    } else ok = false;
                }
                pp.active = false; pp.resolved = true; pp.startEv = nullptr;
                db->curStream = nullptr;
                if (!ok) db->quiesce();  // (an error: nothing of this search stays in flight, and nothing waits for it)
            }
            // (a search outside the queue runs on mainStream behind everything -- main() has joined the second stream -- and has
            // marked it `foreign`: the next queued search waits for all of it)
        }
    } scope{db};
    if (db->asyncTimed && opts().pipeline != 0 && !db->gate.active) {
        OpalxDb::Pipe& pp = db->pipe;
        if (db->asyncHead == db->asyncTail) pp.prevKind = 0;  // (every queued search has been waited for: nothing of theirs is in flight)
        if (!db->mainStream) db->main();
        if (!db->pipeStream) db->pipeStream = opts().pipePrio < 0 ? opal_hip_stream_create_low() : opts().pipePrio > 0 ? opal_hip_stream_create_high() : opal_hip_stream_create();
        if (db->mainStream && db->pipeStream) {
            if (pp.foreign) {
                // something else has been enqueued on mainStream since the last queued search (main() was handed out, and it made
                // mainStream wait for the second stream first): an event behind all of it stands for "the search before"
                if (!pp.done[0]) pp.done[0] = opal_hip_event_create_untimed();
                if (!pp.done[0] || opal_hip_event_record(pp.done[0], db->mainStream)) return OPAL_ERR_NO_SIMD_SUPPORT;
                pp.foreign = false; pp.unjoined = false;
                pp.prevKind = 2; pp.lastIdx = 0;
            }
            scope.piped = true;
            scope.idx = pp.prevKind != 0 ? (pp.lastIdx ^ 1) : 0;
            db->curStream = scope.idx ? db->pipeStream : db->mainStream;
            pp.active = true; pp.resolved = false; pp.untracked = false; pp.launched = false; pp.startEv = nullptr; pp.blocks = 0;
        }
    }
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
        if (!quads.empty() && pipe_resolve(db, st, false)) return OPAL_ERR_NO_SIMD_SUPPORT;
        if (write_degenerate(db, quads, dScore, dEndQ, dEndT)) return OPAL_ERR_NO_SIMD_SUPPORT;
    }
    if (Q == 0 || P == 0) {
        if (opal_hip_stream_sync(st)) return OPAL_ERR_NO_SIMD_SUPPORT;
        scope.ok = true;
        return 0;
    }

    if (ensure_query_buffers(db, query, Q, M, A)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (also zeroes the counters)
    if (db->clockProbe && db->timelineProbe && (!db->timeline.ensure(4 * 65536) || opal_hip_memset(db->timeline.p, 0, 4 * 65536 * sizeof(unsigned long long), st)))
        return OPAL_ERR_NO_SIMD_SUPPORT;

    long long maxS = 0, minS = 0;
    for (int i = 0; i < A * A; i++) { maxS = std::max<long long>(maxS, M[i]); minS = std::min<long long>(minS, M[i]); }
    bool fits16 = Go <= 32767 && Ge <= 32767 && maxS <= 32767 && minS >= -32768;
    // The largest value any cell can hold (H, E, F and the diagonal sum alike: each is the score of an alignment of two prefixes):
    // no alignment scores more than its matched query residues against their best letters, gaps only cost (reference recurrence
    // opal.cpp:748-771). Rounds 1-5 used min(Q, L) * maxS -- twice the true bound for a protein matrix (BLOSUM50: diagonal 5-15), and
    // with it half of a 2000-aa query's targets beyond 2 000 residues left the skewed 16-bit kernels for no reason.
    long long qBound = query_score_bound(query, Q, M, A);
    if (opts().noQueryBound) qBound = (long long)Q * maxS;
    auto hi_bound = [&](long long L) { return std::min<long long>(qBound, L * maxS); };  // (<= min(Q, L) * maxS)
    if (opts().forceWidth == 32) fits16 = false;  // test / profiling hook

    // Narrow rung: 17 = plain 16-bit, one sequence per lane (full-rate VALU, 4-byte LDS profile entries); 16 = packed
    // pairs (half-rate VALU, 2-byte entries: queries whose 4-byte profile exceeds LDS); 18 = plain with 2-byte entries
    // (+0.5 instructions per cell; experimental, on request only).
    int narrow = 17;
    bool chunkedQuery = false;  // no single launch holds the query at any width: row chunks
    if (opts().narrow) narrow = opts().narrow;
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
        const bool force17 = opts().narrowForce == 17;  // run width 17 in row chunks rather than switching width
        // (width 18 measured SLOWER than the packed pairs on the 2000-aa workload, 15.1 vs 14.8 ms, so it is never
        // chosen automatically: OPAL_HIP_NARROW=18 selects it for experiments)
        if (narrow == 17 && !fits(17) && !force17) {
            narrow = 16;
            // ... except where the plain kernels with 2-byte profile entries measure faster (2000-aa query, class kernels,
            // 100 000 / 20 000 sequences: SW + end 4 150 against 3 350 / 3 720 against 2 970 GCUPS, OV 4 270 : 4 130 / 3 810 : 3 660,
            // NW and HW 4 970 : 5 100 / 4 400 : 4 040)
            if (!opts().narrow && fits(18) &&
                ((mode == OPAL_MODE_SW && wantEnd) || mode == OPAL_MODE_OV || (mode != OPAL_MODE_SW && db->numSeqs < 50000)))
                narrow = 18;
        }
        if (narrow == 18 && !fits(18)) narrow = 16;
        // Queries that no single launch holds at any width (more than 64 lanes x 40 rows) run in row chunks: on the plain
        // kernels with 2-byte profile entries, whose chunk profiles leave room for twice the waves of the 4-byte ones and which,
        // unlike the packed pairs, issue at full rate (3000-aa query x 100 000 sequences, class-kernel chunks: SW + end 5 110 against
        // 3 470 GCUPS at width 16, SW score 5 320 against 4 700; what streams its chunks -- stream_kernel.hpp -- does not come here)
        chunkedQuery = !fits(17) && !fits(18) && !fits(16);
        if (!opts().narrow && !force17 && chunkedQuery) {
            narrow = mode == OPAL_MODE_HW ? 17 : 18;  // (HW: no 2-byte-profile chunk kernels beyond 16 rows per lane, inst_s18multi.hip)
            if (opts().chunkWidth) narrow = opts().chunkWidth;
        }
    }
    // SW score-only searches with long queries stream their sequences through persistent waves (plain 16-bit arithmetic)
    StreamPlan splan;
    if (mode == OPAL_MODE_SW && fits16) splan = plan_stream(Q, A, mode, wantEnd, db->residues, db->pairLen.empty() ? 0 : db->pairLen[0], db->numSeqs);
    // A database with a long tail (Swiss-Prot: sequences of 35 000 residues among 350-residue ones): one sequence would pin its
    // lanes for longer than the whole search takes, so plan_stream says no -- to the whole database. Its BULK streams all
    // the same: the longest items [0, headItems) run on the class body (4G lanes each) as the first workgroups of the
    // streaming launch itself, the rest streams behind them. The head is the smallest of 128 / 512 / 2048 items that lets the
    // rest stream. (As a launch of its own on a side stream the head started before or after the stream's persistent workgroups
    // from run to run: P18080 x the Swiss-Prot shape 13.9 ms under the profiler, 14.4 / 15.4 / 18.9 ms without it, against
    // 15.3 ms on the class kernels alone -- profiles/r03_stream_head.txt.)
    int headItems = 0;
    if (mode == OPAL_MODE_SW && fits16 && !splan.G && !chunkedQuery && !opts().noStreamHead) {
        long long headRes = 0;
        int at = 0;
        for (int cut : {128, 512, 2048}) {
            if ((long long)cut * 8 > db->numSeqs) break;
            for (; at < cut; at++) headRes += db->pairLen[(size_t)at];
            if (headRes * 4 > db->residues) break;
            const StreamPlan rest = plan_stream(Q, A, mode, wantEnd, db->residues - headRes, db->pairLen[(size_t)cut], db->numSeqs - cut);
            if (rest.G && rest.passes == 1 && rest.width == 17 && rest.G <= 16 && rest.rows <= (wantEnd ? 36 : 39)) { splan = rest; headItems = cut; break; }  // (the shapes of inst_stream17head.hip)
        }
    }
    if (opts().streamHeadItems > 0) {  // test hook: a head of that many items on a (forced) stream
        const int n = opts().streamHeadItems;
        if (mode == OPAL_MODE_SW && splan.G && splan.passes == 1 && splan.width == 17 && splan.G <= 16 && splan.rows <= (wantEnd ? 36 : 39) && n > 0 && n < 2 * P)
            headItems = n;
    }
    // While the database is still arriving (opalSearchDatabase) a streaming launch would have to wait for the last chunk: at 16
    // lanes per sequence or fewer, where streaming gains 2-4 % over the class kernels, the sliced class launches that start
    // on the first chunks are the better trade (C4 shard: the copy of 218 MB takes a third of the call).
    if (splan.G && splan.G <= 16 && db->gate.active && db->gate.chunks() >= 4 && narrow == 17 && !opts().noSlices) {
        if (opts().stream != 1) { splan = StreamPlan(); headItems = 0; }
    }
    // End locations with CHEAP gaps (scores grow along the whole sequence: the best cell changes in every step, and every step pays
    // a settlement and a store, stream_kernel.hpp): at 16 lanes per sequence or fewer the class kernels are as fast or faster there
    // (published shape, gaps 3/1: 390 aa 14.8 ms against 15.7 streamed, 513 aa 18.6 either way; with gaps 11/1 the stream wins
    // 5 %: P18080 x 625 000 16.9 ms against 17.8). "Cheap" by a rule of thumb on the inputs: opening + 3 extensions below 0.6 of
    // the best match.
    if (splan.G && splan.G <= 16 && wantEnd && mode == OPAL_MODE_SW && opts().stream != 1 && 5 * ((long long)Go + 3LL * Ge) < 3 * maxS) {
        splan = StreamPlan(); headItems = 0;
    }
    const bool plain = narrow != 16 || splan.G;
    const int satLimit = plain ? (int)(32767 - maxS) : 32767;
    SearchCtx c{db, Q, A, Go, Ge, mode, wantEnd, dScore, dEndQ, dEndT, satLimit, mode == OPAL_MODE_SW ? -satLimit : 0, maxS, qBound};
    const int nNarrow = narrow == 16 ? P : 2 * P;  // items of the narrow rung: pairs or sequences
    const LenView pairLenOf{db->pairLen.data(), 2, nullptr};  // width-16 items are pairs
    const LenView seqLenOf{db->pairLen.data(), 1, nullptr};   // width-17/18/32 items are pair*2+half
    void* evStart = nullptr; void* evStop = nullptr;
    const bool timed = kernelMs || db->asyncTimed;  // (asyncTimed: opalx_search_device_async -- opalx_db_wait reads the events)
    db->asyncEvents = false;
    if (db->asyncTimed) {  // (queued searches time themselves with the events of their own slot)
        const int slot = (int)(db->asyncHead % OPALX_ASYNC_DEPTH);
        for (int i = 0; i < 2; i++)
            if (!db->asyncEv[2 * slot + i]) db->asyncEv[2 * slot + i] = opal_hip_event_create();
        evStart = db->asyncEv[2 * slot]; evStop = db->asyncEv[2 * slot + 1];
    } else if (timed) { evStart = db->event(4); evStop = db->event(5); }
    if (scope.piped && evStart) db->pipe.startEv = evStart;  // (recorded behind the wait for the search before it: pipe_resolve)
    else if (evStart) opal_hip_event_record(evStart, st);
    // A 16-bit rung whose scores can reach its limit may leave a re-run list (db->ovfItems, one per handle) and reads its length
    // back: such a search waits for the one before it and is waited for as a whole.
    const bool canSaturate = hi_bound(db->pairLen[0]) >= (long long)satLimit;
    if (mode == OPAL_MODE_SW && (canSaturate || !fits16) && pipe_resolve(db, st, false)) return OPAL_ERR_NO_SIMD_SUPPORT;

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
                a.first_item = headItems; a.num_items = 2 * P - headItems;
                if (debug_on())
                    fprintf(stderr, "[opal] streaming launch width=%d G=%d rows=%d end=%d items=[%d,+%d)\n", splan.width, splan.G, splan.rows, wantEnd,
                            a.first_item, a.num_items);
                a.side_first_item = 0; a.side_num_items = headItems;  // (the long head: class body, first workgroups of the launch)
                a.prio_items = stream_prio_items(db, a.first_item, a.num_items, splan.G);
                if (!gate_items(db, st, 0, 2 * P)) return OPAL_ERR_NO_SIMD_SUPPORT;
                if (pipe_resolve(db, st, splan.passes == 1)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (row chunks share db->boundary)
                if (splan.passes == 1 && db->pipe.active) a.started_counter = db->ovfCount.p + 3;
                if (splan.passes > 1) {
                    // a query beyond one wave: row chunks from the top, the first one partial (as launch_chunked cuts them), a
                    // chunk's bottom row through db->boundary (4 bytes per residue), the hand-out counter zeroed in between
                    if (!db->boundary.ensure(8 * ((size_t)db->residues + 16))) return OPAL_ERR_NO_SIMD_SUPPORT;
                    const int cap = splan.G * splan.rows;
                    const int firstRows = Q - (splan.passes - 1) * cap;
                    for (int ch = 0; ch < splan.passes && !rc; ch++) {
                        a.chunk_flags = (ch > 0 ? 1 : 0) | (ch < splan.passes - 1 ? 2 : 0);
                        a.chunk_row0 = ch == 0 ? 0 : firstRows + (ch - 1) * cap;
                        a.chunk_rows = ch == 0 ? firstRows : cap;
                        a.boundary = db->boundary.p;
                        if (debug_on()) fprintf(stderr, "[opal] streaming chunk %d/%d rows [%d,+%d)\n", ch + 1, splan.passes, a.chunk_row0, a.chunk_rows);
                        if (ch > 0 && opal_hip_memset32(db->ovfCount.p + 2, 0, 1, st)) return OPAL_ERR_NO_SIMD_SUPPORT;
                        rc = opal_hip_launch_stream(&a, db->ovfCount.p + 2) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
                        count(CNT_CHUNK_LAUNCHES);
                    }
                } else {
                    rc = opal_hip_launch_stream(&a, db->ovfCount.p + 2) ? OPAL_ERR_NO_SIMD_SUPPORT : 0;
                    if (!rc) pipe_launched_grid(db);
                }
                count(CNT_STREAM_SW);
            }
            else if (narrow == 16) rc = launch_range(c, 16, wantEnd, 0, nullptr, 0, nNarrow, pairLenOf, db->ovfCount.p);
            else {
                // A short query against a database with a real length tail: the few targets that would outlast the search on
                // the class kernels run with the roles swapped (swap_kernel.hpp) on a high-priority stream of their own,
                // submitted first; everything else as usual beside them.
                int swapRows = 0;
                int swapN = (narrow == 17 && !chunkedQuery) ? plan_swap(Q, A, mode, seqLenOf, 0, nNarrow, db->residues, &swapRows) : 0;
                if (wantEnd && swapRows > 16) swapRows = 16;  // (end cells: the 8- and 16-row shapes)
                if (wantEnd && swapN > 0 && opal_hip_swap_lds_bytes(swapRows, Q, A) > 160 * 1024) swapN = 0;
                void* joinSwap = nullptr;
                // (before the fork: the side stream inherits the wait. A swapped launch shares nothing with the search before this one --
                // its counters live in this search's header slot -- and its workgroups have started before the grid's do: the search
                // stays "one grid" for the search queued behind it)
                if (swapN > 0 && pipe_resolve(db, st, true)) return OPAL_ERR_NO_SIMD_SUPPORT;
                if (swapN > 0) {
                    void* fork = db->event(11);
                    void* side = db->swap();
                    OpalHipLaunch a;
                    fill_launch(a, c, 18, wantEnd, 0, nullptr, db->ovfCount.p);
                    a.rows_per_lane = swapRows; a.first_item = 0; a.num_items = swapN; a.stream = side;
                    a.started_counter = db->ovfCount.p + 3;  // (zeroed with the search header)
                    if (!fork || !side || opal_hip_event_record(fork, st) || opal_hip_stream_wait_event(side, fork) || !gate_items(db, side, 0, swapN) ||
                        opal_hip_launch_swap(&a) || !(joinSwap = db->event(12)) || opal_hip_event_record(joinSwap, side) ||
                        // the grids that follow on `main` reach the dispatcher when the swapped workgroups have their CUs
                        (!opts().noStartedWait && opal_hip_wait_started(db->ovfCount.p + 3, swapN, st))) {
                        db->quiesce();
                        return OPAL_ERR_NO_SIMD_SUPPORT;
                    }
                    count(CNT_SWAPPED, swapN);
                    db->pipe.blocks += swapN;  // (they count themselves in on the same word as the grid behind them)
                }
                rc = launch_range(c, narrow, wantEnd, 0, nullptr, swapN, nNarrow - swapN, seqLenOf, db->ovfCount.p);
                if (rc && swapN > 0) db->quiesce();  // (the swapped launch may still be writing results and the re-run list on its stream)
                if (!rc && joinSwap && opal_hip_stream_wait_event(st, joinSwap)) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
            }
            if (rc) return rc;
            int cnt = 0;
            // The re-run list can only be non-empty if some score can reach the rung's limit at all: with
            // min(Q, longest sequence) * max(matrix) below it (a 300-aa query on BLOSUM50: 4 500 against 32 752) the
            // count need not be read back, and the search has no host round trip between its launches and its end.
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
                const LenView rerunLenOf{db->pairLen.data(), 1, rerun.data()};
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
        // Long queries on large databases: the 16-bit part as a stream of sequences through persistent waves
        // (stream_kernel.hpp, diagonal-skewed kernels). The few sequences whose skewed values could leave 16 bits (the longest
        // ones) run on the unskewed class body in the first workgroups of the same launch; if they hold more than 2 % of the
        // residues the class kernels run everything.
        StreamPlan gplan;
        int q32 = 0;  // pairs [p32, q32): 16-bit, but not safe for the skewed stream kernels -> class kernels beside the stream
        auto plan_rungs = [&](bool queryBound) {
            auto hi_of = [&](long long L) { return queryBound ? hi_bound(L) : std::min<long long>(Q, L) * maxS; };
            p32 = 0;
            if (!fits16) p32 = P;
            else {
                while (p32 < P) {
                    const long long L = db->pairLen[2 * (size_t)p32];
                    const long long lo = 3LL * Go + (Q + L + 2) * (long long)Ge - minS;  // -minS: diag + P must not wrap either
                    const long long hi = hi_of(L);
                    if (lo < 32768 - 1 && hi < 32767) break;
                    p32++;
                }
            }
            gplan = StreamPlan();
            q32 = p32;
            if (fits16 && p32 < P) {
                long long moved = 0;
                auto skew_safe = [&](long long L) {
                    return hi_of(L) + (OPAL_HIP_MAX_ROWS + L + 64 + 4) * (long long)Ge < 32000;
                };
                while (q32 < P && !skew_safe(db->pairLen[2 * (size_t)q32])) { moved += db->pairLen[2 * (size_t)q32] + db->pairLen[2 * (size_t)q32 + 1]; q32++; }
                // (a query in row chunks: the unskewable sequences run on the class kernels' chunks on a side stream BESIDE the
                // streamed ones, not inside the launch -- they may hold a sixth of the residues before they become the longer side)
                // (round 5: a sixth also for unchunked queries -- the longest of them now go to the pair kernel, and the alternative to
                // a stream with a few hundred side workgroups ahead of it is the class kernels at 64 lanes for EVERYTHING: 81 ms
                // against ~50 for a 2000-aa query on the Swiss-Prot shape, whose unskewable targets hold 2.7 % of the residues)
                if (q32 < P && (moved * (chunkedQuery ? 6 : (opts().pair != 0 ? 16 : 50)) <= db->residues || opts().stream == 1) && !opts().noSkew) {
                    gplan = plan_stream(Q, A, mode, wantEnd, db->residues - moved, db->pairLen[2 * (size_t)q32], db->numSeqs - 2 * q32);
                }
                if (!gplan.G) q32 = p32;
            }
        };
        plan_rungs(true);
        // Class-kernel row chunks (a query beyond one wave on a database too small or too long-tailed to stream) keep the looser
        // bound: the 32-bit prefix it sets aside runs on a side stream BESIDE the chunks of the rest, and that overlap is worth more
        // there than the 16-bit rate (2600-aa HW x 20 000 sequences: 4.6 ms with the prefix beside them, 5.9 without; NW / OV 2-7 %).
        if (fits16 && qBound < (long long)Q * maxS && chunkedQuery && !gplan.G) {
            plan_rungs(false);
            c.qBound = (long long)Q * maxS;
        }
        int sideBlocks = 0;  // workgroups of the launches that run beside the main range on other streams
        int taken = 0;       // items [.., taken) have been handed to a pair launch (pair_kernel.hpp): every range below starts behind them
        auto launch_global_stream = [&]() -> int {
            OpalHipLaunch a;
            fill_launch(a, c, gplan.width, 0, 0, nullptr, db->ovfCount.p);
            a.lanes_per_seq = gplan.G; a.rows_per_lane = gplan.rows; a.skew = 1;
            a.first_item = std::max(2 * q32, taken); a.num_items = 2 * P - a.first_item;  // (taken: what a pair launch has, below)
            a.reserve_blocks = std::min(sideBlocks, 64);
            a.prio_items = stream_prio_items(db, a.first_item, a.num_items, gplan.G);
            if (gplan.passes > 1) {
                // a query beyond one wave: row chunks from the top (NW: the last one partial, HW/OV: the first), a chunk's bottom
                // row through the upper half of db->boundary (4 bytes per residue; the class-kernel chunks of the sequences that
                // need 32 bits or cannot be skewed run beside this on a side stream and index the lower half by THEIR residues)
                if (!gate_items(db, st, 0, 2 * P) || pipe_resolve(db, st, false)) return 1;
                if (!db->boundary.ensure(8 * ((size_t)db->residues + 16)) || !db->scratch.ensure(2 * (size_t)std::max(db->numSeqs, 1))) return 1;
                const int cap = gplan.G * gplan.rows;
                const int firstRows = Q - (gplan.passes - 1) * cap;
                count(CNT_STREAM_GLOBAL);
                for (int ch = 0; ch < gplan.passes; ch++) {
                    a.chunk_flags = (ch > 0 ? 1 : 0) | (ch < gplan.passes - 1 ? 2 : 0);
                    if (mode == OPAL_MODE_NW) { a.chunk_row0 = ch * cap; a.chunk_rows = std::min(cap, Q - a.chunk_row0); }
                    else { a.chunk_row0 = ch == 0 ? 0 : firstRows + (ch - 1) * cap; a.chunk_rows = ch == 0 ? firstRows : cap; }
                    a.boundary = db->boundary.p + 4 * ((size_t)db->residues + 16);
                    a.scratch = db->scratch.p;
                    if (debug_on())
                        fprintf(stderr, "[opal] streaming chunk %d/%d mode=%d G=%d rows=%d: query rows [%d,+%d) items=[%d,+%d)\n", ch + 1, gplan.passes,
                                mode, gplan.G, gplan.rows, a.chunk_row0, a.chunk_rows, a.first_item, a.num_items);
                    if (ch > 0 && opal_hip_memset32(db->ovfCount.p + 2, 0, 1, st)) return 1;
                    if (opal_hip_launch_stream(&a, db->ovfCount.p + 2)) return 1;
                    count(CNT_CHUNK_LAUNCHES);
                }
                return 0;
            }
            // the skew-unsafe sequences (the longest ones) ride along in the same launch on the unskewed class body
            a.side_first_item = std::max(2 * p32, taken); a.side_num_items = std::max(0, std::max(2 * q32, taken) - a.side_first_item);
            if (debug_on())
                fprintf(stderr, "[opal] streaming launch mode=%d width=%d G=%d rows=%d items=[%d,+%d)\n", mode, gplan.width, gplan.G, gplan.rows,
                        a.first_item, a.num_items);
            count(CNT_STREAM_GLOBAL);
            if (!gate_items(db, st, 0, 2 * P) || pipe_resolve(db, st, true)) return 1;
            if (db->pipe.active) a.started_counter = db->ovfCount.p + 3;
            if (opal_hip_launch_stream(&a, db->ovfCount.p + 2)) return 1;
            pipe_launched_grid(db);
            return 0;
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
        // The longest targets with the roles swapped (swap_kernel.hpp, opal_swap_gl: int32, the target on the rows of a wave): every
        // sequence the a-priori bounds send to the 32-bit rung -- a 35 213-residue target is 35 276 dependent steps of int32 rows
        // there, 8 ms against a 110-aa query on a search whose bulk takes 3 -- and the 16-bit ones behind them that would outlast
        // the search alone on the widest class. Items [nBad, nBad + swapG) on a high-priority stream of their own, submitted first;
        // the 32-bit rung then has nothing left, the 16-bit range starts behind them.
        // ONE PAIR OVER SEVERAL WAVES (pair_kernel.hpp): the targets the bounds send to the 32-bit rung, and the long 16-bit ones
        // behind them that would outlast the search on one wave, each on a workgroup of its own whose waves share the query's
        // rows -- a 35 213-residue target against a 2000-aa query is 22 ms of one wave's 32-row steps, 6 ms of eight waves'
        // 4-row steps. On the high-priority stream of the swapped launches (which then has nothing to do), submitted first.
        int pairN = 0;
        void* joinPair = nullptr;
        // (not beside class kernels on packed pairs, width 16: their items are PAIRS of sequences; beside a stream it is fine)
        if (!chunkedQuery && (narrow != 16 || gplan.G) && P > 0) {
            int rowsP = 0, wavesP = 0;
            pairN = plan_pair(Q, A, mode, seqLenOf, nBad, 2 * P - nBad, db->residues, 2 * p32 - nBad, fits16, &rowsP, &wavesP);
            while (pairN > 0 && db->pairSeq[(size_t)nBad + pairN - 1] < 0) pairN--;  // (the absent last item of an odd count)
            if (pairN > 0 && pairN < 2 * p32 - nBad) pairN = 0;
            if (pairN > 0 && pipe_resolve(db, st, true)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (before the fork; as for the swapped launches of SW)
            if (pairN > 0) {
                unsigned long long base = 0, words = 64;
                for (int i = 0; i < nBad; i++) base += (unsigned long long)db->pairLen[(size_t)i];
                for (int i = 0; i < pairN; i++) words += (unsigned long long)db->pairLen[(size_t)nBad + i];
                void* fork = db->event(11);
                void* side = db->swap();
                OpalHipLaunch a;
                fill_launch(a, c, 32, 0, 0, nullptr, db->ovfCount.p + 1);
                a.rows_per_lane = rowsP; a.waves_per_block = wavesP; a.first_item = nBad; a.num_items = pairN; a.stream = side;
                a.started_counter = db->ovfCount.p + 3;
                if (debug_on()) fprintf(stderr, "[opal] pair launch mode=%d: items [%d,+%d) on %d waves x %d rows per lane\n", mode, nBad, pairN, wavesP, rowsP);
                // (the rows the waves of a pair hand each other: one half of the buffer per stream of queued searches, which may overlap)
                const size_t bndWords = (size_t)(wavesP > 1 ? wavesP - 1 : 0) * (size_t)words;
                if (wavesP > 1 && db->pairBnd.would_realloc(2 * bndWords)) {
                    // (growing frees the old buffer: nothing of the search before may still be using it)
                    if (db->pipe.active && !db->pipe.resolved && pipe_resolve(db, st, false)) return OPAL_ERR_NO_SIMD_SUPPORT;
                    if (db->pipe.active) { db->pipe.untracked = true; if (db->pipeStream) opal_hip_stream_sync(db->pipeStream); if (db->mainStream) opal_hip_stream_sync(db->mainStream); }
                }
                if ((wavesP > 1 && !db->pairBnd.ensure(2 * bndWords, 1.25)) || !fork || !side || opal_hip_event_record(fork, st) ||
                    opal_hip_stream_wait_event(side, fork) || !gate_items(db, side, nBad, pairN) ||
                    opal_hip_launch_pair(&a, db->pairBnd.p + ((db->pipe.active && db->curStream == db->pipeStream) ? db->pairBnd.cap / 2 : 0), base, words) ||
                    !(joinPair = db->event(12)) || opal_hip_event_record(joinPair, side) ||
                    (!opts().noStartedWait && opal_hip_wait_started(db->ovfCount.p + 3, pairN, st))) {
                    db->quiesce();
                    return OPAL_ERR_NO_SIMD_SUPPORT;
                }
                count(CNT_PAIRED, pairN);
                db->pipe.blocks += pairN;
                taken = nBad + pairN;  // (no CUs reserved for them: the stream behind waits until they have started, and takes their CUs when they end)
            }
        }
        int swapG = 0;
        void* joinSwapG = nullptr;
        if (!chunkedQuery && !gplan.G && narrow != 16 && P > 0 && pairN == 0) {
            int rowsG = 0;
            const int planned = plan_swap(Q, A, mode, seqLenOf, nBad, 2 * P - nBad, db->residues, &rowsG, fits16);
            const int need32 = 2 * p32 - nBad;
            const int want = std::max(planned, (rowsG && opts().swap != 0) ? need32 : 0);
            if (rowsG && want > 0 && want <= 240 && want >= need32) swapG = std::min(want, 2 * P - nBad);
            while (swapG > 0 && db->pairSeq[(size_t)nBad + swapG - 1] < 0) swapG--;  // (the absent last item of an odd count)
            if (swapG > 0 && swapG < need32) swapG = 0;
            // (16 rows per lane -- half the steps -- measured for a handful of targets: 195-aa HW 6.5 -> 5.4 ms at best, but 9 ms in
            // every other run: a 103 KB workgroup finds room only on a CU with at most one bulk workgroup, i.e. only if it gets
            // there first; 54 KB fit beside three. 8 rows stay.)
            if (swapG > 0 && pipe_resolve(db, st, true)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (before the fork; as above)
            if (swapG > 0) {
                void* fork = db->event(11);
                void* side = db->swap();
                OpalHipLaunch a;
                fill_launch(a, c, 32, 0, 0, nullptr, db->ovfCount.p + 1);
                a.rows_per_lane = rowsG; a.first_item = nBad; a.num_items = swapG; a.stream = side;
                a.started_counter = db->ovfCount.p + 3;
                if (debug_on()) fprintf(stderr, "[opal] swapped launch mode=%d: items [%d,+%d) (%d of them beyond the 16-bit bounds), %d rows per lane\n", mode, nBad, swapG, need32, rowsG);
                if (!fork || !side || opal_hip_event_record(fork, st) || opal_hip_stream_wait_event(side, fork) || !gate_items(db, side, nBad, swapG) ||
                    opal_hip_launch_swap(&a) || !(joinSwapG = db->event(12)) || opal_hip_event_record(joinSwapG, side) ||
                    (!opts().noStartedWait && opal_hip_wait_started(db->ovfCount.p + 3, swapG, st))) {
                    db->quiesce();
                    return OPAL_ERR_NO_SIMD_SUPPORT;
                }
                count(CNT_SWAPPED, swapG);
                db->pipe.blocks += swapG;
            }
        }
        // (what the swapped launch took is gone from both rungs: the 32-bit range is empty, the 16-bit one starts behind it)
        const int first16 = std::max(swapG > 0 ? nBad + swapG : 2 * p32, taken);
        if (swapG > 0) { p32 = 0; q32 = 0; nBad = 0; }
        const int first32 = std::max(nBad, taken);  // the 32-bit prefix: what the pair launch has not taken (normally nothing then)
        // A query beyond one wave runs in row chunks, one launch after the other; its 32-bit prefix (a few long sequences: every
        // chunk is one long lone-wave tail, 2 x 2.4 ms of a 26 ms 3000-aa NW search) runs BESIDE the 16-bit chunks on a
        // high-priority side stream. The two ranges share nothing but the buffers they index by their own sequences.
        void* join32 = nullptr;
        SearchCtx c32 = c;
        c32.sideBlocks = &sideBlocks;
        // (round 4: any query. On a database with a real length tail -- Swiss-Prot shape, 110-aa query -- the ten targets beyond
        // 32 000 residues need 8 ms on the 32-bit kernels, and everything else waited behind them on the main stream: 14 ms for a
        // search whose bulk takes 2.8. Unchunked queries use the stream of the swapped launches, which the global modes never
        // need: the side streams 0..2 belong to the long classes of the 16-bit range.)
        // (not beside row chunks at width 16: those index the boundary buffer by the PAIRS' token streams, the 32-bit prefix by
        // its sequences' residues -- two index spaces in one buffer; only test hooks get there, a query in row chunks runs at width 17 / 18)
        if (P > q32 && (2 * p32 > first32 || (chunkedQuery && gplan.G && gplan.passes > 1 && q32 > p32)) && !opts().noSidePrefix &&
            !(chunkedQuery && narrow == 16)) {
            if (pipe_resolve(db, st, false)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (before the fork)
            void* fork = db->event(8);
            void* side = chunkedQuery ? db->stream(0) : db->swap();
            if (fork && side && !opal_hip_event_record(fork, st) && !opal_hip_stream_wait_event(side, fork)) { c32.stream = side; join32 = db->event(9); }
        }
        if (2 * p32 > first32 && launch_range(c32, 32, 0, 0, nullptr, first32, 2 * p32 - first32, seqLenOf, db->ovfCount.p + 1)) {
            db->quiesce();
            return OPAL_ERR_NO_SIMD_SUPPORT;
        }
        void* join16 = nullptr;
        if (gplan.G && gplan.passes > 1 && q32 > p32) {
            // (streamed row chunks: the 16-bit sequences that cannot be skewed take the class kernels' chunks -- on a second side
            // stream, beside the 32-bit ones: each of these launches is one lone-wave tail, 3.5-4 ms next to a stream, and four
            // of them in a row took as long as the three streamed chunks, 15 ms)
            SearchCtx c16 = c32;
            if (c32.stream && 2 * p32 > first32) {
                void* side2 = db->stream(1);
                if (side2 && !opal_hip_stream_wait_event(side2, db->event(8))) { c16.stream = side2; join16 = db->event(10); }
            }
            const int e = narrow == 16 ? launch_range(c16, 16, 0, 0, nullptr, p32, q32 - p32, pairLenOf, db->ovfCount.p)
                                       : launch_range(c16, narrow, 0, 0, nullptr, 2 * p32, 2 * (q32 - p32), seqLenOf, db->ovfCount.p);
            if (e || (join16 && opal_hip_event_record(join16, c16.stream))) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        }
        if (c32.stream && (!join32 || opal_hip_event_record(join32, c32.stream))) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        if (P > q32 && 2 * P > taken) {
            const int e = gplan.G ? launch_global_stream()
                        : narrow == 16 ? launch_range(c, 16, 0, 0, nullptr, p32, P - p32, pairLenOf, db->ovfCount.p)
                                       : launch_range(c, narrow, 0, 0, nullptr, first16, 2 * P - first16, seqLenOf, db->ovfCount.p);
            if (e) { db->quiesce(); return OPAL_ERR_NO_SIMD_SUPPORT; }
        }
        if ((c32.stream && opal_hip_stream_wait_event(st, join32)) || (join16 && opal_hip_stream_wait_event(st, join16)) ||
            (joinSwapG && opal_hip_stream_wait_event(st, joinSwapG)) || (joinPair && opal_hip_stream_wait_event(st, joinPair))) {
            db->quiesce();
            return OPAL_ERR_NO_SIMD_SUPPORT;
        }
    }
    if (scope.piped && !db->pipe.resolved && pipe_resolve(db, st, false)) return OPAL_ERR_NO_SIMD_SUPPORT;  // (nothing launched: the start event still wants recording)
    if (evStop) { opal_hip_event_record(evStop, st); db->asyncEvents = db->asyncTimed; }
    // (a caller that copies the results back on the same stream and waits for that copy needs no wait of its own here)
    if ((finalSync || kernelMs) && opal_hip_stream_sync(st)) return OPAL_ERR_NO_SIMD_SUPPORT;
    if (kernelMs && evStart && evStop) *kernelMs = opal_hip_event_elapsed_ms(evStart, evStop);
    if (db->clockProbe && !db->asyncTimed) {  // (measurement runs only: one more small copy and wait; queued searches: opalx_db_last_clock_ghz reads it)
        unsigned long long v[2] = {0, 0};
        if (!opal_hip_memcpy_d2h(v, db->hdrDev + 16, sizeof v, st) && !opal_hip_stream_sync(st) && v[1])
            db->lastClockGhz = (double)v[0] / (double)v[1] * opal_hip_wall_clock_hz() * 1e-9;
    }
    scope.ok = true;
    return rc;
}

}  // namespace opal_host
