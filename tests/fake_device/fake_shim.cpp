// fake_shim.cpp -- TEST INFRASTRUCTURE ONLY. A stand-in for the extern "C" HIP shim (include/opal_hip.h, part 1) that runs on
// host threads, so that ThreadSanitizer can watch the library's HOST scheduler -- the thread pool, the gather beside the
// search, chunk gates, the speculative search of a re-seen database, the queue of asynchronous searches, the per-GPU slot
// workers of the multi-GPU paths -- which is concurrent code that no GPU test can check for data races.
//
// It is linked ONLY into tests/fake_device/host_stress (never into libopal.so, never built by __graft_entry__.build(), never
// loaded by opal_amd): the product has no CPU path and fails loudly without a gfx950 device.
//
//   * a "stream" is a worker thread with a FIFO of closures; copies, memsets and kernel launches are closures, so they really
//     run asynchronously to the calling thread, in stream order; events carry the ordering between streams and to the host;
//   * "device memory" is host memory;
//   * a kernel launch computes its items with the scalar oracle (oracle/opal_oracle.c -- this file lives under tests/, which
//     may call the checker), with the real kernels' contract: results by caller index, SW scores at the 16-bit rung's limit
//     go to the re-run list instead, row-chunked queries are answered by their last chunk;
//   * the planner's questions (instantiated shapes, LDS bytes, registers) get plausible fixed answers.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "opal.h"
#include "opal_hip.h"

extern "C" int oracle_search(const unsigned char* query, int Q, const unsigned char* const* db, int N, const int* lens, int gapOpen,
                             int gapExt, const int* M, int A, int searchType, int mode, int* score, int* endQ, int* endT);

namespace {

struct Stream {
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m);
                busy = false;
                if (q.empty()) idle.notify_all();
            }
        }
    }
    void push(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); }
        cv.notify_one();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(m);
        idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};

Stream* null_stream() { static Stream* s = new Stream; return s; }
Stream* S(void* p) { return p ? (Stream*)p : null_stream(); }

struct Event {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;  // records issued / records the streams have reached
    std::chrono::steady_clock::time_point at;
};

thread_local int t_device = 0;

// ---------------------------------------------------------------- the "kernels"
struct Range { int first, count; };

void run_items(const OpalHipLaunch& a, Range r, bool useWork) {
    if (a.chunk_flags & 2) return;  // a query in row chunks: the last chunk answers
    const bool pairs = a.width == 16;
    for (int i = 0; i < r.count; i++) {
        const int item = useWork && a.work ? a.work[i] : r.first + i;
        for (int half = 0; half < (pairs ? 2 : 1); half++) {
            const int it = pairs ? 2 * item + half : item;
            const int seq = a.pair_seq[it], len = a.pair_len[it];
            if (seq < 0 || len <= 0) continue;
            const unsigned char* t = a.residues + a.seq_offset[it];
            if (a.residues_out) memcpy(a.residues_out + a.seq_offset[it], t, (size_t)len);  // (the zero-copy kernels write the device copy)
            int s = 0, eq = -1, et = -1;
            const int rc = oracle_search(a.query, a.query_length, &t, 1, &len, a.gap_open, a.gap_ext, a.score_matrix, a.alphabet_length,
                                         a.want_end || a.out_end_query ? 1 : 0, a.mode, &s, &eq, &et);
            if (rc) continue;  // (beyond the oracle's 32-bit bounds: not computed, as the library does)
            if (a.mode == OPAL_MODE_SW && a.width != 32 && s >= (a.sat_limit ? a.sat_limit : 32767)) {
                const int at = __atomic_fetch_add(a.overflow_count, 1, __ATOMIC_RELAXED);
                a.overflow_items[at] = it;
                continue;
            }
            a.out_score[seq] = s;
            if (a.out_end_query) a.out_end_query[seq] = eq;
            if (a.out_end_target) a.out_end_target[seq] = et;
        }
    }
}

int enqueue(void* stream, std::function<void()> f) { S(stream)->push(std::move(f)); return 0; }

}  // namespace

extern "C" {

const char* opal_hip_last_error(void) { return "fake device"; }
int opal_hip_device_count(void) { const char* e = getenv("FAKE_DEVICES"); return e ? atoi(e) : 1; }
int opal_hip_set_device(int device) { if (device < 0 || device >= opal_hip_device_count()) return 1; t_device = device; return 0; }
int opal_hip_get_device(void) { return t_device; }
double opal_hip_wall_clock_hz(void) { return 1e8; }
void* opal_hip_malloc(size_t bytes) { return aligned_alloc(64, (bytes + 63) / 64 * 64); }
void opal_hip_free(void* p) { free(p); }
void* opal_hip_host_malloc(size_t bytes) { return aligned_alloc(64, (bytes + 63) / 64 * 64); }
void* opal_hip_host_malloc_device_cached(size_t bytes) { return opal_hip_host_malloc(bytes); }
void opal_hip_host_free(void* p) { free(p); }
int opal_hip_enable_peer(int) { return 0; }

int opal_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) { return enqueue(stream, [=] { memcpy(dst, src, bytes); }); }
int opal_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) { return enqueue(stream, [=] { memcpy(dst, src, bytes); }); }
int opal_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) { return enqueue(stream, [=] { memcpy(dst, src, bytes); }); }
int opal_hip_memcpy_peer(void* dst, int, const void* src, int, size_t bytes, void* stream) { return enqueue(stream, [=] { memcpy(dst, src, bytes); }); }
int opal_hip_memset(void* dst, int value, size_t bytes, void* stream) { return enqueue(stream, [=] { memset(dst, value, bytes); }); }
int opal_hip_memset32(void* dst, int value, size_t count, void* stream) {
    return enqueue(stream, [=] { int* p = (int*)dst; for (size_t i = 0; i < count; i++) p[i] = value; });
}
int opal_hip_permute(const int* in, const int* perm, int count, int* out, void* stream) {
    return enqueue(stream, [=] { for (int i = 0; i < count; i++) out[perm[i]] = in[i]; });
}
int opal_hip_permute_blocks(const int* in, const int* perm, int blocks, int per, int stride, int* out, void* stream) {
    return enqueue(stream, [=] {
        for (int k = 0; k < blocks; k++)
            for (int j = 0; j < per; j++)
                if (perm[k * per + j] >= 0) out[perm[k * per + j]] = in[(size_t)k * stride + j];
    });
}
void* opal_hip_rccl_init(const int*, int) { return nullptr; }  // (no RCCL on the fake device: the handle keeps its peer copies)
void opal_hip_rccl_destroy(void*) {}
int opal_hip_rccl_gather(void*, const void* const*, void*, size_t, void* const*) { return 1; }
int opal_hip_scatter_results(const int* quads, int count, int* score, int* eq, int* et, void* stream) {
    return enqueue(stream, [=] {
        for (int i = 0; i < count; i++) {
            const int idx = quads[4 * i];
            score[idx] = quads[4 * i + 1];
            if (eq) eq[idx] = quads[4 * i + 2];
            if (et) et[idx] = quads[4 * i + 3];
        }
    });
}
int opal_hip_stream_sync(void* stream) { S(stream)->sync(); return 0; }
void* opal_hip_stream_create(void) { return new Stream; }
void* opal_hip_stream_create_high(void) { return new Stream; }
void* opal_hip_stream_create_low(void) { return new Stream; }
void opal_hip_stream_destroy(void* stream) { if (stream) delete (Stream*)stream; }

void* opal_hip_event_create(void) { return new Event; }
void* opal_hip_event_create_untimed(void) { return new Event; }
void opal_hip_event_destroy(void* ev) { delete (Event*)ev; }
int opal_hip_event_record(void* ev, void* stream) {
    Event* e = (Event*)ev;
    if (!e) return 1;
    { std::lock_guard<std::mutex> lk(e->m); e->recorded++; }
    return enqueue(stream, [e] {
        { std::lock_guard<std::mutex> lk(e->m); e->completed++; e->at = std::chrono::steady_clock::now(); }
        e->cv.notify_all();
    });
}
int opal_hip_event_sync(void* ev) {
    Event* e = (Event*)ev;
    if (!e) return 1;
    std::unique_lock<std::mutex> lk(e->m);
    const uint64_t want = e->recorded;
    e->cv.wait(lk, [&] { return e->completed >= want; });
    return 0;
}
int opal_hip_stream_wait_event(void* stream, void* ev) {
    Event* e = (Event*)ev;
    if (!e) return 1;
    uint64_t want;
    { std::lock_guard<std::mutex> lk(e->m); want = e->recorded; }  // (an event never recorded: nothing to wait for, as in HIP)
    return enqueue(stream, [e, want] {
        std::unique_lock<std::mutex> lk(e->m);
        e->cv.wait(lk, [&] { return e->completed >= want; });
    });
}
float opal_hip_event_elapsed_ms(void* start, void* stop) {
    if (!start || !stop) return -1.f;
    opal_hip_event_sync(stop);
    Event* a = (Event*)start; Event* b = (Event*)stop;
    std::chrono::steady_clock::time_point ta, tb;
    { std::lock_guard<std::mutex> lk(a->m); ta = a->at; }
    { std::lock_guard<std::mutex> lk(b->m); tb = b->at; }
    return std::chrono::duration<float, std::milli>(tb - ta).count();
}

int opal_hip_pack_tokens(const unsigned char*, const uint64_t*, const uint64_t*, const int*, int, uint16_t*, uint64_t, void*) { return 0; }  // (the fake kernels read the residues)
int opal_hip_reorder_residues(const unsigned char* src, const uint64_t* so, const uint64_t* dof, const int* len, int count, unsigned char* dst, void* stream) {
    return enqueue(stream, [=] { for (int i = 0; i < count; i++) memcpy(dst + dof[i], src + so[i], (size_t)len[i]); });
}

int opal_hip_order_max_length(void) { return 39999; }
size_t opal_hip_order_scratch_bytes(int n, int max_len) { return (n > 0 && max_len >= 0 && max_len < 40000) ? 64 : 0; }
int opal_hip_order_by_length(const int* lens, int n, int max_len, int nonEmpty, void* scratch, int* seq, int* len, uint64_t* off, int* empties, void* stream) {
    if (n <= 0) return 0;
    if (max_len < 0 || max_len >= 40000 || !scratch) return 1;
    return enqueue(stream, [=] {
        std::vector<int> idx;
        int ne = 0;
        for (int i = 0; i < n; i++) { if (lens[i] > 0) idx.push_back(i); else empties[ne++] = i; }
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lens[a] > lens[b]; });
        uint64_t at = 0;
        for (size_t k = 0; k < idx.size(); k++) { seq[k] = idx[k]; len[k] = lens[idx[k]]; off[k] = at; at += (uint64_t)lens[idx[k]]; }
        if (nonEmpty & 1) { seq[nonEmpty] = -1; len[nonEmpty] = 0; off[nonEmpty] = at; }
    });
}

int opal_hip_order_cuts(const uint64_t* targets, int T, const uint64_t* off, int nonEmpty, uint64_t total, uint64_t* out, void* stream) {
    return enqueue(stream, [=] {
        for (int t = 0; t < T; t++) {
            const int it = (int)(std::lower_bound(off, off + nonEmpty, targets[t]) - off);
            out[2 * t] = (uint64_t)it;
            out[2 * t + 1] = it < nonEmpty ? off[it] : total;
        }
    });
}

// ---- the planner's questions
int opal_hip_rows_supported(int, int, int, int min_rows) { return min_rows >= 1 && min_rows <= OPAL_HIP_MAX_ROWS ? min_rows : 0; }
int opal_hip_max_waves_per_block(int width) { return width == 18 ? 16 : 8; }
int opal_hip_kernel_vgprs(const OpalHipLaunch*) { return 120; }
size_t opal_hip_lds_bytes(int width, int G, int rows, int A) { return (size_t)(A + 1) * (size_t)G * (size_t)rows * ((width == 16 || width == 18) ? 2 : 4) + 1024; }
size_t opal_hip_swend_ckpt_bytes(int G, int rows, int wpb) { return (size_t)wpb * (64 / G) * ((size_t)G * rows * 2 + 16); }
int opal_hip_uber_supported(int, int want_end, int, int rows) { return rows >= 1 && rows <= ((want_end & 16) ? OPAL_ZERO_COPY_MAX_ROWS : OPAL_HIP_MAX_ROWS); }
int opal_hip_stream_rows_supported(int, int, int, int min_rows) { return min_rows >= 1 && min_rows <= OPAL_HIP_MAX_ROWS ? min_rows : 0; }
size_t opal_hip_stream_lds_bytes(int width, int, int G, int rows, int A) { return opal_hip_lds_bytes(width, G, rows, A) + 8192; }
int opal_hip_swap_rows(int Q, int) { return Q <= 1024 ? 16 : 0; }
size_t opal_hip_swap_lds_bytes(int rows, int Q, int A) { return (size_t)(A + 1) * 64 * (size_t)rows * 2 + 8 * (size_t)Q + 4096; }
int opal_hip_swap_gl_rows(int Q, int, int fits16) { return Q <= 1024 ? (fits16 ? 16 : 8) : 0; }
size_t opal_hip_swap_gl_lds_bytes(int rows, int Q, int A) { return (size_t)(A + 1) * 64 * (size_t)rows * (rows == 16 ? 2 : 4) + 12 * (size_t)Q + 4096; }
int opal_hip_rev_rows_supported(int) { return 0; }    // start cells and traceback: the host's own passes
int opal_hip_rev_launch(const OpalHipRevLaunch*) { return 1; }
int opal_hip_trace_rows_supported(int) { return 0; }
int opal_hip_trace_launch(const OpalHipTraceLaunch*) { return 1; }

// ---- launches
int opal_hip_launch(const OpalHipLaunch* a) {
    const OpalHipLaunch c = *a;
    return enqueue(c.stream, [c] { run_items(c, Range{c.first_item, c.num_items}, true); });
}
int opal_hip_launch_uber(const OpalHipUberLaunch* u) {
    const OpalHipUberLaunch c = *u;
    return enqueue(c.base.stream, [c] {
        run_items(c.base, Range{c.longest_first_item, c.longest_num_items}, false);
        run_items(c.base, Range{c.long_first_item, c.long_num_items}, false);
        run_items(c.base, Range{c.base.first_item, c.base.num_items}, false);
    });
}
int opal_hip_launch_stream(const OpalHipLaunch* a, int*) {
    const OpalHipLaunch c = *a;
    return enqueue(c.stream, [c] {
        run_items(c, Range{c.side_first_item, c.side_num_items}, false);
        run_items(c, Range{c.first_item, c.num_items}, false);
    });
}
int opal_hip_pair_shape(int Q, int, int* rows, int* waves) {
    for (int S : {2, 4, 8}) {
        const int W = (Q + 64 * S - 1) / (64 * S);
        if (W >= 1 && W <= 16) { if (rows) *rows = S; if (waves) *waves = W; return 1; }
    }
    return 0;
}
int opal_hip_launch_pair(const OpalHipLaunch* a, void*, unsigned long long, unsigned long long) {
    OpalHipLaunch c = *a;
    c.width = 32;
    return enqueue(c.stream, [c] { run_items(c, Range{c.first_item, c.num_items}, false); });
}
int opal_hip_wait_started(const int*, int, void*) { return 0; }  // (the fake streams have no dispatcher to race for)
// queued searches on two streams: no wait at all here, so that consecutive one-grid searches really run at the same time on
// their two worker threads -- whatever they share unprotected, ThreadSanitizer sees
int opal_hip_wait_count(const int*, int, void*) { return 0; }
int opal_hip_last_launch_blocks(void) { return 1; }
int opal_hip_wait_timeouts(void) { return 0; }
int opal_hip_launch_swap(const OpalHipLaunch* a) {
    OpalHipLaunch c = *a;
    if (c.mode == OPAL_MODE_SW) c.width = 17;  // (16-bit arithmetic with the re-run list, like the class kernels)
    return enqueue(c.stream, [c] { run_items(c, Range{c.first_item, c.num_items}, false); });
}

}  // extern "C"
