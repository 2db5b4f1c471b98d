// opal_shim.hip -- the thin extern "C" HIP shim declared in include/opal_hip.h (part 1):
// device selection, memory, copies, events and the single kernel-launch entry. No DP logic and
// no scheduling here: the kernels are in dp_kernel.hpp, the scheduler in opal_api.cpp.
#include <hip/hip_runtime.h>
// RCCL: types only -- the library is loaded with dlopen, on request (opal_hip_rccl_init). A ROCm installation without the rccl
// development package still builds libopal.so: the few types the dlsym'ed entry points need are then declared here (the ABI of
// nccl.h: opaque communicator pointer, int enums), and a run-time librccl.so is all the RCCL gather wants.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2 } ncclDataType_t;
#endif
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "dp_kernel.hpp"
#include "stream_kernel.hpp"
#include "swap_kernel.hpp"
#include "pair_kernel.hpp"
#include "opal_hip.h"
#include "options.hpp"

static thread_local char g_err[512] = "";
static thread_local int g_lastBlocks = 0;  // workgroups of the last class / one-launch / streaming grid this thread launched (opal_hip_last_launch_blocks)

static int fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    fprintf(stderr, "[opal-hip] %s\n", g_err);
    return 2;
}

extern "C" const char* opal_hip_last_error(void) { return g_err; }

extern "C" int opal_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int opal_hip_set_device(int device) {
    hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? 0 : fail("hipSetDevice", e);
}

extern "C" void* opal_hip_malloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { fail("hipMalloc", e); return nullptr; }
    return p;
}

extern "C" void opal_hip_free(void* p) { if (p) (void)hipFree(p); }

extern "C" void* opal_hip_host_malloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { fail("hipHostMalloc", e); return nullptr; }
    return p;
}

// Pinned host memory the DEVICE caches (coarse-grained: coherent with the host at kernel boundaries). The residue staging
// buffer is of this kind: the DP kernels of a cold opalSearchDatabase call read it directly over PCIe (one pass over every
// byte: with the default fine-grained mapping every 8-byte batch of a slot would re-fetch its cache line).
extern "C" void* opal_hip_host_malloc_device_cached(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocNonCoherent);
    if (e != hipSuccess) { fail("hipHostMalloc(non-coherent)", e); return nullptr; }
    return p;
}

extern "C" void opal_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Ticks per second of s_memrealtime (the constant-rate counter the clock probe divides shader cycles by).
// Compute units of the calling thread's current device (256 on MI355X), cached per thread and device.
static int opal_hip_cu_count() {
    static thread_local int cus = 0, cusDevice = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (cusDevice != dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cusDevice = dev;
    }
    return cus;
}
extern "C" double opal_hip_wall_clock_hz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0.0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return 1e8;  // gfx9: 100 MHz
    return 1e3 * (double)khz;
}

extern "C" int opal_hip_get_device(void) {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

extern "C" int opal_hip_memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream) {
    hipError_t e = hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemcpyPeerAsync", e);
}

extern "C" int opal_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemcpyAsync(D2D)", e);
}

extern "C" int opal_hip_memset32(void* dst, int value, size_t count, void* stream) {
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)dst, value, count, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemsetD32Async", e);
}

extern "C" int opal_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemcpyAsync(H2D)", e);
}

extern "C" int opal_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemcpyAsync(D2H)", e);
}

extern "C" int opal_hip_memset(void* dst, int value, size_t bytes, void* stream) {
    hipError_t e = hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipMemsetAsync", e);
}

extern "C" int opal_hip_stream_sync(void* stream) {
    // OPAL_HIP_SPIN_SYNC=1 (experiment): poll instead of blocking -- the wake-up of a blocked host thread costs tens of
    // microseconds on a search that takes 1.5 ms
    if (opal_host::opts().spinSync) {
        hipError_t q;
        while ((q = hipStreamQuery((hipStream_t)stream)) == hipErrorNotReady) __builtin_ia32_pause();
        return q == hipSuccess ? 0 : fail("hipStreamQuery", q);
    }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipStreamSynchronize", e);
}

// Non-blocking streams: nothing is ordered against the legacy null stream; the scheduler orders its
// launches with fork/join events only.
extern "C" void* opal_hip_stream_create(void) {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return (void*)st;
}
// A stream of the highest priority the device offers (side launches of long sequences: dispatched ahead of the bulk
// launch, and on hardware queues of their own priority class).
extern "C" void* opal_hip_stream_create_high(void) {
    int least = 0, greatest = 0;
    hipStream_t st;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return opal_hip_stream_create();
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) != hipSuccess) return opal_hip_stream_create();
    return (void*)st;
}
// ... and of the LOWEST (the second stream of a handle's queued searches: a priority class of its own means a hardware queue that
// the handle's main stream, an ordinary one, can never share -- two streams of one class may land on the same queue, where a grid
// waits for the grid before it to END, whatever the streams say)
extern "C" void* opal_hip_stream_create_low(void) {
    int least = 0, greatest = 0;
    hipStream_t st;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) return opal_hip_stream_create();
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) return opal_hip_stream_create();
    return (void*)st;
}
extern "C" int opal_hip_stream_wait_event(void* stream, void* event) {
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    return e == hipSuccess ? 0 : fail("hipStreamWaitEvent", e);
}
extern "C" void opal_hip_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

extern "C" void* opal_hip_event_create(void) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return (void*)ev;
}
// An event used for ordering only (no timestamps: cheaper to record and to wait for).
extern "C" void* opal_hip_event_create_untimed(void) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)ev;
}
extern "C" void opal_hip_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
extern "C" int opal_hip_event_sync(void* ev) {
    hipError_t e = hipEventSynchronize((hipEvent_t)ev);
    return e == hipSuccess ? 0 : fail("hipEventSynchronize", e);
}
extern "C" int opal_hip_event_record(void* ev, void* stream) {
    hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail("hipEventRecord", e);
}
extern "C" float opal_hip_event_elapsed_ms(void* start, void* stop) {
    float ms = -1.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}

// One wave per pair: column c of the pair's token stream = residue of A | residue of B << 8.
__global__ void __launch_bounds__(256) opal_pack_tokens_kernel(const unsigned char* __restrict__ raw,
                                                               const uint64_t* __restrict__ rawOff,
                                                               const uint64_t* __restrict__ pairOff,
                                                               const int* __restrict__ pairLen, int numPairs,
                                                               uint16_t* __restrict__ tokens) {
    const int lane = threadIdx.x & 63;
    const int pair = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (pair >= numPairs) return;
    const int la = pairLen[2 * pair], lb = pairLen[2 * pair + 1];
    const unsigned char* a = raw + rawOff[2 * pair];
    const unsigned char* b = raw + rawOff[2 * pair + 1];
    uint16_t* t = tokens + pairOff[pair];
    for (int c = lane; c < la; c += 64) {
        unsigned v = a[c];
        if (c < lb) v |= (unsigned)b[c] << 8;
        t[c] = (uint16_t)v;
    }
}

extern "C" int opal_hip_pack_tokens(const unsigned char* raw, const uint64_t* raw_off, const uint64_t* pair_off,
                                    const int* pair_len, int num_pairs, uint16_t* tokens, uint64_t tokens_capacity,
                                    void* stream) {
    hipError_t e = hipMemsetAsync(tokens, 0, tokens_capacity * sizeof(uint16_t), (hipStream_t)stream);
    if (e != hipSuccess) return fail("hipMemsetAsync(tokens)", e);
    if (num_pairs <= 0) return 0;
    const int wpb = 4;
    hipLaunchKernelGGL(opal_pack_tokens_kernel, dim3((num_pairs + wpb - 1) / wpb), dim3(wpb * 64), 0, (hipStream_t)stream, raw,
                       raw_off, pair_off, pair_len, num_pairs, tokens);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("pack kernel launch", e);
}

// One wave per sequence: a database handed over as one flat device buffer (caller order) into item order.
__global__ void __launch_bounds__(256) opal_reorder_kernel(const unsigned char* __restrict__ src, const uint64_t* __restrict__ srcOff,
                                                           const uint64_t* __restrict__ dstOff, const int* __restrict__ len, int count,
                                                           unsigned char* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int i = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (i >= count) return;
    const unsigned char* a = src + srcOff[i];
    unsigned char* d = dst + dstOff[i];
    const int n = len[i];
    for (int c = lane; c < n; c += 64) d[c] = a[c];
}

extern "C" int opal_hip_reorder_residues(const unsigned char* src, const uint64_t* src_off, const uint64_t* dst_off, const int* len,
                                         int count, unsigned char* dst, void* stream) {
    if (count <= 0) return 0;
    const int wpb = 4;
    hipLaunchKernelGGL(opal_reorder_kernel, dim3((count + wpb - 1) / wpb), dim3(wpb * 64), 0, (hipStream_t)stream, src, src_off, dst_off,
                       len, count, dst);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("reorder kernel launch", e);
}

// Degenerate shapes (empty targets, empty query) are answered on the host: their values reach the
// caller-order result arrays through one small scatter instead of one copy per sequence.
__global__ void opal_scatter_results_kernel(const int* __restrict__ quad, int count, int* __restrict__ score,
                                            int* __restrict__ endQ, int* __restrict__ endT) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count) return;
    const int idx = quad[4 * i];
    score[idx] = quad[4 * i + 1];
    if (endQ) endQ[idx] = quad[4 * i + 2];
    if (endT) endT[idx] = quad[4 * i + 3];
}

extern "C" int opal_hip_scatter_results(const int* quads, int count, int* score, int* end_query, int* end_target, void* stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(opal_scatter_results_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, quads, count,
                       score, end_query, end_target);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("scatter kernel launch", e);
}

// out[perm[i]] = in[i]: the gathered per-shard result blocks of a multi-GPU search back into database order (root GPU).
__global__ void opal_permute_kernel(const int* __restrict__ in, const int* __restrict__ perm, int count, int* __restrict__ out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < count) out[perm[i]] = in[i];
}
extern "C" int opal_hip_permute(const int* in, const int* perm, int count, int* out, void* stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(opal_permute_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, perm, count, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("permute kernel launch", e);
}

// out[perm[k * per + j]] = in[k * block_stride + j] for k < blocks, j < per (perm < 0: padding): the blocks an RCCL gather of
// equal-sized (padded) shard blocks leaves on the root, back into database order.
__global__ void opal_permute_blocks_kernel(const int* __restrict__ in, const int* __restrict__ perm, int blocks, int per, int blockStride,
                                           int* __restrict__ out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= blocks * per) return;
    const int k = i / per, j = i - k * per;
    const int to = perm[i];
    if (to >= 0) out[to] = in[(size_t)k * (size_t)blockStride + (size_t)j];
}
extern "C" int opal_hip_permute_blocks(const int* in, const int* perm, int blocks, int per, int block_stride, int* out, void* stream) {
    if (blocks <= 0 || per <= 0) return 0;
    hipLaunchKernelGGL(opal_permute_blocks_kernel, dim3((blocks * per + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, perm, blocks, per,
                       block_stride, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("permute-blocks kernel launch", e);
}

// ---- the path's one collective through RCCL (ncclCommInitAll over the device list + one ncclGather per search).
// librccl.so is loaded on request only (dlopen): a process that never asks for it does not pay for loading it.
namespace {
struct RcclGroup {
    void* lib = nullptr;
    int n = 0;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    ncclResult_t (*commInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*groupStart)() = nullptr;
    ncclResult_t (*groupEnd)() = nullptr;
};
}  // namespace

extern "C" void opal_hip_rccl_destroy(void* h) {
    RcclGroup* g = (RcclGroup*)h;
    if (!g) return;
    for (ncclComm_t c : g->comms) if (c && g->commDestroy) g->commDestroy(c);
    // (the library stays loaded: unloading RCCL under a live HIP runtime is not worth the risk)
    delete g;
}

extern "C" void* opal_hip_rccl_init(const int* devices, int n) {
    if (n < 1) return nullptr;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (devices[a] == devices[b]) { snprintf(g_err, sizeof g_err, "RCCL: device %d listed twice (one rank per GPU)", devices[a]); return nullptr; }
    RcclGroup* g = new RcclGroup;
    g->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!g->lib) g->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!g->lib) { snprintf(g_err, sizeof g_err, "RCCL: %s", dlerror()); delete g; return nullptr; }
    g->commInitAll = (decltype(g->commInitAll))dlsym(g->lib, "ncclCommInitAll");
    g->commDestroy = (decltype(g->commDestroy))dlsym(g->lib, "ncclCommDestroy");
    g->gather = (decltype(g->gather))dlsym(g->lib, "ncclGather");
    g->groupStart = (decltype(g->groupStart))dlsym(g->lib, "ncclGroupStart");
    g->groupEnd = (decltype(g->groupEnd))dlsym(g->lib, "ncclGroupEnd");
    if (!g->commInitAll || !g->commDestroy || !g->gather || !g->groupStart || !g->groupEnd) { snprintf(g_err, sizeof g_err, "RCCL: symbols missing"); delete g; return nullptr; }
    g->n = n;
    g->devices.assign(devices, devices + n);
    g->comms.assign((size_t)n, nullptr);
    int cur = 0;
    (void)hipGetDevice(&cur);
    const ncclResult_t r = g->commInitAll(g->comms.data(), n, devices);
    (void)hipSetDevice(cur);
    if (r != ncclSuccess) { snprintf(g_err, sizeof g_err, "ncclCommInitAll failed (%d)", (int)r); g->comms.clear(); opal_hip_rccl_destroy(g); return nullptr; }
    return g;
}

// One gather of `count` ints per rank to rank 0: send[k] on device k (stream streams[k]), recv on device 0 ([n][count]).
extern "C" int opal_hip_rccl_gather(void* h, const void* const* send, void* recv_root, size_t count, void* const* streams) {
    RcclGroup* g = (RcclGroup*)h;
    if (!g) return 1;
    int cur = 0;
    (void)hipGetDevice(&cur);
    ncclResult_t r = g->groupStart();
    for (int k = 0; k < g->n && r == ncclSuccess; k++) {
        (void)hipSetDevice(g->devices[(size_t)k]);
        r = g->gather(send[k], recv_root, count, ncclInt32, 0, g->comms[(size_t)k], (hipStream_t)streams[k]);
    }
    const ncclResult_t e = g->groupEnd();
    (void)hipSetDevice(cur);
    if (r != ncclSuccess || e != ncclSuccess) { snprintf(g_err, sizeof g_err, "ncclGather failed (%d, %d)", (int)r, (int)e); return 1; }
    return 0;
}

// Let the calling thread's current device access `peer`'s memory directly (xGMI peer DMA for the gather of a multi-GPU
// search). Already enabled, or the same device: fine.
extern "C" int opal_hip_enable_peer(int peer) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    if (dev == peer) return 0;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) return 1;
    hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
    return e == hipSuccess ? 0 : fail("hipDeviceEnablePeerAccess", e);
}

// Kernels that need more than 64 KiB of dynamic LDS get the limit raised once per kernel and device (to the full 160 KiB):
// hipFuncSetAttribute is not free while other kernels are running (it delayed a launch beside a running one by ~2 ms).
static int raise_lds_limit(const void* fn) {
    static std::mutex m;
    static std::set<std::pair<int, const void*>> raised;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    std::lock_guard<std::mutex> lk(m);
    if (raised.count({dev, fn})) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { fail("hipFuncSetAttribute", e); return 1; }
    raised.insert({dev, fn});
    return 0;
}

static size_t lds_bytes(int width, int G, int rows, int A) {
    const int es = (width == 16 || width == 18) ? 2 : 4;  // widths 17 and 32 keep 4-byte profile entries
    return (size_t)(A + 1) * (size_t)opal_profile_stride(es, G, rows) + 48;  // (+48: probe / zero-copy slot, dp_kernel.hpp)
}

// SW with end location on the single-launch width-17 kernels: a column buffer per slot of the workgroup behind the profile
// (dp_kernel.hpp, CKPT: two rows per word + the lane number, whole 16-byte stores)
static size_t swend_ckpt_bytes(const OpalHipLaunch* a, int G, int rows, int wpb) {
    if (!(a->width == 17 && a->mode == MODE_SW && a->want_end && !a->chunk_flags)) return 0;
    return (size_t)(wpb * 64 / G) * (size_t)opal_swend_ckpt_words(rows) * 4;
}

extern "C" size_t opal_hip_swend_ckpt_bytes(int lanes_per_seq, int rows_per_lane, int waves_per_block) {
    if (lanes_per_seq < 1 || lanes_per_seq > 64 || rows_per_lane < 1 || waves_per_block < 1) return 0;
    return (size_t)(waves_per_block * 64 / lanes_per_seq) * (size_t)opal_swend_ckpt_words(rows_per_lane) * 4;
}

extern "C" size_t opal_hip_lds_bytes(int width, int lanes_per_seq, int rows_per_lane, int alphabet_length) {
    return lds_bytes(width, lanes_per_seq, rows_per_lane, alphabet_length);
}

static KernelFn pick_kernel(int width, int mode, int flags, int rows) {
    if (flags & FLAG_MULTI) {
        if (width == 16) return opal_kernel_w16_multi(mode, flags, rows);
        if (width == 17) return opal_kernel_s16(mode, flags, rows);
        if (width == 18) return opal_kernel_s18(mode, flags, rows);
        if (width == 32) return opal_kernel_w32_multi(mode, flags, rows);
        return nullptr;
    }
    if (width == 17) return opal_kernel_s16(mode, flags, rows);
    if (width == 18) return opal_kernel_s18(mode, flags, rows);
    if (width == 16) return opal_kernel_w16(mode, flags, rows);
    if (width == 32) return opal_kernel_w32(mode, flags, rows);
    return nullptr;
}

static int launch_flags(const OpalHipLaunch* a) {
    int flags = 0;
    if (a->mode == MODE_SW && a->want_end) flags |= FLAG_SWEND;
    if (a->mode == MODE_SW && a->width == 32 && a->detect_wrap) flags |= FLAG_WRAP;
    if (a->chunk_flags) flags |= FLAG_MULTI;
    if (a->skew && a->width == 17 && a->mode != MODE_SW && !a->chunk_flags) flags |= FLAG_SKEW;
    return flags;
}

extern "C" int opal_hip_rows_supported(int width, int mode, int want_end, int min_rows) {
    OpalHipLaunch a;
    memset(&a, 0, sizeof a);
    a.width = width; a.mode = mode; a.want_end = want_end & 1;
    a.chunk_flags = (want_end & 2) ? 1 : 0;  /* bit1 of want_end: ask about the multi-chunk variants */
    a.skew = (want_end & 4) ? 1 : 0;         /* bit2: the diagonal-skewed variants */
    for (int r = min_rows < 1 ? 1 : min_rows; r <= OPAL_HIP_MAX_ROWS; r++)
        if (pick_kernel(width, mode, launch_flags(&a), r)) return r;
    return 0;
}

// Registers per lane of the kernel that `a` would launch (from the code object, hipFuncGetAttributes):
// the scheduler derives waves per SIMD from it (512 registers per lane per SIMD, granule 8).
extern "C" int opal_hip_kernel_vgprs(const OpalHipLaunch* a) {
    KernelFn fn = pick_kernel(a->width, a->mode, launch_flags(a), a->rows_per_lane);
    if (!fn) return 0;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, (const void*)fn) != hipSuccess) return 0;
    return attr.numRegs;
}

extern "C" int opal_hip_max_waves_per_block(int width) { return opal_max_waves_per_block(width); }

static UberFn pick_uber(int mode, int flags, int rows) {
    if (flags & FLAG_MULTI) return nullptr;
    if (mode == MODE_SW) return opal_uber_sw(flags, rows);
    if (flags == FLAG_SKEW) return opal_uber_skew(mode, rows);
    return nullptr;
}

extern "C" int opal_hip_uber_supported(int mode, int want_end, int skew, int rows_per_lane) {
    OpalHipLaunch a;
    memset(&a, 0, sizeof a);
    a.width = 17; a.mode = mode; a.want_end = want_end & 1; a.skew = skew;
    if (want_end & 16) return mode == MODE_SW && !(want_end & 1) && pick_uber(mode, FLAG_ZC, rows_per_lane) != nullptr;  /* bit 4: the zero-copy variants */
    return pick_uber(mode, launch_flags(&a), rows_per_lane) != nullptr;
}

extern "C" int opal_hip_launch_uber(const OpalHipUberLaunch* u) {
    g_lastBlocks = 0;
    const OpalHipLaunch* a = &u->base;
    const int G0 = a->lanes_per_seq, S0 = a->rows_per_lane;
    int log2G0 = 0;
    while ((1 << log2G0) < G0) log2G0++;
    UberFn fn = nullptr;
    if (a->width == 17 && !a->chunk_flags && !a->work && (1 << log2G0) == G0 && S0 >= 1 && S0 <= OPAL_HIP_MAX_ROWS &&
        (long long)G0 * S0 >= a->query_length && a->query_length >= 1 && a->mode >= 0 && a->mode <= 3)
        fn = pick_uber(a->mode, launch_flags(a) | (a->residues_out ? FLAG_ZC : 0), S0);  /* (zero copy: SW score shapes only, else no kernel) */
    const int items[3] = {u->longest_num_items, u->long_num_items, a->num_items};
    const int firsts[3] = {u->longest_first_item, u->long_first_item, a->first_item};
    const int prios[3] = {u->longest_priority, u->long_priority, a->priority};
    const int lanes[3] = {4 * G0, 2 * G0, G0};
    const int rows[3] = {(S0 + 3) / 4, (S0 + 1) / 2, S0};
    for (int k = 0; k < 3; k++)
        if (items[k] > 0 && lanes[k] > 64) fn = nullptr;
    if (!fn) {
        snprintf(g_err, sizeof g_err, "unlaunchable one-launch shape: mode=%d G0=%d rows=%d Q=%d", a->mode, G0, S0, a->query_length);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    UberParams p;
    memset(&p, 0, sizeof p);
    KParams& b = p.base;
    b.query = a->query; b.matrix = a->score_matrix; b.Q = a->query_length; b.A = a->alphabet_length;
    b.Go = a->gap_open; b.Ge = a->gap_ext;
    b.res = a->residues; b.seqOff = a->seq_offset; b.resOut = a->residues_out;
    b.tokens = a->tokens; b.pairOff = a->pair_token_offset; b.pairLen = a->pair_len; b.pairSeq = a->pair_seq;
    b.work = nullptr;
    b.outScore = a->out_score; b.outEndQ = a->out_end_query; b.outEndT = a->out_end_target;
    b.ovfCount = a->overflow_count; b.ovfItems = a->overflow_items;
    b.chunkRows = a->query_length;
    b.satLimit = a->sat_limit; b.negPad = a->neg_pad;
    b.clk = a->clock_probe; b.tl = a->timeline; b.started = a->started_counter;
    size_t lds = 0;
    int blocks = 0;
    for (int k = 0; k < 3; k++) {
        p.firstBlock[k] = blocks;
        int lg = 0;
        while ((1 << lg) < lanes[k]) lg++;
        p.cls[k] = KClass{lg, firsts[k], items[k] > 0 ? items[k] : 0, prios[k]};
        if (items[k] <= 0) continue;
        const int slots = 64 / lanes[k];
        const int jobs = (items[k] + slots - 1) / slots;
        blocks += (jobs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        const size_t need = lds_bytes(17, lanes[k], rows[k], a->alphabet_length) + swend_ckpt_bytes(a, lanes[k], rows[k], WAVES_PER_BLOCK);
        if (need > lds) lds = need;
    }
    if (blocks == 0) return 0;
    if (a->lds_bytes_min > 0 && (size_t)a->lds_bytes_min > lds && a->lds_bytes_min <= 160 * 1024) lds = (size_t)a->lds_bytes_min;
    if (lds > 160 * 1024) { snprintf(g_err, sizeof g_err, "query profile needs %zu bytes of LDS (>160 KiB)", lds); return 2; }
    if (lds > 64 * 1024 && raise_lds_limit((const void*)fn)) return 1;
    g_lastBlocks = blocks;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(WAVES_PER_BLOCK * 64), lds, (hipStream_t)a->stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("one-launch kernel launch", e);
}

extern "C" int opal_hip_launch(const OpalHipLaunch* a) {
    g_lastBlocks = 0;  // (a call that launches nothing has started nothing: opal_hip_last_launch_blocks)
    const int G = a->lanes_per_seq;
    int log2G = 0;
    while ((1 << log2G) < G) log2G++;
    if (a->mode < 0 || a->mode > 3) { snprintf(g_err, sizeof g_err, "bad mode %d", a->mode); return 3; }
    KernelFn fn = nullptr;
    if ((1 << log2G) == G && G <= 64 && a->rows_per_lane >= 1 && a->rows_per_lane <= OPAL_HIP_MAX_ROWS &&
        (long long)G * a->rows_per_lane >= (a->chunk_flags ? a->chunk_rows : a->query_length) && a->query_length >= 1 &&
        (!a->chunk_flags || (a->boundary && a->chunk_rows >= 1 && a->chunk_row0 >= 0 &&
                             a->chunk_row0 + a->chunk_rows <= a->query_length)))
        fn = pick_kernel(a->width, a->mode, launch_flags(a), a->rows_per_lane);
    if (!fn) {
        snprintf(g_err, sizeof g_err, "unlaunchable shape: width=%d mode=%d G=%d rows=%d Q=%d", a->width, a->mode, G,
                 a->rows_per_lane, a->query_length);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    if (a->num_items <= 0) return 0;
    KParams p;
    memset(&p, 0, sizeof p);
    p.query = a->query; p.matrix = a->score_matrix; p.Q = a->query_length; p.A = a->alphabet_length;
    p.Go = a->gap_open; p.Ge = a->gap_ext;
    p.res = a->residues; p.seqOff = a->seq_offset; p.resOut = nullptr;  /* (zero copy: one-launch kernels only) */
    p.tokens = a->tokens; p.pairOff = a->pair_token_offset; p.pairLen = a->pair_len; p.pairSeq = a->pair_seq;
    p.work = a->work; p.firstItem = a->first_item; p.numItems = a->num_items; p.log2G = log2G;
    p.outScore = a->out_score; p.outEndQ = a->out_end_query; p.outEndT = a->out_end_target;
    p.ovfCount = a->overflow_count; p.ovfItems = a->overflow_items;
    p.prio = a->priority;
    p.q0 = a->chunk_flags ? a->chunk_row0 : 0;
    p.chunkRows = a->chunk_flags ? a->chunk_rows : a->query_length;
    p.chunkFlags = a->chunk_flags;
    p.bnd = (uint2*)a->boundary;
    p.scratch = a->scratch;
    p.satLimit = a->sat_limit;
    p.negPad = a->neg_pad;
    p.clk = a->clock_probe; p.tl = a->timeline; p.started = a->started_counter;
    const int slots = 64 / G;
    const int jobs = (a->num_items + slots - 1) / slots;
    const int maxWpb = opal_max_waves_per_block(a->width);
    int wpb = (a->waves_per_block >= 1 && a->waves_per_block <= maxWpb) ? a->waves_per_block : WAVES_PER_BLOCK;
    if (opal_host::opts().wavesPerBlock >= 1 && opal_host::opts().wavesPerBlock <= maxWpb) wpb = opal_host::opts().wavesPerBlock;
    size_t lds = lds_bytes(a->width, G, a->rows_per_lane, a->alphabet_length) + swend_ckpt_bytes(a, G, a->rows_per_lane, wpb);
    if (lds > 160 * 1024 && wpb > 4) {  // (the plan priced the column buffer of four waves)
        wpb = 4;
        lds = lds_bytes(a->width, G, a->rows_per_lane, a->alphabet_length) + swend_ckpt_bytes(a, G, a->rows_per_lane, wpb);
    }
    const int blocks = (jobs + wpb - 1) / wpb;
    if (a->lds_bytes_min > 0 && (size_t)a->lds_bytes_min > lds && a->lds_bytes_min <= 160 * 1024) lds = (size_t)a->lds_bytes_min;
    if (lds > 160 * 1024) {
        snprintf(g_err, sizeof g_err, "query profile needs %zu bytes of LDS (>160 KiB)", lds);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    if (lds > 64 * 1024 && raise_lds_limit((const void*)fn)) return 1;
    g_lastBlocks = blocks;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(wpb * 64), lds, (hipStream_t)a->stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("kernel launch", e);
}

// ------------------------------------------------------------------ streaming SW score search (stream_kernel.hpp)
static StreamFn pick_stream(int width, int mode, int want_end, int rows, int chunked = 0) {
    if (chunked) {
        if (width != 17) return nullptr;
        if (mode != MODE_SW) return opal_stream_gl17chunk(mode, rows);
        return want_end ? opal_stream_s17chunkend(rows) : opal_stream_s17chunk(rows);
    }
    if (mode != MODE_SW) {
        if (width == 17) return opal_stream_gl17(mode, rows);
        if (width == 18) return opal_stream_gl18(mode, rows);
        return nullptr;
    }
    if (width == 17) return want_end ? opal_stream_s17end(rows) : opal_stream_s17(rows);
    if (width == 18) return want_end ? opal_stream_s18end(rows) : opal_stream_s18(rows);
    return nullptr;
}
static size_t stream_lds_bytes(int width, int mode, int G, int rows, int A) {
    if (mode != MODE_SW) return width == 18 ? opal_gl_stream_lds_bytes<18>(G, rows, A) : opal_gl_stream_lds_bytes<17>(G, rows, A);
    return width == 18 ? opal_stream_lds_bytes<18>(G, rows, A) : opal_stream_lds_bytes<17>(G, rows, A);
}
extern "C" int opal_hip_stream_rows_supported(int width, int mode, int want_end, int min_rows) {
    /* bit1 of want_end: ask about the row-chunk variants (as opal_hip_rows_supported does) */
    for (int r = min_rows < 1 ? 1 : min_rows; r <= OPAL_HIP_MAX_ROWS; r++)
        if (pick_stream(width, mode, want_end & 1, r, want_end & 2)) return r;
    return 0;
}
extern "C" size_t opal_hip_stream_lds_bytes(int width, int mode, int lanes_per_seq, int rows_per_lane, int alphabet_length) {
    return stream_lds_bytes(width, mode, lanes_per_seq, rows_per_lane, alphabet_length);
}
extern "C" int opal_hip_launch_stream(const OpalHipLaunch* a, int* next_counter) {
    g_lastBlocks = 0;
    const int G = a->lanes_per_seq;
    int log2G = 0;
    while ((1 << log2G) < G) log2G++;
    StreamFn fn = nullptr;
    if ((1 << log2G) == G && G <= 64 && a->mode >= 0 && a->mode <= 3 && !a->work && next_counter && !(a->chunk_flags && a->side_num_items > 0) &&
        (!a->want_end || (a->out_end_query && a->out_end_target)) &&
        a->query_length >= 1 && (long long)G * a->rows_per_lane >= (a->chunk_flags ? a->chunk_rows : a->query_length) &&
        (!a->chunk_flags || (a->boundary && a->chunk_rows >= 1 && a->chunk_row0 >= 0 && a->chunk_row0 + a->chunk_rows <= a->query_length &&
                             /* a partial chunk is the first one (SW/HW/OV: padded above) or the last one (NW: padded below) */
                             (a->chunk_rows == G * a->rows_per_lane || (a->chunk_flags & (a->mode == MODE_NW ? 2 : 1)) == 0))))
        fn = (a->mode == MODE_SW && a->side_num_items > 0) ? (a->width == 17 && !a->chunk_flags ? opal_stream_s17head(a->rows_per_lane, a->want_end) : nullptr)
                                                           : pick_stream(a->width, a->mode, a->want_end, a->rows_per_lane, a->chunk_flags);
    if (!fn) {
        snprintf(g_err, sizeof g_err, "unlaunchable streaming shape: width=%d G=%d rows=%d Q=%d", a->width, G, a->rows_per_lane, a->query_length);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    if (a->num_items <= 0) return 0;
    const size_t lds = stream_lds_bytes(a->width, a->mode, G, a->rows_per_lane, a->alphabet_length);
    if (lds > 160 * 1024) { snprintf(g_err, sizeof g_err, "query profile needs %zu bytes of LDS (>160 KiB)", lds); return 2; }
    const int cus = opal_hip_cu_count();
    StreamParams sp;
    memset(&sp, 0, sizeof sp);
    KParams& p = sp.base;
    p.query = a->query; p.matrix = a->score_matrix; p.Q = a->query_length; p.A = a->alphabet_length;
    p.Go = a->gap_open; p.Ge = a->gap_ext;
    p.res = a->residues; p.seqOff = a->seq_offset;
    p.tokens = a->tokens; p.pairOff = a->pair_token_offset; p.pairLen = a->pair_len; p.pairSeq = a->pair_seq;
    p.firstItem = a->first_item; p.numItems = a->num_items; p.log2G = log2G;
    p.outScore = a->out_score; p.outEndQ = a->out_end_query; p.outEndT = a->out_end_target;  /* (SW score only: NULL) */
    p.ovfCount = a->overflow_count; p.ovfItems = a->overflow_items;
    p.q0 = a->chunk_flags ? a->chunk_row0 : 0;
    p.chunkRows = a->chunk_flags ? a->chunk_rows : a->query_length;
    p.chunkFlags = a->chunk_flags;
    p.bnd = (uint2*)a->boundary;
    p.scratch = a->scratch;
    p.satLimit = a->sat_limit; p.negPad = a->neg_pad;
    p.clk = a->clock_probe; p.tl = a->timeline; p.started = a->started_counter;
    sp.next = next_counter;
    sp.prioItems = a->prio_items > 0 ? a->prio_items : 0;
    // persistent workgroups: one per CU (16 waves = 4 per SIMD), fewer when there are not enough sequences to go round
    const int perBlock = STREAM_WAVES * (64 / G);
    int blocks = (a->num_items + perBlock - 1) / perBlock;
    if (blocks > cus) blocks = cus;
    size_t ldsLaunch = lds;
    if (a->mode == MODE_SW && a->side_num_items > 0) {
        /* the head of a long-tailed database on the class body, 4G lanes x ceil(rows/4) rows, first in the grid */
        if (4 * G > 64) { snprintf(g_err, sizeof g_err, "side items: SW streams at 16 lanes per sequence or fewer only"); return 2; }
        const int perSide = STREAM_SIDE_WAVES * (64 / (4 * G));
        sp.side = KClass{log2G + 2, a->side_first_item, a->side_num_items, 0};
        sp.sideBlocks = (a->side_num_items + perSide - 1) / perSide;
        const size_t need = lds_bytes(17, 4 * G, (a->rows_per_lane + 3) / 4, a->alphabet_length);
        if (need > ldsLaunch) ldsLaunch = need;
        if (ldsLaunch > 160 * 1024) { snprintf(g_err, sizeof g_err, "side items: profile needs %zu bytes of LDS", ldsLaunch); return 2; }
    }
    if (a->mode != MODE_SW && a->side_num_items > 0) {
        // a handful of side sequences: one wave per SIMD (they are the critical path); hundreds of them (a long query on a database
        // with a real length tail): every wave of their workgroups works, and the stream starts beside them on the other CUs
        sp.sideWaves = a->side_num_items * G / 64 > cus ? STREAM_WAVES : STREAM_SIDE_WAVES;
        const int perSide = sp.sideWaves * (64 / G);
        sp.side = KClass{log2G, a->side_first_item, a->side_num_items, 0};
        sp.sideBlocks = (a->side_num_items + perSide - 1) / perSide;
        // (every workgroup resident from the start -- but never fewer than half the CUs for the stream: the side workgroups come
        // first in the grid and leave, the persistent ones take their CUs as they do. max(1, ...) here once left a 2000-aa search of
        // the Swiss-Prot shape with ONE streaming workgroup: 10.9 s)
        if (blocks + sp.sideBlocks > cus && blocks > 1) blocks = std::max(std::min(blocks, cus / 2), cus - sp.sideBlocks);
    }
    if (a->reserve_blocks > 0 && blocks + sp.sideBlocks + a->reserve_blocks > cus)  // CUs for what runs beside this launch on other streams
        blocks = std::max(std::min(blocks, cus / 2), cus - sp.sideBlocks - a->reserve_blocks);
    if (opal_host::opts().streamBlocks >= 1) blocks = opal_host::opts().streamBlocks;  // test hook
    if (ldsLaunch > 64 * 1024 && raise_lds_limit((const void*)fn)) return 1;
    g_lastBlocks = blocks + sp.sideBlocks;
    hipLaunchKernelGGL(fn, dim3(blocks + sp.sideBlocks), dim3(STREAM_WAVES * 64), ldsLaunch, (hipStream_t)a->stream, sp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("streaming kernel launch", e);
}

// ------------------------------------------------------------------ one pair over several waves (pair_kernel.hpp)
// Rows per lane and waves per workgroup for a query: the cheapest column step among the instantiated row counts whose profile
// fits LDS (a step is ~9 instructions per row + ~30; up to two waves per SIMD run at the lone-wave rate, four share the pipe).
extern "C" int opal_hip_pair_shape(int query_length, int alphabet_length, int* rows, int* waves) {
    double bestCost = 0;
    int bestS = 0, bestW = 0;
    for (int S : {2, 4, 8}) {
        const int W = (query_length + 64 * S - 1) / (64 * S);
        if (W < 1 || W > PAIR_MAX_WAVES || opal_pair_lds_bytes(W, S, alphabet_length) > 160 * 1024) continue;
        const double cost = (9.0 * S + 40.0) * (W <= 8 ? 5.2 : 8.6);
        if (!bestS || cost < bestCost) { bestCost = cost; bestS = S; bestW = W; }
    }
    if (rows) *rows = bestS;
    if (waves) *waves = bestW;
    return bestS ? 1 : 0;
}

extern "C" int opal_hip_launch_pair(const OpalHipLaunch* a, void* boundary, unsigned long long bnd_base, unsigned long long bnd_stride) {
    const int S = a->rows_per_lane, W = a->waves_per_block;
    PairFn fn = (W >= 1 && W <= PAIR_MAX_WAVES && (long long)W * 64 * S >= a->query_length && a->query_length >= 1 && !a->work &&
                 (W == 1 || boundary) && (!a->want_end || (a->out_end_query && a->out_end_target)))
                    ? opal_pair_fn(a->mode, S) : nullptr;
    if (!fn) {
        snprintf(g_err, sizeof g_err, "unlaunchable pair shape: mode=%d rows=%d waves=%d Q=%d", a->mode, S, W, a->query_length);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    if (a->num_items <= 0) return 0;
    const size_t lds = opal_pair_lds_bytes(W, S, a->alphabet_length);
    if (lds > 160 * 1024) { snprintf(g_err, sizeof g_err, "query profile needs %zu bytes of LDS (>160 KiB)", lds); return 2; }
    PairParams p;
    memset(&p, 0, sizeof p);
    p.query = a->query; p.matrix = a->score_matrix; p.Q = a->query_length; p.A = a->alphabet_length; p.Go = a->gap_open; p.Ge = a->gap_ext;
    p.res = a->residues; p.seqOff = a->seq_offset; p.pairLen = a->pair_len; p.pairSeq = a->pair_seq;
    p.firstItem = a->first_item; p.numItems = a->num_items;
    p.outScore = a->out_score; p.outEndQ = a->out_end_query; p.outEndT = a->out_end_target;
    p.bnd = (unsigned long long*)boundary; p.bndBase = bnd_base; p.bndStride = bnd_stride;
    p.started = a->started_counter;
    p.clk = a->clock_probe;
    if (lds > 64 * 1024 && raise_lds_limit((const void*)fn)) return 1;
    hipLaunchKernelGGL(fn, dim3(a->num_items), dim3(W * 64), lds, (hipStream_t)a->stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("pair kernel launch", e);
}

// One wave that ends when `target` workgroups of a launch on another stream have started (see include/opal_hip.h).
__global__ void __launch_bounds__(64) opal_wait_started_kernel(const int* counter, int target, unsigned long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > timeout_ticks) break;  // never a deadlock, at worst the old order
        __builtin_amdgcn_s_sleep(32);
    }
}
extern "C" int opal_hip_wait_started(const int* counter, int target, void* stream) {
    if (!counter || target <= 0) return 0;
    // (no more workgroups than can be resident at once -- one per CU for the 55-100 KB workgroups this is for: on a part with fewer
    // CUs than workgroups the rest start when the first ones end, and waiting for THEM would cost every search the whole timeout;
    // 2 ms of the constant-rate counter, from its measured rate)
    const int cus = opal_hip_cu_count();
    if (cus > 0 && target > cus) target = cus;
    const unsigned long long ticks = (unsigned long long)(0.002 * opal_hip_wall_clock_hz());
    hipLaunchKernelGGL(opal_wait_started_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, target, ticks);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("wait kernel launch", e);
}

// One wave that ends when `target` workgroups have counted themselves in on `counter` -- the "started" word of ANOTHER search's header
// slot (zeroed with that header; every workgroup of its grid, and of the swapped / pair launches in front of it, adds one when it
// starts: KParams::started). What a queued search puts in front of its grid on its own stream so that the grid reaches the
// dispatcher when the search before it (on the other stream) has no workgroup left to dispatch: it then fills the CUs that one
// leaves as it drains. Gives up after `timeout_ticks` of the constant-rate counter (never a deadlock: at worst the two grids share
// the dispatcher from then on).
__device__ int g_wait_timeouts = 0;  // waits that gave up (opal_hip_wait_timeouts: tests assert that none did)
__global__ void __launch_bounds__(64) opal_wait_count_kernel(const int* counter, int target, unsigned long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > timeout_ticks) { atomicAdd(&g_wait_timeouts, 1); break; }
        __builtin_amdgcn_s_sleep(32);
    }
}
extern "C" int opal_hip_wait_timeouts(void) {  // (synchronizes the device)
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wait_timeouts), sizeof v, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}
extern "C" int opal_hip_wait_count(const int* counter, int target, void* stream) {
    if (!counter || target <= 0) return 0;
    const unsigned long long ticks = (unsigned long long)(0.25 * opal_hip_wall_clock_hz());  // a quarter of a second
    hipLaunchKernelGGL(opal_wait_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, target, ticks);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("wait kernel launch", e);
}
extern "C" int opal_hip_last_launch_blocks(void) { return g_lastBlocks; }

// ------------------------------------------------------------------ role-swapped SW score search (swap_kernel.hpp)
// Rows per lane: a chunk profile of 64 x rows residues for every query letter. The workgroup shares its CU with the
// search's other launches (whose workgroups need the LDS it leaves free: with 40 rows -- 133 KB -- every swapped target
// blocked a whole CU for the bulk launch, 20 % of the chip for most of a Swiss-Prot-shaped search), and its own per-step
// overhead is small, so the default is the largest shape within 64 KB (16 rows for a protein alphabet: 55 KB).
extern "C" int opal_hip_swap_rows(int query_length, int alphabet_length) {
    if (query_length < 1 || query_length > OPAL_HIP_SWAP_MAX_QUERY || alphabet_length < 1) return 0;
    for (int rows : {40, 24, 16, 8})
        if (opal_swap_lds(rows, query_length, alphabet_length) <= 64 * 1024) return rows;
    return opal_swap_lds(8, query_length, alphabet_length) <= 160 * 1024 ? 8 : 0;
}
extern "C" size_t opal_hip_swap_lds_bytes(int rows_per_lane, int query_length, int alphabet_length) {
    return opal_swap_lds(rows_per_lane, query_length, alphabet_length);
}
// 16 rows per lane hold 2-byte profile entries (the caller has checked that the matrix fits 16 bits), 8 rows 4-byte ones: the
// same LDS (54 KB for a protein alphabet).
extern "C" int opal_hip_swap_gl_rows(int query_length, int alphabet_length, int matrix_fits_16) {
    if (query_length < 1 || query_length > OPAL_HIP_SWAP_MAX_QUERY || alphabet_length < 1) return 0;
    if (matrix_fits_16 && opal_swap_gl_lds(16, query_length, alphabet_length) <= 64 * 1024) return 16;
    if (opal_swap_gl_lds(8, query_length, alphabet_length) <= 160 * 1024) return 8;
    return 0;
}
extern "C" size_t opal_hip_swap_gl_lds_bytes(int rows_per_lane, int query_length, int alphabet_length) {
    return opal_swap_gl_lds(rows_per_lane, query_length, alphabet_length);
}
extern "C" int opal_hip_launch_swap(const OpalHipLaunch* a) {
    const bool gl = a->mode != MODE_SW;
    SwapFn fn = gl ? opal_swap_gl_kernel(a->mode, a->rows_per_lane) : opal_swap_kernel(a->rows_per_lane, a->want_end);
    const size_t lds = gl ? opal_swap_gl_lds(a->rows_per_lane, a->query_length, a->alphabet_length)
                          : opal_swap_lds(a->rows_per_lane, a->query_length, a->alphabet_length);
    if (!fn || a->mode < 0 || a->mode > 3 || (a->want_end && (!a->out_end_query || !a->out_end_target)) || a->work || a->query_length < 1 || a->query_length > OPAL_HIP_SWAP_MAX_QUERY || lds > 160 * 1024) {
        snprintf(g_err, sizeof g_err, "unlaunchable swapped shape: rows=%d Q=%d A=%d", a->rows_per_lane, a->query_length, a->alphabet_length);
        fprintf(stderr, "[opal-hip] %s\n", g_err);
        return 2;
    }
    if (a->num_items <= 0) return 0;
    SwapParams p;
    memset(&p, 0, sizeof p);
    p.query = a->query; p.matrix = a->score_matrix; p.Q = a->query_length; p.A = a->alphabet_length;
    p.Go = a->gap_open; p.Ge = a->gap_ext;
    p.res = a->residues; p.seqOff = a->seq_offset; p.pairLen = a->pair_len; p.pairSeq = a->pair_seq;
    p.firstItem = a->first_item; p.numItems = a->num_items;
    p.outScore = a->out_score; p.outEndQ = a->out_end_query; p.outEndT = a->out_end_target;
    p.ovfCount = a->overflow_count; p.ovfItems = a->overflow_items;
    p.satLimit = a->sat_limit; p.negPad = a->neg_pad;
    p.clk = a->clock_probe;
    p.started = a->started_counter;
    if (lds > 64 * 1024 && raise_lds_limit((const void*)fn)) return 1;
    hipLaunchKernelGGL(fn, dim3(a->num_items), dim3(SWAP_WAVES * 64), lds, (hipStream_t)a->stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("swapped kernel launch", e);
}
