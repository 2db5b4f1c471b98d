// order_kernel.hip -- the length ordering of a database on the device: a STABLE counting sort of the sequence lengths,
// longest first, that emits the item arrays the DP kernels index (caller index, length and byte offset of every item).
//
// The reference walks its database in caller order and refills SIMD lanes one by one (reference src/opal.cpp:472-490
// loadNextSequence); the wave geometry here wants the sequences sorted by length. On the host that ordering
// was the critical path of a cold opalSearchDatabase call (0.5 ms of one thread before the first residue could be
// gathered, on a call whose DP takes 1.5 ms); here it is four small launches behind one 400 KB upload:
//
//   count   block b histograms the lengths of its caller range [b*per, (b+1)*per) in LDS -> cnt[b][len]
//   prefix  per length: cnt[b][len] becomes the number of sequences of that length in the blocks BEFORE b; total[len]
//   scan    one workgroup: lengths in descending order -> first item position and first byte offset of every length
//   place   one wave per caller range, 64 sequences per step in caller order: a sequence's position is
//           base[len] + cnt[b][len] + (sequences of that length seen earlier in this range) + (rank among the lanes of
//           this step holding the same length: match-any by ballots) -- stable by construction, no atomics on positions.
//
// All sequences of one length are neighbours in item order, so the byte offset of an item follows from its length's first
// offset and its rank alone: no second scan over the items.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

#include "opal_hip.h"

namespace {

constexpr int ORDER_MAX_KEYS = 40000;  // one LDS counter per length: 160 000 B of the CU's 160 KiB
constexpr int ORDER_MAX_BLOCKS = 256;

__global__ void __launch_bounds__(256) opal_order_count(const int* __restrict__ lens, int n, int per, int B, int* __restrict__ cnt) {
    extern __shared__ int hist[];
    for (int k = threadIdx.x; k < B; k += 256) hist[k] = 0;
    __syncthreads();
    const int i0 = (int)blockIdx.x * per, i1 = min(n, i0 + per);
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) atomicAdd(&hist[lens[i]], 1);
    __syncthreads();
    int* out = cnt + (size_t)blockIdx.x * (size_t)B;
    for (int k = threadIdx.x; k < B; k += 256) out[k] = hist[k];
}

__global__ void __launch_bounds__(256) opal_order_prefix(int* __restrict__ cnt, int nblk, int B, int* __restrict__ total) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= B) return;
    int run = 0;
    for (int b = 0; b < nblk; b++) {
        int* p = cnt + (size_t)b * (size_t)B + k;
        const int c = *p;
        *p = run;
        run += c;
    }
    total[k] = run;
}

// Lengths in descending order (slot j holds length B-1-j): exclusive scan of the counts (item positions) and of
// count * length (byte offsets). Thread t owns a contiguous segment of slots.
__global__ void __launch_bounds__(1024) opal_order_scan(const int* __restrict__ total, int B, int* __restrict__ base,
                                                        unsigned long long* __restrict__ byteBase) {
    __shared__ unsigned long long sItems[1024], sBytes[1024];
    const int t = (int)threadIdx.x;
    const int seg = (B + 1023) / 1024;
    const int j0 = t * seg, j1 = min(B, j0 + seg);
    unsigned long long items = 0, bytes = 0;
    for (int j = j0; j < j1; j++) {
        const int len = B - 1 - j;
        const unsigned long long c = (unsigned long long)total[len];
        items += c;
        bytes += c * (unsigned long long)len;
    }
    sItems[t] = items; sBytes[t] = bytes;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan over the 1024 segment sums
        unsigned long long a = 0, b = 0;
        if (t >= d) { a = sItems[t - d]; b = sBytes[t - d]; }
        __syncthreads();
        sItems[t] += a; sBytes[t] += b;
        __syncthreads();
    }
    unsigned long long atItem = sItems[t] - items, atByte = sBytes[t] - bytes;
    for (int j = j0; j < j1; j++) {
        const int len = B - 1 - j;
        base[len] = (int)atItem;
        byteBase[len] = atByte;
        const unsigned long long c = (unsigned long long)total[len];
        atItem += c;
        atByte += c * (unsigned long long)len;
    }
}

__global__ void __launch_bounds__(64) opal_order_place(const int* __restrict__ lens, int n, int per, int B, const int* __restrict__ cnt,
                                                       const int* __restrict__ base, const unsigned long long* __restrict__ byteBase,
                                                       int nonEmpty, int* __restrict__ itemSeq, int* __restrict__ itemLen,
                                                       unsigned long long* __restrict__ itemOff, int* __restrict__ emptySeqs) {
    extern __shared__ int cur[];  // sequences of each length placed so far by this caller range
    const int lane = (int)threadIdx.x;
    for (int k = lane; k < B; k += 64) cur[k] = 0;
    __syncthreads();
    const int i0 = (int)blockIdx.x * per, i1 = min(n, i0 + per);
    const int* mine = cnt + (size_t)blockIdx.x * (size_t)B;
    const int nbits = 32 - __clz(max(B - 1, 1));
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int s = i0; s < i1; s += 64) {
        const int i = s + lane;
        const bool act = i < i1;
        const int key = act ? lens[i] : 0;
        // the lanes of this step that hold the same length as mine
        unsigned long long same = __ballot(act);
        for (int bit = 0; bit < nbits; bit++) {
            const bool on = (key >> bit) & 1;
            const unsigned long long m = __ballot(on);
            same &= on ? m : ~m;
        }
        const int rank = __popcll(same & below);
        int seen = 0;
        if (act) seen = cur[key];
        __syncthreads();  // (one wave: every lane has read its counter before the first of a group moves it on)
        if (act && rank == 0) cur[key] = seen + __popcll(same);
        __syncthreads();
        if (act) {
            const int within = mine[key] + seen + rank;  // my place among ALL sequences of this length, caller order
            if (key == 0) emptySeqs[within] = i;
            else {
                const int pos = base[key] + within;
                itemSeq[pos] = i;
                itemLen[pos] = key;
                itemOff[pos] = byteBase[key] + (unsigned long long)within * (unsigned long long)key;
            }
        }
    }
    // an odd number of non-empty sequences: the absent partner of the last one (items come in pairs, dp_kernel.hpp)
    if (blockIdx.x == 0 && lane == 0 && (nonEmpty & 1)) {
        itemSeq[nonEmpty] = -1;
        itemLen[nonEmpty] = 0;
        itemOff[nonEmpty] = byteBase[0];
    }
}

// Where the upload chunks and gather pieces of the fill begin (db_pack.cpp): for every byte offset asked for, the first item
// at or behind it and that item's own offset -- a binary search over the item offsets just written, here rather than on the host,
// whose copy of that array would be freshly DMA-written memory: every probe a cache miss (44 us for 272 searches, measured).
__global__ void opal_order_cuts(const unsigned long long* __restrict__ targets, int T, const unsigned long long* __restrict__ itemOff,
                                int nonEmpty, unsigned long long total, unsigned long long* __restrict__ out) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= T) return;
    const unsigned long long want = targets[t];
    int lo = 0, hi = nonEmpty;  // first item in [0, nonEmpty) whose offset is >= want
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (itemOff[mid] < want) lo = mid + 1; else hi = mid;
    }
    out[2 * t] = (unsigned long long)lo;
    out[2 * t + 1] = lo < nonEmpty ? itemOff[lo] : total;
}

// (more than 64 KiB of dynamic LDS: the limit is raised once per kernel and device, as in opal_shim.hip)
int raise_lds_limit(const void* fn) {
    static std::mutex m;
    static std::set<std::pair<int, const void*>> raised;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    std::lock_guard<std::mutex> lk(m);
    if (raised.count({dev, fn})) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    raised.insert({dev, fn});
    return 0;
}

int order_blocks(int n, int B) {
    int nblk = (n + 2047) / 2048;
    if (nblk > ORDER_MAX_BLOCKS) nblk = ORDER_MAX_BLOCKS;
    while (nblk > 1 && (size_t)nblk * (size_t)B * sizeof(int) > ((size_t)64 << 20)) nblk /= 2;  // cnt[][] within 64 MiB
    return nblk < 1 ? 1 : nblk;
}

}  // namespace

extern "C" int opal_hip_order_max_length(void) { return ORDER_MAX_KEYS - 1; }

extern "C" size_t opal_hip_order_scratch_bytes(int n, int max_len) {
    if (n <= 0 || max_len < 0 || max_len >= ORDER_MAX_KEYS) return 0;
    const size_t B = (size_t)max_len + 1;
    return ((size_t)order_blocks(n, (int)B) * B + 2 * B) * sizeof(int) + B * sizeof(unsigned long long) + 16;
}

extern "C" int opal_hip_order_cuts(const uint64_t* d_targets, int num_targets, const uint64_t* d_item_off, int num_nonempty,
                                   uint64_t total_bytes, uint64_t* d_out, void* stream) {
    if (num_targets <= 0) return 0;
    hipLaunchKernelGGL(opal_order_cuts, dim3((num_targets + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(d_targets), num_targets, reinterpret_cast<const unsigned long long*>(d_item_off),
                       num_nonempty, (unsigned long long)total_bytes, reinterpret_cast<unsigned long long*>(d_out));
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int opal_hip_order_by_length(const int* d_lens, int n, int max_len, int num_nonempty, void* d_scratch, int* d_item_seq,
                                        int* d_item_len, uint64_t* d_item_off, int* d_empty_seqs, void* stream) {
    if (n <= 0) return 0;
    if (max_len < 0 || max_len >= ORDER_MAX_KEYS || !d_scratch) return 1;
    const int B = max_len + 1;
    const int nblk = order_blocks(n, B);
    const int per = (n + nblk - 1) / nblk;
    int* cnt = (int*)d_scratch;
    int* total = cnt + (size_t)nblk * (size_t)B;
    int* base = total + B;
    unsigned long long* byteBase = (unsigned long long*)(((uintptr_t)(base + B) + 7) & ~(uintptr_t)7);
    const size_t lds = (size_t)B * sizeof(int);
    if (lds > 48 * 1024 && (raise_lds_limit(reinterpret_cast<const void*>(opal_order_count)) || raise_lds_limit(reinterpret_cast<const void*>(opal_order_place))))
        return 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(opal_order_count, dim3(nblk), dim3(256), lds, st, d_lens, n, per, B, cnt);
    hipLaunchKernelGGL(opal_order_prefix, dim3((B + 255) / 256), dim3(256), 0, st, cnt, nblk, B, total);
    hipLaunchKernelGGL(opal_order_scan, dim3(1), dim3(1024), 0, st, total, B, base, byteBase);
    hipLaunchKernelGGL(opal_order_place, dim3(nblk), dim3(64), lds, st, d_lens, n, per, B, cnt, base, byteBase, num_nonempty, d_item_seq,
                       d_item_len, reinterpret_cast<unsigned long long*>(d_item_off), d_empty_seqs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
