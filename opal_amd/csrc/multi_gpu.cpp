// multi_gpu.cpp -- the devices of a call and the deal of sequences to them (the reference's natural shard
// point is its per-chunk call, reference src/opal_aligner.cpp:125-160: sequences are independent, query and matrix are
// broadcast constants).
#include "opal_internal.hpp"

namespace opal_host {

// ------------------------------------------------------------------------------------ devices of a call
// OPAL_HIP_DEVICES = "all" | "0,1,2,..." makes opalSearchDatabase shard every call over those GPUs. A device
// may be listed more than once (two contexts on one GPU): that is how the sharding is tested on a one-GPU box.
std::vector<int> call_devices() {
    std::vector<int> out;
    const char* env = getenv("OPAL_HIP_DEVICES");
    if (!env || !*env) return out;
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
