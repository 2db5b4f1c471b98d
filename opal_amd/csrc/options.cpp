// options.cpp -- the option table behind options.hpp: one typed struct, parsed once from the environment, changed afterwards
// only through opalx_set_option (include/opal_hip.h).
#include "options.hpp"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>

extern char** environ;

namespace opal_host {

namespace {

typedef std::function<bool(Options&, const char*)> Setter;  // value never NULL here; false = does not parse
typedef std::function<void(Options&, const Options&)> Resetter;  // copy the option's field(s) from a default-constructed struct
struct Desc { const char* name; Setter set; Resetter reset; };

bool parse_ll(const char* v, long long lo, long long hi, long long* out) {
    if (!v || !*v) return false;
    errno = 0;
    char* end = nullptr;
    const long long x = strtoll(v, &end, 10);
    if (errno || end == v || *end || x < lo || x > hi) return false;
    *out = x;
    return true;
}
bool is_on(const char* v) { return v && *v && strcmp(v, "0") != 0; }

template <typename T> Resetter back(T Options::*f) { return [f](Options& o, const Options& d) { o.*f = d.*f; }; }
template <typename T, typename U> Resetter back2(T Options::*f, U Options::*g) { return [f, g](Options& o, const Options& d) { o.*f = d.*f; o.*g = d.*g; }; }

Desc flag(const char* name, bool Options::*f) { return {name, [f](Options& o, const char* v) { o.*f = is_on(v); return true; }, back(f)}; }
Desc integer(const char* name, int Options::*f, long long lo, long long hi) {
    return {name, [f, lo, hi](Options& o, const char* v) { long long x; if (!parse_ll(v, lo, hi, &x)) return false; o.*f = (int)x; return true; }, back(f)};
}
Desc one_of(const char* name, int Options::*f, std::vector<int> allowed) {
    return {name, [f, allowed](Options& o, const char* v) {
        long long x;
        if (!parse_ll(v, -(1LL << 31), (1LL << 31) - 1, &x)) return false;
        for (int a : allowed) if (a == (int)x) { o.*f = a; return true; }
        return false;
    }, back(f)};
}
Desc big(const char* name, long long Options::*f) {
    return {name, [f](Options& o, const char* v) { long long x; if (!parse_ll(v, 1, (1LL << 62), &x)) return false; o.*f = x; return true; }, back(f)};
}
Desc pair_of(const char* name, int Options::*g, int Options::*s) {
    return {name, [g, s](Options& o, const char* v) {
        int a = 0, b = 0;
        char tail = 0;
        if (sscanf(v, "%d,%d%c", &a, &b, &tail) != 2 || a < 1 || a > 64 || (a & (a - 1)) || b < 1 || b > 40) return false;
        o.*g = a; o.*s = b;
        return true;
    }, back2(g, s)};
}

const std::vector<Desc>& table() {
    static const std::vector<Desc> t = {
        flag("OPAL_HIP_DEBUG", &Options::debug),
        {"OPAL_HIP_DEVICES", [](Options& o, const char* v) {
             if (strcmp(v, "all") != 0)
                 for (const char* p = v; *p; p++) if (!((*p >= '0' && *p <= '9') || *p == ',')) return false;
             o.devices = v; return true; }, back(&Options::devices)},
        integer("OPAL_HIP_DEVICE", &Options::device, 0, 1023),
        {"OPAL_HIP_MULTI_GATHER", [](Options& o, const char* v) {
             if (!strcmp(v, "rccl")) o.multiGatherRccl = true; else if (!strcmp(v, "peer")) o.multiGatherRccl = false; else return false;
             return true; }, back(&Options::multiGatherRccl)},
        integer("OPAL_HOST_THREADS", &Options::hostThreads, 1, 4096),
        integer("OPAL_ORDER_THREADS", &Options::orderThreads, 1, 4096),
        one_of("OPAL_HIP_DEVICE_ORDER", &Options::deviceOrder, {0, 1}),
        integer("OPAL_HOST_WIDE", &Options::hostWide, 0, 1),
        big("OPAL_TRACEBACK_BYTES", &Options::tracebackBytes),
        big("OPAL_TRACEBACK_DEVICE_BYTES", &Options::tracebackDeviceBytes),
        flag("OPAL_HIP_SYNC_FILL", &Options::syncFill),
        flag("OPAL_HIP_NO_REUSE", &Options::noReuse),
        flag("OPAL_HIP_NO_SPECULATION", &Options::noSpeculation),
        flag("OPAL_HIP_NO_ZERO_COPY", &Options::noZeroCopy),
        integer("OPAL_HIP_ZC_FROM_SLICE", &Options::zcFromSlice, 0, 64),
        integer("OPAL_HIP_CHUNK_BYTES", &Options::chunkBytes, 4096, 1 << 30),
        {"OPAL_HIP_COPY_STREAMS", [](Options& o, const char* v) { long long x; if (!parse_ll(v, 1, 2, &x)) return false; o.oneCopyStream = x == 1; return true; }, back(&Options::oneCopyStream)},
        flag("OPAL_HIP_SPIN_SYNC", &Options::spinSync),
        flag("OPAL_HIP_HOST_STARTS", &Options::hostStarts),
        flag("OPAL_HIP_HOST_TRACEBACK", &Options::hostTraceback),
        flag("OPAL_HIP_NO_EXACT_ROW", &Options::noExactRow),
        one_of("OPAL_HIP_LANES_PER_SEQ", &Options::lanesPerSeq, {1, 2, 4, 8, 16, 32, 64}),
        one_of("OPAL_HIP_NARROW", &Options::narrow, {16, 17, 18}),
        one_of("OPAL_HIP_NARROW_FORCE", &Options::narrowForce, {17}),
        one_of("OPAL_HIP_FORCE_WIDTH", &Options::forceWidth, {32}),
        flag("OPAL_HIP_FORCE_CHUNKS", &Options::forceChunks),
        one_of("OPAL_HIP_CHUNK_WIDTH", &Options::chunkWidth, {16, 17, 18}),
        pair_of("OPAL_HIP_CHUNK_GEOMETRY", &Options::chunkG, &Options::chunkS),
        one_of("OPAL_HIP_STREAM", &Options::stream, {0, 1}),
        pair_of("OPAL_HIP_STREAM_CHUNKS", &Options::streamChunkG, &Options::streamChunkS),
        integer("OPAL_HIP_STREAM_BLOCKS", &Options::streamBlocks, 1, 1 << 20),
        one_of("OPAL_HIP_STREAM_PRIO", &Options::streamPrio, {0, 1}),
        integer("OPAL_HIP_STREAM_HEAD_ITEMS", &Options::streamHeadItems, 1, 1 << 30),
        flag("OPAL_HIP_NO_STREAM_HEAD", &Options::noStreamHead),
        flag("OPAL_HIP_NO_STREAM_CHUNKS", &Options::noStreamChunks),
        flag("OPAL_HIP_NO_SKEW", &Options::noSkew),
        flag("OPAL_HIP_NO_QUERY_BOUND", &Options::noQueryBound),
        flag("OPAL_HIP_NO_UBER", &Options::noUber),
        one_of("OPAL_HIP_PIPELINE", &Options::pipeline, {0, 1, 2}),
        integer("OPAL_HIP_PIPE_SLACK", &Options::pipeSlack, 0, 1 << 20),
        one_of("OPAL_HIP_PIPE_PRIO", &Options::pipePrio, {-1, 0, 1}),
        flag("OPAL_HIP_UBER_ONLY", &Options::uberOnly),
        flag("OPAL_HIP_UBER_LDS_PAD", &Options::uberLdsPad),
        flag("OPAL_HIP_NO_SLICES", &Options::noSlices),
        {"OPAL_HIP_SLICES", [](Options& o, const char* v) {
             std::vector<int> out;
             for (const char* q = v; *q;) {
                 char* end = nullptr;
                 const long x = strtol(q, &end, 10);
                 if (end == q || x < 1 || x > 100000 || (*end && *end != ',')) return false;
                 out.push_back((int)x);
                 q = *end ? end + 1 : end;
             }
             if (out.empty()) return false;
             o.slices = out; return true; }, back(&Options::slices)},
        {"OPAL_HIP_SLICE_STREAMS", [](Options& o, const char* v) {
             if (!strcmp(v, "serial")) o.sliceStreams = 0; else if (!strcmp(v, "alternate")) o.sliceStreams = 1; else if (!strcmp(v, "separate")) o.sliceStreams = 2; else return false;
             return true; }, back(&Options::sliceStreams)},
        flag("OPAL_HIP_AUX_HIGH_PRIO", &Options::auxHighPrio),
        flag("OPAL_HIP_NO_SIDE_PREFIX", &Options::noSidePrefix),
        one_of("OPAL_HIP_SIDE_STREAMS", &Options::sideStreams, {2, 3}),
        {"OPAL_HIP_SIDE_STREAM_PRIO", [](Options& o, const char* v) { long long x; if (!parse_ll(v, 0, 1, &x)) return false; o.sideStreamsNormalPrio = x == 0; return true; }, back(&Options::sideStreamsNormalPrio)},
        integer("OPAL_HIP_SIDE_PRIO", &Options::sidePrio, 0, 3),
        {"OPAL_HIP_TAIL_FRACTION", [](Options& o, const char* v) {
             char* end = nullptr;
             const double x = strtod(v, &end);
             if (end == v || *end || !(x > 0.0) || x > 100.0) return false;
             o.tailFraction = x; return true; }, back(&Options::tailFraction)},
        {"OPAL_HIP_TAIL_MODEL", [](Options& o, const char* v) {
             if (!strcmp(v, "pipe")) o.tailPipeModel = true; else if (!strcmp(v, "lone")) o.tailPipeModel = false; else return false;
             return true; }, back(&Options::tailPipeModel)},
        integer("OPAL_HIP_WAVES_PER_BLOCK", &Options::wavesPerBlock, 1, 16),
        flag("OPAL_HIP_NO_STARTED_WAIT", &Options::noStartedWait),
        integer("OPAL_HIP_SWAP", &Options::swap, 0, 1 << 20),
        integer("OPAL_HIP_PAIR", &Options::pair, 0, 1 << 20),
        one_of("OPAL_HIP_SWAP_ROWS", &Options::swapRows, {8, 16, 24, 40}),
    };
    return t;
}

std::mutex g_mutex;
Options g_opts;
std::atomic<bool> g_parsed{false};

const Desc* find(const char* name) {
    for (const Desc& d : table())
        if (!strcmp(d.name, name)) return &d;
    return nullptr;
}

void parse_environment_locked() {
    if (g_parsed.load(std::memory_order_acquire)) return;
    for (char** e = environ; e && *e; e++) {
        const char* eq = strchr(*e, '=');
        if (!eq) continue;
        const std::string name(*e, (size_t)(eq - *e));
        if (name.compare(0, 9, "OPAL_HIP_") != 0 && name.compare(0, 10, "OPAL_HOST_") != 0 && name.compare(0, 15, "OPAL_TRACEBACK_") != 0 &&
            name != "OPAL_ORDER_THREADS")
            continue;
        const Desc* d = find(name.c_str());
        if (!d) { fprintf(stderr, "[opal] unknown option %s in the environment (ignored)\n", name.c_str()); continue; }
        if (!d->set(g_opts, eq + 1)) fprintf(stderr, "[opal] option %s: invalid value '%s' (ignored)\n", name.c_str(), eq + 1);
    }
    g_parsed.store(true, std::memory_order_release);
}

}  // namespace

const Options& opts() {
    // (double-checked: after the first call this is one read of a flag that never changes back)
    if (!g_parsed.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lk(g_mutex); parse_environment_locked(); }
    return g_opts;
}

int set_option(const char* name, const char* value) {
    if (!name) return 1;
    std::lock_guard<std::mutex> lk(g_mutex);
    parse_environment_locked();
    const Desc* d = find(name);
    if (!d) return 1;
    Options scratch = g_opts;
    if (!value) { d->reset(g_opts, Options()); return 0; }  // back to the default
    if (!d->set(scratch, value)) {
        fprintf(stderr, "[opal] option %s: invalid value '%s' (ignored)\n", name, value);
        return 2;
    }
    g_opts = scratch;
    return 0;
}

}  // namespace opal_host

namespace opal_host {
// (what a cached launch plan depends on besides the search itself)
std::string Options::plan_key() const {
    char buf[96];
    snprintf(buf, sizeof buf, "%d|%g|%d|%d|%d", lanesPerSeq, tailFraction, (int)tailPipeModel, (int)uberOnly, sideStreams);
    return buf;
}
}  // namespace opal_host
