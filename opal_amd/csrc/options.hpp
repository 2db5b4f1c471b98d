// options.hpp -- the library's tuning / test options as ONE typed struct, parsed once.
//
// Every option has the name of an environment variable (OPAL_HIP_*, OPAL_HOST_*, OPAL_TRACEBACK_*). The environment is read
// ONCE, when the first search (or the first opalx_set_option) needs it: a caller's later setenv() changes nothing, and no
// search reads the environment while another thread may be writing it. After that, opalx_set_option(name, value) is the way
// to change an option from inside the process (tests, a server reconfiguring itself); it validates the value the same way.
// Unknown OPAL_HIP_* names in the environment and values that do not parse are reported on stderr and ignored.
// INTEGRATION.md lists the options; everything that is not named "test hook" there is safe in production.
#pragma once
#include <string>
#include <vector>

namespace opal_host {

struct Options {
    // ---- diagnostics
    bool debug = false;               // OPAL_HIP_DEBUG: scheduler decisions and timings on stderr
    // ---- devices
    std::string devices;              // OPAL_HIP_DEVICES: "all" or "0,1,..": opalSearchDatabase shards every call over these GPUs
    bool multiGatherRccl = false;     // OPAL_HIP_MULTI_GATHER=rccl: opalx_multi_* handles gather with one ncclGather per search instead of peer copies
    int device = -1;                  // OPAL_HIP_DEVICE: the GPU of threads that have not chosen one (-1: the current device)
    // ---- host side
    int hostThreads = 0;              // OPAL_HOST_THREADS: width of the host thread pool (0: from the CPU quota)
    int orderThreads = 0;             // OPAL_ORDER_THREADS: threads of the length ordering (0: automatic)
    int deviceOrder = -1;             // OPAL_HIP_DEVICE_ORDER: the length ordering on the device (order_kernel.hip): 0 never, 1 whenever possible, -1 automatic (large databases)
    int hostWide = -1;                // OPAL_HOST_WIDE: host traceback lanes (-1: automatic)
    long long tracebackBytes = 0;     // OPAL_TRACEBACK_BYTES: memory budget of the host traceback (0: default)
    long long tracebackDeviceBytes = 0;  // OPAL_TRACEBACK_DEVICE_BYTES: flag-matrix budget of the device traceback (0: 8 GiB)
    bool syncFill = false;            // OPAL_HIP_SYNC_FILL: gather and upload the whole database before the search starts
    bool noReuse = false;             // OPAL_HIP_NO_REUSE: never recognise the previous call's database
    bool noSpeculation = false;       // OPAL_HIP_NO_SPECULATION: a re-seen database is compared before it is searched
    bool noZeroCopy = false;          // OPAL_HIP_NO_ZERO_COPY: a new database is uploaded, never read from pinned memory by the kernels
    int zcFromSlice = 0;              // OPAL_HIP_ZC_FROM_SLICE: the slices before this one wait for their chunks' copies, those from it on read pinned memory (0: all zero-copy)
    int chunkBytes = 0;               // OPAL_HIP_CHUNK_BYTES (test hook): size of an upload chunk of a fill (0: 2 MB or a 96th of the database)
    bool oneCopyStream = false;       // OPAL_HIP_COPY_STREAMS=1: one copy stream instead of two
    bool spinSync = false;            // OPAL_HIP_SPIN_SYNC: poll the stream instead of blocking on it
    bool hostStarts = false;          // OPAL_HIP_HOST_STARTS: start cells by the host's reverse pass
    bool hostTraceback = false;       // OPAL_HIP_HOST_TRACEBACK: operation strings by the host
    bool noExactRow = false;          // OPAL_HIP_NO_EXACT_ROW: no second device pass for doubtful start rows
    // ---- scheduling (planner.cpp, ladder.cpp); "test hook": forces a path the scheduler would not take by itself
    int lanesPerSeq = 0;              // OPAL_HIP_LANES_PER_SEQ (test hook): 1..64, power of two
    int narrow = 0;                   // OPAL_HIP_NARROW (test hook): 16 / 17 / 18: width of the 16-bit rung
    int narrowForce = 0;              // OPAL_HIP_NARROW_FORCE (test hook): 17: width 17 in row chunks rather than another width
    int forceWidth = 0;               // OPAL_HIP_FORCE_WIDTH (test hook): 32: everything on the 32-bit rung
    bool forceChunks = false;         // OPAL_HIP_FORCE_CHUNKS (test hook): class-kernel row chunks whatever the query length
    int chunkWidth = 0;               // OPAL_HIP_CHUNK_WIDTH (test hook): 16 / 17 / 18
    int chunkG = 0, chunkS = 0;       // OPAL_HIP_CHUNK_GEOMETRY "G,S" (test hook)
    int stream = -1;                  // OPAL_HIP_STREAM: 0 never stream, 1 whenever a geometry exists (test hook), -1 automatic
    int streamChunkG = 0, streamChunkS = 0;  // OPAL_HIP_STREAM_CHUNKS "G,S" (test hook)
    int streamBlocks = 0;             // OPAL_HIP_STREAM_BLOCKS (test hook): persistent workgroups of a streaming launch
    int streamPrio = 1;               // OPAL_HIP_STREAM_PRIO=0: no raised issue priority for the waves on a streaming launch's longest sequences
    int streamHeadItems = 0;          // OPAL_HIP_STREAM_HEAD_ITEMS (test hook)
    bool noStreamHead = false;        // OPAL_HIP_NO_STREAM_HEAD
    bool noStreamChunks = false;      // OPAL_HIP_NO_STREAM_CHUNKS
    bool noSkew = false;              // OPAL_HIP_NO_SKEW: NW/HW/OV on the unskewed kernels
    bool noQueryBound = false;        // OPAL_HIP_NO_QUERY_BOUND: the a-priori score bound of the 16-bit rung from the largest matrix entry alone (rounds 1-5), not from the query's own residues
    bool noUber = false;              // OPAL_HIP_NO_UBER: length classes as separate launches
    bool uberOnly = false;            // OPAL_HIP_UBER_ONLY: keep the one-launch chain whatever the tail
    bool uberLdsPad = true;           // OPAL_HIP_UBER_LDS_PAD=0: length classes as separate launches when the bulk wants a padded LDS request (round 3)
    bool noSlices = false;            // OPAL_HIP_NO_SLICES: no sliced launches while the database arrives
    std::vector<int> slices;          // OPAL_HIP_SLICES "a,b,..": upload chunks per slice
    int sliceStreams = 1;             // OPAL_HIP_SLICE_STREAMS: serial (0) / alternate (1, default) / separate (2)
    bool auxHighPrio = true;          // OPAL_HIP_AUX_HIGH_PRIO=0: the second slice stream of a cold call is an ordinary stream (default: a high-priority one, a hardware queue of its own: cold C2 call 2.38-2.47 -> 2.29 ms in a process that already owns other streams)
    bool noSidePrefix = false;        // OPAL_HIP_NO_SIDE_PREFIX
    int sideStreams = 2;              // OPAL_HIP_SIDE_STREAMS: 2 or 3 side streams of separate class launches
    bool sideStreamsNormalPrio = false;  // OPAL_HIP_SIDE_STREAM_PRIO=0
    int sidePrio = -1;                // OPAL_HIP_SIDE_PRIO: s_setprio of side classes (-1: 1)
    double tailFraction = 0;          // OPAL_HIP_TAIL_FRACTION (0: 0.7)
    bool tailPipeModel = false;       // OPAL_HIP_TAIL_MODEL=pipe
    int wavesPerBlock = 0;            // OPAL_HIP_WAVES_PER_BLOCK (test hook)
    int pipeline = 1;                 // OPAL_HIP_PIPELINE=0: queued searches (opalx_search_device_async) one after the other on one stream (round 5); 1: in turn on two
                                      // streams, a search's grid reaching the dispatcher when the one before has started its last workgroup; 2 (test hook): in turn on two
                                      // streams with no "started" wait (each waits for the whole search before it: the order of round 5 on two streams)
    int pipePrio = -1;                // OPAL_HIP_PIPE_PRIO: priority class of the second stream of queued searches: -1 lowest (default: a hardware queue of its own), 0 ordinary, 1 highest
    int pipeSlack = 0;                // OPAL_HIP_PIPE_SLACK: a queued search's grid is released when all but this many workgroups of the one before have started
    bool noStartedWait = false;       // OPAL_HIP_NO_STARTED_WAIT: the grids of a search do not wait for its swapped workgroups to have started
    int swap = -1;                    // OPAL_HIP_SWAP: 0 never use the role-swapped kernel, N > 0 (test hook) the N longest targets, -1 automatic
    int pair = -1;                    // OPAL_HIP_PAIR: 0 never spread a pair over several waves, N > 0 (test hook) the N longest targets, -1 automatic
    int swapRows = 0;                 // OPAL_HIP_SWAP_ROWS (test hook): 8 / 16 / 24 / 40
    // the values of the options that are part of a launch-plan cache key, as one string
    std::string plan_key() const;
};

const Options& opts();                                  // parsed from the environment on first use
int set_option(const char* name, const char* value);    // 0 ok, 1 unknown name, 2 invalid value (value NULL: back to the default)

}  // namespace opal_host
