/*
 * opal_hip.h -- the seam between the host C++ library and the HIP device code,
 * plus the additive "prepared database" API.
 *
 * Everything here is extern "C" and POD-only (plain pointers and sizes; no C++
 * or torch types cross it).
 *
 * Part 1 (opal_hip_*) is the thin device shim the host scheduler in
 * opal_amd/csrc/opal_api.cpp drives: memory, copies, and ONE kernel-launch
 * entry that runs the affine-gap DP for a list of packed sequence pairs. It
 * replaces the reference's SIMD kernels
 *   searchDatabaseSW_<SimdSW<T>>      reference src/opal.cpp:164-470
 *   searchDatabase_<Simd<T>, MODE>    reference src/opal.cpp:594-977
 * and their per-lane refill (loadNextSequence, :472-490).
 *
 * Part 2 (opalx_*) is additive to opal.h: it lets a caller that already holds
 * the database in device memory (bench.py, a server answering many queries)
 * reuse the packed database across queries and receive results in device
 * buffers, in caller order. opalSearchDatabase itself is implemented as
 * create -> search -> destroy on top of it.
 */
#ifndef OPAL_HIP_H
#define OPAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ part 1 */

/* One kernel launch over `num_items` work items. All pointers are device pointers. */
/* Zero-copy launches (OpalHipLaunch::residues_out with opal_hip_launch_uber): SW score searches whose bulk class has at most
 * this many rows per lane -- the shapes instantiated with the extra store (opal_hip_uber_supported(mode, 16, ...) tells). */
#define OPAL_ZERO_COPY_MAX_ROWS 38

typedef struct OpalHipLaunch {
    /* what to compute */
    int width;        /* 16: two sequences per lane as packed int16; 17: one sequence per lane, plain 16-bit
                         (full-rate VALU); 32: one sequence per lane, int32 */
    int mode;         /* OPAL_MODE_* */
    int want_end;     /* 0: score only; 1: also end locations (out_end_query / out_end_target must be set) */
    int detect_wrap;  /* width 32, SW only: test ulH+P for 32-bit wrap (reference opal.cpp:302-303) */
    /* wave geometry: each sequence (pair) is spread over `lanes_per_seq` adjacent lanes, each lane
       holding `rows_per_lane` consecutive query rows in registers */
    int lanes_per_seq;   /* power of two, 1..64 */
    int rows_per_lane;   /* 1..OPAL_HIP_MAX_ROWS; lanes_per_seq*rows_per_lane >= query_length */
    /* query */
    const unsigned char* query;  /* [query_length] alphabet indices */
    int query_length;
    const int* score_matrix;     /* [alphabet_length^2], row = query letter */
    int alphabet_length;
    int gap_open, gap_ext;
    /* packed database (OpalxDb in opal_internal.hpp / DESIGN.md "data layout"): the sequences sorted by length, longest
       first; item i = position in that order = pair * 2 + half */
    const unsigned char* residues;  /* one byte per residue, sequence after sequence in item order */
    const uint64_t* seq_offset;     /* [items] start of each sequence in `residues` */
    unsigned char* residues_out;    /* opal_hip_launch_uber, SW score only: NULL, or `residues` is pinned host memory read over PCIe
                                       and the kernel also stores every residue it reads here -- the device copy */
    const uint16_t* tokens;      /* width 16 only. Per pair (items 2p, 2p+1): max(lenA,lenB) entries, low byte = residue of A, high byte = B */
    const uint64_t* pair_token_offset; /* width 16 only. [num_pairs] start of each pair's token stream */
    const int* pair_len;         /* [2*num_pairs] lengths of A and B (0 = absent) */
    const int* pair_seq;         /* [2*num_pairs] caller index of A and B (-1 = absent) */
    /* work list: item i is work[i] if work != NULL else first_item + i.
       width 16: an item is a pair index; width 17 and 32: an item is pair*2 + half */
    const int* work;
    int first_item;
    int num_items;
    /* outputs, indexed by caller sequence index */
    int* out_score;
    int* out_end_query;   /* may be NULL when want_end == 0 */
    int* out_end_target;
    /* overflow re-run list produced by width-16 kernels (items in width-32 encoding) */
    int* overflow_count;  /* device counter, incremented atomically */
    int* overflow_items;  /* capacity >= 2*num_pairs */
    void* stream;         /* hipStream_t or NULL */
    int priority;         /* 0..3: wave issue priority (s_setprio) of this launch's waves */
    /* Queries too long for one wave (lanes_per_seq*rows_per_lane < query_length) run as several launches
       over consecutive row chunks, top to bottom. chunk_flags == 0 means "single launch covers the query".
       bit0: a chunk above exists (its bottom row is read from `boundary`), bit1: a chunk below exists
       (this chunk's bottom row is written to `boundary`). SW/HW/OV: only the FIRST chunk may be shorter
       than lanes_per_seq*rows_per_lane; NW: only the LAST one. */
    int chunk_flags;
    int chunk_row0, chunk_rows;
    void* boundary;       /* 8 bytes per residue of the packed database (width 16: per token) */
    /* width 17 only: additions wrap, so scores >= sat_limit (= 32767 - largest matrix entry) are handed to
       the re-run list; neg_pad is the profile value of padding columns (SW: -sat_limit, else 0) */
    int sat_limit, neg_pad;
    /* width 17, NW/HW/OV, single launch: use the diagonal-skewed kernels (values grow by up to
       (rows_per_lane + steps) * gap_ext; the caller guarantees they stay below 32767) */
    int skew;
    int* scratch;         /* [2 * number of sequences] carried across chunks (OV) */
    /* waves per workgroup, 1..opal_hip_max_waves_per_block(width) (0: default 4). The scheduler picks it from the kernel's register count
       and the LDS size of the query profile so that every SIMD holds an even number (>= 2) of waves. */
    int waves_per_block;
    /* request at least this much LDS per workgroup (0: the query profile's own size): the scheduler pads the
       request to keep an odd number of workgroups from sharing a CU */
    int lds_bytes_min;
    /* opal_hip_launch_stream: items [side_first_item, +side_num_items) run on the class body inside the same launch, ahead
       of the streaming workgroups. NW/HW/OV: sequences too long for the skewed arithmetic of the stream, same geometry;
       SW (width 17, lanes_per_seq <= 16): the longest sequences of a long-tailed database, on 4 x lanes_per_seq lanes */
    int side_first_item, side_num_items;
    /* Clock probe (NULL: off): two device counters; wave 0 of every workgroup adds the shader cycles and the constant-rate
       ticks (opal_hip_wall_clock_hz per second) of its lifetime to [0] / [1] -- their ratio is the clock the chip held
       under this kernel. */
    unsigned long long* clock_probe;
    /* Timeline probe (NULL: off): every workgroup b of the launch writes the constant-rate ticks at its start and end to
       timeline[2b], timeline[2b+1] (capacity: 2 x the grid size). */
    unsigned long long* timeline;
    /* opal_hip_launch_stream: leave this many CUs without a persistent workgroup -- for launches of the same search that run
       BESIDE the stream on other streams (the 32-bit prefix, swapped targets): a persistent workgroup per CU would otherwise
       keep them waiting for the whole stream whenever it reaches the dispatcher first (24 ms or 34 ms per search, from run to run) */
    int reserve_blocks;
    /* opal_hip_launch_stream: the waves working on the first prio_items sequences handed out (the longest of the range) run at
       a raised issue priority (0: none) */
    int prio_items;
    /* NULL, or a device counter (zeroed by the caller: the search header) that every workgroup of the launch increments when it
       starts: opal_hip_wait_started (the grids of a search behind its swapped / pair workgroups), opal_hip_wait_count (the
       grid of the NEXT queued search behind this one's last workgroup) */
    int* started_counter;
} OpalHipLaunch;

/* ONE launch for up to three length classes of a search (plain 16-bit kernels): the bulk of the database on
   base.lanes_per_seq (G0) lanes x base.rows_per_lane (S0) rows [items base.first_item .. +base.num_items), long
   sequences on 2*G0 lanes x ceil(S0/2) rows, the longest on 4*G0 x ceil(S0/4). The workgroups of the longest class
   come first in the grid. No side streams, events or extra hardware queues are involved. */
typedef struct OpalHipUberLaunch {
    OpalHipLaunch base;
    int long_first_item, long_num_items;
    int longest_first_item, longest_num_items;
    int long_priority, longest_priority;   /* s_setprio of those classes' waves (0: none) */
} OpalHipUberLaunch;
/* 1 if a one-launch kernel exists for this variant (width 17, single launch) and bulk rows-per-lane value; bit 4 of
   `want_end` (16) asks for the zero-copy variant (base.residues_out set: SW score only) */
int opal_hip_uber_supported(int mode, int want_end, int skew, int rows_per_lane);
int opal_hip_launch_uber(const OpalHipUberLaunch* args);

/* Streaming searches (stream_kernel.hpp): persistent workgroups, every group of `lanes_per_seq` lanes works through a
   stream of sequences handed out by `next_counter` (a device int, zero before the launch), the next sequence entering
   while the previous one drains. width 17 or 18; SW (score only or with end locations: want_end) or NW/HW/OV (the
   diagonal-skewed kernels: the caller guarantees the 16-bit range as for `skew`); no work list; results and, for SW, the
   overflow list as for opal_hip_launch. */
int opal_hip_stream_rows_supported(int width, int mode, int want_end, int min_rows);   /* smallest instantiated rows_per_lane >= min_rows (0: none) */
size_t opal_hip_stream_lds_bytes(int width, int mode, int lanes_per_seq, int rows_per_lane, int alphabet_length);
int opal_hip_launch_stream(const OpalHipLaunch* args, int* next_counter);

#define OPAL_HIP_MAX_ROWS 40

/* Role-swapped SW score search (swap_kernel.hpp) for the few longest targets of a search with a short query: the TARGET
   on the rows of a wave (row chunks of 64 lanes x rows_per_lane residues, top to bottom), the query on the columns; one
   workgroup per item of [first_item, +num_items). Read from `args`: query / matrix / gaps, the packed database, first_item,
   num_items, rows_per_lane (from opal_hip_swap_rows), want_end (end cells: 8 or 16 rows per lane only), out_score /
   out_end_query / out_end_target, the overflow list, sat_limit / neg_pad, stream, clock_probe. opal_hip_swap_rows: the default rows per lane (8 / 16 / 24 / 40 exist) for this query length and
   alphabet -- the largest shape within 64 KiB of LDS, so that the launch shares its CUs (0: none -- query beyond
   OPAL_HIP_SWAP_MAX_QUERY or alphabet too large). */
#define OPAL_HIP_SWAP_MAX_QUERY 1024
int opal_hip_swap_rows(int query_length, int alphabet_length);
size_t opal_hip_swap_lds_bytes(int rows_per_lane, int query_length, int alphabet_length);  /* (> 160 KiB: not launchable) */
int opal_hip_launch_swap(const OpalHipLaunch* args);
/* One query x target pair spread over several waves of one workgroup (pair_kernel.hpp: NW / HW / OV, int32, 2-byte profile
   entries -- every matrix entry must fit 16 bits): the longest targets of a global-mode search, items [first_item, +num_items),
   one workgroup each. opal_hip_pair_shape: rows per lane (2 / 4 / 8) and waves (1..16) for this query, 0 if none fits LDS.
   opal_hip_launch_pair: rows_per_lane / waves_per_block from it; `boundary`: (waves - 1) x bnd_stride 8-byte words (the bottom
   rows the waves hand each other), bnd_stride >= the residues of the items + 64, bnd_base = the residue offset of the first item.
   Fields used as in opal_hip_launch_swap (started_counter included). */
int opal_hip_pair_shape(int query_length, int alphabet_length, int* rows, int* waves);
int opal_hip_launch_pair(const OpalHipLaunch* args, void* boundary, unsigned long long bnd_base, unsigned long long bnd_stride);
/* Enqueue on `stream` a one-wave kernel that ends when *counter >= target (or after ~2 ms): what is launched behind it on
   that stream reaches the dispatcher only after `target` workgroups of a launch on ANOTHER stream have taken their CUs. A grid
   that fills every CU with workgroups of 39 KB of LDS otherwise starves a 55 KB workgroup arriving a moment later until
   it drains: one retiring workgroup never frees enough (541 762 sequences, 300-aa query: 8.5 ms or 12.5 from run to run). */
int opal_hip_wait_started(const int* counter, int target, void* stream);
/* The same with a longer patience (a quarter of a second), for the "started" word of another SEARCH's header: queued searches
   that take turns on two streams (OpalHipLaunch::started_counter of the class / one-launch / streaming launches counts into the
   same word as the swapped / pair launches in front of them). opal_hip_last_launch_blocks: the grid size of the last class /
   one-launch / streaming launch made by the calling thread (0 if the call launched nothing). */
int opal_hip_wait_count(const int* counter, int target, void* stream);
int opal_hip_last_launch_blocks(void);
int opal_hip_wait_timeouts(void);   /* opal_hip_wait_count calls on the current device that gave up so far (diagnostics; synchronizes) */
/* NW / HW / OV have role-swapped kernels too (int32 arithmetic, scores and end cells: out_end_* may be NULL): the same entry
   with args->mode; rows per lane from opal_hip_swap_gl_rows: 16 (2-byte profile entries: every matrix entry must fit 16
   bits) or 8 (4-byte entries). */
int opal_hip_swap_gl_rows(int query_length, int alphabet_length, int matrix_fits_16);
size_t opal_hip_swap_gl_lds_bytes(int rows_per_lane, int query_length, int alphabet_length);

int opal_hip_device_count(void);
/* Select `device` for the calling thread. 0 on success. */
int opal_hip_set_device(int device);
int opal_hip_get_device(void);             /* the calling thread's current device, -1 on error */
double opal_hip_wall_clock_hz(void);       /* ticks per second of the constant-rate device counter (clock probe) */
void* opal_hip_malloc(size_t bytes);
void opal_hip_free(void* p);
void* opal_hip_host_malloc(size_t bytes);  /* pinned host memory (staging of residues and results) */
void* opal_hip_host_malloc_device_cached(size_t bytes);  /* the same, coarse-grained: kernels may read it directly (cached in L2) */
void opal_hip_host_free(void* p);
int opal_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int opal_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int opal_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* Device-to-device across GPUs (xGMI peer DMA): the gather step of the multi-GPU search (opalx_multi_search_device). */
int opal_hip_memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream);
int opal_hip_enable_peer(int peer);   /* direct access from the current device to `peer` (0: enabled or same device) */
/* out[perm[i]] = in[i] for i < count, on the device */
int opal_hip_permute(const int* in, const int* perm, int count, int* out, void* stream);
/* out[perm[k * per + j]] = in[k * block_stride + j] for k < blocks, j < per; perm < 0 = padding (the layout an RCCL gather of
   equal-sized shard blocks leaves on the root) */
int opal_hip_permute_blocks(const int* in, const int* perm, int blocks, int per, int block_stride, int* out, void* stream);
/* The path's one collective through RCCL: a single-process communicator group over `devices` (all different;
   librccl.so is dlopen()ed here, not linked), and one ncclGather of `count` ints per rank to rank 0 inside a group call --
   send[k] and streams[k] on device k, recv_root ([n][count] ints) on devices[0]. init returns NULL if RCCL is not usable. */
void* opal_hip_rccl_init(const int* devices, int n);
void opal_hip_rccl_destroy(void* group);
int opal_hip_rccl_gather(void* group, const void* const* send, void* recv_root, size_t count, void* const* streams);
int opal_hip_memset(void* dst, int value, size_t bytes, void* stream);
int opal_hip_memset32(void* dst, int value, size_t count, void* stream);   /* `count` 32-bit words */
/* score[quads[4i]] = quads[4i+1], end_query[..] = quads[4i+2], end_target[..] = quads[4i+3]; `quads` on the device. */
int opal_hip_scatter_results(const int* quads, int count, int* score, int* end_query, int* end_target, void* stream);
int opal_hip_stream_sync(void* stream);
void* opal_hip_stream_create(void);       /* non-blocking stream: ordered only through events */
void* opal_hip_stream_create_high(void);  /* the same at the device's highest stream priority */
void* opal_hip_stream_create_low(void);    /* ... of the lowest (a priority class, and with it a hardware queue, of its own) */
void opal_hip_stream_destroy(void* stream);
int opal_hip_stream_wait_event(void* stream, void* event);
/* Interleave residues into the pair token streams of the width-16 kernels, on the device: for every pair p and column c,
   tokens[pair_off[p] + c] = raw[raw_off[2p] + c] | (c < pair_len[2p+1] ? raw[raw_off[2p+1] + c] : 0) << 8. */
int opal_hip_pack_tokens(const unsigned char* raw, const uint64_t* raw_off, const uint64_t* pair_off,
                         const int* pair_len, int num_pairs, uint16_t* tokens, uint64_t tokens_capacity, void* stream);
/* Copy `count` sequences on the device: dst[dst_off[i] + c] = src[src_off[i] + c] for c < len[i] (a database handed over
   as one flat device buffer in caller order, into item order). */
int opal_hip_reorder_residues(const unsigned char* src, const uint64_t* src_off, const uint64_t* dst_off, const int* len,
                              int count, unsigned char* dst, void* stream);
/* The length ordering on the device (order_kernel.hip): a stable counting sort of `n` lengths (caller order, on the device,
   every one in [0, max_len]) by length, longest first. Writes, for every item k < num_nonempty, the caller index, length and
   byte offset (items back to back in item order) of the k-th sequence; the absent item num_nonempty if that is odd
   {-1, 0, total bytes}; and the caller indices of the zero-length sequences, in caller order, into d_empty_seqs.
   max_len must not exceed opal_hip_order_max_length(); d_scratch holds opal_hip_order_scratch_bytes(n, max_len) bytes. */
int opal_hip_order_max_length(void);
size_t opal_hip_order_scratch_bytes(int n, int max_len);
int opal_hip_order_by_length(const int* d_lens, int n, int max_len, int num_nonempty, void* d_scratch, int* d_item_seq,
                             int* d_item_len, uint64_t* d_item_off, int* d_empty_seqs, void* stream);
/* ... and, behind it on the same stream: for every byte offset d_targets[t], d_out[2t] = the first item at or behind it (of the
   num_nonempty real items), d_out[2t+1] = that item's offset (total_bytes if there is none) -- where the fill's upload chunks and
   gather pieces begin. */
int opal_hip_order_cuts(const uint64_t* d_targets, int num_targets, const uint64_t* d_item_off, int num_nonempty,
                        uint64_t total_bytes, uint64_t* d_out, void* stream);
/* Enqueue one DP kernel. 0 on success, non-zero if the shape is not launchable. */
int opal_hip_launch(const OpalHipLaunch* args);
/* Smallest rows_per_lane >= min_rows for which this kernel variant is instantiated (0: none).
   want_end bit0: the SW end-location pass; bit1: the multi-chunk (chunk_flags != 0) variants; bit2: the
   diagonal-skewed variants (width 17, NW/HW/OV). */
int opal_hip_rows_supported(int width, int mode, int want_end, int min_rows);
/* Largest waves_per_block the kernels of this width are compiled for (8; 16 for width 18). */
int opal_hip_max_waves_per_block(int width);
/* VGPRs per lane of the kernel `args` selects (width, mode, want_end, detect_wrap, chunk_flags, skew,
   rows_per_lane are read), 0 if unknown. */
int opal_hip_kernel_vgprs(const OpalHipLaunch* args);
/* Bytes of LDS the launch would need (query profile), for the scheduler's shape choice. */
size_t opal_hip_lds_bytes(int width, int lanes_per_seq, int rows_per_lane, int alphabet_length);
/* ... and what a single-launch width-17 SW search WITH end locations adds to it: the buffer of every slot of a workgroup
   in which the rows holding the slot's best cell wait for their one look-up (one group of row blocks -- a quarter of a
   lane's rows where the kernel keeps four accumulators --, one word per row, then the lane and the group's first row;
   a multiple of 16 bytes). The streaming SW kernels keep two such buffers per slot (two sequences in flight);
   opal_hip_stream_lds_bytes includes them. */
size_t opal_hip_swend_ckpt_bytes(int lanes_per_seq, int rows_per_lane, int waves_per_block);
const char* opal_hip_last_error(void);
/* Timing helpers for bench.py: HIP events on the stream the kernels run on. */
void* opal_hip_event_create(void);
void* opal_hip_event_create_untimed(void);   /* ordering only (hipEventDisableTiming) */
void opal_hip_event_destroy(void* ev);
int opal_hip_event_record(void* ev, void* stream);
int opal_hip_event_sync(void* ev);   /* host-waits for the event */
float opal_hip_event_elapsed_ms(void* start, void* stop); /* synchronises on `stop` */

/* Start locations on the device (rev_kernel.hip): for every listed sequence, the score-limited reverse pass of
   reference src/opal.cpp:1475-1507 from its end cell. All pointers are device pointers. */
typedef struct OpalHipRevLaunch {
    int mode;                     /* OPAL_MODE_* of the search that produced score / end */
    const unsigned char* query; int query_length;
    const int* score_matrix; int alphabet_length;
    int gap_open, gap_ext;
    const unsigned char* residues; const uint64_t* seq_offset;   /* the packed database (see OpalHipLaunch) */
    const int* seq_item;          /* [N] caller index -> item in the packed database */
    const int* items; int num_items;   /* caller indices to process (NULL: 0..num_items-1) */
    const int* score; const int* end_query; const int* end_target;   /* [N], caller order */
    int* start_query; int* start_target;                             /* [N], caller order */
    int* flags;                   /* [2], zeroed by the caller: [0] length of `redo`, [1] sequences with no start found */
    int* redo;                    /* [N] sequences whose start the host must recompute (see rev_kernel.hip) */
    void* stream;
    /* Second pass over a `redo` list (items = that list): the stop COLUMN of every listed sequence is final (start_target holds
       it), only its row is in doubt -- the reference takes the first row of that column holding the score, whatever cell it
       is. exact_row != 0 recomputes the prefix up to that column and writes that row; sequences without a stop column are
       appended to `redo` (counted in flags[0]) for the host. */
    int exact_row;
} OpalHipRevLaunch;
/* rows per lane the reverse kernel would use for this query length (0: longer than 64 x 40 rows, unsupported) */
int opal_hip_rev_rows_supported(int query_length);
int opal_hip_rev_launch(const OpalHipRevLaunch* args);

/* Alignment operations on the device (trace_kernel.hip): for every listed sequence, the traceback of reference
   src/opal.cpp:1369-1409 inside the box between its start and end cells. All pointers are device pointers. */
typedef struct OpalHipTraceLaunch {
    const unsigned char* query; int alphabet_length;
    const int* score_matrix;
    int gap_open, gap_ext;
    const unsigned char* residues; const uint64_t* seq_offset;   /* the packed database (see OpalHipLaunch) */
    const int* seq_item;           /* [N] caller index -> item in the packed database */
    const int* items; int num_items;    /* caller indices of this batch */
    int max_rows;                  /* largest number of box rows (endQ - startQ + 1) in the batch */
    const int* end_query; const int* end_target; const int* start_query; const int* start_target;   /* [N], caller order */
    const uint64_t* dir_offset;    /* [num_items] start of each flag matrix in `dirs`: columns x (rows rounded up to 4) bytes */
    unsigned char* dirs;
    const uint64_t* ops_offset;    /* [num_items] start of each operation string in `ops` (capacity rows + columns) */
    unsigned char* ops;
    int* ops_length;               /* [num_items] out */
    int* box_score;                /* [num_items] out: must equal the alignment's score */
    void* stream;
} OpalHipTraceLaunch;
/* rows per lane the trace kernel would use for boxes of up to `rows` rows (0: more than 64 x 40, unsupported) */
int opal_hip_trace_rows_supported(int rows);
int opal_hip_trace_launch(const OpalHipTraceLaunch* args);

/* ------------------------------------------------------------------ part 2 */

typedef struct OpalxDb OpalxDb; /* opaque: packed database resident in device memory */

/* Pack `db_length` sequences (host pointers, as in opalSearchDatabase) and upload them to the
   current device. Returns NULL on failure (see opal_hip_last_error). */
OpalxDb* opalx_db_create(unsigned char* const db[], int db_length, const int db_seq_lengths[]);
/* Same, from one contiguous residue buffer: sequence j is residues[offsets[j] .. offsets[j+1]). */
OpalxDb* opalx_db_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length);
/* Same with the residue buffer already in DEVICE memory of the current device (offsets stay on the host);
   the buffer is only read during the call. The library works on a private non-blocking stream: whatever produced
   the buffer (a kernel or copy on another stream, e.g. torch's) must be COMPLETE before the call -- synchronise that
   stream first. The same holds for the result buffers handed to opalx_search_device / opalx_search_starts. */
OpalxDb* opalx_db_create_flat_device(const unsigned char* d_residues, const int64_t* offsets, int db_length);
/* opalSearchDatabase keeps its device buffers, pinned staging memory and streams in cached contexts between
   calls; this frees them (optional: call it when no search is running). The library's host threads are never
   stopped: do not dlclose() it. */
void opalx_release_cached(void);
/* Number of searches of this process that ran on the streaming SW kernel (diagnostics and tests). */
long long opalx_stream_launch_count(void);
/* What the searches of this process ran on, cumulative (diagnostics and tests). Fills out[0 .. min(capacity, 10)):
   [0] searches on the streaming SW kernels, [1] on the streaming NW/HW/OV kernels, [2] one-launch class kernels,
   [3] class kernels launched one class at a time, [4] row-chunk launches of queries beyond one wave, [5] launches of the
   32-bit rung, [6] alignments traced on the device, [7] alignments traced or started by the host, [8] opalSearchDatabase
   calls that found their database already on the device, [9] targets searched on the role-swapped kernel. Returns 10. */
int opalx_counters(long long* out, int capacity);
/* Test hook, no device needed: the order the library gives `n` sequences of lengths `lens` (stable, longest first; zero
   lengths left out; an absent last item when their number is odd: seq -1, len 0). Fills seq / len / off (byte offset in
   the packed residue buffer), `capacity` entries each. Returns the number of items or -1 if capacity is too small. */
int opalx_debug_order(const int* lens, int n, int* seq, int* len, unsigned long long* off, int capacity);
/* Test hook (needs the device): the same ordering computed by the device (order_kernel.hip), whatever the size; also the
   caller indices of the zero-length sequences (`empties`, capacity entries; *num_empties their count). Returns the number of
   items, or -1 if the device ordering does not apply (a length beyond opal_hip_order_max_length()) or failed. */
int opalx_debug_order_device(const int* lens, int n, int* seq, int* len, unsigned long long* off, int capacity, int* empties,
                             int* num_empties);
/* Test hook, no device needed: stress of the library's host thread pool (`rounds` jobs of 1..300 tasks, synchronous and
   asynchronous in turn, from `callers` threads at once). 0 if every task of every job ran exactly once. */
int opalx_debug_pool_selftest(int rounds, int callers);
/* Test hook, no device needed: what the scheduler would stream a search on (planner.cpp, plan_stream): out[0..3] = width
   (17 / 18; 0: the class kernels), lanes per sequence, rows per lane, passes (> 1: the query runs in that many row chunks).
   `residues`, `longest`, `num_seqs` describe the database. Returns 0. */
int opalx_debug_plan_stream(int query_length, int alphabet_length, int mode, int want_end, long long residues, int longest,
                            int num_seqs, int* out);
/* Test hook, no device needed: the a-priori bound on every DP value of a search with this query -- the sum over its residues of
   the best score the matrix gives that residue against any letter (score_matrix[query_letter * alphabet_length + target_letter],
   as opal.h), at least 0 per residue -- which decides what may run on the 16-bit kernels (ladder.cpp). */
long long opalx_debug_score_bound(const unsigned char* query, int query_length, const int* score_matrix, int alphabet_length);
/* Test hook, no device needed: where the scheduler would send the LONGEST sequences of a search (planner.cpp, plan_pair then
   plan_swap, as ladder.cpp asks them): out[0] = how many go to the role-swapped kernel (swap_kernel.hpp), out[1] its rows per
   lane; out[2] = how many to the pair kernel (pair_kernel.hpp; global modes only), out[3] its rows per lane, out[4] its waves.
   `lengths_desc`: the database's sequence lengths, longest first (`count` of them; the scheduler looks at the head of the list),
   `residues` their sum over the whole database, `need32` how many of them the a-priori bounds send to the 32-bit rung. Returns 0. */
int opalx_debug_plan_long_targets(int query_length, int alphabet_length, int mode, const int* lengths_desc, int count,
                                  long long residues, int need32, int* out);
/* The library's options (INTEGRATION.md lists them) carry the names of environment variables -- OPAL_HIP_DEVICES,
   OPAL_HIP_DEBUG, the scheduler's test hooks ... The environment is read ONCE, when the library first needs an option; after
   that this call is the way to change one from inside the process (value NULL: back to the default). Values are
   validated as they are for the environment. Returns 0, 1 for an unknown name, 2 for a value that does not parse (nothing
   changes then). Not to be called while a search is running. */
int opalx_set_option(const char* name, const char* value);
void opalx_db_destroy(OpalxDb* db);
/* Clock probe for measurements (bench.py): while enabled, every search on this handle also measures the shader clock
   its DP kernels ran at (cycle counter / constant-rate counter over the lifetime of one wave per workgroup) and reads it
   back after the search (one extra small copy: leave it off in timed loops). opalx_db_last_clock_ghz returns the figure
   of the last probed search, 0 if none. */
void opalx_db_clock_probe(OpalxDb* db, int enable);
double opalx_db_last_clock_ghz(const OpalxDb* db);
/* enable == 2 also records, per workgroup of the LAST DP launch of a search, the constant-rate ticks
   (opal_hip_wall_clock_hz per second) at its start and end; opalx_db_last_timeline copies them out
   ({start, end} per workgroup, in grid order; returns the number of workgroups, at most capacity / 2). */
int opalx_db_last_timeline(OpalxDb* db, unsigned long long* out, int capacity);
/* ... and the shader-clock cycle counter at the same two moments of every workgroup (same layout, same return value): cycles
   per tick x opal_hip_wall_clock_hz = the clock that workgroup saw over its life. */
int opalx_db_last_timeline_cycles(OpalxDb* db, unsigned long long* out, int capacity);
int opalx_db_num_sequences(const OpalxDb* db);
int64_t opalx_db_num_residues(const OpalxDb* db);

/*
 * Search the whole packed database. Results go to DEVICE buffers of db_length ints each, in
 * caller order (d_end_query/d_end_target may be NULL for OPAL_SEARCH_SCORE; whatever last wrote them on another
 * stream must be complete, see opalx_db_create_flat_device). The call enqueues the kernels and waits for them. Returns 0 or an OPAL_ERR_* code. `kernel_ms`, if not NULL,
 * receives the device time of all DP kernels (HIP events on the launch stream).
 * search_type is OPAL_SEARCH_SCORE or OPAL_SEARCH_SCORE_END.
 */
int opalx_search_device(OpalxDb* db, const unsigned char* query, int query_length,
                        int gap_open, int gap_ext, const int* score_matrix, int alphabet_length,
                        int search_type, int mode, int overflow_method,
                        int* d_score, int* d_end_query, int* d_end_target, float* kernel_ms);

/* The same without the wait: the kernels are enqueued on the handle's stream and the call returns, so that the caller
   can do other host work (bench.py: the previous step's gather) while the GPU searches; opalx_db_wait waits for the search
   and reports its kernel time. Up to OPALX_ASYNC_DEPTH - 1 searches may be queued on a handle (each into result buffers
   of its own): they run one after the other on the handle's stream, with no host round trip between them, and every
   opalx_db_wait waits for the OLDEST one not yet waited for (first in, first out); with none outstanding it waits for the
   stream. One more opalx_search_device_async than that returns OPAL_ERR_NO_SIMD_SUPPORT and enqueues nothing. A search that
   needs a host decision midway (a 16-bit rung whose scores can saturate: the re-run list is read back) waits internally
   and is complete on return. Return codes as opalx_search_device (a 32-bit wrap is reported by the call that detects it). */
#define OPALX_ASYNC_DEPTH 4
int opalx_search_device_async(OpalxDb* db, const unsigned char* query, int query_length,
                              int gap_open, int gap_ext, const int* score_matrix, int alphabet_length,
                              int search_type, int mode, int* d_score, int* d_end_query, int* d_end_target);
int opalx_db_wait(OpalxDb* db, float* kernel_ms);
/* Round 6: queued searches take turns on TWO streams of the handle. A search that is one grid (a one-launch class search, a
   single streaming launch; no re-run list possible) lets the grid of the search queued behind it reach the dispatcher as soon as
   its own last workgroup has started, so that the second fills the CUs the first leaves while it drains; anything else waits for
   the whole search before it. Results, return codes and the first-in-first-out order of opalx_db_wait are unchanged; what
   changes is what the times mean: `kernel_ms` of a search is the device time from the release of its first launch to its end --
   and consecutive searches overlap, so these do not add up to the elapsed time. `period_ms` (opalx_db_wait2) is the device time
   between the end of the search waited for before this one and the end of this one: the time the search cost the queue (0 for
   the first search ever waited for on the handle). Option OPAL_HIP_PIPELINE=0: one stream, one search after the other. */
int opalx_db_wait2(OpalxDb* db, float* kernel_ms, float* period_ms);

/* As above with HOST result buffers (copied back after the kernels). */
int opalx_search_host(OpalxDb* db, const unsigned char* query, int query_length,
                      int gap_open, int gap_ext, const int* score_matrix, int alphabet_length,
                      int search_type, int mode, int overflow_method,
                      int* score, int* end_query, int* end_target);

/*
 * One database over SEVERAL GPUs of this process (the counterpart of bench.py's one-process-per-GPU
 * path for a single-process caller): the sequences are dealt to `num_devices` GPUs in 64-sequence groups balancing the
 * residue count, every GPU keeps its shard packed in its own memory, a search runs on all of them at once (one host thread
 * per GPU), every shard's {score[, endQ, endT]} block travels to devices[0] by peer DMA (xGMI) and a permutation kernel
 * there writes the three result vectors in DATABASE order into the caller's buffers on devices[0] -- no result touches
 * the host. A device may be listed more than once (how the path is tested on a one-GPU box).
 */
typedef struct OpalxMultiDb OpalxMultiDb;
OpalxMultiDb* opalx_multi_create_flat(const unsigned char* residues, const int64_t* offsets, int db_length,
                                      const int* devices, int num_devices);
void opalx_multi_destroy(OpalxMultiDb* mdb);
/* d_score / d_end_query / d_end_target: device buffers on devices[0], db_length ints each (end buffers may be NULL for
   OPAL_SEARCH_SCORE). `kernel_ms`: the longest DP time over the shards. Returns 0 or an OPAL_ERR_* code. */
int opalx_multi_search_device(OpalxMultiDb* mdb, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                              const int* score_matrix, int alphabet_length, int search_type, int mode,
                              int* d_score, int* d_end_query, int* d_end_target, float* kernel_ms);
/* The same without the wait, as opalx_search_device_async / opalx_db_wait for the single handle: up to OPALX_ASYNC_DEPTH - 1
   searches may be queued (each into result buffers of its own); they run one after the other on every shard's stream, the
   gather of one behind the root's permutation of the one before, with no host round trip between them. opalx_multi_wait
   waits for the OLDEST one (first in, first out) and reports its kernel time (the longest over the shards). */
int opalx_multi_search_device_async(OpalxMultiDb* mdb, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                                    const int* score_matrix, int alphabet_length, int search_type, int mode,
                                    int* d_score, int* d_end_query, int* d_end_target);
int opalx_multi_wait(OpalxMultiDb* mdb, float* kernel_ms);
/* How this handle gathers its shards' results on the root GPU: 0 = peer copies (hipMemcpyPeerAsync per shard and vector, the
   default), 1 = one RCCL ncclGather per search (option OPAL_HIP_MULTI_GATHER=rccl when the handle was created; needs every
   listed device to be a different GPU -- otherwise, or if librccl.so cannot be loaded, the handle falls back to peer copies). */
int opalx_multi_gather_kind(const OpalxMultiDb* mdb);

/*
 * Host traceback (no device work): for every result that holds a score and an end location, fill
 * start location and alignment operations, exactly as OPAL_SEARCH_ALIGNMENT does after the search
 * (reference src/opal.cpp:1475-1507, findAlignment :1236-1431). `alignment` is malloc()ed.
 */
#define OPALX_ERR_NO_MEMORY 4   /* additive error code: the host traceback could not allocate its matrix */
struct OpalSearchResult;
int opalx_find_alignments(unsigned char query[], int query_length, unsigned char* db[], int db_length,
                          int db_seq_lengths[], int gap_open, int gap_ext, int* score_matrix, int alphabet_length,
                          struct OpalSearchResult* results[], int mode);

/* The same with start cells already known for some sequences (start_query[i] / start_target[i] >= 0, e.g. from
   opalx_search_starts): only the box between start and end is traced for those. NULL arrays or negative
   entries mean "find it" (the full score-limited reverse pass on the host). */
int opalx_find_alignments_from(unsigned char query[], int query_length, unsigned char* db[], int db_length,
                               int db_seq_lengths[], int gap_open, int gap_ext, int* score_matrix, int alphabet_length,
                               struct OpalSearchResult* results[], int mode, const int* start_query, const int* start_target);
/* The same, skipping the results i with skip[i] != 0 (e.g. already traced on the device); skip may be NULL. */
int opalx_find_alignments_masked(unsigned char query[], int query_length, unsigned char* db[], int db_length,
                                 int db_seq_lengths[], int gap_open, int gap_ext, int* score_matrix, int alphabet_length,
                                 struct OpalSearchResult* results[], int mode, const int* start_query, const int* start_target,
                                 const unsigned char* skip);

/*
 * Start locations of alignments, computed on the device (rev_kernel.hip; reference src/opal.cpp:1475-1507):
 * d_score / d_end_query / d_end_target are the device arrays a search with OPAL_SEARCH_SCORE_END on the same
 * database, query, gaps, matrix and mode has filled; d_start_query / d_start_target (db_length ints each)
 * receive the start cells in caller order (-1: no alignment). `redo` (host, capacity db_length, may be NULL) receives
 * the indices whose start is not reliable and must be found by the host pass (opalx_find_alignments_from with a
 * negative start) -- empty in practice; *redo_count their number. Returns 0, OPAL_ERR_NO_SIMD_SUPPORT when the
 * query is longer than the kernel supports (2560 rows).
 */
int opalx_search_starts(OpalxDb* db, const unsigned char* query, int query_length, int gap_open, int gap_ext,
                        const int* score_matrix, int alphabet_length, int mode,
                        const int* d_score, const int* d_end_query, const int* d_end_target,
                        int* d_start_query, int* d_start_target, int* redo, int* redo_count);

#ifdef __cplusplus
}
#endif

#endif /* OPAL_HIP_H */
