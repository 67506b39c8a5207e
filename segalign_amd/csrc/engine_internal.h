// engine_internal.h -- what the host-side translation units of libsegalign_hip.so share: error handling with the reference's
// exit codes, the grow-only device buffers, per-slot and per-device state, the table arena, the engine's global state and the
// functions that cross file boundaries.  Nothing in here is part of the C-ABI (include/segalign_amd.h).
//
//   arena.hip          the VMM table arena (background mapping, trim, teardown)
//   options.hip        engine state, tunables, the option table (sa_set_option ...), knobs
//   pool.hip           slots, the (device, slot) token pool, the persistent host worker pool
//   profile.hip        HIP-event profiling of the engine's own streams (sa_profile_*)
//   front.hip          the front of a call: seed upload / device seeding, neighbourhood table, position probe, drop-in check
//   core.hip           saf_core: iteration plan -> filter levels -> exact extension -> ordering / de-duplication -> return vectors
//   api_setup.hip      InitializeInterface / InitializeProcessor / Shutdown, target + query upload, GenerateSeedPosTable
//   api_calls.hip      SeedAndFilter and its additive forms (range, chunks, interval, call lists), ExtendHits, DeviceMakeSeeds
//   api_rm.hip         repeat-masker entries and the device-side coverage post-processing
//   api_introspect.hip statistics, lookup mode, copies of device state for the tests
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/segalign_amd.h"
#include "kernels.h"
#include "plan.h"
#include "probe.h"
#include "join.h"

namespace sa {

// ------------------------------------------------------------------------------------------------------------------
// error handling -- exit codes of common/cuda_utils.h:4-37 (+15 for launches, which the reference never checks)
// ------------------------------------------------------------------------------------------------------------------
static void die(int code, const char* what, const char* tag, hipError_t err) {
    fprintf(stderr, "Error: %s for %s failed with error \" %s \" \n", what, tag, hipGetErrorString(err));
    exit(code);
}
static inline void check_set_device(int dev, const char* tag) {
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) die(11, "hipSetDevice", tag, e);
}
static inline void* dev_malloc(size_t bytes, const char* tag) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        fprintf(stderr, "Error: hipMalloc of %lu bytes for %s failed with error \" %s \" \n", (unsigned long)bytes, tag,
                hipGetErrorString(e));
        exit(12);
    }
    return p;
}
static inline void check_memcpy(hipError_t e, const char* tag) {
    if (e != hipSuccess) die(13, "hipMemcpy", tag, e);
}
static inline void dev_free(void* p, const char* tag) {
    if (!p) return;
    hipError_t e = hipFree(p);
    if (e != hipSuccess) die(14, "hipFree", tag, e);
}
static inline void check_launch(const char* tag) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) die(15, "kernel launch", tag, e);
}
static inline void check_sync(hipStream_t s, const char* tag) {
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) die(15, "hipStreamSynchronize", tag, e);
}

// A slot's share of the WORK ARENA (arena.hip): device memory that is already mapped and page-cleared when the first calls arrive.
// A fresh hipMalloc costs 25-60 ms per GiB on this platform, and the first pass after engine start allocated ~3 GB per slot inside
// its calls (140-180 ms instead of 95); buffers attached to a region carve their memory out of it instead (bump
// allocation: a buffer that regrows abandons its old piece until the slot is destroyed) and fall back to hipMalloc when the region
// is exhausted or not mapped yet.
struct Arena;
struct WorkRegion {
    Arena* arena = nullptr;
    size_t off = 0, size = 0, used = 0;  // [off, off + size) of the arena's range
    void* take(size_t bytes);            // nullptr: no room (or that part of the arena is not mapped yet)
};

template <typename T>
struct DevBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;  // elements
    WorkRegion* region = nullptr;  // (optional) where the memory comes from
    bool in_region = false;
    void ensure(size_t n, const char* tag, bool keep = false, hipStream_t s = 0) {
        if (n <= cap) return;
        size_t ncap = std::max(n, cap + cap / 2);
        if (!keep && p && !in_region) {  // nothing to carry over: give the old buffer back FIRST (peak = max(old, new), not old + new)
            dev_free(p, tag);
            p = nullptr;
            cap = 0;
        }
        T* np = region ? (T*)region->take(ncap * sizeof(T)) : nullptr;
        const bool nr = np != nullptr;
        if (!np) np = (T*)dev_malloc(ncap * sizeof(T), tag);
        if (keep && p && cap) {
            check_memcpy(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s), tag);
            check_sync(s, tag);
        }
        if (!in_region) dev_free(p, tag);
        p = np;
        cap = ncap;
        in_region = nr;
    }
    void release(const char* tag) {
        if (!in_region) dev_free(p, tag);
        p = nullptr;
        cap = 0;
        in_region = false;
    }
};

struct SeqBuf {  // encoded sequence with SEQ_PAD guard bytes on both sides; the allocation is kept and reused (grow-only):
                 // hipMalloc / hipFree synchronise the device, which would stall the calls running on the other query buffer
    uint8_t* alloc = nullptr;
    uint8_t* codes = nullptr;
    uint32_t len = 0;
    size_t cap = 0;
    void create(uint32_t n, const char* tag, hipStream_t s, bool row_coded = false) {
        size_t bytes = (size_t)n + 2 * SEQ_PAD + 64;  // +64: the k-mer window reads 32 bytes from any position
        if (!alloc || cap < bytes) {
            dev_free(alloc, tag);
            cap = bytes + bytes / 16;
            alloc = (uint8_t*)dev_malloc(cap, tag);
        }
        // guard bytes carry bit 6: OR-ed into a matrix index they select a terminator entry of the extension kernels'
        // 128-entry table, so a window that runs over a block edge stops the walk without any bounds arithmetic.
        // Below the guard bit they hold the code 7 ('E', the record separator) in the buffer's own coding, so a reader
        // that ignores the guard bit still sees a separator there
        check_memcpy(hipMemsetAsync(alloc, row_coded ? 0x78 : 0x47, bytes, s), tag);
        codes = alloc + SEQ_PAD;
        len = n;
    }
    void clear() {  // ClearQuery / ClearRef: the block is gone, the memory stays with the engine
        codes = nullptr;
        len = 0;
    }
    void release(const char* tag) {
        dev_free(alloc, tag);
        alloc = codes = nullptr;
        len = 0;
        cap = 0;
    }
};

extern int g_q2_copies;  // 2-bit copies of a query strand the engine builds: 1 (the class filter's one-copy form), 16 with option cls_one_copy = 2

struct PackedBuf {  // phase copies of a packed sequence (packed X-drop filter): copy k at base + k*stride; grow-only like SeqBuf
    uint8_t* alloc = nullptr;
    uint8_t* base = nullptr;
    size_t stride = 0;
    size_t cap = 0;
    void reserve(size_t bytes, const char* tag) {
        if (alloc && cap >= bytes) return;
        dev_free(alloc, tag);
        cap = bytes + bytes / 16;
        alloc = (uint8_t*)dev_malloc(cap, tag);
    }
    void create(const uint8_t* codes, uint32_t len, int bits, const char* tag, hipStream_t s) {
        if (bits == 2) {  // overlapped-line layout (encode.hip): no separate pads, the layout carries its own bias
            const uint32_t nphys = pack2_phys_bytes(len);
            stride = nphys;
            reserve(stride * 4, tag);
            base = alloc;
            launch_pack2_phases(codes, len, base, stride, nphys, s);  // writes every physical byte (0 outside the block)
            return;
        }
        const uint32_t nbytes = len / 2 + 1;
        stride = ((size_t)nbytes + 2 * PACK_PAD + 127) & ~(size_t)127;
        reserve(stride * PACK4_COPIES, tag);
        // pads read as code 7 in both nibbles (any content keeps the filter's scores upper bounds; this makes a walk that
        // leaves the block die quickly under the default matrices)
        check_memcpy(hipMemsetAsync(alloc, 0x77, stride * PACK4_COPIES, s), tag);
        base = alloc + PACK_PAD;
        launch_pack4_phases(codes, len, base, stride, nbytes, s);
    }
    // the sixteen shifted 2-bit copies of a query strand (class filter, encode.hip): every byte is written, no pads
    void create_q2(const uint8_t* codes, uint32_t len, const char* tag, hipStream_t s) {
        stride = q2_copy_stride(len);
        copies = (uint32_t)g_q2_copies;
        reserve(stride * copies, tag);
        base = alloc;
        launch_pack2_shifted(codes, len, base, stride, copies, s);
    }
    uint32_t copies = 0;  // (2-bit query copies: how many create_q2 built)
    void clear() {
        base = nullptr;
        stride = 0;
    }
    void release(const char* tag) {
        dev_free(alloc, tag);
        alloc = base = nullptr;
        stride = 0;
        cap = 0;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// profiling: HIP events on the engine's own streams
// ------------------------------------------------------------------------------------------------------------------
struct ProfSpan {
    int dev;
    float t0, t1;  // ms since the device's epoch event
};
struct ProfEntry {
    std::string name;
    double total_ms = 0;
    uint64_t launches = 0;
    std::vector<ProfSpan> spans;  // when every launch ran (several slots overlap): sa_profile_busy_ms
};
extern std::mutex g_prof_mu;
extern std::vector<ProfEntry> g_prof;
extern bool g_prof_on;
constexpr int PROF_MAX_DEV = 16;
extern hipEvent_t g_prof_epoch[PROF_MAX_DEV];  // per device: recorded by sa_profile_reset
extern int g_trace_scopes;  // option debug >= 2: synchronise after every kernel scope and name it on stderr (fault localisation)

struct Slot;
struct ProfRec {
    int id;
    hipEvent_t e0, e1;
};
int prof_id(const char* name);

// ------------------------------------------------------------------------------------------------------------------
// per-device state
// ------------------------------------------------------------------------------------------------------------------
constexpr int MAX_SLOTS_PER_DEVICE = 8;
extern uint32_t SPEC_RECS;        // records of the speculative output copy (256 KB); option spec_recs (tests)
extern uint32_t g_dedup_seg_max;  // option dedup_seg_max: records per segment the LDS chain accepts (0 = its LDS capacity; tests)
constexpr int SA_MAX_CHUNKS = 256;  // chunks one multi-chunk call may carry: 2 reference iterations each = MAX_SEGS segments
static_assert(SA_MAX_CHUNKS == (int)JOIN_SEG_FIRST && 2 * SA_MAX_CHUNKS <= MAX_SEGS,
              "a call's chunks x 2 reference iterations = MAX_SEGS segments (d_seg_end holds MAX_SEGS u64); the join path's segment table (join.h) is cut at SA_MAX_CHUNKS");
constexpr int SA_MAX_CHUNKS_GENERAL = 32;  // ... when it takes the general path (per-chunk iteration plans of up to 1000 iterations each)
constexpr int SA_DEFAULT_CHUNKS = 40;  // ... and what the interval entries hand to one call: the 40 chunks of a strand of a 10 Mbp interval
extern int SLOTS_PER_DEVICE;  // calls in flight per device (the reference allows one: token == device); option slots

struct Counters {  // device-side scalars of one slot
    uint32_t survivors;
    uint32_t uniq;
    uint32_t uniq2;
    uint32_t pad;
    unsigned long long examined;         // E: positions the reference algorithm scores
    unsigned long long examined_filter;  // positions the filter kernel scored (partial walks of candidates included)
    uint32_t n_long;  // candidates the filter forwarded to the exact kernel (this batch)
    uint32_t n_ent;   // entropy candidates (this batch)
    uint32_t n_heads; // run heads of the chain shortcut (this batch)
    uint32_t n_l2;    // hits the context filter handed to the second level (this batch)
    uint32_t n_l2_max;  // largest sub-list of them (compared with the sub-list capacity)
    uint32_t n_audit;   // (tests) hits the filters rejected, see the audit option
    uint32_t n_chain_big;  // bucket groups the chain sort left unordered (a bucket above its LDS capacity)
    uint32_t pad2;
};

struct DevCtx;
struct Slot {
    int dev = 0;
    DevCtx* ctx = nullptr;            // the device context this slot belongs to
    hipStream_t stream = nullptr;
    hipStream_t stream_lo = nullptr;  // option filter_prio: a lowest-priority stream the class filter alone is launched on (the slot's own stream
    hipEvent_t ev_lo_a = nullptr, ev_lo_b = nullptr;  // is then created with the highest priority), bracketed by these two events
    DevBuf<uint64_t> seeds;
    DevBuf<uint32_t> start, count, flags, flag_prefix;
    DevBuf<uint64_t> prefix;
    DevBuf<uint8_t> scan_temp, sort_temp, unique_temp;
    DevBuf<Hit> hits;
    DevBuf<HspRec> recA, recB;
    DevBuf<CandRec> cand_list;
    DevBuf<L2Rec> l2_list;
    DevBuf<uint2> audit;                    // (tests) rejected hits of the filter levels
    DevBuf<uint32_t> l2_counts;             // sub-list counters (one 128-byte line each)
    DevBuf<CandRec> chain_tmp, chain_sorted;  // chain shortcut of the exact stage
    DevBuf<uint32_t> chain_is_head, chain_heads, chain_bucket_cnt, chain_bucket_start;
    DevBuf<EntRec> ent_list;
    DevBuf<sa_segment_pair> out16;
    // repeat-masker coverage (coverage.hip): difference array over the block + scan/compaction scratch
    DevBuf<uint32_t> cov_diff, cov_pre, cov_is_start, cov_is_end, cov_sidx, cov_eidx, cov_pairs;
    uint32_t* d_cov_range = nullptr;  // {min query_start, max query_start+len} touched since the last reset
    uint32_t* h_cov = nullptr;        // pinned: range + per-tile totals
    // table-direct path (probe.hip): per-position scratch, compacted non-empty positions, chunk plans
    DevBuf<uint64_t> td_toff;
    DevBuf<uint32_t> td_tcnt;
    DevBuf<TdRec> td_rec;
    DevBuf<uint32_t> td_chunk;
    DevBuf<uint32_t> td_bits;         // head-bit map of the call's hits (class filter)
    DevBuf<uint8_t> td_partial;
    void* d_td_bounds = nullptr;
    TdPlan* d_td_plan = nullptr;
    TdPlan* h_td_plan = nullptr;      // pinned
    IterPlan* d_plan = nullptr;       // SA_MAX_CHUNKS_GENERAL plans (one per chunk of a multi-chunk call on the general path)
    uint64_t* d_seg_end = nullptr;    // segment ends of the running batch (ExtendArgs.seg_end)
    uint64_t* h_seg_end = nullptr;    // pinned
    Counters* d_cnt = nullptr;
    uint32_t* d_verify = nullptr;     // drop-in calls: "the host seed vector is what the device seeder would emit" (seeds.hip)
    uint32_t* h_verify = nullptr;     // pinned
    DevBuf<uint32_t> out_seg;         // segment id of every final record (multi-chunk calls split their output by it)
    uint32_t* d_seg_info = nullptr;   // per-segment counts / offsets of the LDS dedup (dedup.hip)
    uint32_t* h_seg_info = nullptr;   // pinned
    uint32_t* h_seg = nullptr;        // pinned
    size_t h_seg_cap = 0;
    uint32_t* h_bounds = nullptr;     // pinned: flag-prefix values at the chunk boundaries of a multi-chunk call
    // pinned host staging
    IterPlan* h_plan = nullptr;
    Counters* h_cnt = nullptr;
    uint64_t* h_seeds = nullptr;
    size_t h_seeds_cap = 0;
    sa_segment_pair* h_out = nullptr;
    size_t h_out_cap = 0;
    // profiling
    std::vector<ProfRec> prof_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
    size_t events_used = 0;
    // key-ordered calls (join.h / join.hip): the call's positions sorted by key, entries by class, the query field words
    DevBuf<uint32_t> jq_keys, jq_pairs, jq_misc, jq_start, jq_pos, jq_ent_nt, jq_qx;
    DevBuf<uint4> jq_ent;
    DevBuf<unsigned long long> jq_vstart, jq_stats;
    DevBuf<uint8_t> jq_scan;
    JoinHead* d_jhead = nullptr;
    JoinChunk* d_jplan = nullptr;
    JoinChunk* h_jplan = nullptr;     // pinned
    uint32_t jq_chunk = 0;            // chunk size of the running key-ordered call
    WorkRegion work;                  // this slot's share of the device's work arena
};

// ------------------------------------------------------------------------------------------------------------------
// Table arena: the device memory of the neighbourhood table, obtained through the virtual-memory API in 1 GiB chunks that a
// BACKGROUND thread maps behind each other into one reserved address range.  Why: the first allocation of tens of GB in a process
// costs 25-60 ms per GiB on this platform (tools/micro/alloc_cost*.hip: the driver hands out cleared pages) -- 1 s for the 36 GB
// table of a 100 Mbp block, ~6 s for a 500 Mbp block -- and a plain hipMalloc pays it inside GenerateSeedPosTable.  The arena
// starts growing at InitializeProcessor (option arena_gb), i.e. while the host is still reading its FASTA files
// (src/main.cpp:298 comes before :300-549), the table build only waits for the bytes it needs, a larger block just raises the
// goal (no reallocation, no copy), and a block that needs less keeps what is mapped.
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t ARENA_CHUNK = (size_t)1 << 30;
struct Arena {
    int dev = 0;
    uint8_t* base = nullptr;   // reserved virtual range (VMM) or the plain allocation (fallback)
    size_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    size_t mapped = 0;         // bytes usable from `base` on (guarded by mu)
    size_t goal = 0;           // the worker maps until mapped >= goal
    bool failed = false;       // a chunk could not be obtained: out of memory at `mapped`
    bool stop = false;
    bool busy = false;         // the worker thread is running
    bool vmm = true;           // false: no virtual-memory API here -> one synchronous hipMalloc per growth
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
};

void arena_request(Arena& A, size_t bytes);   // ask for `bytes` usable bytes (asynchronously); never shrinks
bool arena_wait(Arena& A, size_t bytes);      // block until `bytes` are usable; false: they cannot be had (out of memory)
void arena_settle(Arena& A, size_t need);     // the block's need is known and mapped: stop mapping ahead
void arena_trim(Arena& A, size_t keep);       // give everything beyond `keep` bytes back to the device (the worker is stopped first)
void arena_destroy(Arena& A);
size_t arena_mapped(Arena& A);
Arena& arena_of(int key, int ordinal);        // one arena per device context slot for the life of the process
void arena_stop_all();                        // stop every background worker (ShutdownProcessor, atexit); what is mapped stays
void arena_release_all();                     // give every arena back to its device (sa_release_arena)

struct DevCtx {
    int dev = 0;                         // HIP device ordinal
    int index = 0;                       // engine device number g (position in g_dev; what sa_call_stats.device reports)
    hipStream_t admin = nullptr;
    size_t total_mem = 0;
    int* d_sub_mat = nullptr;
    SeqBuf ref;
    SeqBuf ref8;                         // row-coded copy (code << 3) read by the extension kernel
    const char* ref_host_ptr = nullptr;  // identity of the block last sent (to skip a second upload for the table)
    PackedBuf ref2;                      // 2-bit phase copies of the target (packed filter)
    PackedBuf query4[SA_BUFFER_DEPTH], query4_rc[SA_BUFFER_DEPTH];  // 4-bit phase copies of the query strands
    PackedBuf query2[SA_BUFFER_DEPTH], query2_rc[SA_BUFFER_DEPTH];  // 2-bit shifted copies of the query strands (class filter)
    PackedBuf ref4, ref4_rc;             // repeat masker: the query IS the target
    PackedBuf refq2, refq2_rc;           // ... and its 2-bit shifted copies
    uint32_t* d_present = nullptr;       // code-presence masks on the device: [0] target, [1 + b] query buffer b
    uint32_t ref_present = 0xFFu;        // which of the 8 codes occur in the resident target / query blocks (class_scores)
    uint32_t query_present[SA_BUFFER_DEPTH] = {0xFFu, 0xFFu};
    SeqBuf ref_rc;                       // repeat masker
    uint32_t* bucket_start = nullptr;    // 4^k + 1
    uint32_t* pos_table = nullptr;
    DevBuf<uint32_t> keep_bucket, keep_pos;  // their memory: kept across target blocks (g_ClearRef only forgets the tables) -- a fresh
    DevBuf<uint64_t> keep_nbr_start;         // hipMalloc pays first-touch page clearing, 25-60 ms per GiB, inside the next block's table build
    uint32_t num_index = 0;
    uint32_t nkeys = 0;
    // sequence upload: ASCII goes through a ring of two pinned buffers into a reused device staging buffer, so the copies
    // are real asynchronous DMA on the admin stream (a pageable hipMemcpyAsync is staged synchronously by the runtime) and
    // nothing is allocated or freed per block (the reference mallocs + frees a temp per call, seed_filter_interface.cu:90-99,
    // src/seed_filter.cu:905-918)
    DevBuf<uint8_t> up_tmp;
    uint8_t* up_pinned[2] = {nullptr, nullptr};
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    // neighbourhood table (probe.hip): per key the concatenation of the buckets of the key's seed words
    std::mutex nbr_mu;
    uint64_t* nbr_start = nullptr;       // nkeys + 1
    uint32_t* nbr_pos = nullptr;         // == pos_table when no transition word exists (nbr_alias); null when nbr_ctx is built
    uint32_t nbr_left_skip = 0;          // what the records of nbr_ctx were cut with (CtxRec): seed_size, or 0 under option ctx_skip_seed = 0
    CtxRec* nbr_ctx = nullptr;           // the runs WITH their target context: 32-byte records in the arena (class filter, extend.hip 1d)
    Arena& work_arena;                   // cleared device memory for the slots' work buffers (WorkRegion), mapped in the background like `arena`
    Arena& arena;                        // memory of the context table: outlives target blocks AND engine contexts (a process-wide
                                         // cache per device ordinal, see arena_of), grown in the background
    DevCtx(Arena& a, Arena& w) : work_arena(w), arena(a) {}
    bool nbr_alias = false;
    uint64_t nbr_total = 0;
    uint32_t nbr_tmask = 0;
    int nbr_state = 0;                   // 0: not built, 1: ready, -1: not available for this table (memory, 32-bit run lengths)
    SeqBuf query[SA_BUFFER_DEPTH], query_rc[SA_BUFFER_DEPTH];
    Slot slots[MAX_SLOTS_PER_DEVICE];
};

extern int g_ndev;
extern std::vector<int> g_selected;
extern std::vector<DevCtx*> g_dev;
extern std::mutex g_mu;
extern std::condition_variable g_cv;
extern std::vector<std::pair<int, int>> g_tokens;
extern bool g_proc_init;
extern int g_transition;
extern uint32_t g_wga_chunk;
extern uint32_t g_seed_size;
extern int g_sub_mat[64];
extern int g_xdrop, g_hspthresh, g_noentropy;
extern int g_log4_double, g_entropy_ulps, g_table_scratch_arena, g_ctx_skip_seed;  // options log4_double (H2), entropy_ulps (H13, tests)
extern int64_t g_max_seeds;
extern int64_t g_max_hits;
extern bool g_max_hits_overridden;
extern bool g_count_examined;
extern int g_fin_batch;
extern int g_bufs_per_wave;
extern int g_long_cap;
extern int g_long_blocks;
extern int g_packed_waves;
extern int g_ctx_waves;
extern uint32_t g_l2_cap_test;
extern int g_nbr_two_stage;
extern int g_table_atomic;
extern int64_t g_arena_gb;
extern int g_ctx_threads;
extern int g_dedup_threads;
extern int g_spec_dedup;
extern int g_l2_blocks;
extern int g_max_waves;
extern int g_fast_filter;
extern int g_packed_filter;
extern int g_chain_sort_threads, g_chain_buckets, g_chain_bucket_target, g_chain_sort_blocks, g_chain_group_max;
extern int g_chunks_per_call;
extern int g_key_order, g_key_order_chunks;
extern int64_t g_key_order_hits, g_key_order_min_pos;
extern int64_t g_call_hits, g_call_hits_max;
extern int g_no_small_dedup;
extern int g_ctx;
extern uint32_t g_audit_cap;
extern int g_td;
extern int g_chain;
extern uint32_t CHAIN_CAP;
extern SeedShape g_shape;
extern uint32_t g_query_len[SA_BUFFER_DEPTH];
extern thread_local sa_call_stats t_stats;
extern thread_local std::vector<uint2> t_audit;
extern thread_local uint32_t t_front_flags;  // SA_PATH_* bits the front of the calling thread's current call has set (front.hip -> core.hip)

int64_t opt_value(const char* name);  // the option table (options.hip)
void resolve_options();
void require_init(const char* who);
void require_proc(const char* who, uint32_t buffer);
void class_scores(uint32_t present_t, uint32_t present_q, int cls[4]);
int max_hits_for_mem(uint64_t total_global_mem);

// ---- profiling scope ------------------------------------------------------------------------------------------------
struct ProfScope {
    Slot* sl;
    bool on;
    const char* nm;
    ProfRec r;
    ProfScope(Slot* s, const char* name) : sl(s), on(g_prof_on), nm(name) {
        if (g_trace_scopes) fprintf(stderr, "[scope] %s ...\n", name);
        if (!on) return;
        if (sl->events_used == sl->event_pool.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            sl->event_pool.push_back({a, b});
        }
        r.id = prof_id(name);
        r.e0 = sl->event_pool[sl->events_used].first;
        r.e1 = sl->event_pool[sl->events_used].second;
        sl->events_used++;
        hipEventRecord(r.e0, sl->stream);
    }
    ~ProfScope() {
        if (g_trace_scopes) {
            hipError_t e = hipStreamSynchronize(sl->stream);
            fprintf(stderr, "[scope] %s done (%s)\n", nm, hipGetErrorString(e));
        }
        if (!on) return;
        hipEventRecord(r.e1, sl->stream);
        sl->prof_pending.push_back(r);
    }
};
void prof_flush(Slot* sl);  // call after the slot's stream has been synchronised

// ---- slots, token pool, worker pool (pool.hip) ----
Slot* acquire_slot();
void release_slot(Slot* s);
void slot_init(Slot& s, DevCtx* dc);
void slot_destroy(Slot& s);
void run_parallel(size_t n, int threads, std::function<void(size_t)> fn);

// ------------------------------------------------------------------------------------------------------------------
// SeedAndFilter core: seeds already in slot->seeds (device), n of them.
// ------------------------------------------------------------------------------------------------------------------
struct CoreArgs {
    const uint8_t* query;
    uint32_t query_len;
    int rm;
    int rm_rev;
    uint32_t rm_win_start, rm_win_end;
    uint32_t q_lo, q_hi;  // query positions of the seed words lie in [q_lo, q_hi) when the caller knows it (0,0 otherwise)
    // repeat-masker coverage accumulation (sa_rm_mask_interval): the final HSPs of the call are counted into this
    // difference array on the device instead of being returned
    uint32_t* cov_diff;
    uint32_t cov_diff_len;
    const PackedBuf* query4;  // 4-bit phase copies of `query` (nullptr: the packed filter is not used for this call)
    // multi-chunk call: the seed vector holds `nchunks` consecutive chunks, chunk c = seeds [seed_bound[c], seed_bound[c+1]);
    // every chunk gets its own iteration plan, dedup scope and output vector (exactly what nchunks separate calls give)
    int nchunks;                          // 0 or 1: ordinary call
    uint32_t seed_bound[SA_MAX_CHUNKS + 1];
    sa_segment_pair** outs;               // [nchunks]
    size_t* counts;                       // [nchunks]
    // table-direct call (td_front has filled sl->h_td_plan and the compacted position arrays): no seed words, no per-word
    // extents, no hit list; the filter reads its anchors out of the neighbourhood table
    int td;
    uint32_t td_words;                    // seed words per valid position (1 + transition positions)
    // sa_extend_hits (introspection): sl->hits already holds raw_hits anchors; one iteration, no lookup, no dedup -- the
    // survivors of the extension stage (find_hsps + done-flag compaction) are returned as they are
    uint64_t raw_hits;
    // class filter: 2-bit shifted copies of this call's strand and of the other strand, code presence of the query block
    const PackedBuf* q2_own;
    const PackedBuf* q2_other;
    uint32_t q_present;
    int join;                             // key-ordered call (join_front has sorted the positions and planned the chunks): td is set as well
};

size_t saf_core(DevCtx* dc, Slot* sl, uint32_t num_seeds, const CoreArgs& ca, sa_segment_pair** out);

// ---- the front of a call (front.hip) ----
void upload_seeds(Slot* sl, const uint64_t* seeds, size_t n);
uint32_t device_seeds(Slot* sl, const uint8_t* qcodes, uint32_t start, uint32_t end, int nb = 0, const uint32_t* bpos = nullptr,
                      uint32_t* bseed = nullptr);
uint32_t seed_tmask();
void nbr_release(DevCtx* dc);
bool ensure_nbr(DevCtx* dc);
bool q2_usable(const PackedBuf* q2_own, const PackedBuf* q2_other);
bool td_eligible(DevCtx* dc, const PackedBuf* query4, const PackedBuf* q2_own, const PackedBuf* q2_other);
uint32_t td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, int K, const uint32_t* bpos, int rm, uint32_t* words_out);
bool join_wanted(DevCtx* dc, int K, uint32_t n_positions);  // should this call take the key-ordered form?
uint32_t join_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, uint32_t qlen, int K, const uint32_t* bpos, const PackedBuf* q2_own, const PackedBuf* q2_other,
                    uint32_t* words_out);
uint32_t dropin_td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, uint32_t qlen, const uint64_t* host_seeds, size_t n,
                         const PackedBuf* q4, const PackedBuf* q2_own, const PackedBuf* q2_other, int rm, uint32_t* first_out,
                         uint32_t* end_out, uint32_t* words_out);
void set_query2(CoreArgs& ca, DevCtx* dc, uint32_t buffer, int rev);
void set_query2_rm(CoreArgs& ca, DevCtx* dc, int rev);
extern int g_seed_upload;
extern size_t g_q2_limit;  // bytes the sixteen 2-bit copies of a query strand may span (option q2_limit_mb; 4 GiB)

// ---- set-up helpers (api_setup.hip) ----
const uint8_t* upload_ascii(DevCtx* dc, const char* src, size_t len, const char* tag);
void presence_of(DevCtx* dc, const uint8_t* codes, uint32_t len, int slot, uint32_t* host_mask);

}  // namespace sa
