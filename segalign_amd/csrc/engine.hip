// engine.hip -- host side of libsegalign_hip.so: device contexts, the (device, slot) token pool, workspace
// management, the SeedAndFilter orchestration and the C-ABI of include/segalign_amd.h.
//
// Mirrors, entry point by entry point, the reference engine:
//   common/seed_filter_interface.cu (InitializeInterface, SendRefWriteRequest, ClearRef)
//   common/seed_pos_table.cu        (GenerateSeedPosTable -- here a DEVICE build)
//   src/seed_filter.cu              (InitializeProcessor, SendQueryWriteRequest, ClearQuery, SeedAndFilter,
//                                    ShutdownProcessor)
//   repeat_masker_src/seed_filter.cu (the self-alignment variant)
// There is no CPU fallback: without a HIP device every entry point exits exactly like the reference does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/segalign_amd.h"
#include "kernels.h"
#include "plan.h"
#include "probe.h"

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

template <typename T>
struct DevBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;  // elements
    void ensure(size_t n, const char* tag, bool keep = false, hipStream_t s = 0) {
        if (n <= cap) return;
        size_t ncap = std::max(n, cap + cap / 2);
        T* np = (T*)dev_malloc(ncap * sizeof(T), tag);
        if (keep && p && cap) {
            check_memcpy(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s), tag);
            check_sync(s, tag);
        }
        dev_free(p, tag);
        p = np;
        cap = ncap;
    }
    void release(const char* tag) {
        dev_free(p, tag);
        p = nullptr;
        cap = 0;
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
        reserve(stride * Q2_COPIES, tag);
        base = alloc;
        launch_pack2_shifted(codes, len, base, stride, s);
    }
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
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;
static bool g_prof_on = false;
constexpr int PROF_MAX_DEV = 16;
static hipEvent_t g_prof_epoch[PROF_MAX_DEV] = {};  // per device: recorded by sa_profile_reset
static int g_trace_scopes = 0;  // option debug >= 2: synchronise after every kernel scope and name it on stderr (fault localisation)

struct Slot;
struct ProfRec {
    int id;
    hipEvent_t e0, e1;
};

static int prof_id(const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = 0; i < g_prof.size(); i++)
        if (g_prof[i].name == name) return (int)i;
    ProfEntry e;
    e.name = name;
    g_prof.push_back(e);
    return (int)g_prof.size() - 1;
}

// ------------------------------------------------------------------------------------------------------------------
// per-device state
// ------------------------------------------------------------------------------------------------------------------
constexpr int MAX_SLOTS_PER_DEVICE = 4;
static uint32_t SPEC_RECS = 16384;    // records of the speculative output copy (256 KB); SEGALIGN_AMD_SPEC_RECS (tests)
static uint32_t g_dedup_seg_max = 0;   // SEGALIGN_AMD_DEDUP_SEG_MAX: records per segment the LDS chain accepts (0 = its LDS capacity; tests)
constexpr int SA_MAX_CHUNKS = 32;  // chunks one multi-chunk call may carry: 2 reference iterations each = MAX_SEGS segments
constexpr int SA_DEFAULT_CHUNKS = 20;  // ... and what the interval entries hand to one call: the 40 chunks of a 10 Mbp strand go as 20 + 20
static int SLOTS_PER_DEVICE = 4;  // calls in flight per device (the reference allows one: token == device); option slots

// Every slot issues its kernels on a stream of its own, next to the upload stream.  The HIP runtime multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4), and two slots that share a queue run one after the other: with four slots the
// small kernels of one call then wait behind another call's filter kernel instead of overlapping it (0.94 -> 1.03 Gbp/s on the
// default workload with 8 queues).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process;
// a value the user has set is left alone.
__attribute__((constructor)) static void default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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
    uint32_t pad2[2];
};

struct Slot {
    int dev = 0;
    hipStream_t stream = nullptr;
    DevBuf<uint64_t> seeds;
    DevBuf<uint32_t> start, count, flags, flag_prefix;
    DevBuf<uint64_t> prefix;
    DevBuf<uint8_t> scan_temp, sort_temp;
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
    IterPlan* d_plan = nullptr;       // SA_MAX_CHUNKS plans (one per chunk of a multi-chunk call)
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

static void arena_worker(Arena* A) {
    hipSetDevice(A->dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = A->dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::unique_lock<std::mutex> lk(A->mu);
    while (!A->stop && !A->failed && A->mapped < A->goal && A->mapped + ARENA_CHUNK <= A->va_bytes) {
        uint8_t* at = A->base + A->mapped;
        lk.unlock();
        hipMemGenericAllocationHandle_t h;
        bool ok = hipMemCreate(&h, ARENA_CHUNK, &prop, 0) == hipSuccess;
        if (ok && hipMemMap(at, ARENA_CHUNK, 0, h, 0) != hipSuccess) { hipMemRelease(h); ok = false; }
        if (ok && hipMemSetAccess(at, ARENA_CHUNK, &acc, 1) != hipSuccess) { hipMemUnmap(at, ARENA_CHUNK); hipMemRelease(h); ok = false; }
        lk.lock();
        if (ok) {
            A->chunks.push_back(h);
            A->mapped += ARENA_CHUNK;
        } else {
            (void)hipGetLastError();
            A->failed = true;
        }
        A->cv.notify_all();
    }
    A->busy = false;
    A->cv.notify_all();
}

// ask for `bytes` usable bytes (asynchronously); never shrinks
static void arena_request(Arena& A, size_t bytes) {
    std::unique_lock<std::mutex> lk(A.mu);
    if (!A.vmm) return;
    if (!A.base) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)288 << 30;
        A.va_bytes = (total_b + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
        void* va = nullptr;
        if (hipMemAddressReserve(&va, A.va_bytes, 0, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            A.vmm = false;
            A.va_bytes = 0;
            return;
        }
        A.base = (uint8_t*)va;
    }
    const size_t want = std::min(A.va_bytes, (bytes + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK);
    if (want > A.goal) {
        A.goal = want;
        A.failed = false;
    } else if (A.failed && want > A.mapped) {
        A.failed = false;  // (memory may have been given back since the last attempt)
    }
    if (!A.busy && !A.failed && A.mapped < A.goal) {
        if (A.worker.joinable()) { lk.unlock(); A.worker.join(); lk.lock(); }
        A.busy = true;
        A.stop = false;
        A.worker = std::thread(arena_worker, &A);
    }
}
// block until `bytes` are usable; false: they cannot be had (out of memory)
static bool arena_wait(Arena& A, size_t bytes) {
    arena_request(A, bytes);
    std::unique_lock<std::mutex> lk(A.mu);
    if (!A.vmm) {  // fallback: a plain allocation of exactly what is needed, kept while it is large enough
        if (A.mapped >= bytes) return true;
        if (A.base) { hipFree(A.base); A.base = nullptr; A.mapped = 0; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        A.base = (uint8_t*)p;
        A.mapped = bytes;
        return true;
    }
    A.cv.wait(lk, [&] { return A.mapped >= bytes || A.failed || !A.busy; });
    return A.mapped >= bytes;
}
// The block's need is known and mapped: stop mapping ahead (the default goal is a guess made before any sequence was seen; what is
// mapped stays).  Mapping is page clearing on the device: left running it takes memory bandwidth from the first query pass.
static void arena_settle(Arena& A, size_t need) {
    std::lock_guard<std::mutex> lk(A.mu);
    const size_t n = (need + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
    if (A.goal > std::max(n, A.mapped)) A.goal = std::max(n, A.mapped);
}
// give everything beyond `keep` bytes back to the device (the worker is stopped first)
static void arena_trim(Arena& A, size_t keep) {
    std::unique_lock<std::mutex> lk(A.mu);
    A.stop = true;
    A.goal = std::min(A.goal, (keep + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK);
    if (A.worker.joinable()) { lk.unlock(); A.worker.join(); lk.lock(); }
    A.stop = false;
    if (!A.vmm) {
        if (keep == 0 && A.base) { hipFree(A.base); A.base = nullptr; A.mapped = 0; }
        return;
    }
    while (A.mapped >= ARENA_CHUNK && A.mapped - ARENA_CHUNK >= keep) {
        A.mapped -= ARENA_CHUNK;
        hipMemUnmap(A.base + A.mapped, ARENA_CHUNK);
        hipMemRelease(A.chunks.back());
        A.chunks.pop_back();
    }
    // On this runtime (ROCm 7.2) the pages of an unmapped + released chunk only go back to the device when the ADDRESS RANGE is
    // freed (tools/micro/vmm_info2.hip: every teardown order leaves hipMemGetInfo and the number of creatable chunks unchanged
    // until hipMemAddressFree).  So giving everything back means giving the range back too; the next request reserves a new one.
    if (A.mapped == 0 && A.base) {
        hipMemAddressFree(A.base, A.va_bytes);
        A.base = nullptr;
        A.va_bytes = 0;
        A.goal = 0;
    }
    A.failed = false;
}
static void arena_destroy(Arena& A) {
    arena_trim(A, 0);
    std::lock_guard<std::mutex> lk(A.mu);
    A.vmm = true;
}
static size_t arena_mapped(Arena& A) {
    std::lock_guard<std::mutex> lk(A.mu);
    return A.mapped;
}
// One arena per device ordinal for the life of the process: cleared device pages cost seconds to get, so they are kept across
// ShutdownProcessor / InitializeInterface cycles (option arena_gb = 0 gives them back at ShutdownProcessor).
static Arena& arena_of(int ordinal) {
    static std::mutex mu;
    static std::vector<Arena*> arenas;  // (never destroyed: worker threads may outlive static destruction)
    std::lock_guard<std::mutex> lk(mu);
    if ((int)arenas.size() <= ordinal) arenas.resize((size_t)ordinal + 1, nullptr);
    if (!arenas[ordinal]) {
        arenas[ordinal] = new Arena();
        arenas[ordinal]->dev = ordinal;
    }
    return *arenas[ordinal];
}

struct DevCtx {
    int dev = 0;
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
    CtxRec* nbr_ctx = nullptr;           // the runs WITH their target context: 32-byte records in the arena (class filter, extend.hip 1d)
    Arena& arena;                        // memory of the context table: outlives target blocks AND engine contexts (a process-wide
                                         // cache per device ordinal, see arena_of), grown in the background
    explicit DevCtx(Arena& a) : arena(a) {}
    bool nbr_alias = false;
    uint64_t nbr_total = 0;
    uint32_t nbr_tmask = 0;
    int nbr_state = 0;                   // 0: not built, 1: ready, -1: not available for this table (memory, 32-bit run lengths)
    SeqBuf query[SA_BUFFER_DEPTH], query_rc[SA_BUFFER_DEPTH];
    Slot slots[MAX_SLOTS_PER_DEVICE];
};

static int g_ndev = 0;
static std::vector<int> g_selected;  // sa_select_devices
static std::vector<DevCtx*> g_dev;
static std::mutex g_mu;  // token pool (seed_filter_interface.cu:7-9)
static std::condition_variable g_cv;
static std::vector<std::pair<int, int>> g_tokens;

static bool g_proc_init = false;
static int g_transition = 1;
static uint32_t g_wga_chunk = 250000;
static uint32_t g_seed_size = 19;
static int g_sub_mat[64];
static int g_xdrop = 910, g_hspthresh = 3000, g_noentropy = 0;
static int64_t g_max_seeds = 0;
static int64_t g_max_hits = 0;
static bool g_max_hits_overridden = false;
static bool g_count_examined = false;
static int g_fin_batch = 48;      // SEGALIGN_AMD_FIN_BATCH
static int g_bufs_per_wave = 8;   // SEGALIGN_AMD_BUFS_PER_WAVE
static int g_long_cap = 128;      // SEGALIGN_AMD_LONG_CAP: bases per side before a hit goes to the long kernel
static int g_long_blocks = 1792;  // SEGALIGN_AMD_LONG_BLOCKS: grid of the long kernel (4 waves per block)
static int g_packed_waves = 4096; // SEGALIGN_AMD_PACKED_WAVES: waves of the packed filter (2 workgroups of 8 waves per CU measured best: 3072 +16 %, 6144 +20 %, 8192 +14 %)
static int g_ctx_waves = 0;       // SEGALIGN_AMD_CTX_WAVES: wave budget of the context filter; 0 = one 4096-hit chunk per wave (measured best)
static uint32_t g_l2_cap_test = 0; // SEGALIGN_AMD_L2_CAP
static int g_nbr_two_stage = 1;   // SEGALIGN_AMD_NBR_ONE_STAGE=1: every table entry cuts its own context out of the target
static int g_table_atomic = 0;    // option table_atomic: build the seed table with the atomic counting sort even where the partition build applies
static int64_t g_arena_gb = 40;   // option arena_gb: GiB of table arena the engine starts mapping at InitializeProcessor (0: on demand only)
static int g_ctx_threads = 0;     // SEGALIGN_AMD_CTX_THREADS: workgroup size of the context filter (0 = kernel default)
static int g_dedup_threads = 0;   // SEGALIGN_AMD_DEDUP_THREADS: workgroup size of the per-segment LDS chain (0 = 1024)
static int g_spec_dedup = 1;      // SEGALIGN_AMD_SPEC_DEDUP=0: wait for the survivor count before the LDS chain (one more host sync)
static int g_l2_blocks = 512;     // SEGALIGN_AMD_L2_BLOCKS: workgroups of the second-level packed filter
static int g_max_waves = 4096;    // SEGALIGN_AMD_MAX_WAVES: waves of the filter kernel (4 per SIMD saturate instruction issue)
static int g_fast_filter = 0;     // derived in InitializeProcessor: xdrop >= 0 && 7*max(M) <= xdrop
static int g_packed_filter = 0;   // derived in InitializeProcessor: the packed upper-bound filter may be used
static int g_chain_sort_threads = 256;  // SEGALIGN_AMD_CHAIN_SORT_THREADS
static int g_chunks_per_call = SA_DEFAULT_CHUNKS;  // SEGALIGN_AMD_CHUNKS_PER_CALL: chunks sa_seed_interval hands to one multi-chunk call
static int g_no_small_dedup = 0;  // SEGALIGN_AMD_NO_SMALL_DEDUP=1: always use the library sorts
static int g_ctx = 1;             // neighbourhood table with target context when it fits (SEGALIGN_AMD_NO_CTX=1: positions only)
static uint32_t g_audit_cap = 0;  // SEGALIGN_AMD_AUDIT_CAP (tests): record up to this many hits the filter levels reject per call
static int g_td = 1;              // table-direct lookup (neighbourhood table + position probe, probe.hip); SEGALIGN_AMD_NO_TD=1 turns it off
static int g_chain = 1;           // chain shortcut of the exact stage (SEGALIGN_AMD_NO_CHAIN=1 turns it off)
static uint32_t CHAIN_CAP = 1u << 22;  // candidates per batch the chain buffers hold (SEGALIGN_AMD_CHAIN_CAP); larger batches fall back
static SeedShape g_shape = {0, 0, 0, {0}};
static uint32_t g_query_len[SA_BUFFER_DEPTH] = {0, 0};

static int64_t opt_value(const char* name);  // the option table (below)

static thread_local sa_call_stats t_stats;
static thread_local std::vector<uint2> t_audit;  // rejected hits of the calling thread's last hot call (audit option)

// Class scores of the class filter (extend.hip 1d): cls[x] bounds every matrix entry a base pair with (target code ^ query
// code) == x can have.  Codes >= 4 are stored as code 0 in the 2-bit copies, so a pair with such a code can show up in ANY
// class: its score joins every cls[x] -- but only for the codes that actually occur in the two resident blocks.
static void class_scores(uint32_t present_t, uint32_t present_q, int cls[4]) {
    for (int x = 0; x < 4; x++) {
        int m = INT32_MIN;
        for (int r = 0; r < 4; r++) m = std::max(m, g_sub_mat[r * 8 + (r ^ x)]);
        cls[x] = m;
    }
    bool any = false;
    int na = INT32_MIN;
    for (int r = 0; r < 8; r++)
        for (int q = 0; q < 8; q++) {
            if (r < 4 && q < 4) continue;
            if (r >= 4 && !((present_t >> r) & 1u)) continue;
            if (q >= 4 && !((present_q >> q) & 1u)) continue;
            na = std::max(na, g_sub_mat[r * 8 + q]);
            any = true;
        }
    if (any) for (int x = 0; x < 4; x++) cls[x] = std::max(cls[x], na);
    // the filter keeps (score, drop) as two int16 halves of one register: 64 bases x |score| must stay below 2^14.  Raising a
    // negative score keeps the bound an upper bound (positive scores are <= 127 wherever the packed filters are eligible)
    for (int x = 0; x < 4; x++) cls[x] = std::max(cls[x], -255);
}

static int max_hits_for_mem(uint64_t total_global_mem) {  // src/seed_filter.cu:832-841, literally
    float global_mem_gb = static_cast<float>(total_global_mem / 1073741824.0f);
    return (int)(4194304 * global_mem_gb);
}

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
static void prof_flush(Slot* sl) {  // call after the slot's stream has been synchronised
    if (sl->prof_pending.empty()) { sl->events_used = 0; return; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : sl->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            g_prof[r.id].total_ms += ms;
            g_prof[r.id].launches += 1;
            float t0 = 0;
            if (sl->dev >= 0 && sl->dev < PROF_MAX_DEV && g_prof_epoch[sl->dev] &&
                hipEventElapsedTime(&t0, g_prof_epoch[sl->dev], r.e0) == hipSuccess) {
                if (g_prof[r.id].spans.size() < ((size_t)1 << 20)) g_prof[r.id].spans.push_back({sl->dev, t0, t0 + ms});  // (bounded: a
                // profile left enabled without a reset keeps its totals, sa_profile_busy_ms then covers the first 2^20 launches)
            }
            else (void)hipGetLastError();
        }
    }
    sl->prof_pending.clear();
    sl->events_used = 0;
}

// ---- token pool ------------------------------------------------------------------------------------------------------
static Slot* acquire_slot() {  // src/seed_filter.cu:699-708
    std::unique_lock<std::mutex> lk(g_mu);
    g_cv.wait(lk, [] { return !g_tokens.empty(); });
    auto t = g_tokens.back();
    g_tokens.pop_back();
    lk.unlock();
    check_set_device(g_dev[t.first]->dev, "SeedAndFilter");
    return &g_dev[t.first]->slots[t.second];
}
static void release_slot(Slot* s) {  // src/seed_filter.cu:798-803
    int di = -1, si = -1;
    for (int d = 0; d < g_ndev; d++)
        for (int k = 0; k < SLOTS_PER_DEVICE; k++)
            if (&g_dev[d]->slots[k] == s) { di = d; si = k; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_tokens.push_back({di, si});
    }
    g_cv.notify_one();
}

static void slot_init(Slot& s, int dev) {
    s.dev = dev;
    hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    s.d_plan = (IterPlan*)dev_malloc(sizeof(IterPlan) * SA_MAX_CHUNKS, "plan");
    s.d_cnt = (Counters*)dev_malloc(sizeof(Counters), "counters");
    s.d_verify = (uint32_t*)dev_malloc(sizeof(uint32_t), "seed verify flag");
    s.d_cov_range = (uint32_t*)dev_malloc(2 * sizeof(uint32_t), "coverage range");
    s.d_td_bounds = dev_malloc(probe_bounds_bytes(), "probe bounds");
    s.d_seg_info = (uint32_t*)dev_malloc(dedup_seg_info_words() * sizeof(uint32_t), "segment info");
    s.d_td_plan = (TdPlan*)dev_malloc(sizeof(TdPlan) * SA_MAX_CHUNKS, "probe plan");
    if (hipHostMalloc((void**)&s.h_cov, 8 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_bounds, (SA_MAX_CHUNKS + 2) * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_plan, sizeof(IterPlan) * SA_MAX_CHUNKS) != hipSuccess ||
        hipHostMalloc((void**)&s.h_td_plan, sizeof(TdPlan) * SA_MAX_CHUNKS) != hipSuccess ||
        hipHostMalloc((void**)&s.h_seg_info, dedup_seg_info_words() * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_verify, sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_cnt, sizeof(Counters)) != hipSuccess) {
        fprintf(stderr, "Error: hipHostMalloc for slot staging failed\n");
        exit(12);
    }
}
static void slot_destroy(Slot& s) {
    s.seeds.release("seeds"); s.start.release("start"); s.count.release("count"); s.flags.release("flags");
    s.flag_prefix.release("flag_prefix"); s.prefix.release("prefix"); s.scan_temp.release("scan_temp");
    s.sort_temp.release("sort_temp"); s.hits.release("hits"); s.recA.release("recA"); s.recB.release("recB");
    s.out16.release("out16");
    s.cand_list.release("candidate list");
    s.l2_list.release("second-level list");
    s.audit.release("audit list");
    s.l2_counts.release("second-level counters");
    s.chain_tmp.release("chain"); s.chain_sorted.release("chain"); s.chain_is_head.release("chain");
    s.chain_heads.release("chain"); s.chain_bucket_cnt.release("chain"); s.chain_bucket_start.release("chain"); s.ent_list.release("entropy list");
    s.cov_diff.release("coverage"); s.cov_pre.release("coverage"); s.cov_is_start.release("coverage");
    s.cov_is_end.release("coverage"); s.cov_sidx.release("coverage"); s.cov_eidx.release("coverage"); s.cov_pairs.release("coverage");
    s.td_toff.release("probe"); s.td_tcnt.release("probe"); s.td_rec.release("probe"); s.td_chunk.release("probe"); s.td_bits.release("probe"); s.td_partial.release("probe");
    dev_free(s.d_td_bounds, "probe bounds"); dev_free(s.d_td_plan, "probe plan"); dev_free(s.d_seg_info, "segment info");
    s.d_td_bounds = nullptr; s.d_td_plan = nullptr; s.d_seg_info = nullptr;
    if (s.h_seg_info) hipHostFree(s.h_seg_info);
    s.h_seg_info = nullptr;
    if (s.h_td_plan) hipHostFree(s.h_td_plan);
    s.h_td_plan = nullptr;
    dev_free(s.d_plan, "plan"); dev_free(s.d_cnt, "counters"); dev_free(s.d_cov_range, "coverage range");
    dev_free(s.d_verify, "seed verify flag");
    if (s.h_verify) hipHostFree(s.h_verify);
    s.d_plan = nullptr; s.d_cnt = nullptr; s.d_cov_range = nullptr; s.d_verify = nullptr; s.h_verify = nullptr;
    if (s.h_cov) hipHostFree(s.h_cov);
    s.h_cov = nullptr;
    s.out_seg.release("out seg");
    if (s.h_seg) hipHostFree(s.h_seg);
    if (s.h_bounds) hipHostFree(s.h_bounds);
    s.h_seg = nullptr; s.h_bounds = nullptr; s.h_seg_cap = 0;
    if (s.h_plan) hipHostFree(s.h_plan);
    if (s.h_cnt) hipHostFree(s.h_cnt);
    if (s.h_seeds) hipHostFree(s.h_seeds);
    if (s.h_out) hipHostFree(s.h_out);
    s.h_plan = nullptr; s.h_cnt = nullptr; s.h_seeds = nullptr; s.h_out = nullptr;
    s.h_seeds_cap = s.h_out_cap = 0;
    for (auto& e : s.event_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    s.event_pool.clear();
    if (s.stream) hipStreamDestroy(s.stream);
    s.stream = nullptr;
}

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
};

static size_t saf_core(DevCtx* dc, Slot* sl, uint32_t num_seeds, const CoreArgs& ca, sa_segment_pair** out) {
    hipStream_t st = sl->stream;
    memset(&t_stats, 0, sizeof(t_stats));
    t_stats.num_seeds = num_seeds;
    t_stats.device = dc->dev;

    uint64_t num_hits = 0;
    uint32_t n_final = 0;
    uint32_t survivors = 0;
    uint64_t n_cand_total = 0, n_ent_total = 0, n_fwd_total = 0;
    // chunks of the seed vector (one for an ordinary call)
    const int K = ca.nchunks > 1 ? ca.nchunks : 1;
    uint32_t sbound[SA_MAX_CHUNKS + 1] = {0, num_seeds};
    if (ca.nchunks > 1) memcpy(sbound, ca.seed_bound, sizeof(uint32_t) * (K + 1));
    uint64_t chunk_hits[SA_MAX_CHUNKS] = {0};
    uint32_t chunk_first_seg[SA_MAX_CHUNKS + 1] = {0};
    bool have_seg = false;  // sl->h_seg holds the segment of every final record

    if (num_seeds > 0) {
        // flat list of reference iterations ("segments") over all chunks: global seed / hit offsets of their ends
        struct SegEnd { int64_t seed_hi; uint64_t hit_hi; };
        std::vector<SegEnd> segs;
        if (ca.raw_hits) {
            check_memcpy(hipMemsetAsync(sl->d_cnt, 0, sizeof(Counters), st), "counters");
            memset(sl->h_cnt, 0, sizeof(Counters));
            segs.push_back({1, ca.raw_hits});
            chunk_hits[0] = num_hits = ca.raw_hits;
            chunk_first_seg[1] = 1;
        } else if (ca.td) {
            // ---- table-direct: td_front has probed the positions and planned every chunk (probe.hip); counters were cleared there ----
            memset(sl->h_cnt, 0, sizeof(Counters));
            for (int c = 0; c < K; c++) {
                const TdPlan& tp = sl->h_td_plan[c];
                chunk_first_seg[c] = (uint32_t)segs.size();
                chunk_hits[c] = tp.num_hits;
                sbound[c + 1] = sbound[c] + tp.num_valid * ca.td_words;
                if (tp.num_hits > 0) {  // num_hits < MAX_HITS: exactly two iterations (:721-724,:732-741)
                    segs.push_back({0, tp.split});
                    segs.push_back({0, tp.hit_base + tp.num_hits});
                    t_stats.num_iter += 2;
                }
                num_hits = tp.hit_base + tp.num_hits;
            }
            chunk_first_seg[K] = (uint32_t)segs.size();
        } else {
        // ---- bucket lookup + prefix (find_num_hits :157-182 ; inclusive_scan :714) ----
        sl->start.ensure(num_seeds, "seed start");
        sl->count.ensure(num_seeds, "seed count");
        sl->prefix.ensure((size_t)num_seeds + 1, "hit prefix");
        sl->scan_temp.ensure(scan_temp_bytes(num_seeds), "scan temp");
        {
            ProfScope p(sl, "seed_lookup");
            launch_seed_lookup(sl->seeds.p, num_seeds, dc->bucket_start, dc->nkeys, sl->start.p, sl->count.p, st);
        }
        {
            ProfScope p(sl, "hit_prefix_scan");
            launch_exclusive_scan_u64(sl->count.p, sl->prefix.p, num_seeds, sl->scan_temp.p, st);
        }
        // ---- iteration plan (:718-745) of every chunk, one D2H ----
        {
            ProfScope p(sl, "iteration_plan");
            for (int c = 0; c < K; c++)
                launch_plan(sl->prefix.p + sbound[c], sbound[c + 1] - sbound[c], (uint64_t)(uint32_t)g_max_hits, ca.rm ? 0 : 1,
                            sl->d_plan + c, st);
        }
        check_launch("lookup/scan/plan");
        check_memcpy(hipMemcpyAsync(sl->h_plan, sl->d_plan, sizeof(IterPlan) * K, hipMemcpyDeviceToHost, st), "plan");
        check_memcpy(hipMemsetAsync(sl->d_cnt, 0, sizeof(Counters), st), "counters");
        check_sync(st, "plan");
        memset(sl->h_cnt, 0, sizeof(Counters));
        {
            uint64_t hit_base = 0;
            for (int c = 0; c < K; c++) {
                const IterPlan& plan = sl->h_plan[c];
                if (plan.overflow) {
                    fprintf(stderr, "Error: SeedAndFilter needs %u iterations (> %u); MAX_HITS=%ld is too small for %lu hits\n",
                            plan.overflow, PLAN_MAX_ITER, (long)g_max_hits, (unsigned long)plan.num_hits);
                    exit(15);
                }
                chunk_first_seg[c] = (uint32_t)segs.size();
                chunk_hits[c] = plan.num_hits;
                if (plan.num_hits > 0)
                    for (uint32_t i = 0; i < plan.num_iter; i++)
                        segs.push_back({(int64_t)sbound[c] + plan.limit_pos[i] + 1, hit_base + plan.upto[i]});
                hit_base += plan.num_hits;
                t_stats.num_iter += plan.num_iter;
            }
            chunk_first_seg[K] = (uint32_t)segs.size();
            num_hits = hit_base;
        }
        }

        auto ensure_host_out = [&](size_t n) {
            if (sl->h_out_cap >= n) return;
            if (sl->h_out) hipHostFree(sl->h_out);
            sl->h_out_cap = std::max<size_t>(n, 1u << 16);
            if (hipHostMalloc((void**)&sl->h_out, sl->h_out_cap * sizeof(sa_segment_pair)) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for hsp_output failed\n");
                exit(12);
            }
        };
        auto ensure_host_seg = [&](size_t n) {
            if (sl->h_seg_cap >= n) return;
            if (sl->h_seg) hipHostFree(sl->h_seg);
            sl->h_seg_cap = std::max<size_t>(n, 1u << 16);
            if (hipHostMalloc((void**)&sl->h_seg, sl->h_seg_cap * sizeof(uint32_t)) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for the segment ids failed\n");
                exit(12);
            }
        };
        bool spec_tried = false; // ... was launched (a segment too large for LDS is not worth a second attempt)
        bool spec_done = false;  // the speculative LDS chain of a single-batch call delivered the final records
        if (num_hits > 0 && !segs.empty()) {
            // ---- batches of consecutive iterations: expand (find_hits) + extend (find_hsps) ----
            const uint64_t HIT_BATCH = 1ull << 27;  // 128 Mi hits (1 GiB of 8-byte hits) per batch unless one iteration is larger
            sl->recA.ensure((size_t)std::max<uint64_t>(1u << 20, std::min<uint64_t>(num_hits, 1ull << 26)), "survivors");
            uint32_t it = 0;
            int64_t seed_lo = 0;
            uint64_t hit_lo = 0;
            while (it < segs.size()) {
                ExtendArgs ea;
                memset(&ea, 0, sizeof(ea));
                int nseg = 0;
                int64_t b_seed_lo = seed_lo, b_seed_hi = seed_lo;
                uint64_t b_hit_lo = hit_lo, b_hit_hi = hit_lo;
                uint32_t it0 = it;
                while (it < segs.size() && nseg < (ca.td ? MAX_SEGS : MAX_SEGS_ABS)) {
                    uint64_t upto = segs[it].hit_hi;
                    if (!ca.td && nseg > 0 && upto - b_hit_lo > HIT_BATCH) break;  // (no hit list in a table-direct call)
                    ea.seg_end[nseg++] = upto;
                    b_seed_hi = std::max(b_seed_hi, segs[it].seed_hi);
                    b_hit_hi = upto;
                    it++;
                }
                seed_lo = b_seed_hi;
                hit_lo = b_hit_hi;
                const uint64_t bh = b_hit_hi - b_hit_lo;
                if (bh == 0 || (!ca.td && b_seed_hi <= b_seed_lo)) continue;  // iterations without hits produce nothing (H5)
                if (ca.td) {
                    ea.td = 1;
                    ea.td_rec = sl->td_rec.p;
                    ea.td_chunk = sl->td_chunk.p;
                    ea.td_m = sl->h_td_plan[K - 1].m_hi;
                    ea.td_pos = dc->nbr_pos;
                    ea.seed_size = g_seed_size;
                    // (class filter: needs the 2-bit copies of both strands, each set below 4 GB -- blocks of up to ~1 Gbp)
                    if (dc->nbr_ctx && ca.q2_own && ca.q2_own->base && ca.q2_other && ca.q2_other->base &&
                        ca.q2_own->stride * Q2_COPIES < ((size_t)1 << 32)) {
                        ea.td_ctx = dc->nbr_ctx;
                        ea.td_bits = reinterpret_cast<const uint64_t*>(sl->td_bits.p);
                        ea.q2_own = ca.q2_own->base;
                        ea.q2_other = ca.q2_other->base;
                        ea.q2_stride = ca.q2_own->stride;
                        class_scores(dc->ref_present, ca.q_present, ea.cls);
                    }
                    if (ea.td_ctx) {
                        // (a sub-list can take a whole chunk; SEGALIGN_AMD_L2_CAP: tests start small to reach the regrow-and-rerun path)
                        sl->l2_list.ensure(g_l2_cap_test ? (size_t)g_l2_cap_test
                                                         : (size_t)std::max<uint64_t>((uint64_t)L2_NSUB * TD_CHUNK_HITS, bh / 8), "second-level list");
                        sl->l2_counts.ensure((size_t)L2_NSUB * L2_CNT_STRIDE, "second-level counters");
                    }
                    ea.l2_count = sl->l2_counts.p;
                    ea.l2_total = &sl->d_cnt->n_l2;
                    ea.l2_max = &sl->d_cnt->n_l2_max;
                    ea.l2_blocks = (uint32_t)g_l2_blocks;
                    ea.ctx_waves = (uint32_t)g_ctx_waves;
                    ea.ctx_threads = (uint32_t)g_ctx_threads;
                } else if (!ca.raw_hits) {
                    sl->hits.ensure((size_t)bh, "hits");
                    ProfScope p(sl, "expand_hits");
                    launch_expand_hits(sl->seeds.p, sl->start.p, sl->count.p, sl->prefix.p, (uint32_t)b_seed_lo,
                                       (uint32_t)b_seed_hi, b_hit_lo, dc->pos_table, g_seed_size, sl->hits.p, st);
                }
                ea.hits = ca.td ? nullptr : sl->hits.p;
                ea.ref8 = dc->ref8.codes;
                ea.fin_batch = g_fin_batch;
                ea.bufs_per_wave = g_bufs_per_wave;
                ea.query = ca.query;
                ea.ref_len = dc->ref.len;
                ea.query_len = ca.query_len;
                ea.sub_mat = dc->d_sub_mat;
                ea.xdrop = g_xdrop;
                ea.hspthresh = g_hspthresh;
                ea.noentropy = g_noentropy;
                ea.num_hits = bh;
                ea.hit_base = b_hit_lo;
                ea.num_segs = nseg;
                ea.seg_base = it0;
                ea.out_count = &sl->d_cnt->survivors;
                ea.examined = g_count_examined ? &sl->d_cnt->examined : nullptr;
                ea.rm = ca.rm;
                ea.rm_rev = ca.rm_rev;
                ea.rm_win_start = ca.rm_win_start;
                ea.rm_win_end = ca.rm_win_end;
                ea.long_cap = (uint32_t)g_long_cap;
                ea.cand_count = &sl->d_cnt->n_long;
                ea.fast_filter = g_fast_filter;
                if (g_packed_filter && ca.query4 && ca.query4->base && dc->ref2.base && !g_count_examined) {
                    ea.fast_filter = 3;  // packed upper-bound filter (extend.hip 1b)
                    ea.ref2 = dc->ref2.base;
                    ea.ref2_stride = dc->ref2.stride;
                    ea.query4 = ca.query4->base;
                    ea.query4_stride = ca.query4->stride;
                }
                ea.ent_count = &sl->d_cnt->n_ent;
                ea.long_blocks = (uint32_t)g_long_blocks;
                ea.max_waves = (uint32_t)(ea.fast_filter == 3 ? g_packed_waves : g_max_waves);
                ea.ent_blocks = 64;
                sl->cand_list.ensure((size_t)std::max<uint64_t>(1u << 16, bh / 16), "candidate list");
                // chain shortcut: valid for the plain X-drop recurrence (xdrop >= 0), needs the 29-bit position field of its
                // sort key, and is off while E is being counted.  The repeat masker takes it too: its window only decides WHICH
                // hits are extended (all candidates lie inside it), and its chain starts with an exact-duplicate unique
                // (rm :819-823), so the duplicates the shortcut never produces would be removed there anyway
                const bool chain_rel = ca.td && ca.q_hi > ca.q_lo && (uint64_t)ca.q_hi - ca.q_lo + g_seed_size < (1u << 26);  // anchors relative to the call's first position fit the key
                const bool chain = g_chain && g_xdrop >= 0 && (chain_rel || (nseg <= MAX_SEGS_ABS && ca.query_len < (1u << 29))) && !g_count_examined;
                ea.chain_q_bits = chain_rel ? 26u : 29u;
                ea.chain_q_base = chain_rel ? ca.q_lo : 0u;
                ea.chain_cap = chain ? CHAIN_CAP : 0u;
                ea.chain_sort_threads = (uint32_t)g_chain_sort_threads;
                if (chain) {
                    sl->chain_tmp.ensure(CHAIN_CAP, "chain candidates");
                    sl->chain_sorted.ensure(CHAIN_CAP, "chain candidates");
                    sl->chain_is_head.ensure(CHAIN_CAP, "chain flags");
                    sl->chain_heads.ensure(CHAIN_CAP, "chain heads");
                    sl->chain_bucket_cnt.ensure(chain_num_buckets(), "chain buckets");
                    sl->chain_bucket_start.ensure(chain_num_buckets() + 1, "chain buckets");
                    ea.chain_tmp = sl->chain_tmp.p;
                    ea.chain_sorted = sl->chain_sorted.p;
                    ea.chain_bucket_cnt = sl->chain_bucket_cnt.p;
                    ea.chain_bucket_start = sl->chain_bucket_start.p;
                    ea.chain_is_head = sl->chain_is_head.p;
                    ea.chain_heads = sl->chain_heads.p;
                    ea.chain_head_count = &sl->d_cnt->n_heads;
                }
                sl->ent_list.ensure((size_t)std::max<uint64_t>(1u << 16, bh / 32), "entropy list");
                Counters before = *sl->h_cnt;  // counters as of the previous batch (zero for the first)
                before.n_long = 0;
                before.n_ent = 0;
                before.n_heads = 0;
                before.n_l2 = 0;
                before.n_l2_max = 0;
                // (a table-direct call is ONE batch, and td_front's clearing kernel has just zeroed the counters, the sub-list counters,
                //  the chain buckets and the segment info: the memsets below only run for later batches and for reruns)
                bool cleared = ca.td && it0 == 0;
                if (!cleared) check_memcpy(hipMemsetAsync(&sl->d_cnt->n_long, 0, 6 * sizeof(uint32_t), st), "counters");
                if (g_audit_cap && ca.td) {
                    sl->audit.ensure(g_audit_cap, "audit list");
                    ea.audit_list = sl->audit.p;
                    ea.audit_count = &sl->d_cnt->n_audit;
                    ea.audit_cap = g_audit_cap;
                }
                before.n_audit = 0;
                for (;;) {  // rerun the batch with larger lists if one overflowed (device writes are guarded)
                    ea.out = sl->recA.p;
                    ea.out_cap = (uint32_t)std::min<size_t>(sl->recA.cap, 0xFFFFFFFFu);
                    ea.cand_list = sl->cand_list.p;
                    ea.cand_cap_recs = (uint32_t)std::min<size_t>(sl->cand_list.cap, 0xFFFFFFFFu);
                    ea.ent_list = sl->ent_list.p;
                    ea.ent_cap_recs = (uint32_t)std::min<size_t>(sl->ent_list.cap, 0xFFFFFFFFu);
                    if (ea.td && ea.td_ctx) {
                        // context / class filter over the table's own records, then the packed filter on what it could not decide
                        ea.l2_list = sl->l2_list.p;
                        ea.l2_cap = (uint32_t)std::min<size_t>(sl->l2_list.cap / L2_NSUB, 0xFFFFFFu);  // per sub-list
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->l2_counts.p, 0, (size_t)L2_NSUB * L2_CNT_STRIDE * sizeof(uint32_t), st), "second-level counters");
                        { ProfScope p(sl, "extend_filter"); launch_extend_filter_cls(ea, st); }
                        ExtendArgs e2 = ea;
                        e2.td = 0;
                        e2.src_cand = 1;
                        { ProfScope p(sl, "extend_filter2"); launch_extend_filter(e2, st); }
                    } else {
                        ProfScope p(sl, "extend_filter");
                        launch_extend_filter(ea, st);
                    }
                    if (ea.chain_cap) {
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->chain_bucket_cnt.p, 0, chain_num_buckets() * sizeof(uint32_t), st), "chain buckets");
                        { ProfScope p(sl, "chain_group"); launch_chain_group(ea, st); }
                        { ProfScope p(sl, "chain_link");  launch_chain_link(ea, st); }
                    }
                    if (ea.chain_cap) { ProfScope p(sl, "extend_exact_chain"); launch_extend_exact_chain(ea, st); }
                    else              { ProfScope p(sl, "extend_exact");       launch_extend_exact(ea, st); }
                    { ProfScope p(sl, "extend_entropy"); launch_extend_entropy(ea, st); }
                    check_launch("expand/extend");
                    // A call that is ONE batch (every table-direct call) does not wait for the survivor count: the per-segment LDS
                    // chain (:776-782) is launched on the device-side count and its first SPEC_RECS records travel with the
                    // counters -- one host sync for extension + chain + output instead of two
                    const bool spec = ca.td && !ca.rm && !ca.raw_hits && it0 == 0 && it == segs.size() &&
                                      segs.size() <= dedup_small_max_segs() && !g_no_small_dedup && g_spec_dedup;
                    const uint32_t seg_words = dedup_seg_info_words();
                    if (spec) {
                        spec_tried = true;
                        sl->out16.ensure(dedup_seg_max_total(), "out16");
                        ensure_host_out(dedup_seg_max_total());
                        ensure_host_seg(std::max<size_t>(dedup_seg_max_total(), seg_words));
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->d_seg_info, 0, seg_words * sizeof(uint32_t), st), "segment info");
                        { ProfScope p(sl, "dedup_seg"); launch_dedup_seg(ea.out, 0, &sl->d_cnt->survivors, (uint32_t)segs.size(), sl->out16.p, sl->d_seg_info, (uint32_t)g_dedup_threads, g_dedup_seg_max, st); }
                        check_launch("dedup seg");
                        check_memcpy(hipMemcpyAsync(sl->h_seg_info, sl->d_seg_info, seg_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "segment info");
                        check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)SPEC_RECS * sizeof(sa_segment_pair), hipMemcpyDeviceToHost, st),
                                     "hsp_output");  // :788
                    }
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "extend");
                    spec_done = spec && sl->h_seg_info[seg_words - 1] == 0;
                    if (ea.chain_cap && sl->h_cnt->n_long > ea.chain_cap && sl->h_cnt->n_long <= ea.cand_cap_recs) {
                        spec_done = false;  // (the chain ran on an unfinished survivor list)
                        // more candidates than the chain buffers hold: the chain kernels left the batch alone (device-side
                        // test on the same counter); extend every candidate on its own
                        ExtendArgs eb = ea;
                        eb.chain_cap = 0;
                        { ProfScope p(sl, "extend_exact");   launch_extend_exact(eb, st); }
                        { ProfScope p(sl, "extend_entropy"); launch_extend_entropy(eb, st); }
                        check_launch("extend (no chain)");
                        check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                        check_sync(st, "extend (no chain)");
                    }
                    const Counters& c = *sl->h_cnt;
                    const bool l2_ok = !(ea.td && ea.td_ctx) || c.n_l2_max <= ea.l2_cap;
                    if (c.survivors <= ea.out_cap && c.n_long <= ea.cand_cap_recs && c.n_ent <= ea.ent_cap_recs && l2_ok) break;
                    if (!l2_ok)  // (the later stages saw a truncated list)
                        sl->l2_list.ensure((size_t)c.n_l2_max * L2_NSUB + ((size_t)c.n_l2_max * L2_NSUB) / 4, "second-level list(grow)");
                    // an overflowing long list also truncates what the later kernels saw: size everything from the
                    // counts of this attempt (upper bounds for the rerun: survivors <= hits, entropy candidates <= hits)
                    if (c.n_long > ea.cand_cap_recs) sl->cand_list.ensure((size_t)c.n_long, "candidate list(grow)");
                    if (c.n_ent > ea.ent_cap_recs || c.n_long > ea.cand_cap_recs)
                        sl->ent_list.ensure((size_t)std::min<uint64_t>(bh, (uint64_t)c.n_ent + c.n_long), "entropy list(grow)");
                    if (c.survivors > ea.out_cap || c.n_long > ea.cand_cap_recs)
                        sl->recA.ensure((size_t)std::min<uint64_t>((uint64_t)before.survivors + bh, (uint64_t)c.survivors + c.n_long + c.n_ent),
                                        "survivors(grow)", true, st);
                    check_memcpy(hipMemcpy(sl->d_cnt, &before, sizeof(Counters), hipMemcpyHostToDevice), "counter reset");
                    cleared = false;  // (the rerun of the batch clears its lists itself)
                }
                survivors = sl->h_cnt->survivors;
                n_cand_total += sl->h_cnt->n_long;
                n_fwd_total += sl->h_cnt->n_l2;
                n_ent_total += sl->h_cnt->n_ent;
            }
            if (g_audit_cap && ca.td) {  // (tests) the rejected hits of the last batch
                const uint32_t na = std::min(sl->h_cnt->n_audit, g_audit_cap);
                t_audit.resize(na);
                if (na) {
                    check_memcpy(hipMemcpyAsync(t_audit.data(), sl->audit.p, (size_t)na * sizeof(uint2), hipMemcpyDeviceToHost, st), "audit");
                    check_sync(st, "audit");
                }
            }
            t_stats.num_examined = sl->h_cnt->examined;
            t_stats.num_examined_filter = sl->h_cnt->examined_filter;
            t_stats.num_candidates = n_cand_total;
            t_stats.num_forwarded = n_fwd_total;
            t_stats.num_entropy = n_ent_total;

            // ---- order + de-duplicate (:776-782 ; rm :819-831) ----
            if (survivors > 0 && spec_done && survivors <= dedup_seg_max_total()) {
                if (survivors > SPEC_RECS) {  // the records beyond the speculative prefix
                    check_memcpy(hipMemcpyAsync(sl->h_out + SPEC_RECS, sl->out16.p + SPEC_RECS, (size_t)(survivors - SPEC_RECS) * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");
                    check_sync(st, "hsp_output");
                }
                const uint32_t S = dedup_small_max_segs();
                size_t pos = 0;
                for (uint32_t g = 0; g < (uint32_t)segs.size(); g++) {  // close the gaps the unique step left
                    const uint32_t m2 = sl->h_seg_info[g], off = sl->h_seg_info[S + g];
                    if (m2 && pos != off) memmove(sl->h_out + pos, sl->h_out + off, (size_t)m2 * sizeof(sa_segment_pair));
                    for (uint32_t i = 0; i < m2; i++) sl->h_seg[pos + i] = g;
                    pos += m2;
                }
                n_final = (uint32_t)pos;
                have_seg = true;
            } else if (survivors > 0 && ca.raw_hits) {  // the extension stage's own output, unordered
                sl->out16.ensure(survivors, "out16");
                ensure_host_out(survivors);
                launch_strip(sl->recA.p, survivors, sl->out16.p, nullptr, st);
                check_launch("strip");
                check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)survivors * sizeof(sa_segment_pair), hipMemcpyDeviceToHost, st),
                             "hsp_output");
                check_sync(st, "hsp_output");
                n_final = survivors;
            } else if (survivors > 0) {
                sl->recB.ensure(std::max<size_t>(survivors, sl->recA.cap), "survivors B");
                size_t tb = sort_temp_bytes(survivors);
                sl->sort_temp.ensure(tb, "sort temp");
                HspRec* fin = nullptr;
                bool done = false;
                if (!ca.rm && survivors <= dedup_seg_max_total() && segs.size() <= dedup_small_max_segs() && !g_no_small_dedup && !spec_tried) {
                    // the whole chain in LDS, one workgroup per segment; one D2H of the (gapped) records + the segment counts
                    const uint32_t words = dedup_seg_info_words();
                    sl->out16.ensure(survivors, "out16");
                    ensure_host_out(survivors);
                    ensure_host_seg(std::max<size_t>(survivors, words));
                    check_memcpy(hipMemsetAsync(sl->d_seg_info, 0, words * sizeof(uint32_t), st), "segment info");
                    { ProfScope p(sl, "dedup_seg"); launch_dedup_seg(sl->recA.p, survivors, nullptr, (uint32_t)segs.size(), sl->out16.p, sl->d_seg_info, (uint32_t)g_dedup_threads, g_dedup_seg_max, st); }
                    check_launch("dedup seg");
                    check_memcpy(hipMemcpyAsync(sl->h_seg_info, sl->d_seg_info, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "segment info");
                    check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)survivors * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");  // :788
                    check_sync(st, "hsp_output");
                    if (sl->h_seg_info[words - 1] == 0) {  // (else a segment was too large for LDS: library sorts below)
                        const uint32_t S = dedup_small_max_segs();
                        size_t pos = 0;
                        for (uint32_t g = 0; g < (uint32_t)segs.size(); g++) {  // close the gaps the unique step left
                            const uint32_t m2 = sl->h_seg_info[g], off = sl->h_seg_info[S + g];
                            if (m2 && pos != off) memmove(sl->h_out + pos, sl->h_out + off, (size_t)m2 * sizeof(sa_segment_pair));
                            for (uint32_t i = 0; i < m2; i++) sl->h_seg[pos + i] = g;
                            pos += m2;
                        }
                        n_final = (uint32_t)pos;
                        have_seg = true;
                        done = true;
                    }
                }
                if (done) {
                    // nothing left to do on the device
                } else if (!ca.rm) {
                    { ProfScope p(sl, "sort_diag");  launch_sort(sl->recA.p, sl->recB.p, survivors, ORDER_DIAG, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");     launch_unique(sl->recB.p, sl->recA.p, survivors, 0, &sl->d_cnt->uniq, st); }
                    check_launch("sort/unique");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique");
                    n_final = sl->h_cnt->uniq;
                    { ProfScope p(sl, "sort_lastz"); launch_sort(sl->recA.p, sl->recB.p, n_final, ORDER_LASTZ, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    fin = sl->recB.p;
                } else {
                    { ProfScope p(sl, "sort_rm_first"); launch_sort(sl->recA.p, sl->recB.p, survivors, ORDER_RM_FIRST, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");        launch_unique(sl->recB.p, sl->recA.p, survivors, 1, &sl->d_cnt->uniq, st); }
                    check_launch("sort/unique");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique");
                    uint32_t n1 = sl->h_cnt->uniq;
                    { ProfScope p(sl, "sort_rm_diag");  launch_sort(sl->recA.p, sl->recB.p, n1, ORDER_RM_DIAG, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");        launch_unique(sl->recB.p, sl->recA.p, n1, 0, &sl->d_cnt->uniq2, st); }
                    check_launch("sort/unique 2");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique 2");
                    n_final = sl->h_cnt->uniq2;
                    if (ca.cov_diff) {  // coverage is order-independent: the final sort (rm :831) and the D2H are not needed
                        ProfScope p(sl, "coverage_add");
                        launch_coverage_add_hsprec(sl->recA.p, n_final, ca.cov_diff, ca.cov_diff_len, sl->d_cov_range, st);
                        check_launch("coverage add");
                    } else {
                        { ProfScope p(sl, "sort_rm_final"); launch_sort(sl->recA.p, sl->recB.p, n_final, ORDER_RM_FINAL, sl->sort_temp.p, sl->sort_temp.cap, st); }
                        fin = sl->recB.p;
                    }
                }
                if (n_final > 0 && fin) {
                    sl->out16.ensure(n_final, "out16");
                    ensure_host_out(n_final);
                    if (K > 1) { sl->out_seg.ensure(n_final, "out seg"); ensure_host_seg(n_final); }
                    { ProfScope p(sl, "strip"); launch_strip(fin, n_final, sl->out16.p, K > 1 ? sl->out_seg.p : nullptr, st); }
                    check_launch("final sort/strip");
                    if (K > 1) {
                        check_memcpy(hipMemcpyAsync(sl->h_seg, sl->out_seg.p, (size_t)n_final * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "hsp segs");
                        have_seg = true;
                    }
                    check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)n_final * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");  // :788
                }
                check_sync(st, "hsp_output");
            }
        }
    }
    prof_flush(sl);

    t_stats.lookup_path = ca.td ? ((dc->nbr_ctx && ca.q2_own && ca.q2_own->base) ? 2 : 1) : 0;
    t_stats.num_hits = num_hits;
    t_stats.num_survivors = survivors;
    t_stats.num_anchors = n_final;
    if (K > 1 && ca.outs) {
        // ---- one return vector per chunk: records are ordered by segment, chunk c owns segments
        //      [chunk_first_seg[c], chunk_first_seg[c+1]) ; a chunk without seeds returns nothing (seeder.cpp:76) ----
        size_t pos = 0;
        for (int c = 0; c < K; c++) {
            size_t n_c = 0;
            if (have_seg)
                while (pos + n_c < n_final && sl->h_seg[pos + n_c] < chunk_first_seg[c + 1]) n_c++;
            if (sbound[c + 1] == sbound[c]) {
                ca.outs[c] = nullptr;
                ca.counts[c] = 0;
            } else {
                sa_segment_pair* r = (sa_segment_pair*)malloc((n_c + 1) * sizeof(sa_segment_pair));
                memset(&r[0], 0, sizeof(sa_segment_pair));
                r[0].len = (uint32_t)n_c;
                r[0].score = (int32_t)(uint32_t)chunk_hits[c];
                if (n_c) memcpy(r + 1, sl->h_out + pos, n_c * sizeof(sa_segment_pair));
                ca.outs[c] = r;
                ca.counts[c] = n_c + 1;
            }
            pos += n_c;
        }
        return (size_t)n_final + K;
    }
    if (out == nullptr) return (size_t)n_final + 1;  // coverage mode: nothing is returned to the host

    // ---- return vector: header + HSPs (:804-827 ; rm :857-861) ----
    sa_segment_pair* res = (sa_segment_pair*)malloc(((size_t)n_final + 1) * sizeof(sa_segment_pair));
    memset(&res[0], 0, sizeof(sa_segment_pair));
    if (!ca.rm) {
        res[0].len = n_final;
        res[0].score = (int32_t)(uint32_t)num_hits;
    } else {
        uint64_t ta = n_final;
        res[0].ref_start = (uint32_t)(num_hits & 0xFFFFFFFFull);
        res[0].query_start = (uint32_t)(num_hits >> 32);
        res[0].len = (uint32_t)(ta & 0xFFFFFFFFull);
        res[0].score = (int32_t)(ta >> 32);
    }
    if (n_final) memcpy(res + 1, sl->h_out, (size_t)n_final * sizeof(sa_segment_pair));
    *out = res;
    return (size_t)n_final + 1;
}

static int g_seed_upload = 0;  // option seed_upload: 0 pinned staging (memcpy + DMA), 1 pageable hipMemcpyAsync, 2 hipHostRegister + DMA
static void upload_seeds(Slot* sl, const uint64_t* seeds, size_t n) {
    sl->seeds.ensure(std::max<size_t>(n, (size_t)g_max_seeds), "seed_offsets");
    if (n == 0) return;
    if (g_seed_upload == 1) {  // the runtime stages the pageable vector itself (chunked, synchronous for the caller)
        ProfScope p(sl, "h2d_seeds");
        check_memcpy(hipMemcpyAsync(sl->seeds.p, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream), "seed_offsets");
        return;
    }
    if (g_seed_upload == 2) {  // pin the caller's pages for the duration of the copy
        const uintptr_t lo = (uintptr_t)seeds & ~(uintptr_t)4095, hi = ((uintptr_t)(seeds + n) + 4095) & ~(uintptr_t)4095;
        if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
            {
                ProfScope p(sl, "h2d_seeds");
                check_memcpy(hipMemcpyAsync(sl->seeds.p, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream), "seed_offsets");
            }
            check_sync(sl->stream, "seed_offsets");
            hipHostUnregister((void*)lo);
            return;
        }
        (void)hipGetLastError();
    }
    if (sl->h_seeds_cap < n) {
        if (sl->h_seeds) hipHostFree(sl->h_seeds);
        sl->h_seeds_cap = std::max<size_t>(n, (size_t)g_max_seeds);
        if (hipHostMalloc((void**)&sl->h_seeds, sl->h_seeds_cap * sizeof(uint64_t)) != hipSuccess) {
            fprintf(stderr, "Error: hipHostMalloc for seed_offsets failed\n");
            exit(12);
        }
    }
    memcpy(sl->h_seeds, seeds, n * sizeof(uint64_t));  // reference copies the vector too (:694-697)
    ProfScope p(sl, "h2d_seeds");
    check_memcpy(hipMemcpyAsync(sl->seeds.p, sl->h_seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream),
                 "seed_offsets");  // :710
}

// device-side seeder (8f-1): fills sl->seeds for query positions [start,end); returns number of seed words
// nb > 0: also reports, for nb positions bpos[] in [start, end], the number of seed words emitted before them
static uint32_t device_seeds(Slot* sl, const uint8_t* qcodes, uint32_t start, uint32_t end, int nb = 0,
                             const uint32_t* bpos = nullptr, uint32_t* bseed = nullptr) {
    for (int b = 0; b < nb; b++) bseed[b] = 0;
    if (end <= start) return 0;
    hipStream_t st = sl->stream;
    const uint32_t n = end - start;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    const uint32_t tmask = g_transition ? (sh.transition_mask & ((1u << sh.weight) - 1u)) : 0u;
    const uint32_t per = 1u + (uint32_t)__builtin_popcount(tmask);
    sl->flags.ensure(n, "seed flags");
    sl->flag_prefix.ensure((size_t)n + 1, "seed flag prefix");
    sl->scan_temp.ensure(scan_temp_bytes(n), "scan temp");
    {
        ProfScope p(sl, "seed_flags");
        launch_seed_flags(qcodes, start, end, sh, sl->flags.p, st);
    }
    {
        ProfScope p(sl, "seed_flag_scan");
        launch_exclusive_scan_u32(sl->flags.p, sl->flag_prefix.p, n, sl->scan_temp.p, st);
    }
    check_launch("seed flags");
    uint32_t nvalid = 0;
    check_memcpy(hipMemcpyAsync(&sl->h_cnt->pad, sl->flag_prefix.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "nvalid");
    for (int b = 0; b < nb; b++)
        check_memcpy(hipMemcpyAsync(&sl->h_bounds[b], sl->flag_prefix.p + (std::min(std::max(bpos[b], start), end) - start), sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, st), "chunk bounds");
    check_sync(st, "seed flags");
    nvalid = sl->h_cnt->pad;
    for (int b = 0; b < nb; b++) bseed[b] = sl->h_bounds[b] * per;
    const uint64_t nseeds = (uint64_t)nvalid * per;
    if (nseeds == 0) return 0;
    sl->seeds.ensure(std::max<size_t>((size_t)nseeds, (size_t)g_max_seeds), "seed_offsets");
    {
        ProfScope p(sl, "seed_emit");
        launch_seed_emit(qcodes, start, end, sh, g_transition, sl->flag_prefix.p, sl->seeds.p, st);
    }
    check_launch("seed emit");
    return (uint32_t)nseeds;
}

// ---- table-direct lookup (probe.hip) -----------------------------------------------------------------------------------
static uint32_t seed_tmask() {
    return g_transition ? (g_shape.transition_mask & ((1u << g_shape.weight) - 1u)) : 0u;
}

__global__ void widen_u32_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}

static void nbr_release(DevCtx* dc) {
    dev_free(dc->nbr_start, "nbr_start");
    if (!dc->nbr_alias) dev_free(dc->nbr_pos, "nbr_pos");
    dc->nbr_start = nullptr;
    dc->nbr_pos = nullptr;
    dc->nbr_ctx = nullptr;  // (the memory stays with the arena)
    dc->nbr_alias = false;
    dc->nbr_total = 0;
    dc->nbr_state = 0;
}

// Builds (once per table and transition mask) the neighbourhood table of the device; false when it is not available:
// no table yet, a merged run longer than 2^32 entries, or not enough free HBM for (words per position) x pos_table.
static bool ensure_nbr(DevCtx* dc) {
    if (!g_td || !dc->bucket_start || !dc->pos_table) return false;
    const uint32_t tmask = seed_tmask();
    std::lock_guard<std::mutex> lk(dc->nbr_mu);
    if (dc->nbr_state != 0 && dc->nbr_tmask == tmask) return dc->nbr_state == 1;
    check_set_device(dc->dev, "neighbourhood table");
    hipStream_t st = dc->admin;
    nbr_release(dc);
    dc->nbr_tmask = tmask;
    dc->nbr_state = -1;
    const uint32_t nkeys = dc->nkeys;
    dc->nbr_start = (uint64_t*)dev_malloc(((size_t)nkeys + 1) * sizeof(uint64_t), "nbr_start");
    uint64_t total = dc->num_index;
    if (tmask == 0) {  // one word per position: the runs ARE the buckets
        hipLaunchKernelGGL(widen_u32_kernel, dim3(4096), dim3(256), 0, st, dc->bucket_start, dc->nbr_start, nkeys + 1);
        check_launch("nbr widen");
        check_sync(st, "nbr widen");
    } else {
        uint32_t* cnt = (uint32_t*)dev_malloc(((size_t)nkeys + 1) * sizeof(uint32_t), "nbr counts");
        void* scan_tmp = dev_malloc(scan_temp_bytes(nkeys), "scan temp");
        check_memcpy(hipMemsetAsync(cnt + nkeys, 0, sizeof(uint32_t), st), "nbr overflow flag");  // cnt[nkeys] doubles as the flag
        launch_nbr_count(dc->bucket_start, nkeys, tmask, g_shape.weight, cnt, cnt + nkeys, st);
        launch_exclusive_scan_u64(cnt, dc->nbr_start, nkeys, scan_tmp, st);
        check_launch("nbr count/scan");
        uint32_t overflow = 0;
        check_memcpy(hipMemcpyAsync(&total, dc->nbr_start + nkeys, sizeof(uint64_t), hipMemcpyDeviceToHost, st), "nbr total");
        check_memcpy(hipMemcpyAsync(&overflow, cnt + nkeys, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "nbr overflow");
        check_sync(st, "nbr count");
        dev_free(cnt, "nbr counts");
        dev_free(scan_tmp, "scan temp");
        if (overflow) {
            dev_free(dc->nbr_start, "nbr_start");
            dc->nbr_start = nullptr;
            return false;
        }
    }
    const bool dbg = opt_value("debug") != 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = total_b = 0;  // (then only the plain lookup modes are tried)
    // keep room for the slots' work buffers: at human-scale hit density a sixteen-chunk call holds ~6 GB of lists per slot
    const size_t reserve = ((size_t)8 << 30) + ((size_t)4 << 30) * (size_t)SLOTS_PER_DEVICE;
    const size_t need_pos = (size_t)std::max<uint64_t>(total, 1) * sizeof(uint32_t);
    // context records (class filter): 32 bytes per entry; the two-stage fill wants num_index records of scratch behind them, which is
    // given up (one-stage fill) when only the table itself fits.  (+ 16 KB of slack: the filter requests two buffers ahead, so
    // its lanes read up to 3 x 64 entries past the last run)
    const size_t rec_b = (size_t)std::max<uint64_t>(total, 1) * sizeof(CtxRec) + 16384;
    const size_t scratch_b = (size_t)dc->num_index * sizeof(CtxRec);
    const size_t have = arena_mapped(dc->arena);  // (already ours: does not count against the free memory)
    bool built = false;
    if (g_ctx && dc->ref2.base && rec_b + reserve <= free_b + have) {
        const bool two_stage = g_nbr_two_stage && tmask != 0 && rec_b + scratch_b + reserve <= free_b + have;
        const size_t need = rec_b + (two_stage ? scratch_b : 0);
        if (arena_wait(dc->arena, need)) {
            arena_settle(dc->arena, need);
            uint8_t* arena = dc->arena.base;
            dc->nbr_ctx = reinterpret_cast<CtxRec*>(arena);
            if (dbg) fprintf(stderr, "neighbourhood table: %.1f M entries, waited %.1f ms for %.1f GB of arena\n", total / 1e6, now() - t_a, need / 1e9);
            const double t_b = now();
            launch_nbr_fill_ctx(dc->bucket_start, dc->pos_table, nkeys, tmask, g_shape.weight, dc->nbr_start, dc->ref2.base, dc->ref2.stride,
                                g_seed_size, dc->nbr_ctx, two_stage ? reinterpret_cast<CtxRec*>(arena + rec_b) : nullptr, (uint32_t)dc->num_index, st);
            check_launch("nbr fill ctx");
            check_sync(st, "nbr fill ctx");
            if (dbg) fprintf(stderr, "neighbourhood table: context fill (%s) took %.1f ms\n", two_stage ? "two-stage" : "one-stage", now() - t_b);
            built = true;
        }
    }
    if (built) {
        // (nothing else to do)
    } else if (tmask == 0) {
        dc->nbr_pos = dc->pos_table;
        dc->nbr_alias = true;
    } else if ([&] {  // positions only: an arena kept from an earlier (smaller) block gives its memory back first
                   if (arena_mapped(dc->arena)) {
                       arena_trim(dc->arena, 0);
                       hipDeviceSynchronize();
                       if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
                   }
                   if (dbg) fprintf(stderr, "neighbourhood table: no room for the context table (%.1f GB + %.1f GB reserve), positions only need %.1f GB, free %.1f GB\n",
                                    rec_b / 1e9, reserve / 1e9, need_pos / 1e9, free_b / 1e9);
                   return need_pos + reserve <= free_b;
               }()) {
        dc->nbr_pos = (uint32_t*)dev_malloc(need_pos, "nbr_pos");
        launch_nbr_fill(dc->bucket_start, dc->pos_table, nkeys, tmask, g_shape.weight, dc->nbr_start, dc->nbr_pos, st);
        check_launch("nbr fill");
        check_sync(st, "nbr fill");
    } else {
        dev_free(dc->nbr_start, "nbr_start");
        dc->nbr_start = nullptr;
        return false;
    }
    dc->nbr_total = total;
    dc->nbr_state = 1;
    return true;
}

// may this call take the table-direct path?  (the anchors are only ever read by the packed filter's TD fetch)
static bool td_eligible(DevCtx* dc, const PackedBuf* query4) {
    return g_td && g_packed_filter && !g_count_examined && query4 && query4->base && dc->ref2.base && ensure_nbr(dc);
}

// Position probe + chunk plans for query positions [bpos[0], bpos[K]) (chunk c = [bpos[c], bpos[c+1])); one D2H, one sync.
// Returns the number of seed words the reference would have been handed (0: nothing to do), or UINT32_MAX when the call must
// take the general path: a chunk with num_hits >= MAX_HITS (more than two reference iterations), hit counts that wrap the
// reference's uint32 arithmetic, or more than 2^32 hits in the call (the filter indexes hits with 32 bits).
static uint32_t td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, int K, const uint32_t* bpos, int rm, uint32_t* words_out) {
    hipStream_t st = sl->stream;
    const uint32_t start = bpos[0], end = bpos[K];
    const uint32_t tmask = seed_tmask();
    const uint32_t words = 1u + (uint32_t)__builtin_popcount(tmask);
    *words_out = words;
    if (end <= start) return 0;
    const uint32_t n = end - start;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    sl->td_toff.ensure(n, "probe scratch");
    sl->td_tcnt.ensure(n, "probe scratch");
    sl->td_rec.ensure((size_t)n + 1, "probe records");
    sl->td_chunk.ensure(TD_CHUNK_CAP, "probe chunk starts");
    sl->td_partial.ensure(probe_partial_bytes(n), "probe partials");
    TdBounds tb;
    tb.nb = K + 1;
    for (int c = 0; c <= K; c++) tb.pos[c] = bpos[c];
    {
        ProfScope p(sl, "seed_probe");
        launch_probe_lookup(qcodes, start, n, sh, dc->nbr_start, dc->nkeys, sl->td_toff.p, sl->td_tcnt.p, sl->td_partial.p, st);
    }
    // head-bit map for the class filter: sized for 128 hits per position; a denser call regrows it (it stays) and repeats the compaction
    const bool want_bits = dc->nbr_ctx != nullptr;
    if (want_bits) sl->td_bits.ensure(std::max<size_t>((size_t)n * 4 + 64, 1u << 16), "probe head bits");
    // the device-side state the later stages of the call expect zeroed is cleared by the probe's own clearing kernel
    sl->l2_counts.ensure((size_t)L2_NSUB * L2_CNT_STRIDE, "second-level counters");
    sl->chain_bucket_cnt.ensure(chain_num_buckets(), "chain buckets");
    ZeroList zl;
    zl.p[0] = reinterpret_cast<uint32_t*>(sl->d_cnt);  zl.n[0] = (uint32_t)(sizeof(Counters) / sizeof(uint32_t));
    zl.p[1] = sl->l2_counts.p;                         zl.n[1] = (uint32_t)(L2_NSUB * L2_CNT_STRIDE);
    zl.p[2] = sl->chain_bucket_cnt.p;                  zl.n[2] = chain_num_buckets();
    zl.p[3] = sl->d_seg_info;                          zl.n[3] = dedup_seg_info_words();
    for (bool first_pass = true;; first_pass = false) {
        {
            ProfScope p(sl, "probe_compact");
            launch_probe_compact(start, n, sl->td_toff.p, sl->td_tcnt.p, sl->td_partial.p, sl->d_td_bounds, sl->td_rec.p, sl->td_chunk.p, TD_CHUNK_CAP,
                                 want_bits ? sl->td_bits.p : nullptr, (uint32_t)std::min<size_t>(sl->td_bits.cap, 0xFFFFFFFFu), zl, tb, first_pass, st);
        }
        {
            ProfScope p(sl, "iteration_plan");
            launch_probe_plan(qcodes, sh, tmask, dc->bucket_start, sl->d_td_bounds, K, sl->td_rec.p, sl->d_td_plan, st);
        }
        check_launch("probe");
        check_memcpy(hipMemcpyAsync(sl->h_td_plan, sl->d_td_plan, sizeof(TdPlan) * K, hipMemcpyDeviceToHost, st), "probe plan");
        check_sync(st, "probe plan");
        const uint64_t call_hits = sl->h_td_plan[K - 1].hit_base + sl->h_td_plan[K - 1].num_hits;
        const uint64_t need_words = ((call_hits + 63) >> 6) * 2 + 16;  // (the filter reads up to six 64-bit words past the last buffer)
        if (!want_bits || need_words <= sl->td_bits.cap || call_hits > 0xFFFFFFFFull) break;
        sl->td_bits.ensure((size_t)need_words + need_words / 4, "probe head bits(grow)");
    }
    uint64_t nvalid = 0;
    for (int c = 0; c < K; c++) {
        const TdPlan& tp = sl->h_td_plan[c];
        nvalid += tp.num_valid;
        if (tp.num_hits >= (uint64_t)(uint32_t)g_max_hits) return 0xFFFFFFFFu;
        if (!rm && tp.num_hits > 0xFFFFFFFFull) return 0xFFFFFFFFu;
    }
    if (sl->h_td_plan[K - 1].hit_base + sl->h_td_plan[K - 1].num_hits >= 0xFFFFFFFFull) return 0xFFFFFFFFu;  // (hit indices are 32-bit, 2^32 - 1 is a sentinel)
    if (nvalid * words >= 0xFFFFFFFFull) return 0xFFFFFFFFu;
    return (uint32_t)(nvalid * words);
}

// ---- ASCII upload through the pinned ring (see DevCtx) -------------------------------------------------------------
constexpr size_t UP_CHUNK = (size_t)32 << 20;
static const uint8_t* upload_ascii(DevCtx* dc, const char* src, size_t len, const char* tag) {
    hipStream_t st = dc->admin;
    dc->up_tmp.ensure(len + 64, tag);
    for (int k = 0; k < 2; k++)
        if (!dc->up_pinned[k]) {
            if (hipHostMalloc((void**)&dc->up_pinned[k], UP_CHUNK) != hipSuccess || hipEventCreateWithFlags(&dc->up_ev[k], hipEventDisableTiming) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for the upload ring failed\n");
                exit(12);
            }
        }
    size_t i = 0;
    for (size_t off = 0; off < len; off += UP_CHUNK, i++) {
        const int k = (int)(i & 1);
        const size_t n = std::min(UP_CHUNK, len - off);
        if (i >= 2) hipEventSynchronize(dc->up_ev[k]);  // the DMA that last used this pinned buffer has finished
        memcpy(dc->up_pinned[k], src + off, n);
        check_memcpy(hipMemcpyAsync(dc->up_tmp.p + off, dc->up_pinned[k], n, hipMemcpyHostToDevice, st), tag);
        hipEventRecord(dc->up_ev[k], st);
    }
    return dc->up_tmp.p;
}

// DROP-IN FAST PATH.  g_SeedAndFilter hands the engine a host seed vector (src/seeder.cpp:57-78).  When that vector is exactly
// what the device seeder would emit for the positions it spans -- checked on the device, one lane per position group, plus the
// probe's own count of valid positions -- the call is the same as sa_seed_and_filter_range(first, last + 1) and takes the
// table-direct path (one probe per position, record-stream filter).  Anything else keeps the reference-shaped path on the
// uploaded words: hand-made vectors, and the reference's minus-strand arena when the query holds other IUPAC letters (H14).
// Returns the seed-word count of the table-direct call, or UINT32_MAX.
static uint32_t dropin_td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, uint32_t qlen, const uint64_t* host_seeds, size_t n,
                                const PackedBuf* q4, int rm, uint32_t* first_out, uint32_t* end_out, uint32_t* words_out) {
    if (n == 0 || !td_eligible(dc, q4)) return 0xFFFFFFFFu;
    const uint32_t tmask = seed_tmask();
    const uint32_t per = 1u + (uint32_t)__builtin_popcount(tmask);
    if (n % per != 0 || n > 0xFFFFFFFFull) return 0xFFFFFFFFu;
    const uint32_t first = (uint32_t)host_seeds[0], last = (uint32_t)host_seeds[n - 1];
    if (last < first || (uint64_t)last + g_seed_size > qlen) return 0xFFFFFFFFu;
    hipStream_t st = sl->stream;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    *sl->h_verify = 0;
    check_memcpy(hipMemsetAsync(sl->d_verify, 0xFF, sizeof(uint32_t), st), "seed verify flag");
    {
        ProfScope p(sl, "seed_verify");
        launch_seed_verify(sl->seeds.p, (uint32_t)(n / per), per, qcodes, qlen, sh, tmask, sl->d_verify, st);
    }
    check_memcpy(hipMemcpyAsync(sl->h_verify, sl->d_verify, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "seed verify flag");
    const uint32_t bp[2] = {first, last + 1u};
    const uint32_t ns = td_front(dc, sl, qcodes, 1, bp, rm, words_out);  // (synchronises the stream: the flag has arrived)
    if (ns == 0xFFFFFFFFu || *sl->h_verify == 0u || (uint64_t)ns != (uint64_t)n) return 0xFFFFFFFFu;
    *first_out = first;
    *end_out = last + 1u;
    return ns;
}

// code presence of a freshly encoded block -> *host_mask (one small D2H; the callers synchronise the admin stream anyway)
static void presence_of(DevCtx* dc, const uint8_t* codes, uint32_t len, int slot, uint32_t* host_mask) {
    if (!dc->d_present) dc->d_present = (uint32_t*)dev_malloc((1 + SA_BUFFER_DEPTH) * sizeof(uint32_t), "code presence");
    check_memcpy(hipMemsetAsync(dc->d_present + slot, 0, sizeof(uint32_t), dc->admin), "code presence");
    launch_code_presence(codes, len, dc->d_present + slot, dc->admin);
    check_memcpy(hipMemcpyAsync(host_mask, dc->d_present + slot, sizeof(uint32_t), hipMemcpyDeviceToHost, dc->admin), "code presence");
}
static void set_query2(CoreArgs& ca, DevCtx* dc, uint32_t buffer, int rev) {  // plain calls: strand copies of query buffer `buffer`
    ca.q2_own = rev ? &dc->query2_rc[buffer] : &dc->query2[buffer];
    ca.q2_other = rev ? &dc->query2[buffer] : &dc->query2_rc[buffer];
    ca.q_present = dc->query_present[buffer];
}
static void set_query2_rm(CoreArgs& ca, DevCtx* dc, int rev) {  // repeat masker: the query is the target
    ca.q2_own = rev ? &dc->refq2_rc : &dc->refq2;
    ca.q2_other = rev ? &dc->refq2 : &dc->refq2_rc;
    ca.q_present = dc->ref_present;
}

// ------------------------------------------------------------------------------------------------------------------
// options: ONE table for every tunable / switch of the engine (documented in include/segalign_amd.h, sa_set_option).
// Resolution at every InitializeProcessor: value set through sa_set_option > environment SEGALIGN_AMD_<NAME> > default.
// ------------------------------------------------------------------------------------------------------------------
struct Option {
    const char* name;
    int64_t def, lo, hi;
    int test_only;      // 1: exists to reach a code path from the test matrix; 0: deployment tuning
    int64_t value;      // resolved value
    int64_t api_value;
    bool api_set;
};
static Option g_opts[] = {
    // deployment
    {"slots", 4, 1, MAX_SLOTS_PER_DEVICE, 0},          // calls in flight per device (the reference allows one: token == device)
    {"chunks_per_call", SA_DEFAULT_CHUNKS, 1, SA_MAX_CHUNKS, 0},  // chunks sa_seed_interval / sa_rm_mask_interval hand to one pass
    {"no_ctx", 0, 0, 1, 0},                            // 1: neighbourhood table without target context (lookup mode 1)
    {"no_td", 0, 0, 1, 0},                             // 1: no neighbourhood table at all (lookup mode 0, the reference-shaped path)
    {"no_chain", 0, 0, 1, 0},                          // 1: every candidate is extended on its own (no chain shortcut)
    {"no_packed_filter", 0, 0, 1, 0},                  // 1: byte-coded filter kernels only (also disables table-direct lookup)
    {"no_fast_filter", 0, 0, 1, 0},                    // 1: exact per-base filter only
    {"arena_gb", 40, 0, 1024, 0},                      // GiB of table arena mapped in the background from InitializeProcessor on
    {"debug", 0, 0, 2, 0},                             // 1: table-build timings on stderr; 2: + sync and name every kernel scope
    // launch geometry (swept by tools/sweep_*.sh; the defaults are the measured optima)
    {"fin_batch", 48, 1, 64, 0}, {"bufs_per_wave", 8, 1, 1 << 20, 0}, {"long_cap", 128, 0, 2 * PACK_PAD, 0},
    {"long_blocks", 1792, 1, 1 << 20, 0}, {"max_waves", 4096, 4, 1 << 20, 0}, {"packed_waves", 4096, 8, 1 << 20, 0},
    {"l2_blocks", 512, 1, 1 << 20, 0}, {"ctx_waves", 0, 0, 1 << 20, 0}, {"ctx_threads", 0, 0, 1024, 0},
    {"chain_sort_threads", 256, 64, 512, 0}, {"dedup_threads", 0, 0, 1024, 0},
    {"nbr_one_stage", 0, 0, 1, 0}, {"table_atomic", 0, 0, 1, 0}, {"seed_upload", 0, 0, 2, 0},
    // test-only: small capacities that force the overflow / fallback branches
    {"l2_cap", 0, 0, 1 << 30, 1}, {"spec_dedup", 1, 0, 1, 1}, {"spec_recs", 16384, 1, 16384, 1}, {"dedup_seg_max", 0, 0, 1 << 30, 1},
    {"no_small_dedup", 0, 0, 1, 1}, {"chain_cap", 1 << 22, 1, 1 << 30, 1}, {"audit_cap", 0, 0, 1 << 28, 1},
};
static Option* find_option(const char* name) {
    for (auto& o : g_opts)
        if (strcmp(o.name, name) == 0) return &o;
    return nullptr;
}
static int64_t opt_value(const char* name) {
    Option* o = find_option(name);
    return o ? o->value : 0;
}
static void resolve_options() {
    for (auto& o : g_opts) {
        int64_t v = o.def;
        char env[96] = "SEGALIGN_AMD_";
        size_t n = strlen(env);
        for (const char* c = o.name; *c && n + 1 < sizeof(env); c++) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = '\0';
        if (o.api_set) v = o.api_value;
        else if (const char* e = getenv(env)) {
            char* endp = nullptr;
            v = strtoll(e, &endp, 10);
            if (endp == e) v = 1;  // a switch set to a non-number ("yes") counts as on
        }
        o.value = std::max(o.lo, std::min(o.hi, v));
    }
    SLOTS_PER_DEVICE = (int)opt_value("slots");
    g_chunks_per_call = (int)opt_value("chunks_per_call");
    g_ctx = opt_value("no_ctx") ? 0 : 1;
    g_td = opt_value("no_td") ? 0 : 1;
    g_chain = opt_value("no_chain") ? 0 : 1;
    g_arena_gb = opt_value("arena_gb");
    g_trace_scopes = opt_value("debug") >= 2 ? 1 : 0;
    g_fin_batch = (int)opt_value("fin_batch");
    g_bufs_per_wave = (int)opt_value("bufs_per_wave");
    g_long_cap = (int)opt_value("long_cap") & ~7;
    g_long_blocks = (int)opt_value("long_blocks");
    g_max_waves = (int)opt_value("max_waves");
    g_packed_waves = (int)opt_value("packed_waves");
    g_l2_blocks = (int)opt_value("l2_blocks");
    g_ctx_waves = (int)opt_value("ctx_waves");
    g_ctx_threads = (int)opt_value("ctx_threads");
    g_chain_sort_threads = (int)opt_value("chain_sort_threads") & ~63;
    g_dedup_threads = (int)opt_value("dedup_threads");
    g_nbr_two_stage = opt_value("nbr_one_stage") ? 0 : 1;
    g_table_atomic = (int)opt_value("table_atomic");
    g_seed_upload = (int)opt_value("seed_upload");
    g_l2_cap_test = opt_value("l2_cap") ? (uint32_t)std::max<int64_t>(L2_NSUB, opt_value("l2_cap")) : 0u;
    g_spec_dedup = (int)opt_value("spec_dedup");
    SPEC_RECS = (uint32_t)opt_value("spec_recs");
    g_dedup_seg_max = (uint32_t)opt_value("dedup_seg_max");
    g_no_small_dedup = (int)opt_value("no_small_dedup");
    CHAIN_CAP = (uint32_t)opt_value("chain_cap");
    g_audit_cap = (uint32_t)opt_value("audit_cap");
}

static void require_init(const char* who) {
    if (g_ndev <= 0) {
        fprintf(stderr, "Error: %s called before InitializeInterface\n", who);
        exit(1);
    }
}
static void require_proc(const char* who, uint32_t buffer) {  // hot entry points: processor initialised, buffer id in range
    require_init(who);
    if (!g_proc_init) {
        fprintf(stderr, "Error: %s called before InitializeProcessor\n", who);
        exit(1);
    }
    if (buffer >= SA_BUFFER_DEPTH) {
        fprintf(stderr, "Error: %s: query buffer %u out of range (BUFFER_DEPTH %d)\n", who, buffer, SA_BUFFER_DEPTH);
        exit(1);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent host worker pool: the engine's own seeder threads (the reference keeps one seeder body per TBB worker,
// src/main.cpp:565-573).  run_parallel(n, threads, fn) executes fn(0) .. fn(n-1) with at most `threads` of them in flight on pool
// threads that live as long as the process -- no std::thread is created per interval call.  Several run_parallel calls may be
// active at once (the host keeps several intervals in flight); each gets its own share of workers.
// ------------------------------------------------------------------------------------------------------------------
struct PoolBatch {
    std::function<void(size_t)> fn;
    size_t n = 0;
    int want = 1;                 // workers this batch may occupy
    int joined = 0;               // workers that took it (guarded by the pool mutex)
    std::atomic<size_t> next{0};
    std::atomic<size_t> done{0};
    std::mutex mu;
    std::condition_variable cv;
};
struct WorkPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<PoolBatch>> batches;
    int workers = 0;
};
static WorkPool* g_pool = new WorkPool();  // (never destroyed: its detached workers wait on it until the process ends)

static void pool_worker() {
    WorkPool& P = *g_pool;
    std::unique_lock<std::mutex> lk(P.mu);
    for (;;) {
        std::shared_ptr<PoolBatch> b;
        for (auto& c : P.batches)
            if (c->joined < c->want && c->next.load() < c->n) { b = c; break; }
        if (!b) {
            P.cv.wait(lk);
            continue;
        }
        b->joined++;
        lk.unlock();
        for (;;) {
            const size_t i = b->next.fetch_add(1);
            if (i >= b->n) break;
            b->fn(i);
            if (b->done.fetch_add(1) + 1 == b->n) {
                std::lock_guard<std::mutex> g(b->mu);
                b->cv.notify_all();
            }
        }
        lk.lock();
    }
}

static void run_parallel(size_t n, int threads, std::function<void(size_t)> fn) {
    if (n == 0) return;
    threads = std::max(1, std::min<int>(threads, (int)n));
    if (threads == 1) {  // the caller's own thread is the one seeder body
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    auto b = std::make_shared<PoolBatch>();
    b->fn = std::move(fn);
    b->n = n;
    b->want = threads;
    {
        WorkPool& P = *g_pool;
        std::lock_guard<std::mutex> lk(P.mu);
        P.batches.push_back(b);
        int wanted = 0;  // one worker per call the active batches may have in flight; the pool grows on demand and stays
        for (auto& c : P.batches) wanted += c->want;
        while (P.workers < std::min(wanted, 64)) {
            std::thread(pool_worker).detach();
            P.workers++;
        }
        P.cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> g(b->mu);
        b->cv.wait(g, [&] { return b->done.load() == b->n; });
    }
    WorkPool& P = *g_pool;
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto it = P.batches.begin(); it != P.batches.end(); ++it)
        if (it->get() == b.get()) { P.batches.erase(it); break; }
}

}  // namespace sa

using namespace sa;

// ====================================================================================================================
// C-ABI
// ====================================================================================================================
extern "C" {

const char* sa_version(void) { return "segalign_amd 0.1 (gfx950)"; }

void sa_select_devices(const int* ids, int n) {
    g_selected.clear();
    for (int i = 0; i < n; i++) g_selected.push_back(ids[i]);
}

void sa_shutdown_processor(void);
static void destroy_interface();

int sa_initialize_interface(int num_gpu) {  // seed_filter_interface.cu:49-80
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        fprintf(stderr, "Error: No GPU device found!\n");
        exit(1);
    }
    if (!g_selected.empty()) {
        for (int id : g_selected)
            if (id < 0 || id >= n) {
                fprintf(stderr, "Requested GPUs greater than available GPUs\n");
                exit(10);
            }
        n = (int)g_selected.size();
    }
    int use;
    if (num_gpu == -1) use = n;
    else if (num_gpu <= n) use = num_gpu;
    else {
        fprintf(stderr, "Requested GPUs greater than available GPUs\n");
        exit(10);
    }
    fprintf(stderr, "Using %d GPU(s)\n", use);
    if (!g_dev.empty()) destroy_interface();  // re-initialisation: release the previous contexts first
    g_ndev = use;
    for (int g = 0; g < use; g++) {
        const int ord = g_selected.empty() ? g : g_selected[g];
        check_set_device(ord, "InitializeInterface");
        DevCtx* dc = new DevCtx(arena_of(ord));
        dc->dev = ord;
        hipStreamCreateWithFlags(&dc->admin, hipStreamNonBlocking);
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, ord);
        dc->total_mem = prop.totalGlobalMem;
        g_dev.push_back(dc);
    }
    return use;
}

void sa_initialize_processor(int transition, uint32_t wga_chunk, uint32_t seed_size, const int* sub_mat, int xdrop,
                             int hspthresh, int noentropy) {  // src/seed_filter.cu:830-897
    require_init("InitializeProcessor");
    resolve_options();
    if (xdrop >= (1 << 25) || xdrop <= -(1 << 25)) {
        fprintf(stderr, "Error: |xdrop| must be below 2^25\n");
        exit(1);
    }
    g_transition = transition ? 1 : 0;
    g_wga_chunk = wga_chunk;
    g_max_seeds = transition ? 13ll * wga_chunk : (int64_t)wga_chunk;  // :836-839
    if (!g_max_hits_overridden) g_max_hits = max_hits_for_mem(g_dev[0]->total_mem);  // :832-841 (device 0)
    g_seed_size = seed_size;
    memcpy(g_sub_mat, sub_mat, sizeof(g_sub_mat));
    g_xdrop = xdrop;
    g_hspthresh = hspthresh;
    g_noentropy = noentropy ? 1 : 0;
    {
        int mx = g_sub_mat[0];
        for (int i = 1; i < 64; i++) mx = std::max(mx, g_sub_mat[i]);
        g_fast_filter = (xdrop >= 0 && (int64_t)7 * std::max(mx, 0) <= (int64_t)xdrop) ? 1 : 0;
        // int16 score arithmetic: the best of a side (<= max(M) * long_cap rounded up to whole 64-base windows) and xdrop itself must stay well inside
        // the saturation range, or a walk could never satisfy the drop test and every hit would become a candidate
        // the 4-bit query copies carry PACK_PAD bytes = 2 * PACK_PAD bases of padding: a capped walk must end inside it
        if (g_long_cap > 2 * PACK_PAD) g_long_cap = 2 * PACK_PAD;
        g_packed_filter = (xdrop >= 0 && xdrop <= 16383 && (int64_t)std::max(mx, 0) * (((int64_t)g_long_cap + 63) / 64 * 64) <= 16383 &&
                           !opt_value("no_packed_filter")) ? 1 : 0;
        if (opt_value("no_fast_filter")) { g_fast_filter = 0; g_packed_filter = 0; }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_tokens.clear();
    for (int g = 0; g < g_ndev; g++) {
        DevCtx* dc = g_dev[g];
        check_set_device(dc->dev, "InitializeProcessor");
        if (!dc->d_sub_mat) dc->d_sub_mat = (int*)dev_malloc(64 * sizeof(int), "sub_mat");
        check_memcpy(hipMemcpy(dc->d_sub_mat, g_sub_mat, 64 * sizeof(int), hipMemcpyHostToDevice), "sub_mat");
        for (int k = 0; k < SLOTS_PER_DEVICE; k++) {
            if (!dc->slots[k].stream) slot_init(dc->slots[k], dc->dev);
            dc->slots[k].seeds.ensure((size_t)g_max_seeds, "seed_offsets");
        }
        // start mapping the table arena now: the host still has its FASTA files to read (src/main.cpp:300-549)
        if (g_arena_gb > 0 && g_td && g_ctx && g_packed_filter) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t reserve = ((size_t)8 << 30) + ((size_t)4 << 30) * (size_t)SLOTS_PER_DEVICE;
                const size_t have = arena_mapped(dc->arena);
                const size_t room = free_b + have > reserve ? free_b + have - reserve : 0;
                arena_request(dc->arena, std::min<size_t>((size_t)g_arena_gb << 30, room));
            }
        }
    }
    // LIFO pool like available_gpus (:895): slot-major so that concurrent callers spread over devices first
    for (int k = SLOTS_PER_DEVICE - 1; k >= 0; k--)
        for (int g = g_ndev - 1; g >= 0; g--) g_tokens.push_back({g, k});
    g_proc_init = true;
}

// everything the engine holds on one device except the context itself (what the reference's cudaDeviceReset() wipes, :939)
static void release_device_state(DevCtx* dc) {
    check_set_device(dc->dev, "ShutdownProcessor");
    hipDeviceSynchronize();
    for (int k = 0; k < MAX_SLOTS_PER_DEVICE; k++) if (dc->slots[k].stream) slot_destroy(dc->slots[k]);
    dc->ref.release("d_ref_seq");
    dc->ref8.release("d_ref_seq rows");
    dc->ref2.release("d_ref_seq 2-bit");
    dc->ref_rc.release("d_seq_rc");
    dc->ref4.release("d_seq 4-bit");
    dc->ref4_rc.release("d_seq_rc 4-bit");
    dc->refq2.release("d_seq 2-bit shifted");
    dc->refq2_rc.release("d_seq_rc 2-bit shifted");
    dev_free(dc->d_present, "code presence");
    dc->d_present = nullptr;
    dc->ref_host_ptr = nullptr;
    nbr_release(dc);
    // (the table arena stays mapped: it is a cache of cleared device pages that cost seconds to get -- destroy_interface and
    //  option arena_gb = 0 give it back)
    if (g_arena_gb == 0) arena_destroy(dc->arena);
    dev_free(dc->bucket_start, "d_index_table");
    dev_free(dc->pos_table, "d_pos_table");
    dc->bucket_start = dc->pos_table = nullptr;
    dc->num_index = 0;
    for (int b = 0; b < SA_BUFFER_DEPTH; b++) {
        dc->query[b].release("d_query_seq");
        dc->query_rc[b].release("d_query_rc_seq");
        dc->query4[b].release("d_query_seq 4-bit");
        dc->query4_rc[b].release("d_query_rc_seq 4-bit");
        dc->query2[b].release("d_query_seq 2-bit");
        dc->query2_rc[b].release("d_query_rc_seq 2-bit");
    }
    dc->up_tmp.release("upload staging");
    for (int k = 0; k < 2; k++) {
        if (dc->up_pinned[k]) hipHostFree(dc->up_pinned[k]);
        if (dc->up_ev[k]) hipEventDestroy(dc->up_ev[k]);
        dc->up_pinned[k] = nullptr;
        dc->up_ev[k] = nullptr;
    }
    dev_free(dc->d_sub_mat, "sub_mat");
    dc->d_sub_mat = nullptr;
}

// g_ShutdownProcessor (src/seed_filter.cu:932-940): the reference clears its device vectors and resets the device, which
// also drops the target and the tables.  Same here; the INTERFACE (device list, contexts) stays, so a host may run
// InitializeProcessor / SendRefWriteRequest again without a second InitializeInterface.
void sa_shutdown_processor(void) {
    for (auto* dc : g_dev) release_device_state(dc);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_tokens.clear();
    }
    g_proc_init = false;
    for (uint32_t b = 0; b < SA_BUFFER_DEPTH; b++) g_query_len[b] = 0;
}

static void destroy_interface() {  // re-initialisation of the interface: contexts go too
    sa_shutdown_processor();
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "InitializeInterface");
        if (dc->admin) hipStreamDestroy(dc->admin);
        delete dc;
    }
    g_dev.clear();
    g_ndev = 0;
}

// ---- target ---------------------------------------------------------------------------------------------------------
void sa_send_ref_write_request(const char* seq, size_t addr, uint32_t len) {  // seed_filter_interface.cu:82-101
    require_init("SendRefWriteRequest");
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendRefWriteRequest");
        const uint8_t* tmp = upload_ascii(dc, seq + addr, len, "ref_seq");
        dc->ref.create(len, "ref_seq", dc->admin);
        launch_encode(tmp, dc->ref.codes, len, dc->admin);
        dc->ref8.create(len, "ref_seq rows", dc->admin, true);
        launch_row_code(dc->ref.codes, dc->ref8.codes, len, dc->admin);
        dc->ref2.create(dc->ref.codes, len, 2, "ref_seq 2-bit", dc->admin);
        presence_of(dc, dc->ref.codes, len, 0, &dc->ref_present);
        check_launch("compress_string");
        check_sync(dc->admin, "SendRefWriteRequest");
        dc->ref_host_ptr = seq + addr;
        nbr_release(dc);  // a neighbourhood table built for another block must not survive (its context records are target bases)
    }
}

void sa_clear_ref(void) {  // seed_filter_interface.cu:103-113
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "ClearRef");
        dc->ref.release("d_ref_seq");
        dc->ref8.release("d_ref_seq rows");
        dc->ref2.release("d_ref_seq 2-bit");
        dc->ref_host_ptr = nullptr;
        nbr_release(dc);
        dev_free(dc->bucket_start, "d_index_table");
        dev_free(dc->pos_table, "d_pos_table");
        dc->bucket_start = dc->pos_table = nullptr;
        dc->num_index = 0;
    }
}

int sa_generate_shape_pos(const char* shape) {  // ntcoding.cpp:21-37
    SeedShape sh;
    memset(&sh, 0, sizeof(sh));
    int n = 0, span = 0;
    for (int i = 0; shape[i] != '\0'; i++, span++) {
        if (shape[i] == '1' || shape[i] == 'T') {
            if (n >= MAX_CARE) {
                fprintf(stderr, "Error: seed weight above %d is not supported\n", MAX_CARE - 1);
                exit(1);
            }
            sh.pos[n] = (uint8_t)i;
            if (shape[i] == 'T') sh.transition_mask |= 1u << n;
            n++;
        }
    }
    if (span > 32) {
        fprintf(stderr, "Error: seed span above 32 is not supported\n");
        exit(1);
    }
    sh.weight = n;
    sh.span = span;
    g_shape = sh;
    return n;
}

void sa_generate_seed_pos_table(const char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step, int shape_size,
                                int kmer_size) {  // seed_pos_table.cu:49-109
    require_init("GenerateSeedPosTable");
    if (!(kmer_size <= 15 && kmer_size > 3)) {  // asserts at :51-52
        fprintf(stderr, "Error: GenerateSeedPosTable requires 3 < kmer_size <= 15\n");
        exit(1);
    }
    if (step == 0) step = 1;
    const uint32_t offset = (uint32_t)(shape_size + 1) % step;                       // :58
    const uint32_t start_offset = step - offset;                                     // :59
    const uint32_t nkeys = (uint32_t)1 << (2 * kmer_size);                           // :61
    const uint32_t num_steps = ref_length >= (uint32_t)shape_size ? (ref_length - (uint32_t)shape_size + offset) / step : 0;  // :64
    SeedShape sh = g_shape;
    sh.span = shape_size;
    // every device builds its own copy of the tables (the reference builds once on the host and replicates, seed_pos_table.cu:
    // 33-47); the builds are independent, so with several devices they run CONCURRENTLY, one host thread per device
    auto build_on = [&](DevCtx* dc) {
        check_set_device(dc->dev, "GenerateSeedPosTable");
        hipStream_t st = dc->admin;
        const uint8_t* codes = dc->ref.codes;
        SeqBuf tmp_codes;
        if (!(dc->ref.codes && dc->ref_host_ptr == ref_str + start_addr && dc->ref.len == ref_length)) {
            // not the resident block: encode a private copy
            const uint8_t* tmp = upload_ascii(dc, ref_str + start_addr, ref_length, "table seq");
            tmp_codes.create(ref_length, "table codes", st);
            launch_encode(tmp, tmp_codes.codes, ref_length, st);
            check_sync(st, "table encode");
            codes = tmp_codes.codes;
        }
        nbr_release(dc);
        dev_free(dc->bucket_start, "d_index_table");
        dev_free(dc->pos_table, "d_pos_table");
        dc->bucket_start = (uint32_t*)dev_malloc(((size_t)nkeys + 1) * sizeof(uint32_t), "index_table");
        uint32_t num_index = 0;
        if (table_partition_build_supported(kmer_size) && !g_table_atomic) {
            // PARTITION build (table.hip): keys + coarse histogram -> offsets of the 4096 coarse partitions -> two LDS-staged
            // partition passes -> one workgroup per partition finishes its slice of bucket_start and pos_table in LDS
            const size_t pw = table_partition_part_start_words();
            uint32_t* keys = (uint32_t*)dev_malloc((size_t)std::max<uint32_t>(num_steps, 1) * sizeof(uint32_t), "kmer keys");
            uint32_t* coarse = (uint32_t*)dev_malloc(3 * pw * sizeof(uint32_t) + 4096, "coarse histogram");  // hist | part_start | cursor | flags
            uint32_t* part_start = coarse + pw;
            uint32_t* cursor = part_start + pw;
            uint8_t* part_unsorted = reinterpret_cast<uint8_t*>(cursor + pw);
            void* scan_tmp = dev_malloc(scan_temp_bytes(pw), "scan temp");
            check_memcpy(hipMemsetAsync(coarse, 0, pw * sizeof(uint32_t), st), "coarse histogram");
            launch_table_keys(codes, num_steps, start_offset, step, sh, keys, coarse, st);
            launch_exclusive_scan_u32(coarse, part_start, pw - 1, scan_tmp, st);
            check_launch("table keys/scan");
            check_memcpy(hipMemcpyAsync(&num_index, part_start + (pw - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st), "num_index");
            check_sync(st, "table keys");
            const size_t np = std::max<uint32_t>(num_index, 1);
            dc->pos_table = (uint32_t*)dev_malloc(np * sizeof(uint32_t), "pos_table");
            uint32_t* pairs = (uint32_t*)dev_malloc(4 * np * sizeof(uint32_t), "partition pairs");  // key_a | pos_a | key_b | pos_b
            launch_table_partition_build(keys, num_steps, start_offset, step, kmer_size, part_start, num_index, cursor, pairs, pairs + np,
                                         pairs + 2 * np, pairs + 3 * np, part_unsorted, dc->bucket_start, dc->pos_table, st);
            check_launch("table partition");
            check_sync(st, "table partition");
            dev_free(pairs, "partition pairs");
            dev_free(keys, "kmer keys");
            dev_free(coarse, "coarse histogram");
            dev_free(scan_tmp, "scan temp");
        } else {
            // ATOMIC build: histogram + scatter with one global atomic per position (any seed weight)
            uint32_t* hist = (uint32_t*)dev_malloc(((size_t)nkeys + 1) * sizeof(uint32_t), "kmer histogram");
            void* scan_tmp = dev_malloc(scan_temp_bytes(nkeys), "scan temp");
            check_memcpy(hipMemsetAsync(hist, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), st), "histogram");
            launch_table_count(codes, num_steps, start_offset, step, sh, hist, st);
            launch_exclusive_scan_u32(hist, dc->bucket_start, nkeys, scan_tmp, st);
            check_launch("table count/scan");
            check_memcpy(hipMemcpyAsync(&num_index, dc->bucket_start + nkeys, sizeof(uint32_t), hipMemcpyDeviceToHost, st),
                         "num_index");
            check_sync(st, "table count");
            dc->pos_table = (uint32_t*)dev_malloc((size_t)std::max<uint32_t>(num_index, 1) * sizeof(uint32_t), "pos_table");
            check_memcpy(hipMemsetAsync(hist, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), st), "cursor");
            launch_table_fill(codes, num_steps, start_offset, step, sh, dc->bucket_start, hist, dc->pos_table, st);
            launch_table_sort_buckets(dc->bucket_start, nkeys, dc->pos_table, st);
            check_launch("table fill/sort");
            check_sync(st, "table fill");
            dev_free(hist, "kmer histogram");
            dev_free(scan_tmp, "scan temp");
        }
        tmp_codes.release("table codes");
        dc->num_index = num_index;
        dc->nkeys = nkeys;
        // the neighbourhood table belongs to the table build when the processor parameters are already known (the reference
        // calls InitializeProcessor first, src/main.cpp:298 before :621); otherwise the first table-direct call builds it
        if (g_proc_init && g_packed_filter) ensure_nbr(dc);
    };
    if (g_dev.size() <= 1) {
        for (auto* dc : g_dev) build_on(dc);
    } else {
        std::vector<std::thread> builders;
        for (auto* dc : g_dev) builders.emplace_back(build_on, dc);
        for (auto& t : builders) t.join();
    }
}

// ---- query ----------------------------------------------------------------------------------------------------------
void sa_send_query_write_request(const char* query_buffer, size_t addr, uint32_t len, uint32_t buffer) {  // :899-919
    require_init("SendQueryWriteRequest");
    if (buffer >= SA_BUFFER_DEPTH) {
        fprintf(stderr, "Error: query buffer %u out of range\n", buffer);
        exit(1);
    }
    g_query_len[buffer] = len;
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendQueryWriteRequest");
        hipStream_t st = dc->admin;
        const uint8_t* tmp = upload_ascii(dc, query_buffer + addr, len, "query_seq");
        dc->query[buffer].create(len, "query_seq", st);
        dc->query_rc[buffer].create(len, "query_rc_seq", st);
        launch_encode_rev_comp(tmp, dc->query[buffer].codes, dc->query_rc[buffer].codes, len, st);
        dc->query4[buffer].create(dc->query[buffer].codes, len, 4, "query_seq 4-bit", st);
        dc->query4_rc[buffer].create(dc->query_rc[buffer].codes, len, 4, "query_rc_seq 4-bit", st);
        dc->query2[buffer].create_q2(dc->query[buffer].codes, len, "query_seq 2-bit", st);
        dc->query2_rc[buffer].create_q2(dc->query_rc[buffer].codes, len, "query_rc_seq 2-bit", st);
        presence_of(dc, dc->query[buffer].codes, len, 1 + (int)buffer, &dc->query_present[buffer]);
        check_launch("compress_string_rev_comp");
        check_sync(st, "SendQueryWriteRequest");
    }
}

void sa_clear_query(uint32_t buffer) {  // :921-930
    if (buffer >= SA_BUFFER_DEPTH) return;
    for (auto* dc : g_dev) {
        // the reference frees here (:921-930); the engine keeps the allocations for the next block of this buffer: a
        // hipFree would synchronise the device under the calls that are running on the OTHER query buffer
        dc->query[buffer].clear();
        dc->query_rc[buffer].clear();
        dc->query4[buffer].clear();
        dc->query4_rc[buffer].clear();
        dc->query2[buffer].clear();
        dc->query2_rc[buffer].clear();
    }
}

// ---- hot calls ------------------------------------------------------------------------------------------------------
size_t sa_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_proc("SeedAndFilter", buffer);
    if ((int64_t)num_seeds > g_max_seeds) {  // :688-692
        printf("MAX_SEEDS exceeded\n");
        fflush(stdout);
        fprintf(stderr, "Assertion `num_seeds <= MAX_SEEDS' failed.\n");
        abort();
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    upload_seeds(sl, seeds, num_seeds);
    CoreArgs ca = {rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes, g_query_len[buffer], 0, 0, 0, 0, 0, 0, nullptr, 0,
                   rev ? &dc->query4_rc[buffer] : &dc->query4[buffer]};  // :762-767
    set_query2(ca, dc, buffer, rev);
    uint32_t lo = 0, hi = 0, words = 0;
    if (dropin_td_front(dc, sl, ca.query, ca.query_len, seeds, num_seeds, ca.query4, 0, &lo, &hi, &words) != 0xFFFFFFFFu) {
        ca.td = 1;
        ca.td_words = words;
        ca.q_lo = lo;
        ca.q_hi = hi;
    }
    size_t n = saf_core(dc, sl, (uint32_t)num_seeds, ca, out);
    release_slot(sl);
    return n;
}

size_t sa_seed_and_filter_range(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_proc("SeedAndFilterRange", buffer);
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    uint32_t qlen = g_query_len[buffer];
    // a seed window must lie inside the block: positions j with j + span <= len
    uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    if (end > lim) end = lim;
    const PackedBuf* q4 = rev ? &dc->query4_rc[buffer] : &dc->query4[buffer];
    uint32_t ns = 0xFFFFFFFFu, words = 0;
    if (td_eligible(dc, q4)) {
        const uint32_t bp[2] = {start, std::max(start, end)};
        ns = td_front(dc, sl, q, 1, bp, 0, &words);
    }
    const bool td = ns != 0xFFFFFFFFu;
    if (!td) ns = device_seeds(sl, q, start, end);
    size_t n = 0;
    *out = nullptr;
    if (ns > 0) {  // seeder.cpp:76: the engine is only called for a non-empty seed vector
        CoreArgs ca = {q, qlen, 0, 0, 0, 0, start, end, nullptr, 0, q4};
        set_query2(ca, dc, buffer, rev);
        ca.td = td ? 1 : 0;
        ca.td_words = words;
        n = saf_core(dc, sl, ns, ca, out);
    } else {
        prof_flush(sl);
        memset(&t_stats, 0, sizeof(t_stats));
    }
    release_slot(sl);
    return n;
}

// Up to SA_MAX_CHUNKS consecutive wga_chunk-sized chunks of one strand in ONE pass over the kernels: the chunks share the
// seeding, lookup, expansion, extension, grouping and ordering launches and the host syncs, while every chunk keeps its own
// iteration plan, dedup scope and return vector -- bit for bit what one sa_seed_and_filter_range call per chunk returns.
int sa_max_chunks_per_call(void) { return SA_MAX_CHUNKS; }
int sa_get_chunks_per_call(void) { return g_chunks_per_call; }  // what sa_seed_interval hands to one call (SEGALIGN_AMD_CHUNKS_PER_CALL)
size_t sa_seed_and_filter_chunks(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** outs, size_t* counts) {
    require_proc("SeedAndFilterChunks", buffer);
    const uint32_t chunk = g_wga_chunk;
    const int K = end > start ? (int)(((uint64_t)end - start + chunk - 1) / chunk) : 0;
    if (K > SA_MAX_CHUNKS) {
        fprintf(stderr, "Error: SeedAndFilterChunks takes at most %d chunks per call\n", SA_MAX_CHUNKS);
        exit(1);
    }
    for (int c = 0; c < K; c++) { outs[c] = nullptr; counts[c] = 0; }
    if (K == 0) return 0;
    if (K == 1) {
        counts[0] = sa_seed_and_filter_range(start, end, rev, buffer, &outs[0]);
        return counts[0];
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    const uint32_t qlen = g_query_len[buffer];
    const uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    const uint32_t send = std::min(end, lim);  // a seed window must lie inside the block
    uint32_t bpos[SA_MAX_CHUNKS + 1], bseed[SA_MAX_CHUNKS + 1];
    for (int c = 0; c <= K; c++) bpos[c] = (uint32_t)std::min<uint64_t>((uint64_t)start + (uint64_t)c * chunk, send);
    const PackedBuf* q4 = rev ? &dc->query4_rc[buffer] : &dc->query4[buffer];
    uint32_t ns = 0xFFFFFFFFu, words = 0;
    if (td_eligible(dc, q4)) ns = td_front(dc, sl, q, K, bpos, 0, &words);
    const bool td = ns != 0xFFFFFFFFu;
    if (!td) ns = device_seeds(sl, q, start, send, K + 1, bpos, bseed);
    size_t total = 0;
    if (ns > 0) {
        CoreArgs ca = {q, qlen, 0, 0, 0, 0, start, send, nullptr, 0, q4};
        set_query2(ca, dc, buffer, rev);
        ca.nchunks = K;
        ca.td = td ? 1 : 0;
        ca.td_words = words;
        for (int c = 0; c <= K; c++) ca.seed_bound[c] = td ? 0u : bseed[c];  // (a table-direct call derives them from its chunk plans)
        ca.outs = outs;
        ca.counts = counts;
        saf_core(dc, sl, ns, ca, nullptr);
        for (int c = 0; c < K; c++) total += counts[c];
    } else {
        prof_flush(sl);
        memset(&t_stats, 0, sizeof(t_stats));
    }
    release_slot(sl);
    return total;
}

void sa_free_segments(sa_segment_pair* p) { free(p); }

// Introspection (tests): the extension stage alone -- find_hsps + compaction of the passing hits (src/seed_filter.cu:232-680)
// -- for caller-supplied anchors {ref_loc, query_loc} on the resident target / query strand.  Returns 1 + the number of
// passing hits; out[0] is a header {len = count}, the records follow in no particular order and are NOT de-duplicated,
// except that exact duplicates may already be merged (the chain shortcut extends one member of a run of anchors that
// provably produce the identical record).
size_t sa_extend_hits(const uint32_t* ref_query_pairs, size_t num_hits, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_init("ExtendHits");
    if (buffer >= SA_BUFFER_DEPTH) {
        fprintf(stderr, "Error: query buffer %u out of range\n", buffer);
        exit(1);
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    size_t n = 0;
    *out = nullptr;
    if (num_hits > 0) {
        sl->hits.ensure(num_hits, "hits");
        check_memcpy(hipMemcpyAsync(sl->hits.p, ref_query_pairs, num_hits * sizeof(Hit), hipMemcpyHostToDevice, sl->stream), "hits");
        CoreArgs ca = {rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes, g_query_len[buffer], 0, 0, 0, 0, 0, 0, nullptr, 0,
                       rev ? &dc->query4_rc[buffer] : &dc->query4[buffer]};
        set_query2(ca, dc, buffer, rev);
        ca.raw_hits = num_hits;
        n = saf_core(dc, sl, 1, ca, out);
    }
    release_slot(sl);
    return n;
}

// seeder_body::operator() of src/seeder.cpp:12-127 for one query interval: plus-strand chunks [start, end) in steps of
// wga_chunk, then the minus-strand chunks of the same interval in reverse-complement coordinates (:33-34,89-91), every
// chunk through sa_seed_and_filter_range; `threads` chunk calls are kept in flight (the reference keeps one per TBB
// worker).  HSPs are concatenated per strand in chunk order, headers removed (:80-85,115-120).
size_t sa_seed_interval(uint32_t start, uint32_t end, uint32_t q_len, int strands, uint32_t buffer, int threads,
                        sa_segment_pair** out_fw, size_t* n_fw, sa_segment_pair** out_rc, size_t* n_rc, sa_call_stats* totals) {
    require_proc("SeedInterval", buffer);
    struct Job { uint32_t a, b; int rev; int k; sa_segment_pair* res[SA_MAX_CHUNKS]; size_t n[SA_MAX_CHUNKS]; };
    std::vector<Job> jobs;
    const int per_job = g_chunks_per_call;  // chunks of one strand that share one pass over the kernels
    for (int rev = 0; rev < 2; rev++) {
        if (!(strands & (rev ? SA_STRAND_MINUS : SA_STRAND_PLUS))) continue;
        const uint32_t a = rev ? q_len - end : start, b = rev ? q_len - start : end;
        // equal groups: 40 chunks go as 14 + 14 + 12, not 16 + 16 + 8 (the calls of an interval finish together)
        const uint64_t nchunks = b > a ? ((uint64_t)b - a + g_wga_chunk - 1) / g_wga_chunk : 0;
        const uint64_t ncalls = (nchunks + per_job - 1) / per_job;
        const uint64_t group = ncalls ? (nchunks + ncalls - 1) / ncalls : 1;
        for (uint64_t i = a; i < b; i += (uint64_t)g_wga_chunk * group) {
            Job jb;
            memset(&jb, 0, sizeof(jb));
            jb.a = (uint32_t)i;
            jb.b = (uint32_t)std::min<uint64_t>(i + (uint64_t)g_wga_chunk * group, b);
            jb.rev = rev;
            jb.k = (int)(((uint64_t)jb.b - jb.a + g_wga_chunk - 1) / g_wga_chunk);
            jobs.push_back(jb);
        }
    }
    sa_call_stats tot;
    memset(&tot, 0, sizeof(tot));
    std::mutex mu;
    run_parallel(jobs.size(), threads, [&](size_t j) {  // (pool threads of the engine: nothing is created per interval)
        Job& jb = jobs[j];
        sa_seed_and_filter_chunks(jb.a, jb.b, jb.rev, buffer, jb.res, jb.n);
        std::lock_guard<std::mutex> lk(mu);
        tot.num_seeds += t_stats.num_seeds;
        tot.num_hits += t_stats.num_hits;
        tot.num_survivors += t_stats.num_survivors;
        tot.num_anchors += t_stats.num_anchors;
        tot.num_examined += t_stats.num_examined;
        tot.num_examined_filter += t_stats.num_examined_filter;
        tot.num_candidates += t_stats.num_candidates;
        tot.num_forwarded += t_stats.num_forwarded;
        tot.num_entropy += t_stats.num_entropy;
        tot.num_iter += t_stats.num_iter;
        tot.device = t_stats.device;
        tot.lookup_path = t_stats.lookup_path;
    });
    size_t cnt[2] = {0, 0};
    for (const Job& jb : jobs)
        for (int c = 0; c < jb.k; c++) if (jb.n[c] > 1) cnt[jb.rev] += jb.n[c] - 1;
    sa_segment_pair* dst[2];
    for (int r = 0; r < 2; r++) dst[r] = (sa_segment_pair*)malloc(std::max<size_t>(cnt[r], 1) * sizeof(sa_segment_pair));
    size_t off[2] = {0, 0};
    for (Job& jb : jobs)
        for (int c = 0; c < jb.k; c++) {
            if (jb.n[c] > 1) {
                memcpy(dst[jb.rev] + off[jb.rev], jb.res[c] + 1, (jb.n[c] - 1) * sizeof(sa_segment_pair));
                off[jb.rev] += jb.n[c] - 1;
            }
            free(jb.res[c]);
        }
    *out_fw = dst[0]; *n_fw = cnt[0];
    *out_rc = dst[1]; *n_rc = cnt[1];
    if (totals) *totals = tot;
    return cnt[0] + cnt[1];
}

// A list of independent calls -- each up to sa_max_chunks_per_call() consecutive chunks of one strand -- run with `threads` of
// them in flight on the engine's worker pool.  This is the unit a multi-GPU host deals out: any subset of the calls of a pass
// may run on any device.  results[i]: the HSPs of call i, its chunks concatenated in order, headers removed.
size_t sa_seed_calls(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, sa_call_result* results,
                     sa_call_stats* totals) {
    require_proc("SeedCalls", buffer);
    sa_call_stats tot;
    memset(&tot, 0, sizeof(tot));
    std::mutex mu;
    run_parallel(num_calls, threads, [&](size_t i) {
        sa_segment_pair* res[SA_MAX_CHUNKS];
        size_t cnt[SA_MAX_CHUNKS];
        const sa_call_desc& c = calls[i];
        const int K = c.end > c.start ? (int)(((uint64_t)c.end - c.start + g_wga_chunk - 1) / g_wga_chunk) : 0;
        sa_seed_and_filter_chunks(c.start, c.end, c.rev, buffer, res, cnt);
        size_t n = 0;
        for (int k = 0; k < K; k++)
            if (cnt[k] > 0) n += cnt[k] - 1;
        sa_segment_pair* out = (sa_segment_pair*)malloc(std::max<size_t>(n, 1) * sizeof(sa_segment_pair));
        size_t off = 0;
        for (int k = 0; k < K; k++) {
            if (cnt[k] > 1) {
                memcpy(out + off, res[k] + 1, (cnt[k] - 1) * sizeof(sa_segment_pair));
                off += cnt[k] - 1;
            }
            free(res[k]);
        }
        results[i].hsps = out;
        results[i].num_hsps = n;
        results[i].num_hits = t_stats.num_hits;
        std::lock_guard<std::mutex> lk(mu);
        tot.num_seeds += t_stats.num_seeds;
        tot.num_hits += t_stats.num_hits;
        tot.num_survivors += t_stats.num_survivors;
        tot.num_anchors += t_stats.num_anchors;
        tot.num_examined += t_stats.num_examined;
        tot.num_examined_filter += t_stats.num_examined_filter;
        tot.num_candidates += t_stats.num_candidates;
        tot.num_forwarded += t_stats.num_forwarded;
        tot.num_entropy += t_stats.num_entropy;
        tot.num_iter += t_stats.num_iter;
        tot.device = t_stats.device;
        tot.lookup_path = t_stats.lookup_path;
    });
    size_t total = 0;
    for (size_t i = 0; i < num_calls; i++) total += results[i].num_hsps;
    if (totals) *totals = tot;
    return total;
}

size_t sa_device_make_seeds(uint32_t start, uint32_t end, int rev, uint32_t buffer, uint64_t* dst, size_t cap) {
    require_proc("DeviceMakeSeeds", buffer);
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    uint32_t qlen = g_query_len[buffer];
    uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    if (end > lim) end = lim;
    uint32_t ns = device_seeds(sl, q, start, end);
    size_t ncopy = std::min<size_t>(ns, cap);
    if (ncopy) {
        check_memcpy(hipMemcpyAsync(dst, sl->seeds.p, ncopy * sizeof(uint64_t), hipMemcpyDeviceToHost, sl->stream), "seeds d2h");
        check_sync(sl->stream, "seeds d2h");
    }
    prof_flush(sl);
    release_slot(sl);
    return ns;
}

// ---- repeat masker --------------------------------------------------------------------------------------------------
void sa_rm_send_query_write_request(void) {  // rm :951-961
    require_init("SendQueryWriteRequest");
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendQueryWriteRequest");
        dc->ref_rc.create(dc->ref.len, "seq_rc", dc->admin);
        launch_rev_comp_codes(dc->ref.codes, dc->ref_rc.codes, dc->ref.len, dc->admin);
        dc->ref4.create(dc->ref.codes, dc->ref.len, 4, "seq 4-bit", dc->admin);
        dc->ref4_rc.create(dc->ref_rc.codes, dc->ref.len, 4, "seq_rc 4-bit", dc->admin);
        dc->refq2.create_q2(dc->ref.codes, dc->ref.len, "seq 2-bit shifted", dc->admin);
        dc->refq2_rc.create_q2(dc->ref_rc.codes, dc->ref.len, "seq_rc 2-bit shifted", dc->admin);
        check_launch("rev_comp_string");
        check_sync(dc->admin, "SendQueryWriteRequest");
    }
}
void sa_rm_clear_query(void) {  // rm :964-972
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "ClearQuery");
        dc->ref_rc.release("d_seq_rc");
        dc->ref4.release("d_seq 4-bit");
        dc->ref4_rc.release("d_seq_rc 4-bit");
        dc->refq2.release("d_seq 2-bit shifted");
        dc->refq2_rc.release("d_seq_rc 2-bit shifted");
    }
}
size_t sa_rm_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t ref_start, uint32_t ref_end,
                             sa_segment_pair** out) {  // rm :724-876
    require_init("SeedAndFilter");
    if ((int64_t)num_seeds > g_max_seeds) {
        printf("MAX_SEEDS exceeded\n");
        fflush(stdout);
        fprintf(stderr, "Assertion `num_seeds <= MAX_SEEDS' failed.\n");
        abort();
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    upload_seeds(sl, seeds, num_seeds);
    CoreArgs ca = {rev ? dc->ref_rc.codes : dc->ref.codes, dc->ref.len, 1, rev ? 1 : 0, ref_start, ref_end, 0, 0, nullptr, 0,
                   rev ? &dc->ref4_rc : &dc->ref4};  // rm :805-810
    set_query2_rm(ca, dc, rev);
    uint32_t lo = 0, hi = 0, words = 0;
    if (dropin_td_front(dc, sl, ca.query, ca.query_len, seeds, num_seeds, ca.query4, 1, &lo, &hi, &words) != 0xFFFFFFFFu) {
        ca.td = 1;
        ca.td_words = words;
        ca.q_lo = lo;
        ca.q_hi = hi;
    }
    size_t n = saf_core(dc, sl, (uint32_t)num_seeds, ca, out);
    release_slot(sl);
    return n;
}

// ---- repeat-masker post-processing on the device (8f-4) -------------------------------------------------------------
namespace sa {

constexpr uint32_t COV_TILE = 1u << 26;  // positions per scan tile (bounds the scratch at 5 x 256 MiB)

// make the slot's difference array cover `block_len` positions, all zero, and reset the touched range
static void coverage_begin(Slot* sl, uint32_t block_len) {
    hipStream_t st = sl->stream;
    const size_t need = (size_t)block_len + 1;
    if (sl->cov_diff.cap < need) {  // a fresh allocation is cleared once; afterwards only the touched range is re-cleared
        sl->cov_diff.ensure(need, "coverage diff");
        check_memcpy(hipMemsetAsync(sl->cov_diff.p, 0, sl->cov_diff.cap * sizeof(uint32_t), st), "coverage diff");
    }
    launch_coverage_range_reset(sl->d_cov_range, st);
    check_launch("coverage begin");
}

// runs with (depth mod 256) >= M over the touched range -> malloc-ed sa_interval list; leaves the array zeroed again
static size_t coverage_finish(Slot* sl, uint32_t block_len, uint32_t M, uint64_t num_hsps, sa_interval** out) {
    hipStream_t st = sl->stream;
    *out = nullptr;
    check_memcpy(hipMemcpyAsync(sl->h_cov, sl->d_cov_range, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage range");
    check_sync(st, "coverage range");
    const uint32_t lo = sl->h_cov[0], hi = sl->h_cov[1];
    if (lo == 0xFFFFFFFFu) return 0;  // nothing was counted: depth 0 everywhere (M == 0: one unterminated run, see below)
    size_t n_out = 0;
    // M == 0 makes every position of the block "covered": one run that reaches the end of the block and is therefore
    // never written (seeder.cpp:168-186 has no flush after the loop)
    if (M > 0 && M <= 255) {
        // every run boundary sits on a position with a non-zero difference, so there are at most 2 per HSP
        const uint64_t cap64 = std::min<uint64_t>(2 * num_hsps + 2, (uint64_t)block_len + 1);
        const uint32_t cap = (uint32_t)std::min<uint64_t>(cap64, 0x7FFFFFFFull);
        sl->cov_pairs.ensure((size_t)cap * 2, "coverage intervals");
        const uint32_t span = hi - lo + 1;  // depth is 0 before lo and from hi on
        const uint32_t tile_cap = std::min(span, COV_TILE);
        sl->cov_pre.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->cov_is_start.ensure(tile_cap, "coverage scan");
        sl->cov_is_end.ensure(tile_cap, "coverage scan");
        sl->cov_sidx.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->cov_eidx.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->scan_temp.ensure(scan_temp_bytes(tile_cap), "scan temp");
        uint32_t depth = 0, nstart = 0, nend = 0;
        for (uint64_t off = 0; off < span; off += COV_TILE) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(COV_TILE, span - off);
            const uint32_t pos0 = lo + (uint32_t)off;
            const uint32_t* d = sl->cov_diff.p + pos0;
            ProfScope p(sl, "coverage_runs");
            launch_exclusive_scan_u32(d, sl->cov_pre.p, n, sl->scan_temp.p, st);
            launch_coverage_flags(d, sl->cov_pre.p, n, depth, M, sl->cov_is_start.p, sl->cov_is_end.p, st);
            launch_exclusive_scan_u32(sl->cov_is_start.p, sl->cov_sidx.p, n, sl->scan_temp.p, st);
            launch_exclusive_scan_u32(sl->cov_is_end.p, sl->cov_eidx.p, n, sl->scan_temp.p, st);
            launch_coverage_emit(sl->cov_is_start.p, sl->cov_is_end.p, sl->cov_sidx.p, sl->cov_eidx.p, n, pos0, nstart, nend, cap,
                                 sl->cov_pairs.p, st);
            check_launch("coverage runs");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[2], sl->cov_pre.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[3], sl->cov_sidx.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[4], sl->cov_eidx.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_sync(st, "coverage totals");
            depth += sl->h_cov[2];
            nstart += sl->h_cov[3];
            nend += sl->h_cov[4];
        }
        // the depth returns to 0 at `hi`, so every run that started has ended (nstart == nend)
        n_out = std::min(nend, cap);
        if (n_out > 0) {
            launch_coverage_finish(sl->cov_pairs.p, (uint32_t)n_out, st);
            check_launch("coverage finish");
            sa_interval* res = (sa_interval*)malloc(n_out * sizeof(sa_interval));
            if (!res) {
                fprintf(stderr, "Error: malloc for the interval list failed\n");
                exit(12);
            }
            check_memcpy(hipMemcpyAsync(res, sl->cov_pairs.p, n_out * sizeof(sa_interval), hipMemcpyDeviceToHost, st), "intervals");
            check_sync(st, "intervals");
            *out = res;
        }
    }
    // leave the difference array zeroed for the next interval
    check_memcpy(hipMemsetAsync(sl->cov_diff.p + lo, 0, ((size_t)hi - lo + 1) * sizeof(uint32_t), st), "coverage clear");
    check_sync(st, "coverage clear");
    return n_out;
}

}  // namespace sa

size_t sa_rm_mask_interval(uint32_t start_pos, uint32_t end_pos, uint32_t ref_start, uint32_t ref_end, int strands, uint32_t M,
                           sa_interval** out, uint64_t* totals) {  // repeat_masker_src/seeder.cpp:28-195
    require_init("MaskInterval");
    Slot* sl = acquire_slot();
    DevCtx* dc = g_dev[0];
    for (auto* d : g_dev) if (d->dev == sl->dev) dc = d;
    const uint32_t block_len = dc->ref.len;
    if (!dc->ref_rc.codes && (strands & SA_STRAND_MINUS)) {
        fprintf(stderr, "Error: MaskInterval on the minus strand before SendQueryWriteRequest\n");
        exit(1);
    }
    coverage_begin(sl, block_len);
    // a seed window must lie inside the block (the reference reads its host arena past the block end for the last
    // positions of the minus strand of a block's first interval: undefined there, no seed here)
    const uint32_t lim = block_len >= g_seed_size ? block_len - g_seed_size + 1 : 0;
    const uint32_t end_pos_rc = block_len - 1 - start_pos;  // seeder.cpp:46-47
    uint64_t tot_seeds = 0, tot_hits = 0, tot_hsps = 0, tot_ex = 0, tot_exf = 0, tot_cand = 0;
    int path = 0;
    // The reference walks the plus-strand chunks and derives a minus-strand chunk from each (:73-150).  Coverage counting is
    // order independent and every chunk keeps its own iteration plan and dedup scope, so the chunks of a strand are grouped:
    // consecutive chunks that tile a range go through ONE table-direct pass (up to g_chunks_per_call of them), the rest --
    // the minus-strand chunk of a short last plus chunk overlaps its neighbour (:118-119) -- go on their own.
    struct Range { uint32_t s0, s1; };
    for (int rev = 0; rev < 2; rev++) {
        if (!(strands & (rev ? SA_STRAND_MINUS : SA_STRAND_PLUS))) continue;
        std::vector<Range> rs;
        for (uint64_t i = start_pos; i < end_pos; i += g_wga_chunk) {  // :73
            const uint32_t start = (uint32_t)i;
            const uint32_t end = (uint32_t)std::min<uint64_t>(i + g_wga_chunk, end_pos);  // :76-77
            uint32_t s0 = start, s1 = end;
            if (rev) {  // :118-119: the minus-strand chunk is derived from the plus-strand chunk END
                s0 = block_len - 1 - end;
                s1 = (uint32_t)std::min<uint64_t>((uint64_t)s0 + g_wga_chunk, end_pos_rc);
            }
            if (s1 > lim) s1 = lim;
            if (s1 > s0) rs.push_back({s0, s1});
        }
        if (rev) std::reverse(rs.begin(), rs.end());  // ascending positions
        const uint8_t* q = rev ? dc->ref_rc.codes : dc->ref.codes;
        const PackedBuf* q4 = rev ? &dc->ref4_rc : &dc->ref4;
        const bool td_ok = td_eligible(dc, q4);
        size_t a = 0;
        while (a < rs.size()) {
            size_t b = a + 1;
            while (td_ok && b < rs.size() && (int)(b - a) < g_chunks_per_call && rs[b].s0 == rs[b - 1].s1) b++;
            int Kc = (int)(b - a);
            uint32_t bp[SA_MAX_CHUNKS + 1];
            for (int c = 0; c < Kc; c++) bp[c] = rs[a + c].s0;
            bp[Kc] = rs[b - 1].s1;
            uint32_t ns = 0xFFFFFFFFu, words = 0;
            if (td_ok) ns = td_front(dc, sl, q, Kc, bp, 1, &words);
            if (ns == 0xFFFFFFFFu && Kc > 1) {  // one of the chunks needs the general path (MAX_HITS): one chunk per call
                b = a + 1;
                Kc = 1;
                bp[1] = rs[a].s1;
                ns = td_front(dc, sl, q, 1, bp, 1, &words);
            }
            const bool td = ns != 0xFFFFFFFFu;
            if (!td) ns = device_seeds(sl, q, rs[a].s0, rs[a].s1);
            if (ns != 0) {  // :103,140
                CoreArgs ca = {q, block_len, 1, rev, ref_start, ref_end, bp[0], bp[Kc], sl->cov_diff.p, block_len + 1, q4};
                set_query2_rm(ca, dc, rev);
                ca.td = td ? 1 : 0;
                ca.td_words = words;
                if (Kc > 1) {
                    ca.nchunks = Kc;
                    for (int c = 0; c <= Kc; c++) ca.seed_bound[c] = 0u;  // (a table-direct call derives them from its chunk plans)
                }
                saf_core(dc, sl, ns, ca, nullptr);
                tot_seeds += ns;
                tot_hits += t_stats.num_hits;
                tot_hsps += t_stats.num_anchors;
                tot_ex += t_stats.num_examined;
                tot_exf += t_stats.num_examined_filter;
                tot_cand += t_stats.num_candidates;
                path = t_stats.lookup_path;
            }
            a = b;
        }
    }
    const size_t n = coverage_finish(sl, block_len, M, tot_hsps, out);
    prof_flush(sl);
    release_slot(sl);
    if (totals) { totals[0] = tot_seeds; totals[1] = tot_hits; totals[2] = tot_hsps; }
    // sa_get_last_call_stats after an interval call: the sums over its SeedAndFilter passes
    t_stats.num_seeds = tot_seeds;
    t_stats.num_hits = tot_hits;
    t_stats.num_anchors = tot_hsps;
    t_stats.num_examined = tot_ex;
    t_stats.num_examined_filter = tot_exf;
    t_stats.num_candidates = tot_cand;
    t_stats.lookup_path = path;
    return n;
}

size_t sa_rm_coverage_intervals(const sa_segment_pair* hsps, size_t num_hsps, uint32_t block_len, uint32_t M, sa_interval** out) {
    require_init("CoverageIntervals");  // repeat_masker_src/seeder.cpp:153-188
    Slot* sl = acquire_slot();
    hipStream_t st = sl->stream;
    coverage_begin(sl, block_len);
    const size_t BATCH = 1u << 24;
    for (size_t off = 0; off < num_hsps; off += BATCH) {
        const size_t n = std::min(BATCH, num_hsps - off);
        sl->out16.ensure(n, "out16");
        check_memcpy(hipMemcpyAsync(sl->out16.p, hsps + off, n * sizeof(sa_segment_pair), hipMemcpyHostToDevice, st), "hsps h2d");
        launch_coverage_add_pairs(reinterpret_cast<const SegPair16*>(sl->out16.p), (uint32_t)n, sl->cov_diff.p, block_len + 1,
                                  sl->d_cov_range, st);
        check_launch("coverage add");
        check_sync(st, "coverage add");
    }
    const size_t n = coverage_finish(sl, block_len, M, num_hsps, out);
    prof_flush(sl);
    release_slot(sl);
    return n;
}

void sa_free_intervals(sa_interval* p) { free(p); }

// ---- knobs ----------------------------------------------------------------------------------------------------------
void sa_set_max_hits(int64_t max_hits) {
    if (max_hits <= 0) {
        g_max_hits_overridden = false;
        if (g_ndev > 0) g_max_hits = max_hits_for_mem(g_dev[0]->total_mem);
    } else {
        g_max_hits = max_hits;
        g_max_hits_overridden = true;
    }
}
int64_t sa_get_max_hits(void) { return g_max_hits; }

// One documented switchboard (include/segalign_amd.h): takes effect at the next InitializeProcessor.
int sa_set_option(const char* name, int64_t value) {
    Option* o = name ? find_option(name) : nullptr;
    if (!o) return -1;
    o->api_value = value;
    o->api_set = true;
    return 0;
}
int sa_reset_option(const char* name) {  // back to environment / default; NULL resets every option
    if (!name) {
        for (auto& o : g_opts) o.api_set = false;
        return 0;
    }
    Option* o = find_option(name);
    if (!o) return -1;
    o->api_set = false;
    return 0;
}
int64_t sa_get_option(const char* name) {  // the value the engine resolved at the last InitializeProcessor
    Option* o = name ? find_option(name) : nullptr;
    return o ? o->value : INT64_MIN;
}
int sa_option_count(void) { return (int)(sizeof(g_opts) / sizeof(g_opts[0])); }
const char* sa_option_name(int i, int* test_only) {
    if (i < 0 || i >= sa_option_count()) return nullptr;
    if (test_only) *test_only = g_opts[i].test_only;
    return g_opts[i].name;
}
// (tests, option audit_cap) the hits the X-drop filter levels REJECTED in the calling thread's last table-direct call, as
// {ref_loc, query_loc} pairs; returns how many were recorded (<= audit_cap)
size_t sa_get_audit(uint32_t* dst_pairs, size_t cap_pairs) {
    const size_t n = std::min(cap_pairs, t_audit.size());
    if (n) memcpy(dst_pairs, t_audit.data(), n * sizeof(uint2));
    return t_audit.size();
}
int sa_max_hits_for_mem(uint64_t total_global_mem) { return max_hits_for_mem(total_global_mem); }

// ---- introspection --------------------------------------------------------------------------------------------------
void sa_get_last_call_stats(sa_call_stats* o) { *o = t_stats; }
void sa_set_count_examined(int on) { g_count_examined = on != 0; }
int sa_get_filter_mode(void) {  // which X-drop filter kernel the next plain (non repeat-masker) call uses
    if (g_count_examined) return g_fast_filter ? 1 : 0;
    return g_packed_filter ? 3 : g_fast_filter;
}
int sa_get_lookup_mode(void) {  // how device-seeded calls look seeds up on device 0 right now (builds the table if needed)
    if (g_ndev <= 0 || !g_proc_init) return 0;
    DevCtx* dc = g_dev[0];
    check_set_device(dc->dev, "lookup mode");
    if (!(g_td && g_packed_filter && !g_count_examined && dc->ref2.base && ensure_nbr(dc))) return 0;
    return dc->nbr_ctx ? 2 : 1;
}
uint64_t sa_get_neighbourhood_entries(void) { return (g_ndev > 0 && g_dev[0]->nbr_state == 1) ? g_dev[0]->nbr_total : 0; }
void sa_profile_enable(int on) { g_prof_on = on != 0; }
void sa_profile_reset(void) {
    for (auto* dc : g_dev) {  // a fresh epoch per device: launch times are kept relative to it
        if (dc->dev < 0 || dc->dev >= PROF_MAX_DEV) continue;
        check_set_device(dc->dev, "profile reset");
        if (!g_prof_epoch[dc->dev] && hipEventCreate(&g_prof_epoch[dc->dev]) != hipSuccess) { g_prof_epoch[dc->dev] = nullptr; continue; }
        hipEventRecord(g_prof_epoch[dc->dev], dc->admin);
        hipEventSynchronize(g_prof_epoch[dc->dev]);
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) { e.total_ms = 0; e.launches = 0; e.spans.clear(); }
}
// Time during which AT LEAST ONE launch of scope `name` was running (union of its launches' [start, end] over all slots; summed
// over devices).  With several calls in flight a launch's own duration says how long it shared the GPU, not how fast it is.
double sa_profile_busy_ms(const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        if (e.name != name) continue;
        std::vector<ProfSpan> v = e.spans;
        std::sort(v.begin(), v.end(), [](const ProfSpan& a, const ProfSpan& b) { return a.dev != b.dev ? a.dev < b.dev : a.t0 < b.t0; });
        double busy = 0;
        size_t i = 0;
        while (i < v.size()) {
            float lo = v[i].t0, hi = v[i].t1;
            size_t j = i + 1;
            while (j < v.size() && v[j].dev == v[i].dev && v[j].t0 <= hi) { hi = std::max(hi, v[j].t1); j++; }
            busy += hi - lo;
            i = j;
        }
        return busy;
    }
    return 0.0;
}
int sa_profile_num_entries(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}
int sa_profile_get(int i, char* name_buf, size_t name_cap, double* total_ms, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (i < 0 || i >= (int)g_prof.size()) return -1;
    if (name_buf && name_cap) {
        strncpy(name_buf, g_prof[i].name.c_str(), name_cap - 1);
        name_buf[name_cap - 1] = '\0';
    }
    if (total_ms) *total_ms = g_prof[i].total_ms;
    if (launches) *launches = g_prof[i].launches;
    return 0;
}

uint32_t sa_get_ref_len(void) { return g_ndev ? g_dev[0]->ref.len : 0; }
uint32_t sa_get_num_index(void) { return g_ndev ? g_dev[0]->num_index : 0; }
uint32_t sa_get_index_table_size(void) { return g_ndev ? g_dev[0]->nkeys : 0; }
uint32_t sa_get_query_len(uint32_t buffer) { return buffer < SA_BUFFER_DEPTH ? g_query_len[buffer] : 0; }

static DevCtx* ctx_of(int dev) {
    if (dev < 0 || dev >= g_ndev) {
        fprintf(stderr, "Error: device %d out of range\n", dev);
        exit(11);
    }
    check_set_device(g_dev[dev]->dev, "copy");
    return g_dev[dev];
}
void sa_copy_ref_codes(int dev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->ref.codes, dc->ref.len, hipMemcpyDeviceToHost), "ref codes");
}
void sa_copy_index_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->bucket_start + 1, (size_t)dc->nkeys * sizeof(uint32_t), hipMemcpyDeviceToHost), "index table");
}
void sa_copy_pos_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->pos_table, (size_t)dc->num_index * sizeof(uint32_t), hipMemcpyDeviceToHost), "pos table");
}
void sa_copy_query_codes(int dev, uint32_t buffer, int rev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    SeqBuf& b = rev ? dc->query_rc[buffer] : dc->query[buffer];
    check_memcpy(hipMemcpy(dst, b.codes, b.len, hipMemcpyDeviceToHost), "query codes");
}

}  // extern "C"
