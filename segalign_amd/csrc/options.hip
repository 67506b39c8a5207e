// options.hip -- the engine's global state, its tunables and the ONE option table behind sa_set_option / SEGALIGN_AMD_*.
#include "engine_internal.h"

namespace sa {

// ---- engine state and tunables ----
int g_ndev = 0;
std::vector<int> g_selected;  // sa_select_devices
std::vector<DevCtx*> g_dev;
std::mutex g_mu;  // token pool (seed_filter_interface.cu:7-9)
std::condition_variable g_cv;
std::vector<std::pair<int, int>> g_tokens;

bool g_proc_init = false;
int g_transition = 1;
uint32_t g_wga_chunk = 250000;
uint32_t g_seed_size = 19;
int g_sub_mat[64];
int g_xdrop = 910, g_hspthresh = 3000, g_noentropy = 0;
int g_log4_double = 0, g_entropy_ulps = 0;
int g_table_scratch_arena = 1;  // option table_scratch_arena
int g_ctx_skip_seed = 1;        // option ctx_skip_seed
int64_t g_max_seeds = 0;
int64_t g_max_hits = 0;
bool g_max_hits_overridden = false;
bool g_count_examined = false;
int g_fin_batch = 48;      // SEGALIGN_AMD_FIN_BATCH
int g_bufs_per_wave = 8;   // SEGALIGN_AMD_BUFS_PER_WAVE
int g_long_cap = 128;      // SEGALIGN_AMD_LONG_CAP: bases per side before a hit goes to the long kernel
int g_long_blocks = 1792;  // SEGALIGN_AMD_LONG_BLOCKS: grid of the long kernel (4 waves per block)
int g_packed_waves = 4096; // SEGALIGN_AMD_PACKED_WAVES: waves of the packed filter (2 workgroups of 8 waves per CU measured best: 3072 +16 %, 6144 +20 %, 8192 +14 %)
int g_ctx_waves = 0;       // SEGALIGN_AMD_CTX_WAVES: wave budget of the context filter; 0 = one 4096-hit chunk per wave (measured best)
uint32_t g_l2_cap_test = 0; // SEGALIGN_AMD_L2_CAP
int g_nbr_two_stage = 1;   // SEGALIGN_AMD_NBR_ONE_STAGE=1: every table entry cuts its own context out of the target
int g_table_atomic = 0;    // option table_atomic: build the seed table with the atomic counting sort even where the partition build applies
int64_t g_arena_gb = 40;   // option arena_gb: GiB of table arena the engine starts mapping at InitializeProcessor (0: on demand only)
int g_ctx_threads = 0;     // SEGALIGN_AMD_CTX_THREADS: workgroup size of the context filter (0 = kernel default)
int g_dedup_threads = 0;   // SEGALIGN_AMD_DEDUP_THREADS: workgroup size of the per-segment LDS chain (0 = 1024)
int g_spec_dedup = 1;      // SEGALIGN_AMD_SPEC_DEDUP=0: wait for the survivor count before the LDS chain (one more host sync)
int g_l2_blocks = 512;     // SEGALIGN_AMD_L2_BLOCKS: workgroups of the second-level packed filter
int g_max_waves = 4096;    // SEGALIGN_AMD_MAX_WAVES: waves of the filter kernel (4 per SIMD saturate instruction issue)
int g_fast_filter = 0;     // derived in InitializeProcessor: xdrop >= 0 && 7*max(M) <= xdrop
int g_packed_filter = 0;   // derived in InitializeProcessor: the packed upper-bound filter may be used
int g_chain_sort_threads = 256;  // SEGALIGN_AMD_CHAIN_SORT_THREADS
int g_chain_buckets = 0, g_chain_bucket_target = 32, g_chain_sort_blocks = 0, g_chain_group_max = 1024;
int64_t g_call_hits_max = 1ll << 30;  // option call_hits_max
int64_t g_call_hits = 128ll << 20;  // option call_hits: seed hits a call is sized for when the resident target's hits are sparse (0: chunks_per_call only)
int g_key_order = 0, g_key_order_chunks = 200;  // options key_order, key_order_chunks (join.h)
int64_t g_key_order_hits = 3ll << 30, g_key_order_min_pos = 0;  // options key_order_hits, key_order_min_pos
int g_chunks_per_call = SA_DEFAULT_CHUNKS;  // SEGALIGN_AMD_CHUNKS_PER_CALL: chunks sa_seed_interval hands to one multi-chunk call
int g_no_small_dedup = 0;  // SEGALIGN_AMD_NO_SMALL_DEDUP=1: always use the library sorts
int g_ctx = 1;             // neighbourhood table with target context when it fits (SEGALIGN_AMD_NO_CTX=1: positions only)
uint32_t g_audit_cap = 0;  // SEGALIGN_AMD_AUDIT_CAP (tests): record up to this many hits the filter levels reject per call
int g_td = 1;              // table-direct lookup (neighbourhood table + position probe, probe.hip); SEGALIGN_AMD_NO_TD=1 turns it off
int g_chain = 1;           // chain shortcut of the exact stage (SEGALIGN_AMD_NO_CHAIN=1 turns it off)
uint32_t CHAIN_CAP = 1u << 23;  // candidates per batch the chain buffers hold (SEGALIGN_AMD_CHAIN_CAP); larger batches fall back
SeedShape g_shape = {0, 0, 0, {0}};
uint32_t g_query_len[SA_BUFFER_DEPTH] = {0, 0};

thread_local sa_call_stats t_stats;
thread_local std::vector<uint2> t_audit;
thread_local uint32_t t_front_flags = 0;  // rejected hits of the calling thread's last hot call (audit option)
uint32_t SPEC_RECS = 16384;
uint32_t g_dedup_seg_max = 0;
int SLOTS_PER_DEVICE = 4;
size_t g_q2_limit = (size_t)1 << 32;
int g_q2_copies = 1;
int g_seed_upload = 0;  // option seed_upload: 0 pinned staging (memcpy + DMA), 1 pageable hipMemcpyAsync, 2 hipHostRegister + DMA

// Class scores of the class filter (extend.hip 1d): cls[x] bounds every matrix entry a base pair with (target code ^ query
// code) == x can have.  Codes >= 4 are stored as code 0 in the 2-bit copies, so a pair with such a code can show up in ANY
// class: its score joins every cls[x] -- but only for the codes that actually occur in the two resident blocks.
void class_scores(uint32_t present_t, uint32_t present_q, int cls[4]) {
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

int max_hits_for_mem(uint64_t total_global_mem) {  // src/seed_filter.cu:832-841, literally
    float global_mem_gb = static_cast<float>(total_global_mem / 1073741824.0f);
    return (int)(4194304 * global_mem_gb);
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
    {"call_hits", 128 << 20, 0, 1ll << 31, 0},
    {"call_hits_max", 1 << 30, 0, 1ll << 32, 0},      // ... and lowered when they are dense: chunks per call <= call_hits_max / estimated hits per chunk (0: no cap)        // seed hits a call is sized for when hits are sparse: chunks per call = max(chunks_per_call, call_hits / hits per chunk)
    {"key_order", 0, 0, 2, 0},                          // key-ordered calls (join.h): 0 never (default: the streamed pass is the faster one as a whole, profiles/r05/key_order_ab.txt), 1 when a call holds at least ~one position per seed key and the target's hits are dense enough, 2 whenever possible (tests)
    {"key_order_chunks", 200, 1, SA_MAX_CHUNKS, 0},    // chunks sa_get_chunks_per_call() hands to one call when key-ordered calls are on (a call's positions per key set the record reuse)
    {"key_order_hits", 3ll << 30, 1 << 20, 1ll << 34, 0},  // ... capped so that a call stays below about this many seed hits (its lists are sized by them)
    {"key_order_min_pos", 0, 0, 1ll << 31, 0},         // positions a call must hold to go key-ordered under key_order = 1 (0: the number of seed keys)
    {"filter_prio", 0, 0, 1, 0},                       // experiment: slot streams at the highest queue priority, the class filter alone on a lowest-priority stream per slot (wants GPU_MAX_HW_QUEUES >= 2 x slots + 1); measured and left off: profiles/r06/
    {"l2_right_state", 0, 0, 1, 0},                    // 1: a hit whose right side alone survived the class filter reaches the second level with the right walk's packed state and resumes behind the 54 context bases (one 64-base window instead of two from the anchor).  Parity-green and audited, and without effect on any clock (profiles/r06/ab_l2state_prio.txt: the second level is priced in random lines, not in windows): default 0 = round 5's form
    {"ctx_skip_seed", 1, 0, 1, 0},                     // context records hold the 58 bases in FRONT of the seed window, which the class filter bounds by seed_size x the largest class score (kernels.h CtxRec); 0: the bases left of the anchor, seed window included (A/B)
    {"clear_ref_frees", 0, 0, 1, 0},                   // 1: g_ClearRef hipFrees the index / position / extent tables like the reference (seed_filter_interface.cu:103-113); 0 (default): it forgets the tables and KEEPS their buffers for the next target block (a fresh allocation pays first-touch page clearing inside the next GenerateSeedPosTable); ShutdownProcessor frees them either way
    {"table_scratch_arena", 1, 0, 1, 0},               // scratch of the seed table build (keys, pair arrays: ~10 GB per 500 Mbp block) carved from the mapped table arena instead of fresh hipMallocs (first-touch page clearing inside every GenerateSeedPosTable)
    {"log4_double", 0, 0, 1, 0},                       // entropy divisor (src/seed_filter.cu:623, hazard H2): 0 = (double)logf(4.0f) as nvcc compiles `log(4.0f)`, 1 = log(4.0) (a host compiler without <cmath>'s float overload in scope)
    {"entropy_ulps", 0, -4, 4, 1},                     // tests (hazard H13): entropy factor moved by this many ulps before the truncating multiplies
    {"no_ctx", 0, 0, 1, 0},                            // 1: neighbourhood table without target context (lookup mode 1)
    {"no_td", 0, 0, 1, 0},                             // 1: no neighbourhood table at all (lookup mode 0, the reference-shaped path)
    {"no_chain", 0, 0, 1, 0},                          // 1: every candidate is extended on its own (no chain shortcut)
    {"no_packed_filter", 0, 0, 1, 0},                  // 1: byte-coded filter kernels only (also disables table-direct lookup)
    {"no_fast_filter", 0, 0, 1, 0},                    // 1: exact per-base filter only
    {"arena_gb", 40, 0, 1024, 0},                      // GiB of table arena mapped in the background from InitializeProcessor on
    {"work_gb", 3, 0, 64, 0},                          // GiB of work arena per slot, mapped in the background from InitializeProcessor on (0: the slots' buffers are plain allocations)
    {"arena_vmm", 1, 0, 1, 0},                         // 0: the table arena is one plain hipMalloc per growth (A/B against the mapped 1 GiB chunks)
    {"debug", 0, 0, 2, 0},                             // 1: table-build timings on stderr; 2: + sync and name every kernel scope
    // launch geometry (swept by tools/sweep_*.sh; the defaults are the measured optima)
    {"fin_batch", 48, 1, 64, 0}, {"bufs_per_wave", 8, 1, 1 << 20, 0}, {"long_cap", 128, 0, 2 * PACK_PAD, 0},
    {"long_blocks", 1792, 1, 1 << 20, 0}, {"max_waves", 4096, 4, 1 << 20, 0}, {"packed_waves", 4096, 8, 1 << 20, 0},
    {"l2_blocks", 512, 1, 1 << 20, 0}, {"ctx_waves", 0, 0, 1 << 20, 0}, {"ctx_threads", 0, 0, 1024, 0},
    {"chain_sort_threads", 256, 64, 512, 0}, {"dedup_threads", 0, 0, 1024, 0},
    {"chain_buckets", 0, 0, 262144, 0},                // chain hash buckets: 0 = picked on the device, ~chain_bucket_target candidates each; else forced (rounded to a power of two in 64 .. 262144; small counts crowd the buckets: tests)
    {"chain_bucket_target", 32, 1, 4096, 0}, {"chain_sort_blocks", 0, 0, 1 << 16, 0}, {"chain_group_max", 1024, 64, 4096, 0},
    {"cls_one_copy", 0, 0, 2, 0},                      // class filter's query windows: 0 / 1 unshifted copy + funnel shifts (default), 2 the sixteen shifted copies (round 3's form)
    {"nbr_one_stage", 0, 0, 1, 0}, {"table_atomic", 0, 0, 1, 0}, {"seed_upload", 0, 0, 2, 0},
    // test-only: small capacities that force the overflow / fallback branches
    {"l2_cap", 0, 0, 1 << 30, 1}, {"spec_dedup", 1, 0, 1, 1}, {"spec_recs", 16384, 1, 16384, 1}, {"dedup_seg_max", 0, 0, 1 << 30, 1},
    {"q2_limit_mb", 4096, 1, 4096, 1},                 // (tests) bytes the 16 two-bit copies of a query strand may span before calls leave the table-direct path
    {"no_small_dedup", 0, 0, 1, 1}, {"chain_no_link", 0, 0, 1, 1}, {"chain_cap", 1 << 23, 1, 1 << 30, 1}, {"audit_cap", 0, 0, 1 << 28, 1},
};
static Option* find_option(const char* name) {
    for (auto& o : g_opts)
        if (strcmp(o.name, name) == 0) return &o;
    return nullptr;
}
int64_t opt_value(const char* name) {
    Option* o = find_option(name);
    return o ? o->value : 0;
}
void resolve_options() {
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
    g_key_order = (int)opt_value("key_order");
    g_key_order_chunks = (int)opt_value("key_order_chunks");
    g_key_order_hits = opt_value("key_order_hits");
    g_key_order_min_pos = opt_value("key_order_min_pos");
    g_call_hits = opt_value("call_hits");
    g_call_hits_max = opt_value("call_hits_max");
    g_table_scratch_arena = (int)opt_value("table_scratch_arena");
    g_ctx_skip_seed = (int)opt_value("ctx_skip_seed");
    g_log4_double = (int)opt_value("log4_double");
    g_entropy_ulps = (int)opt_value("entropy_ulps");
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
    g_chain_buckets = 0;
    if (opt_value("chain_buckets")) {
        g_chain_buckets = 64;
        while (g_chain_buckets < 262144 && g_chain_buckets < (int)opt_value("chain_buckets")) g_chain_buckets <<= 1;
    }
    g_chain_bucket_target = (int)opt_value("chain_bucket_target");
    g_chain_sort_blocks = (int)opt_value("chain_sort_blocks");
    g_chain_group_max = (int)opt_value("chain_group_max") & ~63;
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
    g_q2_limit = (size_t)opt_value("q2_limit_mb") << 20;
    g_q2_copies = opt_value("cls_one_copy") == 2 ? Q2_COPIES : 1;
}

void require_init(const char* who) {
    if (g_ndev <= 0) {
        fprintf(stderr, "Error: %s called before InitializeInterface\n", who);
        exit(1);
    }
}
void require_proc(const char* who, uint32_t buffer) {  // hot entry points: processor initialised, buffer id in range
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

}  // namespace sa

using namespace sa;

extern "C" {

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

}  // extern "C"
