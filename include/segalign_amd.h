/*
 * segalign_amd.h -- C-ABI of libsegalign_hip.so, the MI355X (gfx950) seed -> filter -> extend engine.
 *
 * This is the drop-in boundary for SegAlign's engine.  The reference boundary is C++-ABI (function pointers
 * that pass std::vector by value); every entry point below is the plain-C form of one reference
 * symbol -- plain pointers and sizes, no C++/torch types -- and include/segalign_amd_compat.hpp rebuilds the
 * exact reference symbols (g_InitializeInterface ... g_SeedAndFilter, GenerateSeedPosTable) on top of it so the
 * reference host (src/main.cpp, src/seeder.cpp) links unchanged.  See INTEGRATION.md.
 *
 * Error behaviour mirrors common/cuda_utils.h:4-37 and seed_filter_interface.cu:53-70: a message on stderr and
 * exit(code): 1 no device, 10 too many GPUs requested, 11 set-device, 12 malloc, 13 memcpy, 14 free,
 * 15 kernel launch/synchronise (the reference never checks launches; this engine does).
 *
 * Threading mirrors the reference: sa_seed_and_filter* may be called from many host threads concurrently; the
 * engine hands each call a (device, slot) token from a pool (reference: seed_filter_interface.cu:7-9 +
 * src/seed_filter.cu:699-706,798-803).  All other entry points are called from one thread at a time
 * (the reader lambda of src/main.cpp:601-737).
 */
#ifndef SEGALIGN_AMD_H
#define SEGALIGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_BUFFER_DEPTH 2 /* src/graph.h:14 */

/* src/graph.h:25-30 -- hit / HSP record; len = bases - 1 */
typedef struct sa_segment_pair {
    uint32_t ref_start;
    uint32_t query_start;
    uint32_t len;
    int32_t score;
} sa_segment_pair;

/* ---- engine lifecycle ------------------------------------------------------------------------------------ */

/* g_InitializeInterface, common/seed_filter_interface.h:3,8 ; def common/seed_filter_interface.cu:49-80.
 * num_gpu = -1 -> all visible devices.  Returns the number of devices the engine will use. */
int sa_initialize_interface(int num_gpu);

/* One-process-per-GPU deployments: restrict the NEXT sa_initialize_interface to these HIP device ordinals
 * (engine device g = ids[g]).  n = 0 restores the reference behaviour (devices 0..num_gpu-1). */
void sa_select_devices(const int* ids, int n);
/* (An ordinal may appear more than once in ids[]: every entry becomes an engine device of its own -- context, streams, target,
 *  tables, arena, token-pool slots -- on that GPU.  tests/test_gpu_multi_device.py exercises the multi-device paths that way on a
 *  one-GPU box.) */

/* g_InitializeProcessor, src/seed_filter.h:4,10 ; def src/seed_filter.cu:830-897.
 * sub_mat: 64 ints, sub_mat[r*8+q] over codes A0 C1 G2 T3 L4 N5 X6 E7 (common/parameters.h:4-13). */
void sa_initialize_processor(int transition, uint32_t wga_chunk, uint32_t seed_size, const int* sub_mat, int xdrop,
                             int hspthresh, int noentropy);

/* g_ShutdownProcessor, src/seed_filter.h:8,14 ; def src/seed_filter.cu:932-940. */
void sa_shutdown_processor(void);
/* (additive) The reference's shutdown resets the device (src/seed_filter.cu:939); the engine keeps ONE thing across
 * ShutdownProcessor: the table arena, a cache of cleared device pages that cost seconds to obtain.  This gives it back
 * to the device(s); call after sa_shutdown_processor.  Option arena_gb = 0 makes ShutdownProcessor do it itself. */
void sa_release_arena(void);

/* ---- target block ---------------------------------------------------------------------------------------- */

/* g_SendRefWriteRequest, common/seed_filter_interface.h:4 ; def seed_filter_interface.cu:82-101.
 * ASCII at seq+addr (borrowed for the call) is uploaded to every device and encoded there. */
void sa_send_ref_write_request(const char* seq, size_t addr, uint32_t len);

/* g_ClearRef, common/seed_filter_interface.h:5 ; def seed_filter_interface.cu:103-113 (target + both tables). */
void sa_clear_ref(void);

/* GenerateShapePos, common/ntcoding.h:5 ; def common/ntcoding.cpp:21-37.  Shape string of '1'/'T' (care,
 * 'T' = transition allowed) and anything else (don't care).  Returns the seed weight (kmer_size).
 * State used by sa_generate_seed_pos_table and sa_seed_and_filter_range. */
int sa_generate_shape_pos(const char* shape);

/* GenerateSeedPosTable, common/ntcoding.h:9 ; def common/seed_pos_table.cu:49-109.
 * Built ON THE DEVICE from the encoded target; when (ref_str+start_addr, ref_length) is the block last sent by
 * sa_send_ref_write_request (the only way src/main.cpp:615-621 calls it) nothing is uploaded again. */
void sa_generate_seed_pos_table(const char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step,
                                int shape_size, int kmer_size);

/* ---- query block ----------------------------------------------------------------------------------------- */

/* g_SendQueryWriteRequest, src/seed_filter.h:5 ; def src/seed_filter.cu:899-919.  The reference reads the host
 * global query_DRAM->buffer (src/store.h:7, seed_filter.cu:910); the C form takes that base pointer explicitly. */
void sa_send_query_write_request(const char* query_buffer, size_t addr, uint32_t len, uint32_t buffer);

/* g_ClearQuery, src/seed_filter.h:7 ; def src/seed_filter.cu:921-930. */
void sa_clear_query(uint32_t buffer);

/* ---- the hot call ---------------------------------------------------------------------------------------- */

/* g_SeedAndFilter, src/seed_filter.h:6 ; def src/seed_filter.cu:682-828.
 * seeds[i] = (key << 32) + query_position (src/seeder.cpp:60-61).  A vector that is what src/seeder.cpp:57-74 emits for the
 * positions it spans (checked on the device) is looked up table-direct like sa_seed_and_filter_range; any other vector takes the
 * reference-shaped path (sa_call_stats.lookup_path tells which).  Returns the element count of *out
 * (>= 1): out[0] is the header {len = total anchors, score = (int)num_hits} (seed_filter.cu:806-809), followed by
 * the HSPs of each iteration in order.  *out is owned by the caller; release with sa_free_segments. */
size_t sa_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t buffer, sa_segment_pair** out);

/* ADDITIVE: same as sa_seed_and_filter, but the seed words of query positions [start, end) of
 * the resident block `buffer` (strand `rev`) are generated on the device exactly as src/seeder.cpp:57-74 /
 * :94-109 would on the host (same order), so no seed vector crosses PCIe.  If that chunk has no valid seed the
 * reference does not call the engine at all (seeder.cpp:76); this returns 0 and *out = NULL in that case. */
size_t sa_seed_and_filter_range(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** out);

void sa_free_segments(sa_segment_pair* p);

/* Additive: seeder_body::operator() of src/seeder.cpp:12-127 for one query interval [start, end) of the block in
 * `buffer` (q_len = block length - seed size, src/main.cpp:708): the plus-strand chunks, then the minus-strand chunks in
 * reverse-complement coordinates (:33-34), each through sa_seed_and_filter_range with `threads` calls in flight.
 * *out_fw / *out_rc receive the HSPs per strand in chunk order without the header elements (free with
 * sa_free_segments); totals (optional) sums the per-call statistics.  Returns the number of HSPs. */
/* Additive: up to sa_max_chunks_per_call() consecutive wga_chunk-sized chunks [start, start+chunk), ... of one strand in one
 * pass over the kernels.  outs[c] / counts[c] receive exactly what sa_seed_and_filter_range returns for chunk c (own
 * iteration plan, own dedup scope, own header; NULL / 0 for a chunk without seeds).  Only the slots of the chunks the range
 * covers (ceil((end - start) / wga_chunk)) are written.  Returns the sum of the counts. */
int sa_max_chunks_per_call(void);
int sa_get_chunks_per_call(void);   /* chunks sa_seed_interval groups into one call (option chunks_per_call) */
size_t sa_seed_and_filter_chunks(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** outs, size_t* counts);

struct sa_call_stats; /* defined below */
size_t sa_seed_interval(uint32_t start, uint32_t end, uint32_t q_len, int strands, uint32_t buffer, int threads,
                        sa_segment_pair** out_fw, size_t* n_fw, sa_segment_pair** out_rc, size_t* n_rc,
                        struct sa_call_stats* totals);

/* Additive: a LIST of independent calls, each up to sa_max_chunks_per_call() consecutive chunks [start, end) of strand `rev` (in
 * that strand's coordinates), run with `threads` of them in flight on the engine's persistent worker pool (the reference keeps one
 * seeder body per TBB worker in flight, src/main.cpp:565-573).  results[i].hsps: the HSPs of call i -- its chunks concatenated in
 * order, headers removed -- owned by the caller (sa_free_segments).  This is the unit a multi-GPU host deals out: every call of a
 * pass is independent, so any partition of the list over devices or processes gives identical results. */
typedef struct sa_call_desc {
    uint32_t start, end;
    int rev;
} sa_call_desc;
typedef struct sa_call_result {
    sa_segment_pair* hsps;
    size_t num_hsps;
    uint64_t num_hits;
    int32_t device;   /* engine device (0 .. devices in use - 1) the call ran on */
    int32_t reserved;
} sa_call_result;
size_t sa_seed_calls(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, sa_call_result* results,
                     struct sa_call_stats* totals /* nullable: sums over the calls */);
/* Seed hits of every call of such a list, WITHOUT filtering or extending them: lookup only (the table-direct position probe and its
 * chunk plans, ~0.4 ms per forty-chunk call against ~5 ms for the call itself).  A multi-GPU host weighs the calls of a pass with
 * these counts before it deals them out (longest first); every rank computes the same numbers from the same resident blocks. */
void sa_count_call_hits(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, uint64_t* hits);
/* ... and per wga_chunk piece: chunk_hits[i * sa_max_chunks_per_call() + c] = hits of chunk c of call i (same lookups, no more launches) */
void sa_count_chunk_hits(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, uint64_t* hits, uint64_t* chunk_hits);
uint32_t sa_get_wga_chunk(void); /* the chunk size InitializeProcessor was given */

/* ---- repeat-masker variant (repeat_masker_src/seed_filter.h:4-8) ----------------------------------------- */

/* SendQueryWriteRequest() of the repeat masker: the query IS the target; builds its reverse complement on the
 * device from the encoded target (repeat_masker_src/seed_filter.cu:951-961). */
void sa_rm_send_query_write_request(void);
void sa_rm_clear_query(void); /* repeat_masker_src/seed_filter.cu:964-972 */
/* SeedAndFilter(seeds, rev, ref_start, ref_end), repeat_masker_src/seed_filter.cu:724-876; header packs the 64-bit
 * hit / anchor counts as {ref_start,query_start} / {len,score} (:857-861). */
size_t sa_rm_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t ref_start, uint32_t ref_end,
                             sa_segment_pair** out);

/* ---- repeat-masker post-processing on the device (additive) ---------------------------------- */

/* struct Segment, repeat_masker_src/graph.h:32-35: a run of masked positions, block-relative. */
typedef struct sa_interval {
    uint32_t query_start;
    uint32_t len; /* number of positions in the run; the printer writes start .. start+len+1 (segment_printer.cpp:56) */
} sa_interval;

#define SA_STRAND_PLUS 1
#define SA_STRAND_MINUS 2
#define SA_STRAND_BOTH 3

/* Device-side form of seeder_body::operator() of the repeat masker (repeat_masker_src/seeder.cpp:28-195) for ONE
 * interval of the resident block: for every wga_chunk of [start_pos, end_pos) and every selected strand, seed words
 * are extracted on the device from the encoded block (seeder.cpp:84-101,123-138), SeedAndFilter runs with the target
 * window [ref_start, ref_end] (:105,142), every returned HSP adds 1 to the coverage of positions
 * query_start .. query_start+len-1 (uint8_t counters, :155-159) and the runs with coverage >= M become intervals
 * (:168-186).  Neither seeds nor HSPs cross the PCIe link; only the intervals are returned.
 * Returns the number of intervals; *out is malloc-ed (sa_free_intervals).  Optional totals: seeds, hits, HSPs
 * (the reference's num_seeds / num_seed_hits / num_hsps statistics, seeder.cpp:104-110). */
size_t sa_rm_mask_interval(uint32_t start_pos, uint32_t end_pos, uint32_t ref_start, uint32_t ref_end, int strands, uint32_t M,
                           sa_interval** out, uint64_t* totals /* [3] or NULL */);
/* The counting + run extraction alone (seeder.cpp:153-188) for a host that keeps the reference's chunk loop and
 * hands over the HSPs it collected for one interval (headers already removed). */
size_t sa_rm_coverage_intervals(const sa_segment_pair* hsps, size_t num_hsps, uint32_t block_len, uint32_t M, sa_interval** out);
void sa_free_intervals(sa_interval* p);

/* ---- knobs the reference derives from its GPU (hazard H4) ------------------------------------------------- */

/* MAX_HITS (src/seed_filter.cu:832-841) decides how SeedAndFilter splits a call into iterations, and dedup scope
 * is per iteration.  Default: the reference formula applied to THIS device's memory.  Override to reproduce the
 * output of the reference on a given CUDA GPU (e.g. sa_max_hits_for_mem(16945512448) for a 16 GB V100). */
void sa_set_max_hits(int64_t max_hits);
int64_t sa_get_max_hits(void);
int sa_max_hits_for_mem(uint64_t total_global_mem);

/* ---- options: the engine's one switchboard -------------------------------------------------------------------
 * Every tunable and every switch of the engine lives in one table.  A value is resolved at each sa_initialize_processor:
 * sa_set_option(name, v)  >  environment variable SEGALIGN_AMD_<NAME IN UPPER CASE>  >  default; values are clamped to the
 * option's range.  sa_set_option returns -1 for an unknown name.  Results are bit-identical under every setting; the options
 * choose between implementations of the same reference semantics (src/seed_filter.cu:682-828) or size their launches.
 *
 * Deployment options
 *   slots             calls in flight per device (default 4, at most 8; the reference allows 1: its token IS the device).  Every slot
 *                     has its own stream, and slots that share one of the runtime's default four hardware queues run one after the
 *                     other: the HOST exports GPU_MAX_HW_QUEUES=8 before its first HIP call (the library leaves the environment alone
 *                     and says so once on stderr when the slots + the upload stream outnumber the queues)
 *   chunks_per_call   wga_chunk-sized chunks sa_seed_interval / sa_rm_mask_interval hand to one pass (default 40: a strand's chunks of a 10 Mbp interval; maximum 256; the repeat masker at most 20).
 *                     sa_get_chunks_per_call() adapts it to the resident target: more when seed hits are sparse (option call_hits,
 *                     default 128 M hits per call), fewer when they are dense (option call_hits_max, default 1 G)
 *   no_ctx            1: neighbourhood table without target context (lookup mode 1)
 *   no_td             1: no neighbourhood table (lookup mode 0: seed words -> buckets -> hit list, the reference's shape)
 *   arena_gb          GiB of table arena the engine starts mapping in the background at sa_initialize_processor (default 40:
 *                     the table of a ~100 Mbp block; 0: only on demand).  A larger block raises the goal by itself; setting it
 *                     beforehand (e.g. 180 for 500 Mbp blocks) takes the allocation off the table build's critical path
 *   clear_ref_frees   1: g_ClearRef frees the index / position / extent tables like the reference's clearRef (seed_filter_interface.cu:103-113);
 *                     0 (default): it forgets the tables and keeps their buffers for the next target block (ShutdownProcessor frees them)
 *   key_order, key_order_chunks, key_order_hits, key_order_min_pos   key-ordered calls (off by default)
 *   ctx_skip_seed, table_scratch_arena, log4_double                  INTEGRATION.md 4
 *   l2_right_state    1: the second filter level resumes an open RIGHT walk behind the class filter's context from its packed state
 *                     (default 0: measured without effect, profiles/r06/ab_l2state_prio.txt)
 *   filter_prio       1 (experiment): slot streams at the highest queue priority, the class filter on a lowest-priority stream of its own
 *                     (small kernels 2-3 x faster, the pass 2 % slower: default 0; wants GPU_MAX_HW_QUEUES >= 2 x slots + 1)
 *   no_chain          1: every candidate is extended on its own (no chain shortcut, DESIGN.md 4.5')
 *   no_packed_filter / no_fast_filter   1: fall back to the byte-coded / the exact per-base X-drop filter kernels
 *   debug             1: table-build timings on stderr; 2: + synchronise after every kernel scope and name it (fault localisation)
 *   seed_upload       how sa_seed_and_filter brings the host seed vector over: 0 copy into a pinned staging buffer + DMA (default),
 *                     1 hipMemcpyAsync from the pageable vector (the runtime stages it), 2 hipHostRegister the vector + DMA
 * Launch geometry (defaults are the measured optima, tools/sweep_*.sh)
 *   fin_batch, bufs_per_wave, long_cap, long_blocks, max_waves, packed_waves, l2_blocks, ctx_waves, ctx_threads,
 *   chain_sort_threads, chain_sort_blocks, chain_group_max (candidates a chain workgroup sorts in LDS at a time), chain_bucket_target
 *   (candidates per chain hash bucket the device sizes the bucket count for; chain_buckets forces a count), cls_one_copy,
 *   dedup_threads, nbr_one_stage, table_atomic (1: seed table by the atomic counting sort even for seed weights 9..14, where the
 *   LDS-staged partition build is the default), work_gb (GiB of work arena per slot), arena_vmm, call_hits, call_hits_max
 * Test-only options (small capacities that force the overflow / fallback branches of the orchestration)
 *   l2_cap, spec_dedup, spec_recs, dedup_seg_max, no_small_dedup, chain_cap, chain_no_link, q2_limit_mb, audit_cap
 */
int sa_set_option(const char* name, int64_t value);
int sa_reset_option(const char* name);     /* NULL: every option back to environment / default */
int64_t sa_get_option(const char* name);   /* value resolved at the last sa_initialize_processor (INT64_MIN: unknown name) */
int sa_option_count(void);
const char* sa_option_name(int i, int* test_only /* nullable */);

/* ---- introspection (tests, bench, profiling; not part of the reference surface) --------------------------- */

/* With option audit_cap = N > 0: every hit the X-drop FILTER levels reject in a table-direct call is recorded (up to N per call).
 * Returns how many the calling thread's last call recorded and copies min(that, cap_pairs) {ref_loc, query_loc} pairs.  The
 * parity tests extend each of them with the CPU restatement of find_hsps and require that none passes: the filters' bounds are
 * upper bounds. */
size_t sa_get_audit(uint32_t* dst_pairs, size_t cap_pairs);

typedef struct sa_call_stats {
    uint64_t num_seeds;
    uint64_t num_hits;      /* H */
    uint64_t num_survivors; /* records handed to the dedup stage: the reference's survivors (done = 1) minus the exact
                               duplicates the chain shortcut never extends; equal to the reference count while
                               sa_set_count_examined(1) is on (the shortcut is off then) */
    uint64_t num_anchors;   /* returned HSPs */
    uint64_t num_examined;  /* E: scored positions; only filled while sa_set_count_examined(1) */
    uint64_t num_examined_filter; /* positions scored by the filter kernel alone (same condition) */
    uint64_t num_candidates; /* hits the X-drop filter forwarded to the exact kernel */
    uint64_t num_entropy;   /* hits that needed the entropy factor */
    uint32_t num_iter;
    int device;
    int lookup_path;        /* seed lookup path the call took: 0 general (seed words -> buckets -> hit list), 1 table-direct,
                               2 table-direct with target context (sa_get_lookup_mode) */
    uint32_t path_flags;    /* SA_PATH_*: which of the engine's rare branches the call took (diagnostics; the results never depend on them).
                               Summed statistics (sa_seed_calls, sa_seed_interval) carry the OR over their calls */
    uint64_t num_forwarded; /* context-table calls: hits the class filter (level 1) handed to the second level; 0 otherwise */
} sa_call_stats;
#define SA_PATH_LIST_REGROWN 1u          /* a device list (second-level / candidate / entropy / survivor) overflowed: regrown, batch rerun */
#define SA_PATH_DEDUP_FALLBACK 2u        /* a segment held more survivors than the LDS chain takes: library sorts + unique */
#define SA_PATH_CHAIN_BUCKET_OVERFLOW 4u /* a chain bucket above its LDS capacity was left unsorted (costs extensions, never results) */
#define SA_PATH_CHAIN_SLICED 8u          /* more candidates than the chain buffers hold: the chain stages ran over the list slice by slice */
#define SA_PATH_HEAD_BITS_REGROWN 16u    /* the head-bit map of the call's hits was regrown and the compaction repeated */
#define SA_PATH_KEY_ORDERED 64u          /* the call ran key-ordered: positions sorted by seed key, hits enumerated per key (join.h) */
#define SA_PATH_GENERAL_FALLBACK 32u     /* a device-seeded call could not take the table-direct path (a chunk of 6 x MAX_HITS hits or more, > 2^32 hits, ...) */
void sa_get_last_call_stats(sa_call_stats* out); /* stats of the calling thread's most recent hot call */
void sa_set_count_examined(int on);
/* X-drop filter kernel selected by InitializeProcessor for plain calls: 0 = exact per-base walk, 1 = fast per-base
 * (7*max(M) <= xdrop), 3 = packed 2-bit/4-bit upper-bound filter. */
int sa_get_filter_mode(void);

/* Seed lookup path of the device-seeded entry points (sa_seed_and_filter_range / _chunks, sa_seed_interval,
 * sa_rm_mask_interval) on device 0: 0 = general path (seed words -> find_num_hits / find_hits shape, also used by the drop-in
 * sa_seed_and_filter), 1 = table-direct (neighbourhood table + position probe, no seed words, no hit list), 2 = table-direct
 * with target context in the table (the X-drop filter streams 32-byte records).  Chosen by available HBM;
 * options no_ctx / no_td force 1 / 0.  Results are identical on every path. */
int sa_get_lookup_mode(void);
uint64_t sa_get_neighbourhood_entries(void); /* run entries of the neighbourhood table (0 when not built) */

/* Per-kernel HIP-event timing on the engine's own streams (bench.py's roofline leg). */
void sa_profile_enable(int on);
void sa_profile_reset(void);
int sa_profile_num_entries(void);
/* returns 0 on success; name is NUL-terminated into name_buf */
int sa_profile_get(int i, char* name_buf, size_t name_cap, double* total_ms, uint64_t* launches);
/* ms during which at least one launch of scope `name` was running (union over the slots' streams, summed over devices) since
 * the last sa_profile_reset: with several calls in flight the launches of one kernel overlap, and bytes / this time is the rate
 * the kernel sustained while it ran. */
double sa_profile_busy_ms(const char* name);

/* Device -> host copies of engine state on device `dev` (parity tests). */
uint32_t sa_get_ref_len(void);
uint32_t sa_get_num_index(void);       /* entries in the position table */
uint32_t sa_get_index_table_size(void); /* 4^kmer_size */
void sa_copy_ref_codes(int dev, uint8_t* dst);
void sa_copy_index_table(int dev, uint32_t* dst); /* INCLUSIVE bucket ends == reference d_index_table */
void sa_copy_pos_table(int dev, uint32_t* dst);
void sa_copy_query_codes(int dev, uint32_t buffer, int rev, uint8_t* dst);
uint32_t sa_get_query_len(uint32_t buffer);
/* seed words the device seeder produces for [start,end) (8f-1), for comparison with src/seeder.cpp's vector */
size_t sa_device_make_seeds(uint32_t start, uint32_t end, int rev, uint32_t buffer, uint64_t* dst, size_t cap);

/* The extension stage alone (find_hsps + compaction of the passing hits, src/seed_filter.cu:232-680) for caller-supplied
 * anchors: ref_query_pairs[2i] = ref_loc, [2i+1] = query_loc on the resident target / strand `rev` of query `buffer`.
 * out[0] = header {len = count}; the passing records follow unordered and not de-duplicated (exact duplicates of one
 * run of anchors may already be merged).  Used to check the extension kernels against golden vectors. */
size_t sa_extend_hits(const uint32_t* ref_query_pairs, size_t num_hits, int rev, uint32_t buffer, sa_segment_pair** out);

/* Test entry: the ordering stage alone on `n` records as ONE dedup scope -- stable_sort(hspComp) -> unique_copy(hspEqual) ->
 * stable_sort(hspCompLastz), src/seed_filter.cu:47-108,776-782; rm != 0: the repeat masker's five-step chain,
 * repeat_masker_src/seed_filter.cu:45-135,819-831.  path 0 = the engine's per-segment LDS chain (rm == 0, n <= 2048), path 1 = its
 * library-sort chain.  *out is malloc'ed (sa_free_segments); returns the number of records kept.  Held against rocThrust's own
 * stable_sort / unique_copy in tests/test_gpu_thrust_order.py (hazard H3: unique_copy = head flags on adjacent INPUT pairs). */
size_t sa_order_hsps(const sa_segment_pair* in, size_t n, int rm, int path, sa_segment_pair** out);

const char* sa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SEGALIGN_AMD_H */
