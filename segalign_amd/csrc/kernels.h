// kernels.h -- launcher declarations of the gfx950 kernels of the seed -> filter -> extend engine.
// Every launcher enqueues on the given HIP stream and returns; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sa {

// Sequences live in HBM as one code byte per base (A0 C1 G2 T3 L4 N5 X6 E7, common/parameters.h:4-13) inside
// an allocation padded by SEQ_PAD bytes on both sides, so the 8-byte window loads of the extension kernel may
// over-read without faulting.  Guard bytes are 0x40: bit 6 of a matrix index selects a terminator table entry, so a
// walk that leaves the block stops exactly at the edge (src/seed_filter.cu:332,:482) with no bounds arithmetic.
constexpr int SEQ_PAD = 64;

constexpr int MAX_CARE = 16;  // seed weight limit; reference asserts 3 < kmer_size <= 15 (seed_pos_table.cu:51-52)
constexpr int MAX_SEGS = 512;  // reference iterations handled by one extension batch (256 chunks x 2 iterations of a table-direct call)
constexpr int MAX_SEGS_ABS = 8;  // ... of a batch whose chain sort key carries absolute query positions (general path)

struct SeedShape {            // device copy of the state GenerateShapePos keeps (ntcoding.cpp:6-8)
    int weight;               // # care positions (kmer_size)
    int span;                 // seed_size (19 for 12of19)
    uint32_t transition_mask; // bit t set <=> IsTransitionAtPos(t) (flag of the t-th care position in shape order,
                              // ntcoding.cpp:21-37); the seeder applies it as key ^ (2 << 2t) (seeder.cpp:65-68, H10)
    uint8_t pos[MAX_CARE];    // care offsets inside the span, in shape order (first = most significant 2 bits)
};

struct Hit {                  // 8 B form of the reference's 16 B hit record (len/score are always 0 there)
    uint32_t ref_loc;         // bucket position + seed_size (seed_filter.cu:220)
    uint32_t query_loc;       // query position + seed_size (seed_filter.cu:204)
};

struct HspRec {               // survivor: reference segmentPair + the reference iteration it belongs to
    uint32_t ref_start;
    uint32_t query_start;
    uint32_t len;
    int32_t score;
    uint32_t seg;
};

struct CandRec {              // a hit the X-drop filter could not reject: extended exactly by the exact kernel
    uint32_t ref_loc, query_loc;
    uint32_t hidx;            // index of the hit inside the launch (for the segment id)
};

// What the class filter (level 1) hands to the second level: the anchor plus what level 1 already knows, so that level 2 walks only
// the side(s) that were still alive at the end of the context bases (54 right, seed window + 58 left; rounds 2-4: 48 + 64).  20 bytes:
//   state: the packed (score : drop) register of level 1's LEFT walk at the end of its context (seed window + CTX_L_BASES; extend.hip 1d)
//   meta : known (16 bits) | flags (2 bits) << 16
//     flags bit 0: right side undecided, bit 1: left side undecided
//     flags 1: known = bestL (the left side is settled); the right side is walked from the anchor -- or, with ExtendArgs::l2_right_state,
//              resumes behind level 1's CTX_R_BASES context from `state`, which then holds the RIGHT walk's register (round 6)
//     flags 2: known = bestR, the left walk continues behind level 1's context from `state`
//     flags 3: both sides from the anchor (also: both settled but the bound passes -- the exact pair scores of level 2 get a say)
struct L2Rec {
    uint32_t ref_loc, query_loc;
    uint32_t hidx;
    uint32_t state;
    uint32_t meta;
};

constexpr int L2_NSUB = 256;        // sub-lists of the second-level list
constexpr int L2_CNT_STRIDE = 32;   // dwords between their counters (one 128-byte line each)

struct EntRec {               // finished hit with hspthresh <= total <= 3*hspthresh: needs the entropy factor (:608)
    uint32_t ref_loc, query_loc;
    int32_t bposR, boffL;     // best offsets right / left of the anchor
    int32_t total;
    uint32_t seg;
};

constexpr uint32_t TD_CHUNK_HITS = 4096;  // hits one wave of the context filter handles (64 buffers of 64)
constexpr uint32_t TD_CHUNK_CAP = 1u << 20;  // chunk starts a call can record: 2^20 * 4096 = 2^32 hits, the table-direct limit

struct TdRec {                // table-direct lookup (probe.hip): a non-empty query position of the call
    uint32_t prefix;          // call-wide index of the position's first hit (a table-direct call holds < 2^32 hits)
    uint32_t qpos;            // query position (seed start)
    uint64_t off;             // offset of the position's run in the neighbourhood table
};

// Neighbourhood table WITH target context (probe.hip; class filter, extend.hip 1d): every run entry carries, next to its seed
// position, 28 bytes of 2-bit target bases -- the X-drop filter never touches the target.  The left context is stored in WALKING
// order with whole 2-bit fields reversed and COMPLEMENTED -- the reverse of a strand is the complement of its reverse-complement
// strand, so l ^ (the other query strand's forward window) is the per-base XOR of target and query in walking order without
// any query-side reversal.  32 bytes, 16-byte aligned: two aligned 16-byte loads per hit and a shift for the address (a 28-byte
// record with the position in a side array was measured: its 64-bit multiply-by-28 and the extra gather for the ~4 % of
// forwarded hits cost more VALU issue and wait time than the 4 bytes save in a kernel that is not bound by bytes).
// The left context does NOT hold the seed window itself (round 5): the left walk starts at anchor - 1 = the last base of the seed
// window and crosses its seed_size bases first.  Those carry next to no information -- the care positions match by construction --
// so the class filter bounds them by seed_size x (largest class score), a valid upper bound that costs no lookup, and the record's
// 64 left bases are the ones IN FRONT of the seed: pos - 1, pos - 2, ...  With 45 flank bases (64 minus the 19 of 12of19) 2.5 % of
// random hits were still alive at the end of the left context; with 64: 0.23 % (half as many hits forwarded to the second level).
// The 112 context bases are cut 54 right + 58 left (round 5; rounds 2-4: 48 + 64): with the seed window out of the left context both
// sides walk pure flank, the share of random hits still alive after n flank bases falls steeply and convexly in n (48: 1.8 %, 54:
// 0.84 %, 58: 0.48 %, 64: 0.22 %), so an even split forwards least -- 54 + 58 keeps whole six-base fields: nine on the right, nine
// and the four-base tail on the left, one 224-bit string of eighteen 12-bit fields and an 8-bit tail.
constexpr int CTX_R_BASES = 54, CTX_L_BASES = 58;
struct CtxRec {
    uint32_t pos;             // seed START position in the target (+ seed_size = anchor)
    uint32_t w[7];            // bits 0..107: the 54 bases from the anchor on (anchor+k in bits 2k, 2k+1); bits 108..223: ~(the 58 bases in
                              // front of the seed start, base pos-1-k in bits 108+2k, 108+2k+1)
};
static_assert(sizeof(CtxRec) == 32, "CtxRec is a 32-byte record");

constexpr int Q2_COPIES = 16;  // 2-bit query copies per strand: 4 base phases x 4 byte shifts (encode.hip)
constexpr int Q2_TAIL = 32;    // zero bytes a window may read past the last base of a copy

struct ExtendArgs {
    const uint8_t* ref2;      // packed filter: 2-bit target, phase copy k at ref2 + k*ref2_stride, overlapped-line layout
    size_t ref2_stride;
    const uint8_t* query4;    // filters: 4-bit query of this call's strand, copy (base phase k, byte shift s) at query4 + (k + 2*s)*query4_stride
    size_t query4_stride;
    const uint8_t* ref8;      // ROW-CODED target: byte = code << 3 (points past the front pad)
    const uint8_t* query;     // encoded query, fwd or rc
    uint32_t ref_len;
    uint32_t query_len;
    const int* sub_mat;       // 64 ints in HBM
    int xdrop;
    int hspthresh;
    int noentropy;
    uint32_t l2_right_state;  // 1: a hit whose RIGHT side alone is open reaches level 2 with the right walk's packed state (L2Rec), and level 2 resumes
                              // behind the CTX_R_BASES context instead of at the anchor (option l2_right_state)
    uint32_t left_skip;       // bases of the seed window the class filter bounds instead of walking (CtxRec): seed_size, or 0 (option ctx_skip_seed = 0: round 4's layout)
    int log4_double;          // entropy divisor: 0 = (double)logf(4.0f), what the reference's `log(4.0f)` is under nvcc (hazard H2); 1 = log(4.0)
    int entropy_ulps;         // tests (hazard H13): the entropy factor moved by this many ulps (nextafter) before it is used
    int fin_batch;            // finished lanes a wave accumulates before it finalises + refills them
    int bufs_per_wave;        // launch heuristic: 64-hit buffers each wave should own at least
    uint32_t long_cap;        // bases per side the filter walks before it forwards the hit to the exact kernel
    uint32_t chain_sort_threads;  // workgroup size of the chain sort + link kernel (0: 256)
    int fast_filter;          // 1: xdrop >= 0 && 7*max(M) <= xdrop: the filter may skip the sticky select (extend.hip)
                              // 3: the packed 2-bit / 4-bit upper-bound filter is used (extend.hip 1b)
    CandRec* cand_list;
    uint32_t* cand_count;
    uint32_t cand_cap_recs;
    EntRec* ent_list;
    uint32_t* ent_count;
    uint32_t ent_cap_recs;
    // chain shortcut of the exact stage (extend.hip 2b); chain_cap == 0 disables it
    uint32_t chain_cap;                      // candidates per batch the chain buffers hold
    uint32_t chain_buckets;                  // 0: the device picks the hash buckets from the candidate count (chain_buckets_of); else forced (power of two)
    uint32_t chain_bucket_target;            // candidates a bucket should hold (> 0)
    uint32_t chain_group_max;                // entries the sort + link kernel holds in LDS at a time (64 .. 4096)
    uint32_t chain_no_link;                  // (measurement) 1: no link test, every candidate is a run head
    uint32_t chain_sort_blocks;              // workgroups of the sort + link kernel (0: 4096); they loop over the bucket groups
    uint32_t cand_sliced, cand_first;        // cand_sliced: the chain stages work on candidates [cand_first, cand_first + chain_cap) of the list
                                             // (a batch with more candidates than the chain buffers hold is run slice by slice)
    uint32_t* chain_bucket_cnt;              // [buckets] counters, then scatter cursors (zero on entry)
    uint32_t* chain_bucket_start;            // [buckets + 1]
    CandRec* chain_tmp;                      // [chain_cap] candidates dealt into buckets
    CandRec* chain_sorted;                   // [chain_cap] every bucket sorted by (iteration, diagonal, position)
    uint32_t* chain_is_head;                 // [chain_cap]
    uint32_t* chain_heads;                   // [chain_cap] indices of run heads
    uint32_t* chain_head_count;
    uint32_t* chain_big;                     // -> number of bucket groups the sort left unordered (bucket above its LDS capacity): diagnostics
    uint32_t max_waves;       // wave budget of the main kernel (resident waves of the chip)
    uint32_t long_blocks, ent_blocks;  // grid sizes of the long / entropy kernels (they read their counts on device)
    const Hit* hits;
    // TD ("table direct", probe.hip): no hit list -- hit g of the call is entry g - td_prefix[m] of the run of the m-th
    // non-empty query position; the packed filter reads its anchors straight out of the neighbourhood table
    int td;
    const uint32_t* td_chunk;   // [ceil(num_hits / TD_CHUNK_HITS)] record that holds the first hit of every chunk
    const TdRec* td_rec;        // [td_m + 1] one record per NON-EMPTY query position, in query order; td_rec[td_m].prefix = num_hits
    uint32_t td_m;
    const uint32_t* td_pos;     // neighbourhood table runs: seed START positions in the target (+ seed_size = anchor, :220)
    // class filter (extend.hip 1d), != null: the runs carry their target context (then td_pos is unused); the 2-bit shifted
    // copies of this call's query strand (right windows) and of the OTHER strand (left windows)
    const CtxRec* td_ctx;
    const uint64_t* td_bits;    // head-bit map of the call's hits (probe.hip): bit g set <=> a record (query position) starts at hit g
    const uint8_t* q2_own;
    const uint8_t* q2_other;
    size_t q2_stride;
    int cls[4];                 // score of a base pair by class = (target code ^ query code): upper bounds of the matrix (engine.hip)
    // audit (tests): every hit the filters REJECT is appended here as {ref_loc, query_loc}
    uint2* audit_list;
    uint32_t* audit_count;
    uint32_t audit_cap;
    // hits the context filter could not decide (walk alive at the end of the context, or bound passes): L2_NSUB sub-lists of
    // l2_cap records each, sub-list s appended to through counter l2_count[s * L2_CNT_STRIDE] (a wave uses sub-list
    // wave id % L2_NSUB).  ONE list with ONE counter serialised the whole filter on a single L2 atomic address: 125 k
    // appends of 64 records per call at ~11 ns each cost 0.67 ms of a 2.3 ms kernel.
    L2Rec* l2_list;
    uint32_t* l2_count;
    uint32_t l2_cap;
    uint32_t* l2_total;         // -> number of records in all sub-lists
    uint32_t* l2_max;           // -> largest sub-list count (> l2_cap: records were dropped, the host regrows and reruns)
    int src_cand;               // packed filter reads its anchors from l2_list / *l2_count instead of `hits`
    uint32_t ctx_waves;         // wave budget of the context filter (0: one chunk of TD_CHUNK_HITS hits per wave)
    uint32_t chain_q_bits;      // 32: table-direct call -- the chain sort key (chain_key32) hashes the diagonal alone, the hits of a diagonal are in
                                // iteration order anyway and the link test compares the iterations itself; else (general path, several iterations
                                // per batch) the iteration goes into the hash
    uint32_t ctx_threads;       // workgroup size of the context filter (0: its default)
    uint32_t cls_one_copy;      // 0 / 1: query windows from the unshifted copy (default), 2: from the sixteen shifted copies
    uint32_t l2_blocks;         // grid of the second level (its hit count is known on the device only)
    uint32_t seed_size;
    uint64_t num_hits;
    uint64_t hit_base;        // global index of hits[0] inside the call (segment boundaries are global)
    int num_segs;
    const uint64_t* seg_end;  // [num_segs] device array: exclusive global hit index where segment s ends (ascending)
    uint32_t seg_base;        // segment id of the first segment of this batch
    HspRec* out;              // survivors, appended
    uint32_t out_cap;         // capacity of out[]; the counter keeps counting past it, writes are dropped
    uint32_t* out_count;      // device counter
    unsigned long long* examined; // optional (null = do not count): [0] = E of the call, [1] = bases scored by the filter
    // key-ordered call (join.h): hits carry no index -- seg_end points at the per-chunk table join_plan_kernel wrote ({p_last : e_thr}
    // x 256, then the chunks' first segments x 256), L2Rec / CandRec::hidx is the hit's entry index inside its key's run
    int join;
    uint32_t join_q_lo;       // first query position of the call
    uint32_t join_chunk;      // chunk size (wga_chunk)
    // repeat-masker deltas (repeat_masker_src/seed_filter.cu:239-244, 305-333, 705-708)
    int rm;
    uint32_t rm_win_start, rm_win_end;
    int rm_rev;
};

// ---- encode.hip ------------------------------------------------------------------------------------------------
void launch_encode(const uint8_t* ascii, uint8_t* codes, uint32_t len, hipStream_t s);
void launch_encode_rev_comp(const uint8_t* ascii, uint8_t* codes, uint8_t* codes_rc, uint32_t len, hipStream_t s);
void launch_rev_comp_codes(const uint8_t* codes, uint8_t* codes_rc, uint32_t len, hipStream_t s);
// row-coded copy of the target for the extension kernel: out[i] = codes[i] << 3
void launch_row_code(const uint8_t* codes, uint8_t* out, uint32_t len, hipStream_t s);
// phase copies for the packed filter: out + k*copy_stride is copy k (4 copies at 2 bit/base, 2 copies at 4 bit/base)
constexpr int PACK_PAD = 64;  // pad bytes in front of / behind every 4-bit copy
constexpr int PACK4_COPIES = 8;  // 4-bit copies: 2 base phases x 4 byte shifts (encode.hip)
constexpr int PACK4_FRONT = 4;   // bytes below 0 of every 4-bit copy that hold real bases of the block start (<= PACK_PAD)
// 2-bit copies use overlapped 128-byte lines (encode.hip): 96 new bytes + the first 32 of the next line; logical byte =
// PACK2_BIAS + group index, physical byte of logical jj in the line chosen for logical jb = jj + 32 * (jb / 96)
constexpr int PACK2_PAYLOAD = 96;
constexpr int PACK2_BIAS = 96;
uint32_t pack2_phys_bytes(uint32_t len);
void launch_pack2_phases(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t nphys, hipStream_t s);
void launch_pack4_phases(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t nbytes, hipStream_t s);
// 2-bit query copies of the class filter: copy (p, s) byte j = bases [4 (j + s) + p, +4), codes >= 4 as 0, zero past the end
size_t q2_copy_stride(uint32_t len);
void launch_pack2_shifted(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t copies /* 1 or Q2_COPIES */, hipStream_t s);
// *mask |= 1 << code for every code that occurs in codes[0, len)
void launch_code_presence(const uint8_t* codes, uint32_t len, uint32_t* mask, hipStream_t s);

// ---- scan.hip --------------------------------------------------------------------------------------------------
// exclusive prefix of n u32 values; out_excl[n] receives the total.  OutT = uint32_t or uint64_t.
size_t scan_temp_bytes(uint64_t n);
void launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out_excl, uint64_t n, void* temp, hipStream_t s);
void launch_exclusive_scan_u64(const uint32_t* in, uint64_t* out_excl, uint64_t n, void* temp, hipStream_t s);

// ---- table.hip -------------------------------------------------------------------------------------------------
// histogram of valid k-mers of positions start_offset + i*step, i < num_steps  (seed_pos_table.cu:69-81)
void launch_table_count(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh,
                        uint32_t* hist, hipStream_t s);
// scatter positions into buckets (seed_pos_table.cu:89-101); cursor must be zero on entry
void launch_table_fill(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh,
                       const uint32_t* bucket_start, uint32_t* cursor, uint32_t* pos_table, hipStream_t s);
// canonical (ascending) order inside every bucket -- the reference order is atomic arrival order (H7)
void launch_table_sort_buckets(const uint32_t* bucket_start, uint32_t nkeys, uint32_t* pos_table, hipStream_t s);

// PARTITION build (table.hip): LDS-staged MSD radix partition, for seed weights table_partition_build_supported() accepts
bool table_partition_build_supported(int weight);
size_t table_partition_part_start_words();
void launch_table_keys(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh, uint32_t* keys,
                       uint32_t* coarse_hist /* 4096 words, zero on entry */, hipStream_t s);
void launch_table_partition_build(const uint32_t* keys, uint32_t num_steps, uint32_t start_offset, uint32_t step, int weight,
                                  const uint32_t* part_start /* 4097 */, uint32_t num_index, uint32_t* cursor /* 4096 */, uint32_t* key_a,
                                  uint32_t* pos_a, uint32_t* key_b, uint32_t* pos_b, uint8_t* part_unsorted /* 4096 */,
                                  uint32_t* bucket_start, uint32_t* pos_table, uint32_t* fine, void* scan_tmp, uint32_t** err_flag, hipStream_t s);
size_t table_partition_fine_words(int weight);

// ---- seeds.hip -------------------------------------------------------------------------------------------------
// device-side seeder: valid flags for query positions [start,end), then ordered emission
void launch_seed_flags(const uint8_t* query, uint32_t start, uint32_t end, SeedShape sh, uint32_t* flags, hipStream_t s);
void launch_seed_emit(const uint8_t* query, uint32_t start, uint32_t end, SeedShape sh, int transition,
                      const uint32_t* flag_prefix_excl, uint64_t* seeds, hipStream_t s);
// *ok &= (the host seed vector of ngroups x per words equals what the device seeder emits for the same positions)
void launch_seed_verify(const uint64_t* seeds, uint32_t ngroups, uint32_t per, const uint8_t* query, uint32_t query_len, SeedShape sh,
                        uint32_t tmask, uint32_t* ok, hipStream_t s);
// per seed: bucket start + bucket size  (find_num_hits, seed_filter.cu:157-182)
void launch_seed_lookup(const uint64_t* seeds, uint32_t num_seeds, const uint32_t* bucket_start /*4^k+1*/, uint32_t nkeys,
                        uint32_t* start_out, uint32_t* count_out, hipStream_t s);
// load-balanced bucket expansion (find_hits, seed_filter.cu:184-230)
void launch_expand_hits(const uint64_t* seeds, const uint32_t* start, const uint32_t* count,
                        const uint64_t* hit_prefix_excl, uint32_t seed_lo, uint32_t seed_hi, uint64_t hit_base,
                        const uint32_t* pos_table, uint32_t seed_size, Hit* hits, hipStream_t s);
// iteration plan (lower_bound chain of seed_filter.cu:718-745) computed on the device into *plan
struct IterPlan;
void launch_plan(const uint64_t* hit_prefix_excl, uint32_t num_seeds, uint64_t max_hits, int wrap32, IterPlan* plan,
                 hipStream_t s);

// ---- extend.hip ------------------------------------------------------------------------------------------------
void launch_extend_filter(const ExtendArgs& a, hipStream_t s);   // hits -> candidates
void launch_extend_filter_cls(const ExtendArgs& a, hipStream_t s);  // table-direct hits + their 32-byte context records -> l2_list (1d)
void launch_extend_exact(const ExtendArgs& a, hipStream_t s);    // candidates -> survivors + entropy records
// chain shortcut: keys -> (sort, dedup.hip) -> links/run heads -> one exact extension per run
void launch_chain_group(const ExtendArgs& a, hipStream_t s);
uint32_t chain_num_buckets();
void launch_extend_exact_chain(const ExtendArgs& a, hipStream_t s);
void launch_extend_entropy(const ExtendArgs& a, hipStream_t s);  // entropy records -> survivors

// ---- dedup.hip -------------------------------------------------------------------------------------------------
enum SortOrder { ORDER_DIAG = 0, ORDER_LASTZ = 1, ORDER_RM_FIRST = 2, ORDER_RM_DIAG = 3, ORDER_RM_FINAL = 4 };
size_t sort_temp_bytes(size_t n);
void launch_sort(const HspRec* in, HspRec* out, size_t n, SortOrder order, void* temp, size_t temp_bytes, hipStream_t s);
// adjacent-pair unique (thrust::unique_copy on device, hazard H3) within each segment; order preserving.
// exact = 0: hspEqual of seed_filter.cu:47-52 ; exact = 1: field equality (repeat masker :80-85)
size_t unique_temp_bytes(uint32_t n);
void launch_unique(const HspRec* in, HspRec* out, uint32_t n, int exact, uint32_t* out_count, void* tile_tmp, hipStream_t s);
void launch_strip(const HspRec* in, uint32_t n, void* out_segment_pairs, uint32_t* out_seg /*nullable*/, hipStream_t s);
uint32_t dedup_small_max_segs();
// sort(diag) -> unique -> sort(lastz) -> 16-byte records in LDS, one workgroup per segment id (< dedup_small_max_segs());
// the results land at the segments' input offsets
uint32_t dedup_seg_max_total();
uint32_t dedup_seg_info_words();
void launch_dedup_seg(const HspRec* in, uint32_t n, const uint32_t* n_dev, uint32_t nsegs, void* out_segment_pairs, uint32_t* seg_info,
                      uint32_t threads, uint32_t seg_max, hipStream_t s);

// ---- coverage.hip (repeat-masker post-processing, repeat_masker_src/seeder.cpp:153-188) ------------------------
struct SegPair16 { uint32_t ref_start, query_start, len; int32_t score; };  // layout of sa_segment_pair / segmentPair
void launch_coverage_range_reset(uint32_t* range /*{min start, max end}*/, hipStream_t s);
void launch_coverage_add_hsprec(const HspRec* hsps, uint32_t n, uint32_t* diff, uint32_t diff_len, uint32_t* range, hipStream_t s);
void launch_coverage_add_pairs(const SegPair16* hsps, uint32_t n, uint32_t* diff, uint32_t diff_len, uint32_t* range, hipStream_t s);
void launch_coverage_flags(const uint32_t* diff, const uint32_t* pre, uint32_t n, uint32_t carry_depth, uint32_t M,
                           uint32_t* is_start, uint32_t* is_end, hipStream_t s);
void launch_coverage_emit(const uint32_t* is_start, const uint32_t* is_end, const uint32_t* start_idx, const uint32_t* end_idx,
                          uint32_t n, uint32_t pos0, uint32_t start_base, uint32_t end_base, uint32_t cap, uint32_t* out_pairs,
                          hipStream_t s);
void launch_coverage_finish(uint32_t* pairs, uint32_t n, hipStream_t s);

}  // namespace sa
