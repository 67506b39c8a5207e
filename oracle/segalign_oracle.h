/*
 * segalign_oracle.h -- CPU restatement (plain C) of SegAlign's seed -> filter -> ungapped-extend path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may build, link, load or call anything under oracle/.  The product
 * (segalign_amd/, libsegalign_hip.so) never routes through it and has no CPU fallback.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - k-mer / shape / transition / RevComp functions: PINNED against the real reference object
 *     oracle/_ref/libntcoding_ref.so (g++ on /root/reference/common/ntcoding.cpp as it lies) and against
 *     the golden vectors generated from it (tests/golden/ntcoding_golden.json).
 *   - the FASTA reader the hosts use (segalign_amd/fasta.py, segalign_amd/host/host_common.hpp -- not part of this library): PINNED against
 *     oracle/_ref/kseq_dump (the reference's vendored common/kseq.h compiled as it lies, door oracle/kseq_ref.cpp) and
 *     tests/golden/kseq_golden.json generated from it.
 *   - everything else (table build, find_hsps, SeedAndFilter orchestration): PARITY UNPINNED by the
 *     reference -- the reference ships no tests, golden vectors or fixtures, and its .cu files cannot be
 *     compiled here (no nvcc / CUDA headers / TBB).  Each function below follows the cited reference lines;
 *     the scalar X-drop form is cross-checked against an independent tile-by-tile restatement of the
 *     32-lane kernel (orc_extend_hit_tiled) in tests/test_oracle_extend.py.
 *     Second routes (none a pin: each needs stand-ins for what the image lacks): the reference's kernels and
 *     host files compiled as they lie with the CUDA runtime / thrust / TBB stood in for and the kernels under SIMT emulation --
 *     stage by stage (tests/golden/make_*_golden.py) and the five hot-path files as ONE program end to end
 *     (tests/golden/make_path_golden.py: every g_SeedAndFilter return, iteration plans over a small MAX_HITS included);
 *     this restatement returns all of those vectors.
 *
 * Every function cites the /root/reference file:line it restates.
 */
#ifndef SEGALIGN_ORACLE_H
#define SEGALIGN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/graph.h:25-30 */
typedef struct {
    uint32_t ref_start;
    uint32_t query_start;
    uint32_t len;
    int32_t score;
} orc_segment;

/* common/parameters.h:4-13 */
enum { ORC_A = 0, ORC_C = 1, ORC_G = 2, ORC_T = 3, ORC_L = 4, ORC_N = 5, ORC_X = 6, ORC_E = 7, ORC_NUC = 8 };
#define ORC_INVALID_KMER 0x80000000u /* common/parameters.h:17 */

/* ---- scoring matrix: src/main.cpp:187-268 ------------------------------------------------------------ */
/* ambiguous_mode: 0 = "x" (default), 1 = "n", 2 = "iupac"; reward/penalty as parsed at main.cpp:194-204
 * (for "n"/"iupac" without explicit numbers the caller passes 0,0). */
void orc_build_sub_mat(int* sub_mat /*64*/, int xdrop, int ambiguous_mode, int ambiguous_reward, int ambiguous_penalty);

/* ---- sequence encoding: common/seed_filter_interface.cu:18-47, src/seed_filter.cu:110-155 ------------- */
void orc_encode(const char* src, size_t len, uint8_t* dst);
void orc_encode_rev_comp(const char* src, size_t len, uint8_t* dst, uint8_t* dst_rc);
/* repeat_masker_src/seed_filter.cu:137-167 (rc of an already encoded sequence) */
void orc_rev_comp_codes(const uint8_t* src, size_t len, uint8_t* dst_rc);
/* host ASCII reverse complement: common/ntcoding.cpp:63-105 (same argument order) */
void orc_rev_comp_ascii(char* dst_buffer, const char* src_buffer, size_t rc_start, size_t start, size_t len);

/* ---- seed shape + k-mer: common/ntcoding.cpp:6-61 ----------------------------------------------------- */
int orc_generate_shape_pos(const char* shape);
int orc_is_transition_at_pos(int t);
uint32_t orc_kmer_index_at_pos(const char* sequence, size_t pos, uint32_t seed_size);

/* ---- seed position table: common/seed_pos_table.cu:49-109 --------------------------------------------- */
/* index_table_out: 4^kmer_size entries, INCLUSIVE end offset per key (what the reference uploads,
 * seed_pos_table.cu:103).  pos_table_out: caller passes capacity >= ref_length; returns num_index.
 * Bucket order: ascending position (the reference's order is atomic arrival order = nondeterministic). */
uint32_t orc_generate_seed_pos_table(const char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step,
                                     int shape_size, int kmer_size, uint32_t* index_table_out,
                                     uint32_t* pos_table_out);

/* ---- host seeding loop: src/seeder.cpp:57-74 ---------------------------------------------------------- */
/* seeds_out capacity >= 13*(e-i) (or (e-i) without transitions); returns number of seed words. */
size_t orc_make_seeds(const char* query_buffer, size_t q_block_start, uint32_t i, uint32_t e, uint32_t seed_size,
                      int kmer_size, int transition, uint64_t* seeds_out);

/* ---- ungapped X-drop extension of ONE hit: src/seed_filter.cu:232-652 --------------------------------- */
typedef struct {
    const uint8_t* ref;   /* encoded target, ref_len codes */
    const uint8_t* query; /* encoded query (fwd or rc), query_len codes */
    uint32_t ref_len;
    uint32_t query_len;
    const int* sub_mat; /* 64 */
    int xdrop;
    int hspthresh;
    int noentropy;
    int log4_is_float; /* 1 = divisor is (double)logf(4.0f) as written at seed_filter.cu:623 (default) */
    int entropy_ulps;  /* tests (hazard H13): the entropy factor moved by this many ulps (nextafter) before :633/:637 use it */
} orc_extend_params;

/* Returns 1 if the hit passes (d_done = 1) and fills *out; 0 otherwise (out = zeroed record as at :641-647).
 * examined (may be NULL) += number of (ref,query) positions scored on both sides. */
int orc_extend_hit(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, orc_segment* out,
                   uint64_t* examined);
/* Independent tile-by-tile restatement with `tile` lanes (32 = the reference's warp; any power of two <= 64). */
int orc_extend_hit_tiled(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, int tile,
                         orc_segment* out);

/* ---- SeedAndFilter: src/seed_filter.cu:682-828 -------------------------------------------------------- */
typedef struct {
    uint64_t num_hits;       /* total seed hits */
    uint64_t num_survivors;  /* hits with d_done = 1 (before sort/unique) */
    uint64_t num_examined;   /* E of SURVEY.md 8(d) */
    uint32_t num_iter;       /* iterations executed */
} orc_saf_stats;

typedef struct {
    orc_extend_params ext;
    const uint32_t* index_table; /* inclusive ends, 4^k */
    const uint32_t* pos_table;
    uint32_t seed_size;
    int64_t max_hits; /* MAX_HITS of seed_filter.cu:841 */
    int num_threads;  /* OpenMP threads for the per-hit loop (results independent of it) */
} orc_saf_params;

/* Returns number of orc_segment written to *out_vec (>= 1: element 0 is the header, seed_filter.cu:806-809).
 * *out_vec is malloc'ed; free with orc_free. */
size_t orc_seed_and_filter(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, orc_segment** out_vec,
                           orc_saf_stats* stats);

/* repeat-masker variant: repeat_masker_src/seed_filter.cu:724-876 (query = target itself; ref window filter,
 * rc coordinate flip, different sort/unique chain, 64-bit header). ext.query must be the target (rev=0) or its
 * encoded reverse complement (rev=1); ext.query_len == ext.ref_len. */
size_t orc_seed_and_filter_rm(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, int rev,
                              uint32_t ref_start, uint32_t ref_end, orc_segment** out_vec, orc_saf_stats* stats);

/* The same calls with their intermediate lists kept (tests/test_oracle_rm_golden.py): hits = the hit list as find_hits leaves it
 * (score 0, or -1 outside the repeat masker's target window), ext/done = records and flags as find_hsps leaves them, reduced = the
 * compacted list as compress_output leaves it; each concatenated over the call's iterations.  rm = 0: src/, rm = 1: repeat masker. */
typedef struct {
    orc_segment* hits; orc_segment* ext; uint8_t* done; orc_segment* reduced;
    size_t n_hits, n_ext, n_reduced, cap_hits, cap_ext, cap_reduced;
} orc_stage_trace;
size_t orc_seed_and_filter_traced(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, int rm, int rev, uint32_t ref_start,
                                  uint32_t ref_end, orc_segment** out_vec, orc_saf_stats* stats, orc_stage_trace* tr);
void orc_stage_trace_free(orc_stage_trace* tr);
/* The ordering chain alone on `n` records as one dedup scope: src/seed_filter.cu:776-782 (rm = 0) or
 * repeat_masker_src/seed_filter.cu:819-831 (rm = 1).  *out is malloc'ed (orc_free); returns the records kept. */
size_t orc_order_hsps(const orc_segment* in, size_t n, int rm, orc_segment** out);

/* repeat-masker post-processing, repeat_masker_src/seeder.cpp:153-188: per-position uint8_t coverage counters
 * incremented over query_start .. query_start+len-1 of every HSP, then runs with count >= M (a run still open at the
 * end of the block is not written, :168-186).  Returns the number of {query_start,len} pairs stored in *out. */
typedef struct { uint32_t query_start, len; } orc_interval;
size_t orc_rm_coverage_intervals(const orc_segment* hsps, size_t num_hsps, uint32_t block_len, uint32_t M, orc_interval** out);

/* repeat-masker interval plan, repeat_masker_src/main.cpp:316-436: blocks with neighbour overlap and, per
 * lastz_interval of each block, the seed range [start,end) and the target window [ref_start, ref_end]. */
typedef struct { uint32_t block_index; uint64_t block_start; uint32_t block_len, start, end, ref_start, ref_end; } orc_rm_task;
size_t orc_rm_plan(uint64_t seq_len, uint32_t seq_block_size, uint32_t lastz_interval_size, float prop_neigh_interval,
                   uint32_t seed_size, orc_rm_task** out);

void orc_free(void* p);

/* MAX_HITS exactly as computed at src/seed_filter.cu:832-841 from a device's totalGlobalMem. */
int orc_max_hits_for_mem(uint64_t total_global_mem);

#ifdef __cplusplus
}
#endif
#endif
