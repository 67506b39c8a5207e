// join.h -- the KEY-ORDERED form of a table-direct call (join.hip, extend.hip 1e): the seed hits of a call are enumerated per seed
// key as (the key's run of context records) x (the query positions of the call that carry the key) instead of as a stream of records
// in query order.  What that buys (tools/micro/join_proto.hip): a lane keeps ONE context record in registers and
// scores it against the c positions of its key -- the record is fetched once for c hits --, the LDS address of a class field is one
// XOR of two pre-masked field words (record side once per tile, query side stored with the position), and the waves claim their
// work dynamically: 215 G hits/s against 140-148 for the streamed class filter on the same chip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace sa {

constexpr int JOIN_CMAX = 16;        // query positions one entry pairs a run with (a key with more is cut into several entries)
constexpr int JOIN_QX_DW = 20;       // dwords of a QRecX: query position + 8 right + 10 left field words + the left tail field
constexpr uint32_t JOIN_TAIL_OFF = 16384u;  // byte offset of the four-base tail table behind the 4096 six-base entries (extend.hip cls_table_init)
constexpr uint32_t JOIN_GRAIN = 256;        // work units (64-hit steps) a wave claims at a time
// The segment table of a key-ordered call (ExtendArgs::seg_end, staged in LDS by SEG_TABLE(), read by seg_of): {p_last : e_thr} for chunk
// c at [c], the chunk's first segment id at [JOIN_SEG_FIRST + c].  One call carries at most SA_MAX_CHUNKS = JOIN_SEG_FIRST chunks, two
// segments each: the table fills the MAX_SEGS u64 the candidate stages stage (engine_internal.h asserts SA_MAX_CHUNKS against it).
constexpr uint32_t JOIN_SEG_FIRST = 256;
static_assert(2 * JOIN_SEG_FIRST <= (uint32_t)MAX_SEGS, "join segment table: {p_last, e_thr} x chunks | first segment x chunks must fit MAX_SEGS entries");

// Per chunk of the call (device; the host reads them with the call's one synchronisation).  Only what the reference's iteration plan
// needs: for num_hits < MAX_HITS the plan is "everything before the last hit-bearing seed word / that word's hits"
// (src/seed_filter.cu:718-745), i.e. a hit belongs to the chunk's second iteration iff it sits at the chunk's last non-empty
// position AND at or behind entry e_thr of that position's run.
struct JoinChunk {
    unsigned long long hits;   // seed hits of the chunk
    uint32_t valid;            // valid seed positions (x words per position = seed words the reference would have been handed)
    uint32_t p_last;           // last position with hits
    uint32_t e_thr;            // first run entry of the last hit-bearing seed word of p_last
    uint32_t seg0;             // id of the chunk's first segment (two per chunk with hits, in chunk order)
};

// Written and read on the device: class layout of the entry list and the work the filter's waves claim.
struct JoinHead {
    uint32_t cls_count[JOIN_CMAX + 2];             // entries of class c (c = 1..JOIN_CMAX)
    uint32_t cls_first[JOIN_CMAX + 2];             // first entry of class c; [JOIN_CMAX + 1] = number of entries
    uint32_t cls_cursor[JOIN_CMAX + 2];            // scatter cursors
    uint32_t n_entries;
    uint32_t pad;
    unsigned long long vbase[JOIN_CMAX + 2];       // virtual index (record index in entry order) where class c starts
    unsigned long long work_base[JOIN_CMAX + 2];   // work units in front of class c (classes are worked from JOIN_CMAX down to 1)
    unsigned long long work_total;
    unsigned long long work_next;                  // next unclaimed work unit
    unsigned long long total_hits;
};

struct JoinEnt {              // one (run, up to JOIN_CMAX query positions of its key) pair
    uint32_t run_lo, run_hi;  // offset of the key's run in the context table
    uint32_t q_first;         // index of the first of its c positions in the key-sorted position list (= QRecX index)
    uint32_t n_t;             // run length
};

struct JoinArgs {             // what the filter (extend.hip 1e) needs on top of ExtendArgs
    const JoinHead* head;
    JoinHead* head_rw;                       // (work_next)
    const uint4* ent;                        // [n_entries] JoinEnt
    const unsigned long long* vstart;        // [n_entries + 1] exclusive prefix of n_t in entry order
    const uint32_t* qx;                      // [positions] QRecX, JOIN_QX_DW dwords each
};

// ---- join.hip launchers ----
void launch_join_stats(const uint32_t* qk_start, const uint32_t* qpos, const uint64_t* nbr_start, uint32_t nkeys, uint32_t start, uint32_t chunk, int K,
                       unsigned long long* hits, uint32_t* valid, uint32_t* last1, hipStream_t s);
void launch_join_plan(const uint8_t* query, SeedShape sh, uint32_t tmask, const uint32_t* bucket_start, const unsigned long long* hits, const uint32_t* valid,
                      const uint32_t* last1, int K, JoinChunk* plan, uint64_t* seg_table /* [2 * JOIN_SEG_FIRST]: {p_last, e_thr} per chunk | seg0 per chunk */, JoinHead* head, hipStream_t s);
void launch_join_entries(const uint32_t* qk_start, const uint64_t* nbr_start, uint32_t nkeys, JoinHead* head, uint4* ent, uint32_t* ent_nt, uint32_t ent_cap,
                         hipStream_t s);  // count -> layout -> scatter; head->cls_count must be zero on entry, ent_nt zero-filled
void launch_join_finish(JoinHead* head, const unsigned long long* vstart, hipStream_t s);
void launch_join_qx(const uint32_t* qk_start, uint32_t nkeys, const uint32_t* qpos, const uint8_t* q2_own, const uint8_t* q2_other, uint32_t query_len,
                    uint32_t seed_size, uint32_t left_skip, uint32_t* qx, hipStream_t s);
struct ExtendArgs;
void launch_join_filter(const ExtendArgs& a, const JoinArgs& j, hipStream_t s);  // extend.hip 1e

}  // namespace sa
