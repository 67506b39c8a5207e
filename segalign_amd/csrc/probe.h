// probe.h -- launcher declarations of probe.hip (neighbourhood table + position probe; see the file header there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace sa {

constexpr int TD_MAX_BOUNDS = 260; // chunk boundaries of one call: SA_MAX_CHUNKS + 1 <= 260

struct TdBounds {                 // chunk boundaries of a call: boundary c (c < nb) is query position min(start + c * chunk, end) -- the
    int nb;                       // chunks of a call tile [start, end) with wga_chunk-sized pieces, the last one may be short or empty
    uint32_t start, end, chunk;
};

constexpr int TD_MAX_ITER = 8;    // reference iterations a chunk of a table-direct call may be planned in: num_hits / MAX_HITS + 2 <= 8, i.e. chunks of up
                                  // to 6 x MAX_HITS hits (a 500 Mbp block's 60 M-hit chunks under an 8 GiB GPU's 33.5 M: three); beyond: the general path

struct TdPlan {                   // per chunk, written by probe_plan_kernel
    uint64_t hit_base;            // offset of the chunk's first hit inside the call
    uint64_t num_hits;            // src/seed_filter.cu:716
    uint64_t upto[TD_MAX_ITER];   // call-wide hit offsets where the chunk's reference iterations END (:718-745): [0] = where the last iteration starts and
                                  // [1] = hit_base + num_hits when num_hits < MAX_HITS (always two iterations), the greedy groups of :725-741 otherwise
    uint32_t num_valid;           // valid seed positions (seed words = num_valid * words per position)
    uint32_t m_lo, m_hi;          // the chunk's range of compacted (non-empty) positions
    uint32_t n_iter;              // iterations in upto[] (0: no hits); TD_PLAN_OVERFLOW: the chunk needs more than TD_MAX_ITER, or its counts wrap the
};                                // reference's uint32 arithmetic: the general path plans it
constexpr uint32_t TD_PLAN_OVERFLOW = 0xFFFFFFFFu;

struct ZeroList {                 // dword regions the per-call clearing kernel zeroes next to the head-bit map
    static constexpr int N = 4;
    uint32_t* p[N];
    uint32_t n[N];
};

// neighbourhood table build
void launch_nbr_count(const uint32_t* bucket_start, uint32_t nkeys, uint32_t tmask, int weight, uint32_t* cnt, uint32_t* overflow,
                      hipStream_t s);
void launch_nbr_fill(const uint32_t* bucket_start, const uint32_t* pos_table, uint32_t nkeys, uint32_t tmask, int weight,
                     const uint64_t* nbr_start, uint32_t* nbr_pos, hipStream_t s);

// the table as context records (kernels.h CtxRec); scratch: room for num_index records (nullptr: every entry cuts its context out
// of the target itself)
void launch_nbr_fill_ctx(const uint32_t* bucket_start, const uint32_t* pos_table, uint32_t nkeys, uint32_t tmask, int weight,
                         const uint64_t* nbr_start, const uint8_t* ref2, size_t ref2_stride, uint32_t seed_size, uint32_t left_skip, CtxRec* ctx,
                         CtxRec* scratch, uint32_t num_index, hipStream_t s);

// position probe of n = end - start query positions; t_off/t_cnt: n entries of scratch; c_rec: n + 1 records
size_t probe_partial_bytes(uint32_t n);
size_t probe_bounds_bytes();
void launch_probe_lookup(const uint8_t* query, uint32_t start, uint32_t n, SeedShape sh, const uint64_t* nbr_start, uint32_t nkeys,
                         uint64_t* t_off, uint32_t* t_cnt, void* partial_buf, hipStream_t s);
// chunk_rec[k] (k < chunk_cap) = index of the record that holds hit k * TD_CHUNK_HITS of the call
// head_bits (nullable): bit g of the map is set <=> a record starts at hit g of the call; only hits below 32 * head_words get a bit
void launch_probe_compact(uint32_t start, uint32_t n, const uint64_t* t_off, const uint32_t* t_cnt, void* partial_buf, void* bounds_buf,
                          TdRec* c_rec, uint32_t* chunk_rec, uint32_t chunk_cap, uint32_t* head_bits, uint32_t head_words,
                          const ZeroList& zero, const TdBounds& bpos, bool first_pass, hipStream_t s);
void launch_call_clear(const ZeroList& zero, hipStream_t s);
// max_hits: MAX_HITS in force (src/seed_filter.cu:832-841); wrap32: the src/ binary's uint32 hit arithmetic (the repeat masker's is 64-bit);
// seg_end: [MAX_SEGS] the iteration ends of the call in hit order (sum of n_iter over the chunks; the host checks that it fits)
void launch_probe_plan(const uint8_t* query, SeedShape sh, uint32_t tmask, const uint32_t* bucket_start, const void* bounds_buf, int nchunks,
                       const TdRec* c_rec, uint64_t max_hits, int wrap32, TdPlan* plan, uint64_t* seg_end, hipStream_t s);

}  // namespace sa
