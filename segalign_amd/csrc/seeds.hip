// seeds.hip -- seed words -> bucket lookup -> hit expansion, plus the device-side seeder.
//   seed_lookup   : find_num_hits (src/seed_filter.cu:157-182)
//   expand_hits   : find_hits     (src/seed_filter.cu:184-230), load balanced: the reference launches one
//                   128-thread block per seed and uses 4 of its lanes; here a block owns 256 consecutive seeds,
//                   stages their bucket extents in LDS and every lane materialises hits, so pos_table reads
//                   are contiguous runs and hit writes are fully coalesced 8-byte records.
//   iteration plan: SeedAndFilter's lower_bound chain (src/seed_filter.cu:718-745) as one tiny kernel, so the
//                   host needs a single D2H instead of ~6-10 implicit device_vector[] reads.
//   seed_flags/emit: src/seeder.cpp:57-74 on the device (same word format, same order).
#include <algorithm>

#include "kernels.h"
#include "kmer_dev.h"
#include "plan.h"

namespace sa {

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_flags_kernel(const uint8_t* __restrict__ query, uint32_t start, uint32_t end,
                                                         SeedShape sh, uint32_t* __restrict__ flags) {
    const uint32_t n = end - start;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t key;
        flags[i] = kmer_at(query, start + i, sh, key) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void seed_emit_kernel(const uint8_t* __restrict__ query, uint32_t start, uint32_t end,
                                                        SeedShape sh, int transition,
                                                        const uint32_t* __restrict__ flag_prefix_excl,
                                                        uint64_t* __restrict__ seeds) {
    const uint32_t n = end - start;
    const uint32_t tmask = transition ? (sh.transition_mask & ((1u << sh.weight) - 1u)) : 0u;
    const uint32_t per = 1u + (uint32_t)__builtin_popcount(tmask);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t key;
        const uint32_t j = start + i;
        if (kmer_at(query, j, sh, key)) {
            uint64_t* o = seeds + (uint64_t)flag_prefix_excl[i] * per;
            *o++ = ((uint64_t)key << 32) + j;  // seeder.cpp:60-61
            for (int t = 0; t < sh.weight; t++)
                if ((tmask >> t) & 1u) *o++ = ((uint64_t)(key ^ (2u << (2 * t))) << 32) + j;  // seeder.cpp:64-69
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Is a HOST seed vector (src/seeder.cpp:57-74: per valid position the k-mer word, then one word per transition position) the
// vector the device would emit for the same strand?  One lane per GROUP of `per` words: the group's words are
// {key, key ^ (2 << 2t)...} of the k-mer that really stands at the group's query position, and the positions ascend strictly.
// Together with "#groups == #valid positions in [first, last]" (counted by the position probe) this makes the vector EQUAL to the
// device-emitted one, so the call may take the table-direct path; any deviation (hand-made words, the reference's shifted
// minus-strand arena of hazard H14) clears *ok and the call keeps the reference-shaped path.
__global__ __launch_bounds__(256) void seed_verify_kernel(const uint64_t* __restrict__ seeds, uint32_t ngroups, uint32_t per,
                                                          const uint8_t* __restrict__ query, uint32_t query_len, SeedShape sh,
                                                          uint32_t tmask, uint32_t* __restrict__ ok) {
    bool good = true;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        const uint64_t* w = seeds + (uint64_t)g * per;
        const uint64_t w0 = w[0];
        const uint32_t qpos = (uint32_t)w0;
        uint32_t key = 0;
        if ((uint64_t)qpos + (uint32_t)sh.span > query_len || !kmer_at(query, qpos, sh, key) || (uint32_t)(w0 >> 32) != key) good = false;
        if (g > 0 && (uint32_t)w[-(int64_t)per] >= qpos) good = false;  // strictly ascending positions
        uint32_t j = 1;
        for (int t = 0; t < sh.weight; t++)
            if ((tmask >> t) & 1u) {
                if (w[j] != (((uint64_t)(key ^ (2u << (2 * t))) << 32) + qpos)) good = false;  // seeder.cpp:64-69
                j++;
            }
    }
    if (!good) atomicAnd(ok, 0u);
}

void launch_seed_verify(const uint64_t* seeds, uint32_t ngroups, uint32_t per, const uint8_t* query, uint32_t query_len, SeedShape sh,
                        uint32_t tmask, uint32_t* ok, hipStream_t s) {
    if (ngroups == 0) return;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)ngroups + 255) / 256, 4096);
    hipLaunchKernelGGL(seed_verify_kernel, dim3(blocks), dim3(256), 0, s, seeds, ngroups, per, query, query_len, sh, tmask, ok);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_lookup_kernel(const uint64_t* __restrict__ seeds, uint32_t num_seeds,
                                                          const uint32_t* __restrict__ bucket_start, uint32_t nkeys,
                                                          uint32_t* __restrict__ start_out, uint32_t* __restrict__ count_out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < num_seeds; i += gridDim.x * blockDim.x) {
        const uint32_t key = (uint32_t)(seeds[i] >> 32);  // :172
        uint32_t b = 0, e = 0;
        if (key < nkeys) {  // the two words are adjacent: same 64-byte line except at line boundaries
            b = bucket_start[key];
            e = bucket_start[key + 1];
        }
        start_out[i] = b;
        count_out[i] = e - b;  // :175-180
    }
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int EXP_SEEDS = 256;  // seeds per block == threads per block

__global__ __launch_bounds__(EXP_SEEDS) void expand_hits_kernel(const uint64_t* __restrict__ seeds,
                                                                const uint32_t* __restrict__ start,
                                                                const uint32_t* __restrict__ count,
                                                                const uint64_t* __restrict__ hit_prefix_excl,
                                                                uint32_t seed_lo, uint32_t seed_hi, uint64_t hit_base,
                                                                const uint32_t* __restrict__ pos_table, uint32_t seed_size,
                                                                Hit* __restrict__ hits) {
    __shared__ uint32_t s_off[EXP_SEEDS + 1];  // block-local exclusive hit offsets
    __shared__ uint32_t s_start[EXP_SEEDS];
    __shared__ uint32_t s_qloc[EXP_SEEDS];
    __shared__ uint32_t s_wave[EXP_SEEDS / 64];

    const uint32_t first = seed_lo + blockIdx.x * EXP_SEEDS;
    const uint32_t sid = first + threadIdx.x;
    const bool in = sid < seed_hi;
    const uint32_t c = in ? count[sid] : 0u;
    s_start[threadIdx.x] = in ? start[sid] : 0u;
    s_qloc[threadIdx.x] = in ? (uint32_t)(seeds[sid] & 0xFFFFFFFFull) + seed_size : 0u;  // :204

    // block exclusive scan of c (wave shuffles + 4 partials)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < EXP_SEEDS / 64; w++) if (w < wave) base += s_wave[w];
    s_off[threadIdx.x] = base + inc - c;
    if (threadIdx.x == EXP_SEEDS - 1) s_off[EXP_SEEDS] = base + inc;
    __syncthreads();

    const uint32_t total = s_off[EXP_SEEDS];
    Hit* out = hits + (hit_prefix_excl[first] - hit_base);
    for (uint32_t j = threadIdx.x; j < total; j += EXP_SEEDS) {
        // largest idx with s_off[idx] <= j  (8 LDS probes)
        uint32_t lo = 0, hi = EXP_SEEDS;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= j) lo = mid; else hi = mid;
        }
        const uint32_t k = j - s_off[lo];
        Hit h;
        h.ref_loc = pos_table[s_start[lo] + k] + seed_size;  // :220
        h.query_loc = s_qloc[lo];
        out[j] = h;  // slot order inside a seed is irrelevant: every consumer sorts (SURVEY a-6)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Iteration plan, restating src/seed_filter.cu:718-745 literally on the inclusive prefix incl[i] = excl[i+1].
// max_hits is the reference's MAX_HITS (an int compared and added as unsigned).  wrap32 = 1 reproduces the
// uint32_t arithmetic of src/ (the repeat masker uses 64-bit, wrap32 = 0).
__global__ void plan_kernel(const uint64_t* __restrict__ excl, uint32_t num_seeds, uint64_t max_hits, int wrap32,
                            IterPlan* __restrict__ plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // `excl` may point into the prefix of a larger seed vector (several chunks planned from one scan): all values are
    // taken relative to excl[0], which is 0 for a stand-alone call
    const uint64_t* incl = excl + 1;
    const uint64_t base = excl[0];
    uint64_t num_hits = num_seeds ? incl[num_seeds - 1] - base : 0;  // :716
    if (wrap32) num_hits &= 0xFFFFFFFFull;
    plan->num_hits = num_hits;
    plan->num_iter = 0;
    plan->overflow = 0;
    if (num_seeds == 0 || num_hits == 0) return;
    uint32_t num_iter;
    uint64_t limit;
    if (num_hits < max_hits) { num_iter = 2; limit = num_hits; }            // :721-724
    else { num_iter = (uint32_t)(num_hits / max_hits + 2); limit = max_hits; }  // :725-728
    if (num_iter > PLAN_MAX_ITER) { plan->overflow = num_iter; return; }
    for (uint32_t i = 0; i + 1 < num_iter; i++) {  // :732-739
        uint32_t lo = 0, hi = num_seeds;
        while (lo < hi) {
            uint32_t mid = lo + (hi - lo) / 2;
            uint64_t v = incl[mid] - base;
            if (wrap32) v &= 0xFFFFFFFFull;
            if (v < limit) lo = mid + 1; else hi = mid;
        }
        int64_t pos = (int64_t)lo - 1;  // -1 == the reference's wrapped index (hazard H5): treated as "0 hits"
        plan->limit_pos[i] = pos;
        uint64_t at = pos >= 0 ? incl[pos] - base : 0;
        limit = at + max_hits;
        if (wrap32) limit &= 0xFFFFFFFFull;
        if (limit > num_hits) limit = num_hits;
    }
    plan->limit_pos[num_iter - 1] = (int64_t)num_seeds - 1;                         // :741
    if (plan->limit_pos[num_iter - 1] == plan->limit_pos[num_iter - 2]) num_iter--;  // :743
    for (uint32_t i = 0; i < num_iter; i++) {
        int64_t p = plan->limit_pos[i];
        plan->upto[i] = p >= 0 ? incl[p] - base : 0;
    }
    plan->num_iter = num_iter;
}

// ---------------------------------------------------------------------------------------------------------------
static inline int grid_for(uint64_t work_items, int block, int max_blocks = 256 * 16) {
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

void launch_seed_flags(const uint8_t* query, uint32_t start, uint32_t end, SeedShape sh, uint32_t* flags, hipStream_t s) {
    if (end <= start) return;
    hipLaunchKernelGGL(seed_flags_kernel, dim3(grid_for(end - start, 256)), dim3(256), 0, s, query, start, end, sh, flags);
}
void launch_seed_emit(const uint8_t* query, uint32_t start, uint32_t end, SeedShape sh, int transition,
                      const uint32_t* flag_prefix_excl, uint64_t* seeds, hipStream_t s) {
    if (end <= start) return;
    hipLaunchKernelGGL(seed_emit_kernel, dim3(grid_for(end - start, 256)), dim3(256), 0, s, query, start, end, sh,
                       transition, flag_prefix_excl, seeds);
}
void launch_seed_lookup(const uint64_t* seeds, uint32_t num_seeds, const uint32_t* bucket_start, uint32_t nkeys,
                        uint32_t* start_out, uint32_t* count_out, hipStream_t s) {
    if (num_seeds == 0) return;
    hipLaunchKernelGGL(seed_lookup_kernel, dim3(grid_for(num_seeds, 256)), dim3(256), 0, s, seeds, num_seeds, bucket_start,
                       nkeys, start_out, count_out);
}
void launch_expand_hits(const uint64_t* seeds, const uint32_t* start, const uint32_t* count,
                        const uint64_t* hit_prefix_excl, uint32_t seed_lo, uint32_t seed_hi, uint64_t hit_base,
                        const uint32_t* pos_table, uint32_t seed_size, Hit* hits, hipStream_t s) {
    if (seed_hi <= seed_lo) return;
    uint32_t nblocks = (seed_hi - seed_lo + EXP_SEEDS - 1) / EXP_SEEDS;
    hipLaunchKernelGGL(expand_hits_kernel, dim3(nblocks), dim3(EXP_SEEDS), 0, s, seeds, start, count, hit_prefix_excl,
                       seed_lo, seed_hi, hit_base, pos_table, seed_size, hits);
}
void launch_plan(const uint64_t* hit_prefix_excl, uint32_t num_seeds, uint64_t max_hits, int wrap32, IterPlan* plan,
                 hipStream_t s) {
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(64), 0, s, hit_prefix_excl, num_seeds, max_hits, wrap32, plan);
}

}  // namespace sa
