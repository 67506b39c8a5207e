// coverage.hip -- repeat-masker post-processing on the device.
//
// Replaces the host loops of repeat_masker_src/seeder.cpp:153-188:
//     for every HSP: for (j = query_start; j < query_start + len; j++) int_count[j]++;      (:155-159, uint8_t counters)
//     runs of positions with int_count[i] >= M  ->  Segment{query_start, len}                (:168-186)
// The device form keeps a DIFFERENCE array (one +1 / -1 pair per HSP instead of `len` increments), turns it into the
// coverage count with one prefix scan over the touched range only, and compacts the run boundaries with a second scan.
// Reference quirks kept on purpose:
//   * the counters are uint8_t, so the depth compared with M is the true depth mod 256;
//   * `len` is bases-1, so an HSP covers query_start .. query_start+len-1 (its last base is not counted);
//   * a run is emitted when the first uncovered position AFTER it is seen (:177-184), so a run that reaches the end
//     of the block is never written: the host drops a last run that ends at block_len (api_rm.hip coverage_finish; only
//     HSPs with query_start + len == block_len make one); with M == 0 the whole block is one unterminated run and nothing is written.
#include "kernels.h"

namespace sa {

constexpr int COV_THREADS = 256;

// diff[query_start] += 1, diff[query_start+len] -= 1 ; range[0] = min start, range[1] = max end (atomic)
template <typename Rec>
__global__ __launch_bounds__(COV_THREADS) void coverage_add_kernel(const Rec* __restrict__ hsps, uint32_t n,
                                                                   uint32_t* __restrict__ diff, uint32_t diff_len,
                                                                   uint32_t* __restrict__ range) {
    const uint32_t i = blockIdx.x * COV_THREADS + threadIdx.x;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    if (i < n) {
        const uint32_t qs = hsps[i].query_start, len = hsps[i].len;
        const uint64_t e = (uint64_t)qs + len;
        if (len > 0 && e < diff_len) {  // e <= block_len always holds for an HSP inside the block (diff_len = block_len+1)
            atomicAdd(&diff[qs], 1u);
            atomicAdd(&diff[e], 0xFFFFFFFFu);
            lo = qs;
            hi = (uint32_t)e;
        }
    }
    // wave-level min/max, one atomic pair per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && lo != 0xFFFFFFFFu) {
        atomicMin(&range[0], lo);
        atomicMax(&range[1], hi);
    }
}

// pre[i] = exclusive prefix of diff over the tile, so carry_depth + pre[i] is the depth of the position before lo+i;
// is_start[i] / is_end[i] mark the first position of a run and the first uncovered position after a run.
__global__ __launch_bounds__(COV_THREADS) void coverage_flags_kernel(const uint32_t* __restrict__ diff, const uint32_t* __restrict__ pre,
                                                                     uint32_t n, uint32_t carry_depth, uint32_t M,
                                                                     uint32_t* __restrict__ is_start, uint32_t* __restrict__ is_end) {
    const uint32_t i = blockIdx.x * COV_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t before = carry_depth + pre[i];
    const uint32_t here = before + diff[i];
    const bool a = (before & 0xFFu) >= M;  // int_count[i-1] >= M with uint8_t counters (:57-58,157,172)
    const bool b = (here & 0xFFu) >= M;
    is_start[i] = (b && !a) ? 1u : 0u;
    is_end[i] = (!b && a) ? 1u : 0u;
}

__global__ __launch_bounds__(COV_THREADS) void coverage_emit_kernel(const uint32_t* __restrict__ is_start, const uint32_t* __restrict__ is_end,
                                                                    const uint32_t* __restrict__ start_idx, const uint32_t* __restrict__ end_idx,
                                                                    uint32_t n, uint32_t pos0, uint32_t start_base, uint32_t end_base,
                                                                    uint32_t cap, uint32_t* __restrict__ out_pairs /* {start, end} */) {
    const uint32_t i = blockIdx.x * COV_THREADS + threadIdx.x;
    if (i >= n) return;
    if (is_start[i]) {
        const uint32_t k = start_base + start_idx[i];
        if (k < cap) out_pairs[2 * (size_t)k] = pos0 + i;
    }
    if (is_end[i]) {
        const uint32_t k = end_base + end_idx[i];
        if (k < cap) out_pairs[2 * (size_t)k + 1] = pos0 + i;
    }
}

// {start, end} -> Segment{query_start, len = number of covered positions} (:170-183)
__global__ __launch_bounds__(COV_THREADS) void coverage_finish_kernel(uint32_t* __restrict__ pairs, uint32_t n) {
    const uint32_t i = blockIdx.x * COV_THREADS + threadIdx.x;
    if (i < n) pairs[2 * (size_t)i + 1] -= pairs[2 * (size_t)i];
}

__global__ __launch_bounds__(COV_THREADS) void coverage_range_reset_kernel(uint32_t* range) {
    if (threadIdx.x == 0) { range[0] = 0xFFFFFFFFu; range[1] = 0u; }
}

static inline dim3 cov_grid(uint32_t n) { return dim3((n + COV_THREADS - 1) / COV_THREADS); }

void launch_coverage_add_hsprec(const HspRec* hsps, uint32_t n, uint32_t* diff, uint32_t diff_len, uint32_t* range, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(coverage_add_kernel<HspRec>, cov_grid(n), dim3(COV_THREADS), 0, s, hsps, n, diff, diff_len, range);
}
void launch_coverage_add_pairs(const SegPair16* hsps, uint32_t n, uint32_t* diff, uint32_t diff_len, uint32_t* range, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(coverage_add_kernel<SegPair16>, cov_grid(n), dim3(COV_THREADS), 0, s, hsps, n, diff, diff_len, range);
}
void launch_coverage_range_reset(uint32_t* range, hipStream_t s) {
    hipLaunchKernelGGL(coverage_range_reset_kernel, dim3(1), dim3(64), 0, s, range);
}
void launch_coverage_flags(const uint32_t* diff, const uint32_t* pre, uint32_t n, uint32_t carry_depth, uint32_t M,
                           uint32_t* is_start, uint32_t* is_end, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(coverage_flags_kernel, cov_grid(n), dim3(COV_THREADS), 0, s, diff, pre, n, carry_depth, M, is_start, is_end);
}
void launch_coverage_emit(const uint32_t* is_start, const uint32_t* is_end, const uint32_t* start_idx, const uint32_t* end_idx,
                          uint32_t n, uint32_t pos0, uint32_t start_base, uint32_t end_base, uint32_t cap, uint32_t* out_pairs,
                          hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(coverage_emit_kernel, cov_grid(n), dim3(COV_THREADS), 0, s, is_start, is_end, start_idx, end_idx, n, pos0,
                       start_base, end_base, cap, out_pairs);
}
void launch_coverage_finish(uint32_t* pairs, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(coverage_finish_kernel, cov_grid(n), dim3(COV_THREADS), 0, s, pairs, n);
}

}  // namespace sa
