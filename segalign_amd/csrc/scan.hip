// scan.hip -- exclusive prefix sum of u32 counts (reduce / scan-of-partials / downsweep, 3 launches).
// Replaces thrust::inclusive_scan at src/seed_filter.cu:714 (hit counts) and the serial host scan of the
// 4^k-entry bucket histogram at common/seed_pos_table.cu:23.  HBM traffic: 2 reads + 1 write of the array.
#include "kernels.h"

namespace sa {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;  // consecutive items per thread (two 16-byte loads)
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// returns the exclusive prefix of `v` over the block and the block total (all threads)
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t& total) {
    __shared__ uint64_t wave_sum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint64_t s = wave_sum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__device__ __forceinline__ void load_tile(const uint32_t* __restrict__ in, uint64_t n, uint64_t base, uint32_t v[SCAN_ITEMS]) {
    const uint64_t i0 = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
    if (i0 + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(in + i0) & 15) == 0)) {
        const uint4* p = reinterpret_cast<const uint4*>(in + i0);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) v[j] = (i0 + j < n) ? in[i0 + j] : 0u;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                   uint64_t* __restrict__ block_sums) {
    uint32_t v[SCAN_ITEMS];
    load_tile(in, n, (uint64_t)blockIdx.x * SCAN_TILE, v);
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += v[j];
    uint64_t total;
    block_exclusive_scan(s, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one block: exclusive scan of the per-tile sums in place; total -> *total_out
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials_kernel(uint64_t* __restrict__ block_sums, uint32_t nblocks,
                                                                     uint64_t* __restrict__ total_out) {
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += SCAN_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < nblocks ? block_sums[i] : 0;
        uint64_t total;
        uint64_t ex = block_exclusive_scan(v, total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

template <typename OutT>
__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                      const uint64_t* __restrict__ block_offs,
                                                                      const uint64_t* __restrict__ total_in,
                                                                      OutT* __restrict__ out) {
    uint32_t v[SCAN_ITEMS];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    load_tile(in, n, base, v);
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += v[j];
    uint64_t total;
    uint64_t run = block_exclusive_scan(s, total) + block_offs[blockIdx.x];
    const uint64_t i0 = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (i0 + j < n) out[i0 + j] = (OutT)run;
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (OutT)(*total_in);
}

size_t scan_temp_bytes(uint64_t n) {
    uint64_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(nblocks + 2) * sizeof(uint64_t);
}

template <typename OutT>
static void scan_impl(const uint32_t* in, OutT* out, uint64_t n, void* temp, hipStream_t s) {
    uint64_t* partial = reinterpret_cast<uint64_t*>(temp);
    uint32_t nblocks = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nblocks == 0) nblocks = 1;  // n == 0: still writes out[0] = 0
    uint64_t* total = partial + nblocks;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, s, in, n, partial);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, partial, nblocks, total);
    hipLaunchKernelGGL(scan_downsweep_kernel<OutT>, dim3(nblocks), dim3(SCAN_THREADS), 0, s, in, n, partial, total, out);
}

void launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* temp, hipStream_t s) {
    scan_impl<uint32_t>(in, out, n, temp, s);
}
void launch_exclusive_scan_u64(const uint32_t* in, uint64_t* out, uint64_t n, void* temp, hipStream_t s) {
    scan_impl<uint64_t>(in, out, n, temp, s);
}

}  // namespace sa
