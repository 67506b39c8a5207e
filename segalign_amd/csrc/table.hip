// table.hip -- seed position table build ON THE DEVICE.
// Replaces the CPU/TBB two-pass counting sort + serial host scan + per-GPU H2D replication of
// GenerateSeedPosTable (common/seed_pos_table.cu:49-109).  Layout kept in HBM:
//   bucket_start[4^k + 1]  exclusive bucket offsets (bucket_start[key+1] == the reference's d_index_table[key])
//   pos_table[num_index]   block-relative seed start positions, ascending inside a bucket
// Position rule (hazard H6): positions start_offset + i*step, i < num_steps, with
//   offset = (span+1) % step, start_offset = step - offset, num_steps = (len - span + offset)/step  (:58-64)
#include "kernels.h"
#include "kmer_dev.h"

namespace sa {

__global__ __launch_bounds__(256) void table_count_kernel(const uint8_t* __restrict__ ref, uint32_t num_steps,
                                                          uint32_t start_offset, uint32_t step, SeedShape sh,
                                                          uint32_t* __restrict__ hist) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < num_steps; i += gridDim.x * blockDim.x) {
        uint32_t key;
        if (kmer_at(ref, start_offset + i * step, sh, key)) atomicAdd(&hist[key], 1u);  // :77-78
    }
}

__global__ __launch_bounds__(256) void table_fill_kernel(const uint8_t* __restrict__ ref, uint32_t num_steps,
                                                         uint32_t start_offset, uint32_t step, SeedShape sh,
                                                         const uint32_t* __restrict__ bucket_start,
                                                         uint32_t* __restrict__ cursor, uint32_t* __restrict__ pos_table) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < num_steps; i += gridDim.x * blockDim.x) {
        uint32_t key;
        uint32_t pos = start_offset + i * step;
        if (kmer_at(ref, pos, sh, key)) {
            uint32_t rank = atomicAdd(&cursor[key], 1u);
            pos_table[bucket_start[key] + rank] = pos;  // :95-96
        }
    }
}

// Canonical order: ascending positions inside each bucket (the reference's order is atomic arrival order, hazard H7;
// no consumer depends on it).  One lane sorts one bucket -- right for the typical occupancy T/4^k of a few entries.
// Buckets above SORT_MAX (low-complexity repeats: poly-A, microsatellites) keep their arrival order: a single lane
// would need O(n^1.3) steps on them and the order cannot influence any result.
constexpr uint32_t SORT_MAX = 512;

__global__ __launch_bounds__(256) void table_sort_buckets_kernel(const uint32_t* __restrict__ bucket_start, uint32_t nkeys,
                                                                 uint32_t* __restrict__ pos_table) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nkeys; k += gridDim.x * blockDim.x) {
        uint32_t b = bucket_start[k], e = bucket_start[k + 1];
        uint32_t n = e - b;
        if (n < 2 || n > SORT_MAX) continue;
        uint32_t* a = pos_table + b;
        // shell sort (gaps n/2, n/4, ... 1); plain insertion sort for the tiny typical bucket
        for (uint32_t gap = n > 8 ? n / 2 : 1; gap > 0; gap /= 2) {
            for (uint32_t i = gap; i < n; i++) {
                uint32_t v = a[i];
                uint32_t j = i;
                while (j >= gap && a[j - gap] > v) { a[j] = a[j - gap]; j -= gap; }
                a[j] = v;
            }
        }
    }
}

static inline int grid_for(uint64_t work_items, int block, int max_blocks = 256 * 16) {
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

void launch_table_count(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh,
                        uint32_t* hist, hipStream_t s) {
    if (num_steps == 0) return;
    hipLaunchKernelGGL(table_count_kernel, dim3(grid_for(num_steps, 256)), dim3(256), 0, s, ref, num_steps, start_offset,
                       step, sh, hist);
}
void launch_table_fill(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh,
                       const uint32_t* bucket_start, uint32_t* cursor, uint32_t* pos_table, hipStream_t s) {
    if (num_steps == 0) return;
    hipLaunchKernelGGL(table_fill_kernel, dim3(grid_for(num_steps, 256)), dim3(256), 0, s, ref, num_steps, start_offset,
                       step, sh, bucket_start, cursor, pos_table);
}
void launch_table_sort_buckets(const uint32_t* bucket_start, uint32_t nkeys, uint32_t* pos_table, hipStream_t s) {
    hipLaunchKernelGGL(table_sort_buckets_kernel, dim3(grid_for(nkeys, 256)), dim3(256), 0, s, bucket_start, nkeys,
                       pos_table);
}

}  // namespace sa
