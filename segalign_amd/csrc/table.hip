// table.hip -- seed position table build ON THE DEVICE.
// Replaces the CPU/TBB two-pass counting sort + serial host scan + per-GPU H2D replication of
// GenerateSeedPosTable (common/seed_pos_table.cu:49-109).
//
// Two builds of the same table:
//   * PARTITION build (seed weight 9..12, i.e. keys of 18..24 bits: the reference's default 12of19) -- an MSD radix partition
//     staged through LDS, no global atomic per position and no scattered 4-byte writes:
//       table_keys_kernel       k-mer of every indexed position (coalesced code reads) -> keys[]; coarse histogram (top 12 key bits)
//                               privatised in LDS, one global atomic per non-empty bin and 64 Ki-position tile
//       (scan of the 4096 coarse counts = the table offsets of the coarse partitions)
//       table_partition_kernel  twice: by the top 6 bits, then -- on input already grouped by those -- by the top 12 bits.  A tile of
//                               8192 elements is counted in LDS, reserves its share of every bin with ONE global atomic per
//                               (tile, non-empty bin), is staged in LDS bin by bin and written out in runs: consecutive lanes
//                               write consecutive addresses ("LDS bucket staging")
//       table_finish_kernel     one workgroup per coarse partition (~20 k positions): fine histogram of the low key bits in LDS,
//                               LDS scan = the partition's slice of bucket_start, positions placed and every bucket sorted
//                               in LDS, the slice of pos_table written in one coalesced sweep
//     Traffic: 1 T (codes) + 4 T (keys) + 3 x 8 T_valid (pairs) + 4 T_valid (positions) + 4^k x 4 -- ~3 GB per 100 Mbp block
//     against 15.6 GB of line read-modify-writes of the scatter below (profiles/r02/traffic.json).
//   * ATOMIC build (any other weight, e.g. 14of22's 28-bit keys): histogram and scatter with one global atomic per position.
// Layout kept in HBM:
//   bucket_start[4^k + 1]  exclusive bucket offsets (bucket_start[key+1] == the reference's d_index_table[key])
//   pos_table[num_index]   block-relative seed start positions, ascending inside a bucket
// Position rule (hazard H6): positions start_offset + i*step, i < num_steps, with
//   offset = (span+1) % step, start_offset = step - offset, num_steps = (len - span + offset)/step  (:58-64)
#include <string.h>

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

// min_n: buckets below this size are already in order (the partition build sorts them in LDS); part_unsorted / part_shift: a coarse
// partition that did not fit the finish kernel's LDS left all its buckets in arrival order
__global__ __launch_bounds__(256) void table_sort_buckets_kernel(const uint32_t* __restrict__ bucket_start, uint32_t nkeys,
                                                                 uint32_t* __restrict__ pos_table, uint32_t min_n,
                                                                 const uint8_t* __restrict__ part_unsorted, uint32_t part_shift) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nkeys; k += gridDim.x * blockDim.x) {
        uint32_t b = bucket_start[k], e = bucket_start[k + 1];
        uint32_t n = e - b;
        if (n < 2 || n > SORT_MAX) continue;
        if (n < min_n && !(part_unsorted && part_unsorted[k >> part_shift])) continue;
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
                       pos_table, 0u, (const uint8_t*)nullptr, 0u);
}

// ---------------------------------------------------------------------------------------------------------------------
// PARTITION build
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t TB_INVALID = 0xFFFFFFFFu;
constexpr int TB_KEYS_THREADS = 1024;
constexpr int TB_KEYS_PER = 64;                       // positions per thread: 64 Ki positions per tile
constexpr int TB_COARSE_BITS = 12;                    // coarse partitions = top 12 key bits
constexpr int TB_PART_THREADS = 1024;
constexpr int TB_PART_TILE = 8192;                    // elements per partition tile (64 KB of staged pairs)
constexpr uint32_t TB_FIN_CAP = 22528;                // positions a coarse partition may hold and still be finished in LDS (88 KB)
constexpr uint32_t TB_FIN_CAP_FINE = 8192;            // ... a fine partition of the three-level build (keys above 24 bits)

// k-mers -> keys[], coarse histogram (top TB_COARSE_BITS bits of the key) privatised in LDS
__global__ __launch_bounds__(TB_KEYS_THREADS) void table_keys_kernel(const uint8_t* __restrict__ ref, uint32_t num_steps, uint32_t start_offset,
                                                                     uint32_t step, SeedShape sh, int low_bits, uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ coarse_hist) {
    __shared__ uint32_t s_h[1 << TB_COARSE_BITS];
    for (int i = threadIdx.x; i < (1 << TB_COARSE_BITS); i += TB_KEYS_THREADS) s_h[i] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)blockIdx.x * (TB_KEYS_THREADS * TB_KEYS_PER);
#pragma unroll 4
    for (int j = 0; j < TB_KEYS_PER; j++) {
        const uint64_t i = tile0 + (uint64_t)j * TB_KEYS_THREADS + threadIdx.x;  // consecutive lanes = consecutive positions
        if (i >= num_steps) break;
        uint32_t key;
        const bool ok = kmer_at(ref, start_offset + (uint32_t)i * step, sh, key);  // :77
        keys[i] = ok ? key : TB_INVALID;
        if (ok) atomicAdd(&s_h[key >> low_bits], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << TB_COARSE_BITS); i += TB_KEYS_THREADS)
        if (s_h[i]) atomicAdd(&coarse_hist[i], s_h[i]);
}

// cursor[b] = start of bin b of a pass with `bits` bins bits (bin = coarse partition >> (TB_COARSE_BITS - bits))
__global__ void table_cursor_init_kernel(const uint32_t* __restrict__ part_start, int bits, uint32_t* __restrict__ cursor) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < (1u << bits)) cursor[b] = part_start[b << (TB_COARSE_BITS - bits)];
}

// One partition pass.  FROM_KEYS: the input is keys[] indexed by position step (pos = start_offset + i * step, invalid entries
// skipped); otherwise (key, pos) pairs.  bin = key >> shift, nbins = 1 << bits (<= 4096).  Dynamic LDS: keys + positions of the
// tile (2 x TB_PART_TILE dwords) + per-bin count / local offset / global base (3 x nbins dwords).
template <bool FROM_KEYS>
__global__ __launch_bounds__(TB_PART_THREADS) void table_partition_kernel(const uint32_t* __restrict__ in_key, const uint32_t* __restrict__ in_pos,
                                                                          uint32_t n, uint32_t start_offset, uint32_t step, int shift, int bits,
                                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ out_key,
                                                                          uint32_t* __restrict__ out_pos) {
    extern __shared__ uint32_t s_dyn[];
    const uint32_t nbins = 1u << bits;
    uint32_t* s_key = s_dyn;
    uint32_t* s_pos = s_dyn + TB_PART_TILE;
    uint32_t* s_cnt = s_dyn + 2 * TB_PART_TILE;
    uint32_t* s_loc = s_cnt + nbins;
    uint32_t* s_base = s_loc + nbins;
    __shared__ uint32_t s_wave[TB_PART_THREADS / 64];
    constexpr int PER = TB_PART_TILE / TB_PART_THREADS;
    for (uint32_t i = threadIdx.x; i < nbins; i += TB_PART_THREADS) s_cnt[i] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)blockIdx.x * TB_PART_TILE;
    uint32_t key[PER], pos[PER], rank[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t i = tile0 + (uint64_t)j * TB_PART_THREADS + threadIdx.x;
        key[j] = TB_INVALID;
        pos[j] = 0;
        rank[j] = 0;
        if (i < n) {
            key[j] = in_key[i];
            pos[j] = FROM_KEYS ? start_offset + (uint32_t)i * step : in_pos[i];
        }
        if (key[j] != TB_INVALID) rank[j] = atomicAdd(&s_cnt[key[j] >> shift], 1u);  // rank inside (tile, bin): arrival order
    }
    __syncthreads();
    // exclusive scan of the bin counts (nbins <= 4096 = 4 per thread) + one global reservation per non-empty bin
    {
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            c[q] = b < nbins ? s_cnt[b] : 0u;
            sum += c[q];
        }
        uint32_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; w++) base += s_wave[w];
        uint32_t run = base + inc - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            if (b < nbins) {
                s_loc[b] = run;
                s_base[b] = c[q] ? atomicAdd(&cursor[b], c[q]) : 0u;
                run += c[q];
            }
        }
    }
    __syncthreads();
    // stage bin by bin
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (key[j] != TB_INVALID) {
            const uint32_t slot = s_loc[key[j] >> shift] + rank[j];
            s_key[slot] = key[j];
            s_pos[slot] = pos[j];
        }
    __syncthreads();
    // write out: consecutive slots of a bin go to consecutive addresses
    uint32_t total = 0;
    for (int w = 0; w < TB_PART_THREADS / 64; w++) total += s_wave[w];
    for (uint32_t slot = threadIdx.x; slot < total; slot += TB_PART_THREADS) {
        const uint32_t k = s_key[slot], b = k >> shift;
        const uint32_t dst = s_base[b] + (slot - s_loc[b]);
        out_key[dst] = k;
        out_pos[dst] = s_pos[slot];
    }
}

// ---- keys of more than 24 bits (14of22: 28): a THIRD partition level ---------------------------------------------------------------
// The finish kernel keeps a partition's fine histogram in LDS (<= 4096 bins), so 28-bit keys need 2^16 or more partitions.  After the
// two passes above the pairs are grouped by their top 12 key bits; each of those 4096 groups is cut once more by its next 6 bits:
//   table_subhist_kernel      one workgroup per 12-bit group: histogram of the next TB_FINE_SUB bits in LDS -> hist18[group * 64 + b]
//   (scan of the 2^18 counts = the offsets of the 2^18 fine partitions)
//   table_partition_rel_kernel  the partition pass again, with bins counted RELATIVE to the tile's first 12-bit group: a tile of 8192
//                             consecutive pairs spans a few groups (the input is grouped by them), i.e. a few x 64 bins; a tile that
//                             spans 64 groups or more -- a tiny or wildly skewed table -- raises a flag and the host falls back to the
//                             atomic build
// and the finish kernel then runs once per FINE partition (~300 positions of a 100 Mbp block, 1024 bins of 10 low bits).
constexpr int TB_FINE_SUB = 6;
constexpr int TB_FINE_BITS = TB_COARSE_BITS + TB_FINE_SUB;  // 18

__global__ __launch_bounds__(256) void table_subhist_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ part_start, int shift,
                                                            uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[1 << TB_FINE_SUB];
    if (threadIdx.x < (1 << TB_FINE_SUB)) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = part_start[blockIdx.x], hi = part_start[blockIdx.x + 1];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&s_h[(key[i] >> shift) & ((1u << TB_FINE_SUB) - 1u)], 1u);
    __syncthreads();
    if (threadIdx.x < (1 << TB_FINE_SUB)) hist[((size_t)blockIdx.x << TB_FINE_SUB) + threadIdx.x] = s_h[threadIdx.x];
}

__global__ __launch_bounds__(256) void table_copy_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}

// bin = key >> shift (TB_FINE_BITS bits), counted relative to the first bin of the tile's first 12-bit group; at most 4096 relative bins
__global__ __launch_bounds__(TB_PART_THREADS) void table_partition_rel_kernel(const uint32_t* __restrict__ in_key, const uint32_t* __restrict__ in_pos,
                                                                              uint32_t n, int shift, uint32_t* __restrict__ cursor,
                                                                              uint32_t* __restrict__ out_key, uint32_t* __restrict__ out_pos,
                                                                              uint32_t* __restrict__ err) {
    extern __shared__ uint32_t s_dyn[];
    constexpr uint32_t NB = 4096;
    uint32_t* s_key = s_dyn;
    uint32_t* s_pos = s_dyn + TB_PART_TILE;
    uint32_t* s_cnt = s_dyn + 2 * TB_PART_TILE;
    uint32_t* s_loc = s_cnt + NB;
    uint32_t* s_base = s_loc + NB;
    __shared__ uint32_t s_wave[TB_PART_THREADS / 64];
    __shared__ uint32_t s_bad;
    constexpr int PER = TB_PART_TILE / TB_PART_THREADS;
    for (uint32_t i = threadIdx.x; i < NB; i += TB_PART_THREADS) s_cnt[i] = 0;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)blockIdx.x * TB_PART_TILE;
    const uint32_t bin0 = ((in_key[tile0] >> shift) >> TB_FINE_SUB) << TB_FINE_SUB;  // first bin of the tile's first 12-bit group
    uint32_t key[PER], pos[PER], rank[PER], rel[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t i = tile0 + (uint64_t)j * TB_PART_THREADS + threadIdx.x;
        key[j] = TB_INVALID;
        pos[j] = rank[j] = rel[j] = 0;
        if (i < n) {
            key[j] = in_key[i];
            pos[j] = in_pos[i];
            rel[j] = (key[j] >> shift) - bin0;
            if (rel[j] >= NB) { s_bad = 1; key[j] = TB_INVALID; }
            else rank[j] = atomicAdd(&s_cnt[rel[j]], 1u);
        }
    }
    __syncthreads();
    if (s_bad) {  // (the host rebuilds the table with the atomic build)
        if (threadIdx.x == 0) *err = 1u;
        return;
    }
    {
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { c[q] = s_cnt[threadIdx.x * 4 + q]; sum += c[q]; }
        uint32_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; w++) base += s_wave[w];
        uint32_t run = base + inc - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            s_loc[b] = run;
            s_base[b] = c[q] ? atomicAdd(&cursor[bin0 + b], c[q]) : 0u;
            run += c[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (key[j] != TB_INVALID) {
            const uint32_t slot = s_loc[rel[j]] + rank[j];
            s_key[slot] = key[j];
            s_pos[slot] = pos[j];
        }
    __syncthreads();
    uint32_t total = 0;
    for (int w = 0; w < TB_PART_THREADS / 64; w++) total += s_wave[w];
    for (uint32_t slot = threadIdx.x; slot < total; slot += TB_PART_THREADS) {
        const uint32_t k = s_key[slot], b = (k >> shift) - bin0;
        const uint32_t dst = s_base[b] + (slot - s_loc[b]);
        out_key[dst] = k;
        out_pos[dst] = s_pos[slot];
    }
}

// One workgroup per coarse partition: fine histogram (low key bits) in LDS -> the partition's slice of bucket_start; positions
// placed bucket by bucket in LDS, every bucket sorted (ascending positions: the canonical order, hazard H7), one coalesced write.
__global__ __launch_bounds__(1024) void table_finish_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ pos,
                                                            const uint32_t* __restrict__ part_start, int low_bits,
                                                            uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ pos_table,
                                                            uint8_t* __restrict__ part_unsorted, uint32_t fin_cap) {
    extern __shared__ uint32_t s_dyn[];  // [nlow] offsets / cursors, [nlow] bucket sizes, [fin_cap] positions
    __shared__ uint32_t s_wave[16];
    const uint32_t nlow = 1u << low_bits, lowmask = nlow - 1u;
    uint32_t* s_off = s_dyn;
    uint32_t* s_n = s_dyn + nlow;
    uint32_t* s_out = s_dyn + 2 * nlow;
    const uint32_t p = blockIdx.x;
    const uint32_t lo = part_start[p], m = part_start[p + 1] - lo;
    for (uint32_t i = threadIdx.x; i < nlow; i += blockDim.x) s_off[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) atomicAdd(&s_off[key[lo + i] & lowmask], 1u);
    __syncthreads();
    // exclusive scan over nlow (<= 4096) counts: 4 per thread
    {
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            c[q] = b < nlow ? s_off[b] : 0u;
            sum += c[q];
        }
        uint32_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; w++) base += s_wave[w];
        uint32_t run = base + inc - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = threadIdx.x * 4 + q;
            if (b < nlow) {
                s_off[b] = run;
                s_n[b] = c[q];
                bucket_start[((size_t)p << low_bits) + b] = lo + run;  // exclusive offsets (bucket_start[key + 1] = d_index_table[key], :103)
                run += c[q];
            }
        }
    }
    __syncthreads();
    const bool fits = m <= fin_cap;
    if (threadIdx.x == 0) part_unsorted[p] = fits ? 0 : 1;
    // place (cursor = s_off advanced; the bucket starts are recovered from s_off - s_n afterwards)
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t slot = atomicAdd(&s_off[key[lo + i] & lowmask], 1u);
        if (fits) s_out[slot] = pos[lo + i];
        else pos_table[lo + slot] = pos[lo + i];  // (a partition too large for LDS: placed directly, sorted by the global pass)
    }
    __syncthreads();
    if (!fits) return;
    // sort every bucket ascending (a handful of entries: insertion sort by one lane; large buckets keep arrival order here and
    // are sorted by the global pass, which looks at buckets of >= 33 entries)
    for (uint32_t b = threadIdx.x; b < nlow; b += blockDim.x) {
        const uint32_t n = s_n[b];
        if (n < 2 || n > 32) continue;
        uint32_t* a = s_out + (s_off[b] - n);
        for (uint32_t i = 1; i < n; i++) {
            const uint32_t v = a[i];
            uint32_t j = i;
            while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; j--; }
            a[j] = v;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) pos_table[lo + i] = s_out[i];
}

// The partition build stages its tiles in up to ~120 KB of dynamic LDS; a device that cannot grant that (anything but gfx950's 160 KB)
// keeps the atomic build.  Asked once per process and device.
static bool lds_budget_ok() {
    static int cached[64] = {0};  // 0 unknown, 1 ok, -1 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (cached[dev] == 0) {
        const int want = 160 * 1024 - 256;
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&table_partition_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(&table_partition_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(&table_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        // (the runtime accepts the attribute on devices that cannot honour it; ask the device as well)
        int max_lds = 0;
        hipDeviceProp_t prop;
        const bool have_prop = hipGetDeviceProperties(&prop, dev) == hipSuccess;
        if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); max_lds = 0; }
        const bool gfx950 = have_prop && strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        ok = ok && (max_lds >= want || gfx950);
        cached[dev] = ok ? 1 : -1;
    }
    return cached[dev] == 1;
}
bool table_partition_build_supported(int weight) { return weight >= 9 && weight <= 14 && lds_budget_ok(); }
// scratch of the third partition level (keys above 24 bits), in dwords: hist | part_start (+1) | cursor | unsorted flags | error flag
size_t table_partition_fine_words(int weight) {
    if (2 * weight <= 2 * TB_COARSE_BITS) return 0;
    const size_t n = (size_t)1 << TB_FINE_BITS;
    return 3 * n + 1 + n / 4 + 16;
}
size_t table_partition_part_start_words() { return (1u << TB_COARSE_BITS) + 1; }

void launch_table_keys(const uint8_t* ref, uint32_t num_steps, uint32_t start_offset, uint32_t step, SeedShape sh, uint32_t* keys,
                       uint32_t* coarse_hist, hipStream_t s) {
    if (num_steps == 0) return;
    const int low_bits = 2 * sh.weight - TB_COARSE_BITS;
    const uint32_t tile = TB_KEYS_THREADS * TB_KEYS_PER;
    hipLaunchKernelGGL(table_keys_kernel, dim3((num_steps + tile - 1) / tile), dim3(TB_KEYS_THREADS), 0, s, ref, num_steps, start_offset, step,
                       sh, low_bits, keys, coarse_hist);
}

// keys[] -> pairs A (top 6 bits) -> pairs B (top 12 bits) -> bucket_start + pos_table.  part_start = exclusive scan of the coarse
// histogram (4097 words); cursor: 4096 words of scratch; part_unsorted: 4096 bytes.
void launch_table_partition_build(const uint32_t* keys, uint32_t num_steps, uint32_t start_offset, uint32_t step, int weight,
                                  const uint32_t* part_start, uint32_t num_index, uint32_t* cursor, uint32_t* key_a, uint32_t* pos_a,
                                  uint32_t* key_b, uint32_t* pos_b, uint8_t* part_unsorted, uint32_t* bucket_start, uint32_t* pos_table,
                                  uint32_t* fine /* table_partition_fine_words(weight) dwords */, void* scan_tmp /* scan_temp_bytes(2^18) */,
                                  uint32_t** err_flag /* out: device flag, != 0 after the build = fall back to the atomic build */, hipStream_t s) {
    const int keybits = 2 * weight, low_bits = keybits - TB_COARSE_BITS;
    if (err_flag) *err_flag = nullptr;
    const uint32_t nkeys = 1u << keybits;
    // (dynamic LDS above 64 KB has to be announced per kernel and device)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&table_partition_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&table_partition_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&table_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    // pass 1: top 6 bits, straight from the keys
    hipLaunchKernelGGL(table_cursor_init_kernel, dim3(1), dim3(64), 0, s, part_start, 6, cursor);
    if (num_steps)
        hipLaunchKernelGGL((table_partition_kernel<true>), dim3((num_steps + TB_PART_TILE - 1) / TB_PART_TILE), dim3(TB_PART_THREADS),
                           (2 * TB_PART_TILE + 3 * 64) * sizeof(uint32_t), s, keys, (const uint32_t*)nullptr, num_steps, start_offset, step,
                           keybits - 6, 6, cursor, key_a, pos_a);
    // pass 2: top 12 bits, input grouped by the top 6
    hipLaunchKernelGGL(table_cursor_init_kernel, dim3(16), dim3(256), 0, s, part_start, TB_COARSE_BITS, cursor);
    if (num_index)
        hipLaunchKernelGGL((table_partition_kernel<false>), dim3((num_index + TB_PART_TILE - 1) / TB_PART_TILE), dim3(TB_PART_THREADS),
                           (2 * TB_PART_TILE + 3 * (1u << TB_COARSE_BITS)) * sizeof(uint32_t), s, key_a, pos_a, num_index, 0u, 0u, low_bits,
                           TB_COARSE_BITS, cursor, key_b, pos_b);
    if (keybits > 2 * TB_COARSE_BITS) {
        // third level: 2^18 fine partitions (see above); the pairs go back into the A buffers
        const uint32_t nfine = 1u << TB_FINE_BITS;
        const int fine_low = keybits - TB_FINE_BITS;
        uint32_t* hist18 = fine;
        uint32_t* start18 = hist18 + nfine;
        uint32_t* cursor18 = start18 + nfine + 1;
        uint8_t* unsorted18 = reinterpret_cast<uint8_t*>(cursor18 + nfine);
        uint32_t* err = reinterpret_cast<uint32_t*>(unsorted18 + nfine);
        if (err_flag) *err_flag = err;
        (void)hipMemsetAsync(err, 0, sizeof(uint32_t), s);
        hipLaunchKernelGGL(table_subhist_kernel, dim3(1u << TB_COARSE_BITS), dim3(256), 0, s, key_b, part_start, fine_low, hist18);
        launch_exclusive_scan_u32(hist18, start18, nfine, scan_tmp, s);
        hipLaunchKernelGGL(table_copy_kernel, dim3(256), dim3(256), 0, s, start18, cursor18, nfine);
        if (num_index)
            hipLaunchKernelGGL(table_partition_rel_kernel, dim3((num_index + TB_PART_TILE - 1) / TB_PART_TILE), dim3(TB_PART_THREADS),
                               (2 * TB_PART_TILE + 3 * 4096) * sizeof(uint32_t), s, key_b, pos_b, num_index, fine_low, cursor18, key_a, pos_a, err);
        // a fine partition holds a few hundred positions: 8192 of them in LDS (a partition above that is placed directly)
        hipLaunchKernelGGL(table_finish_kernel, dim3(nfine), dim3(256), (2 * (1u << fine_low) + TB_FIN_CAP_FINE) * sizeof(uint32_t), s, key_a, pos_a,
                           start18, fine_low, bucket_start, pos_table, unsorted18, TB_FIN_CAP_FINE);
        (void)hipMemcpyAsync(bucket_start + nkeys, part_start + (1u << TB_COARSE_BITS), sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        hipLaunchKernelGGL(table_sort_buckets_kernel, dim3(grid_for(nkeys, 256)), dim3(256), 0, s, bucket_start, nkeys, pos_table, 33u, unsorted18,
                           (uint32_t)fine_low);
        return;
    }
    // finish: one workgroup per coarse partition
    hipLaunchKernelGGL(table_finish_kernel, dim3(1u << TB_COARSE_BITS), dim3(1024), (2 * (1u << low_bits) + TB_FIN_CAP) * sizeof(uint32_t), s, key_b,
                       pos_b, part_start, low_bits, bucket_start, pos_table, part_unsorted, TB_FIN_CAP);
    // bucket_start[nkeys] = num_index; buckets of more than 32 entries (and the partitions that did not fit) get the global sort
    (void)hipMemcpyAsync(bucket_start + nkeys, part_start + (1u << TB_COARSE_BITS), sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(table_sort_buckets_kernel, dim3(grid_for(nkeys, 256)), dim3(256), 0, s, bucket_start, nkeys, pos_table, 33u,
                       part_unsorted, (uint32_t)low_bits);
}

}  // namespace sa
