// dedup.hip -- ordering and de-duplication of the surviving HSPs.
// Replaces thrust::stable_sort(hspComp) -> thrust::unique_copy(hspEqual) -> thrust::stable_sort(hspCompLastz)
// (src/seed_filter.cu:776-782) and the five-step chain of the repeat masker (repeat_masker_src/seed_filter.cu:819-831).
// The reference runs the chain once per iteration; here every record carries its iteration (`seg`) as the most
// significant sort key and as an extra inequality in the unique predicate, so one pass over all iterations of a
// call yields exactly the concatenation the reference builds at :811-822.
// Sorting uses rocPRIM's merge sort (AMD's native device primitive -- what thrust::stable_sort lowers to on ROCm);
// survivors are orders of magnitude fewer than hits, so this stage is launch-latency, not bandwidth, bound.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace sa {

// ---- comparators (all compare seg first) --------------------------------------------------------------------------
struct LessDiag {  // hspComp, seed_filter.cu:54-80: (diag as wrapped u32 [H8], ref_start, len, score desc)
    __host__ __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.seg != y.seg) return x.seg < y.seg;
        const uint32_t dx = x.ref_start - x.query_start, dy = y.ref_start - y.query_start;
        if (dx != dy) return dx < dy;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        return x.score > y.score;
    }
};
struct LessLastz {  // hspCompLastz, seed_filter.cu:82-108: (query_start, ref_start, len, score desc)
    __host__ __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.seg != y.seg) return x.seg < y.seg;
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        return x.score > y.score;
    }
};
struct LessRmFirst {  // repeat masker hspComp, rm :109-135: (query_start, len desc, ref_start, score desc)
    __host__ __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.seg != y.seg) return x.seg < y.seg;
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.len != y.len) return x.len > y.len;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        return x.score > y.score;
    }
};
struct LessRmDiag {  // hspDiagComp, rm :52-78: (diag, ref_start, query_start, score desc)
    __host__ __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.seg != y.seg) return x.seg < y.seg;
        const uint32_t dx = x.ref_start - x.query_start, dy = y.ref_start - y.query_start;
        if (dx != dy) return dx < dy;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        return x.score > y.score;
    }
};
struct LessRmFinal {  // hspFinalComp, rm :87-107: (query_start, score desc, ref_start desc)
    __host__ __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.seg != y.seg) return x.seg < y.seg;
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.score != y.score) return x.score > y.score;
        return x.ref_start > y.ref_start;
    }
};

// hspEqual, seed_filter.cu:47-52 (== hspDiagEqual, rm :45-50): same diagonal and one interval contains the other
__device__ __forceinline__ bool hsp_contained(const HspRec& x, const HspRec& y) {
    return ((uint32_t)(x.ref_start - x.query_start) == (uint32_t)(y.ref_start - y.query_start)) &&
           (((x.ref_start >= y.ref_start) && ((uint32_t)(x.ref_start + x.len) <= (uint32_t)(y.ref_start + y.len))) ||
            ((y.ref_start >= x.ref_start) && ((uint32_t)(y.ref_start + y.len) <= (uint32_t)(x.ref_start + x.len))));
}
__device__ __forceinline__ bool hsp_same(const HspRec& x, const HspRec& y) {  // rm hspEqual :80-85
    return x.ref_start == y.ref_start && x.query_start == y.query_start && x.len == y.len && x.score == y.score;
}

template <class Less>
static void sort_impl(const HspRec* in, HspRec* out, size_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    size_t bytes = temp_bytes;
    (void)rocprim::merge_sort(temp, bytes, in, out, n, Less(), s);
}

size_t sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::merge_sort(nullptr, bytes, (const HspRec*)nullptr, (HspRec*)nullptr, n, LessDiag(), (hipStream_t)0);
    return bytes + 256;
}

void launch_sort(const HspRec* in, HspRec* out, size_t n, SortOrder order, void* temp, size_t temp_bytes, hipStream_t s) {
    if (n == 0) return;
    switch (order) {
        case ORDER_DIAG: sort_impl<LessDiag>(in, out, n, temp, temp_bytes, s); break;
        case ORDER_LASTZ: sort_impl<LessLastz>(in, out, n, temp, temp_bytes, s); break;
        case ORDER_RM_FIRST: sort_impl<LessRmFirst>(in, out, n, temp, temp_bytes, s); break;
        case ORDER_RM_DIAG: sort_impl<LessRmDiag>(in, out, n, temp, temp_bytes, s); break;
        case ORDER_RM_FINAL: sort_impl<LessRmFinal>(in, out, n, temp, temp_bytes, s); break;
    }
}

// ---- adjacent-pair unique, order preserving (thrust::unique_copy device semantics, hazard H3) --------------------
// keep[i] = i == 0 || seg differs || !pred(in[i-1], in[i]).  One workgroup walks the (small) array in tiles and
// carries the running output offset; ballot + popcount gives the in-tile rank.
constexpr int UNQ_THREADS = 1024;
constexpr int UNQ_ITEMS = 8;  // consecutive records per thread per tile

__global__ __launch_bounds__(UNQ_THREADS) void unique_kernel(const HspRec* __restrict__ in, HspRec* __restrict__ out,
                                                             uint32_t n, int exact, uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wave_cnt[UNQ_THREADS / 64];
    __shared__ uint32_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += UNQ_THREADS * UNQ_ITEMS) {
        const uint32_t i0 = base + threadIdx.x * UNQ_ITEMS;
        HspRec rec[UNQ_ITEMS];
        uint32_t keepmask = 0;
        HspRec prev;
        prev.ref_start = prev.query_start = prev.len = 0; prev.score = 0; prev.seg = 0xFFFFFFFFu;
        if (i0 > 0 && i0 < n) prev = in[i0 - 1];
#pragma unroll
        for (int j = 0; j < UNQ_ITEMS; j++) {
            const uint32_t i = i0 + j;
            if (i < n) {
                rec[j] = in[i];
                const bool keep = (i == 0) || (prev.seg != rec[j].seg) ||
                                  !(exact ? hsp_same(prev, rec[j]) : hsp_contained(prev, rec[j]));
                keepmask |= keep ? (1u << j) : 0u;
                prev = rec[j];
            }
        }
        const uint32_t mine = (uint32_t)__builtin_popcount(keepmask);
        // wave inclusive scan of the per-thread counts, then block offsets
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_cnt[wave] = inc;
        __syncthreads();
        uint32_t off = carry + inc - mine;
        for (int w = 0; w < wave; w++) off += wave_cnt[w];
#pragma unroll
        for (int j = 0; j < UNQ_ITEMS; j++)
            if ((keepmask >> j) & 1u) out[off++] = rec[j];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < UNQ_THREADS / 64; w++) t += wave_cnt[w];
            carry += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = carry;
}

__global__ __launch_bounds__(256) void strip_kernel(const HspRec* __restrict__ in, uint32_t n, uint4* __restrict__ out,
                                                    uint32_t* __restrict__ out_seg) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const HspRec r = in[i];
        out[i] = make_uint4(r.ref_start, r.query_start, r.len, (uint32_t)r.score);
        if (out_seg) out_seg[i] = r.seg;
    }
}

// ---- the whole chain for a SMALL survivor set in one workgroup -----------------------------------------------------
// After the chain shortcut a call leaves a few hundred survivors; three library sorts + unique + strip then cost ~65 us of
// pure launch latency and one extra host sync.  For n <= DEDUP_SMALL_MAX one workgroup does
//   stable sort by LessDiag -> adjacent-pair unique (hsp_contained, H3) -> stable sort by LessLastz -> 16-byte records
// in LDS (src/seed_filter.cu:776-782, all iterations -- and all chunks of a multi-chunk call -- at once through `seg`).
constexpr int DEDUP_SMALL_MAX = 1024;  // one record per thread in the unique step
constexpr int DEDUP_SMALL_THREADS = 1024;

constexpr int DEDUP_SMALL_SEGS = 16;  // distinct segment ids (reference iterations x chunks of a call) the small path handles

// Stable sort of a[0..n) by `Less` (segment id first) in LDS, result back in a[]; tmp[] is scratch of the same size.
// Two steps: (1) group the records by segment (counting sort over <= DEDUP_SMALL_SEGS ids), (2) RANK sort inside every
// segment: rank(i) = #{j in the segment : rec[j] < rec[i]} + #{j < i : rec[j] equivalent to rec[i]} is exactly the
// position a stable sort gives element i.  O(n * segment size) comparisons spread over the workgroup (the threads are
// split into `parts` groups that each count over a slice of the segment), no barrier chain.  Ends with a barrier.
template <class Less>
__device__ __forceinline__ void seg_rank_sort_lds(HspRec* __restrict__ a, HspRec* __restrict__ tmp, uint32_t n,
                                                  uint32_t* __restrict__ s_rank, uint32_t* __restrict__ s_seg /*[2*SEGS+1]*/) {
    Less less;
    uint32_t* s_cnt = s_seg;                       // [SEGS] records per segment, then running cursors
    uint32_t* s_beg = s_seg + DEDUP_SMALL_SEGS;    // [SEGS+1] first slot of every segment
    if (threadIdx.x < DEDUP_SMALL_SEGS) s_cnt[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) s_rank[k] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&s_cnt[a[k].seg & (DEDUP_SMALL_SEGS - 1)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int g = 0; g < DEDUP_SMALL_SEGS; g++) { s_beg[g] = run; run += s_cnt[g]; s_cnt[g] = s_beg[g]; }
        s_beg[DEDUP_SMALL_SEGS] = run;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        const HspRec r = a[k];
        tmp[atomicAdd(&s_cnt[r.seg & (DEDUP_SMALL_SEGS - 1)], 1u)] = r;  // order inside a segment is arbitrary here:
    }                                                                   // records that compare equal are identical
    __syncthreads();
    const uint32_t nceil = (n + 63u) & ~63u;                  // elements padded to whole waves
    const uint32_t parts = nceil ? blockDim.x / nceil : 1u;    // >= 1 because n <= blockDim.x
    const uint32_t i = threadIdx.x % (nceil ? nceil : 1u), p = threadIdx.x / (nceil ? nceil : 1u);
    if (i < n && p < parts) {
        const HspRec me = tmp[i];
        const uint32_t g = me.seg & (DEDUP_SMALL_SEGS - 1);
        const uint32_t b0 = s_beg[g], m = s_beg[g + 1] - b0;
        const uint32_t j0 = b0 + (uint32_t)((uint64_t)m * p / parts), j1 = b0 + (uint32_t)((uint64_t)m * (p + 1) / parts);
        uint32_t rank = p == 0 ? b0 : 0u;
        for (uint32_t j = j0; j < j1; j++) {
            const HspRec o = tmp[j];
            rank += (less(o, me) || (j < i && !less(me, o))) ? 1u : 0u;
        }
        atomicAdd(&s_rank[i], rank);
    }
    __syncthreads();
    if (threadIdx.x < n) a[s_rank[threadIdx.x]] = tmp[threadIdx.x];
    __syncthreads();
}

__global__ __launch_bounds__(DEDUP_SMALL_THREADS) void dedup_small_kernel(const HspRec* __restrict__ in, uint32_t n,
                                                                          uint4* __restrict__ out, uint32_t* __restrict__ out_seg,
                                                                          uint32_t* __restrict__ out_count) {
    __shared__ HspRec s_a[DEDUP_SMALL_MAX];
    __shared__ HspRec s_b[DEDUP_SMALL_MAX];
    __shared__ uint32_t s_rank[DEDUP_SMALL_MAX];
    __shared__ uint32_t s_seg[2 * DEDUP_SMALL_SEGS + 1];
    __shared__ uint32_t s_wave[DEDUP_SMALL_THREADS / 64];
    __shared__ uint32_t s_m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_a[i] = in[i];
    __syncthreads();
    seg_rank_sort_lds<LessDiag>(s_a, s_b, n, s_rank, s_seg);  // :776
    // adjacent-pair unique on the sorted INPUT sequence (:778-780), order preserving; one element per thread
    {
        const uint32_t i = threadIdx.x;
        bool keep = false;
        HspRec me;
        me.ref_start = me.query_start = me.len = 0; me.score = 0; me.seg = 0;
        if (i < n) {
            me = s_a[i];
            keep = i == 0 || s_a[i - 1].seg != me.seg || !hsp_contained(s_a[i - 1], me);
        }
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int w = 0; w < DEDUP_SMALL_THREADS / 64; w++) {
            const uint32_t c = s_wave[w];
            if (w < wave) base += c;
            total += c;
        }
        if (keep) s_b[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = me;
        if (threadIdx.x == 0) s_m = total;
        __syncthreads();
    }
    const uint32_t m = s_m;
    seg_rank_sort_lds<LessLastz>(s_b, s_a, m, s_rank, s_seg);  // :782
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const HspRec r = s_b[i];
        out[i] = make_uint4(r.ref_start, r.query_start, r.len, (uint32_t)r.score);
        if (out_seg) out_seg[i] = r.seg;
    }
    if (threadIdx.x == 0) *out_count = m;
}

uint32_t dedup_small_max() { return DEDUP_SMALL_MAX; }
uint32_t dedup_small_max_segs() { return DEDUP_SMALL_SEGS; }
void launch_dedup_small(const HspRec* in, uint32_t n, void* out_segment_pairs, uint32_t* out_seg, uint32_t* out_count, hipStream_t s) {
    hipLaunchKernelGGL(dedup_small_kernel, dim3(1), dim3(DEDUP_SMALL_THREADS), 0, s, in, n, reinterpret_cast<uint4*>(out_segment_pairs), out_seg, out_count);
}

void launch_unique(const HspRec* in, HspRec* out, uint32_t n, int exact, uint32_t* out_count, hipStream_t s) {
    hipLaunchKernelGGL(unique_kernel, dim3(1), dim3(UNQ_THREADS), 0, s, in, out, n, exact, out_count);
}
void launch_strip(const HspRec* in, uint32_t n, void* out_segment_pairs, uint32_t* out_seg, hipStream_t s) {
    if (n == 0) return;
    uint32_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(strip_kernel, dim3(g), dim3(256), 0, s, in, n, reinterpret_cast<uint4*>(out_segment_pairs), out_seg);
}

}  // namespace sa
