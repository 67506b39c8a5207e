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

// The same predicate over MANY tiles: a call on repeat-rich sequence leaves millions of survivors (every diagonal of a microsatellite
// pair is an HSP of its own), and one workgroup walking them took 150 ms per call.  A record's verdict only needs its INPUT
// predecessor (hazard H3), so tiles are independent: count per tile -> exclusive scan of the tile counts -> write.
__device__ __forceinline__ uint32_t unique_tile_mask(const HspRec* __restrict__ in, uint32_t n, int exact, uint32_t i0, HspRec (&rec)[UNQ_ITEMS]) {
    uint32_t keepmask = 0;
    HspRec prev;
    prev.ref_start = prev.query_start = prev.len = 0; prev.score = 0; prev.seg = 0xFFFFFFFFu;
    if (i0 > 0 && i0 < n) prev = in[i0 - 1];
#pragma unroll
    for (int j = 0; j < UNQ_ITEMS; j++) {
        const uint32_t i = i0 + j;
        if (i < n) {
            rec[j] = in[i];
            const bool keep = (i == 0) || (prev.seg != rec[j].seg) || !(exact ? hsp_same(prev, rec[j]) : hsp_contained(prev, rec[j]));
            keepmask |= keep ? (1u << j) : 0u;
            prev = rec[j];
        }
    }
    return keepmask;
}
__global__ __launch_bounds__(UNQ_THREADS) void unique_tile_count_kernel(const HspRec* __restrict__ in, uint32_t n, int exact, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t s_tot;
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    HspRec rec[UNQ_ITEMS];
    uint32_t mine = (uint32_t)__builtin_popcount(unique_tile_mask(in, n, exact, blockIdx.x * (UNQ_THREADS * UNQ_ITEMS) + threadIdx.x * UNQ_ITEMS, rec));
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_tot, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_tot;
}
__global__ __launch_bounds__(UNQ_THREADS) void unique_tile_write_kernel(const HspRec* __restrict__ in, HspRec* __restrict__ out, uint32_t n, int exact,
                                                                        const uint32_t* __restrict__ tile_base, uint32_t ntiles, uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wave_cnt[UNQ_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HspRec rec[UNQ_ITEMS];
    const uint32_t keepmask = unique_tile_mask(in, n, exact, blockIdx.x * (UNQ_THREADS * UNQ_ITEMS) + threadIdx.x * UNQ_ITEMS, rec);
    const uint32_t mine = (uint32_t)__builtin_popcount(keepmask);
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_cnt[wave] = inc;
    __syncthreads();
    uint32_t off = tile_base[blockIdx.x] + inc - mine;
    for (int w = 0; w < wave; w++) off += wave_cnt[w];
#pragma unroll
    for (int j = 0; j < UNQ_ITEMS; j++)
        if ((keepmask >> j) & 1u) out[off++] = rec[j];
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = tile_base[ntiles];
}

__global__ __launch_bounds__(256) void strip_kernel(const HspRec* __restrict__ in, uint32_t n, uint4* __restrict__ out,
                                                    uint32_t* __restrict__ out_seg) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const HspRec r = in[i];
        out[i] = make_uint4(r.ref_start, r.query_start, r.len, (uint32_t)r.score);
        if (out_seg) out_seg[i] = r.seg;
    }
}

constexpr int DEDUP_SMALL_SEGS = 512;  // distinct segment ids (reference iterations x chunks of a call) the LDS path handles = MAX_SEGS

// ---- the whole chain in LDS, ONE WORKGROUP PER SEGMENT -----------------------------------------------------------------
// After the chain shortcut a call leaves a few hundred to a few thousand survivors; three library sorts + unique + strip then
// cost ~65 us of pure launch latency and an extra host sync (src/seed_filter.cu:776-782 per iteration).
// A multi-chunk call carries up to 8 segments (4 chunks x 2 reference iterations) whose chains are independent (:776-782 run
// per iteration).  One workgroup per segment id picks its records out of the survivor list, runs sort -> unique -> sort in
// LDS and writes its result into the slot range the segment's INPUT records would occupy (offset = number of survivors in
// lower segments); the host closes the gaps while it splits the output per chunk anyway.  Eight small workgroups on eight
// CUs instead of one 1024-thread workgroup walking all records: the stage's latency drops from ~200 us (500 us next to a
// running filter kernel) to ~20 us, and the limit rises from 1024 survivors per call to 2048 per segment.
constexpr int DEDUP_SEG_THREADS = 1024;
constexpr int DEDUP_SEG_MAX = 2048;      // records per segment (40 KB of LDS)
constexpr int DEDUP_SEG_TOTAL = 131072;  // survivors per call (every workgroup scans the whole list once)

// In-place bitonic sort of m <= DEDUP_SEG_MAX records of ONE segment in LDS.  The records' seg field (constant inside a
// segment) carries the input index while sorting: it is the last key, which makes the keys unique and the result the stable
// order thrust::stable_sort gives (:776,:782); padding entries up to the next power of two compare above everything.
// (The quadratic rank sort this replaces took 0.65 ms per 2048 records; the network is m log^2 m / 2 compare-exchanges.)
constexpr uint32_t SORT_PAD = 0xFFFFFFFFu;
struct KeyDiag {   // hspComp :54-80 inside one segment, then the input index
    __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        const uint32_t dx = x.ref_start - x.query_start, dy = y.ref_start - y.query_start;
        if (dx != dy) return dx < dy;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        if (x.score != y.score) return x.score > y.score;
        return x.seg < y.seg;
    }
};
struct KeyLastz {  // hspCompLastz :82-108 inside one segment, then the input index
    __device__ bool operator()(const HspRec& x, const HspRec& y) const {
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        if (x.score != y.score) return x.score > y.score;
        return x.seg < y.seg;
    }
};
template <class Key>
__device__ __forceinline__ void bitonic_sort_lds(HspRec* __restrict__ a, uint32_t m, uint32_t seg) {
    Key less;
    uint32_t P = 1;
    while (P < m) P <<= 1;
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) a[i].seg = i < m ? i : SORT_PAD;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));  // pair t: a zero bit inserted at bit log2(j)
                const uint32_t hi = lo | j;
                const HspRec x = a[lo], y = a[hi];
                const bool xp = x.seg == SORT_PAD, yp = y.seg == SORT_PAD;
                const bool y_lt_x = (xp || yp) ? (xp && !yp) : less(y, x);
                const bool x_lt_y = (xp || yp) ? (yp && !xp) : less(x, y);
                if (((lo & k) == 0) ? y_lt_x : x_lt_y) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) a[i].seg = seg;
    __syncthreads();
}

__global__ __launch_bounds__(DEDUP_SEG_THREADS) void dedup_seg_kernel(const HspRec* __restrict__ in, uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                      uint4* __restrict__ out, uint32_t* __restrict__ seg_info /* [2 * SEGS + 1] */,
                                                                      uint32_t seg_max /* <= DEDUP_SEG_MAX */) {
    if (n_dev) {  // speculative launch: the survivor count is still on the device
        n = *n_dev;
        if (n > (uint32_t)DEDUP_SEG_TOTAL) {
            if (threadIdx.x == 0) seg_info[2 * DEDUP_SMALL_SEGS] = 1u;
            return;
        }
    }
    __shared__ HspRec s_a[DEDUP_SEG_MAX];  // ONE buffer (40 KB): the unique step compacts in place, so the workgroup fits into the
                                           // LDS a single retiring filter workgroup frees
    __shared__ uint32_t s_cnt[DEDUP_SMALL_SEGS];
    __shared__ uint32_t s_m, s_wave[DEDUP_SEG_THREADS / 64];
    const uint32_t g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t t = threadIdx.x; t < (uint32_t)DEDUP_SMALL_SEGS; t += blockDim.x) s_cnt[t] = 0;
    if (threadIdx.x == 0) s_m = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const HspRec r = in[i];
        const uint32_t sg = r.seg & (DEDUP_SMALL_SEGS - 1);
        atomicAdd(&s_cnt[sg], 1u);
        if (sg == g) {
            const uint32_t k = atomicAdd(&s_m, 1u);
            if (k < DEDUP_SEG_MAX) s_a[k] = r;  // order inside the segment is arbitrary here: equal records are identical
        }
    }
    __syncthreads();
    // offset of the segment's slot range = survivors in lower segments: a workgroup-wide sum of s_cnt[0 .. g)
    __shared__ uint32_t s_off;
    if (threadIdx.x == 0) s_off = 0;
    __syncthreads();
    {
        uint32_t part = 0;
        for (uint32_t t = threadIdx.x; t < g; t += blockDim.x) part += s_cnt[t];
        for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
        if (lane == 0 && part) atomicAdd(&s_off, part);
    }
    __syncthreads();
    const uint32_t off = s_off;
    const uint32_t m = s_m;
    if (m > seg_max) {  // the host falls back to the library sorts
        if (threadIdx.x == 0) { seg_info[2 * DEDUP_SMALL_SEGS] = 1u; seg_info[g] = 0; seg_info[DEDUP_SMALL_SEGS + g] = off; }
        return;
    }
    const uint32_t my_seg = m ? s_a[0].seg : 0u;  // (every record of the workgroup carries the same segment id)
    __syncthreads();
    bitonic_sort_lds<KeyDiag>(s_a, m, my_seg);  // :776
    // adjacent-pair unique on the sorted sequence (:778-780, hazard H3), order preserving
    uint32_t carry = 0;
    for (uint32_t base = 0; base < m; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        bool keep = false;
        HspRec me;
        me.ref_start = me.query_start = me.len = 0; me.score = 0; me.seg = 0;
        if (i < m) {
            me = s_a[i];
            keep = i == 0 || !hsp_contained(s_a[i - 1], me);  // the INPUT neighbour (H3): slot i - 1 still holds it, see below
        }
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
        __syncthreads();  // every read of this tile is done before anything is written
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            const uint32_t c = s_wave[w];
            if (w < wave) pre += c;
            tot += c;
        }
        // in place: kept records move left, so a slot is only ever overwritten by a record from its own or a later position;
        // the last slot of the tile (the next tile's input neighbour) is rewritten only with itself
        if (keep) s_a[carry + pre + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = me;
        carry += tot;
        __syncthreads();
    }
    const uint32_t m2 = carry;
    bitonic_sort_lds<KeyLastz>(s_a, m2, my_seg);  // :782
    for (uint32_t i = threadIdx.x; i < m2; i += blockDim.x) {
        const HspRec r = s_a[i];
        out[off + i] = make_uint4(r.ref_start, r.query_start, r.len, (uint32_t)r.score);
    }
    if (threadIdx.x == 0) { seg_info[g] = m2; seg_info[DEDUP_SMALL_SEGS + g] = off; }
}

uint32_t dedup_seg_max_total() { return DEDUP_SEG_TOTAL; }
uint32_t dedup_seg_info_words() { return 2 * DEDUP_SMALL_SEGS + 1; }
// seg_info[g] = records of segment g after the chain, seg_info[SEGS + g] = their first slot in out; seg_info[2 * SEGS] != 0:
// a segment held more than DEDUP_SEG_MAX records (nothing usable was written); must be zero on entry
// threads: workgroup size (0 = DEDUP_SEG_THREADS)
// n_dev != nullptr: the number of records is read from the device (n ignored); more than dedup_seg_max_total() sets the flag
// seg_max: records per segment the LDS path accepts (0 = DEDUP_SEG_MAX; tests lower it to reach the fallback)
void launch_dedup_seg(const HspRec* in, uint32_t n, const uint32_t* n_dev, uint32_t nsegs, void* out_segment_pairs, uint32_t* seg_info,
                      uint32_t threads, uint32_t seg_max, hipStream_t s) {
    seg_max = seg_max ? std::min<uint32_t>(seg_max, DEDUP_SEG_MAX) : (uint32_t)DEDUP_SEG_MAX;
    threads = threads ? std::min<uint32_t>(DEDUP_SEG_THREADS, std::max<uint32_t>(64, threads & ~63u)) : (uint32_t)DEDUP_SEG_THREADS;
    hipLaunchKernelGGL(dedup_seg_kernel, dim3(nsegs), dim3(threads), 0, s, in, n, n_dev, reinterpret_cast<uint4*>(out_segment_pairs), seg_info, seg_max);
}

uint32_t dedup_small_max_segs() { return DEDUP_SMALL_SEGS; }

// tile_tmp: 2 * (tiles + 1) dwords + scan_temp_bytes(tiles) bytes of scratch (unique_temp_bytes); small inputs take the one-workgroup
// kernel and need none
size_t unique_temp_bytes(uint32_t n) {
    const uint32_t tiles = (n + UNQ_THREADS * UNQ_ITEMS - 1) / (UNQ_THREADS * UNQ_ITEMS);
    return (size_t)2 * (tiles + 1) * sizeof(uint32_t) + scan_temp_bytes(tiles) + 64;
}
void launch_unique(const HspRec* in, HspRec* out, uint32_t n, int exact, uint32_t* out_count, void* tile_tmp, hipStream_t s) {
    const uint32_t tiles = (n + UNQ_THREADS * UNQ_ITEMS - 1) / (UNQ_THREADS * UNQ_ITEMS);
    if (tiles <= 4 || !tile_tmp) {
        hipLaunchKernelGGL(unique_kernel, dim3(1), dim3(UNQ_THREADS), 0, s, in, out, n, exact, out_count);
        return;
    }
    uint32_t* cnt = reinterpret_cast<uint32_t*>(tile_tmp);
    uint32_t* base = cnt + (tiles + 1);
    void* scan_tmp = reinterpret_cast<void*>(((uintptr_t)(base + (tiles + 1)) + 63) & ~(uintptr_t)63);
    hipLaunchKernelGGL(unique_tile_count_kernel, dim3(tiles), dim3(UNQ_THREADS), 0, s, in, n, exact, cnt);
    launch_exclusive_scan_u32(cnt, base, tiles, scan_tmp, s);
    hipLaunchKernelGGL(unique_tile_write_kernel, dim3(tiles), dim3(UNQ_THREADS), 0, s, in, out, n, exact, base, tiles, out_count);
}
void launch_strip(const HspRec* in, uint32_t n, void* out_segment_pairs, uint32_t* out_seg, hipStream_t s) {
    if (n == 0) return;
    uint32_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(strip_kernel, dim3(g), dim3(256), 0, s, in, n, reinterpret_cast<uint4*>(out_segment_pairs), out_seg);
}

}  // namespace sa
