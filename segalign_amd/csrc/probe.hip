// probe.hip -- the NEIGHBOURHOOD seed table and the position probe: seed lookup without seed words.
//
// Reference shape (src/seeder.cpp:57-74 + src/seed_filter.cu:157-230): the host emits, per valid query position, 13 seed
// words (the k-mer and its 12 single-transition neighbours, key ^ (2 << 2t)); find_num_hits gathers one bucket extent per
// WORD from the 4^k-entry index table, a scan turns the counts into offsets, find_hits copies every bucket into a hit list.
// On a 100 Mbp target every one of those 13 gathers touches its own 128-byte line of the index table and its own line of
// the position table to use ~8 + ~20 bytes: the lookup moves ~8x its algorithmic bytes (profiles/r01).
//
// MI355X shape: 288 GB of HBM buy a table in which the 13 buckets a query position needs are ONE contiguous run.
//   nbr_start[key] .. nbr_start[key+1]  =  bucket(key) ++ bucket(key ^ (2<<2t0)) ++ bucket(key ^ (2<<2t1)) ++ ...
// in exactly the order src/seeder.cpp:60-69 emits the seed words, so the concatenation of the runs of consecutive query
// positions IS the reference's hit list order (up to the order inside a bucket, which no consumer depends on, SURVEY a-6).
// One probe per query POSITION (one 16-byte extent, one contiguous run) replaces 13 probes per position; seed words,
// per-word extents and per-word prefix sums are never materialised, and the X-drop filter reads its anchors straight
// out of the runs (extend.hip, TD fetch), so there is no hit list either.  Table size = (1 + #transition positions) x
// pos_table (4.2 GB for a 100 Mbp block, ~26 GB for a 500 Mbp block) + 8 bytes per key.
//
// Per call (positions [start, end) of one strand, up to SA_MAX_CHUNKS chunks):
//   probe_kernel    position -> k-mer (kmer_dev.h) -> {run offset, run length}; per-block sums of (hits, non-empty, valid)
//   probe_partials  exclusive scan of the block sums (one workgroup)
//   probe_compact   order-preserving compaction of the NON-EMPTY positions into 16-byte records {hit offset of the position's
//                   first hit inside the call, query position, run offset}; prefix values at the chunk boundaries
//   probe_plan      per chunk: number of hits, and the reference's iteration split (src/seed_filter.cu:718-745 for
//                   num_hits < MAX_HITS: everything before the LAST HIT-BEARING SEED WORD / that word's hits) -- found from
//                   the last non-empty position and the plain bucket sizes of its 13 words
#include "kernels.h"
#include "kmer_dev.h"
#include "probe.h"

namespace sa {

// ---------------------------------------------------------------------------------------------------------------------
// neighbourhood table build (once per target block, from the plain table of table.hip)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bucket_len(const uint32_t* __restrict__ bucket_start, uint32_t key) {
    return bucket_start[key + 1] - bucket_start[key];
}

// merged run length per key; *overflow is set if a run does not fit 32 bits (the probe keeps run lengths in u32)
__global__ __launch_bounds__(256) void nbr_count_kernel(const uint32_t* __restrict__ bucket_start, uint32_t nkeys, uint32_t tmask,
                                                        int weight, uint32_t* __restrict__ cnt, uint32_t* __restrict__ overflow) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nkeys; k += gridDim.x * blockDim.x) {
        uint64_t c = bucket_len(bucket_start, k);
        for (int t = 0; t < weight; t++)
            if ((tmask >> t) & 1u) c += bucket_len(bucket_start, k ^ (2u << (2 * t)));
        if (c > 0xFFFFFFFFull) { atomicOr(overflow, 1u); c = 0xFFFFFFFFull; }
        cnt[k] = (uint32_t)c;
    }
}

// One 16-lane group per key copies the key's 1 + popcount(tmask) buckets into its run, sub-run after sub-run (seed word
// order).  Buckets are short (T / 4^k entries), so a group streams a ~250-byte run while its 16 lanes share the loads.
constexpr int NBR_GROUP = 16;
__global__ __launch_bounds__(256) void nbr_fill_kernel(const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ pos_table,
                                                       uint32_t nkeys, uint32_t tmask, int weight,
                                                       const uint64_t* __restrict__ nbr_start, uint32_t* __restrict__ nbr_pos) {
    const uint32_t gl = threadIdx.x & (NBR_GROUP - 1);
    const uint32_t groups = gridDim.x * (blockDim.x / NBR_GROUP);
    for (uint32_t k = blockIdx.x * (blockDim.x / NBR_GROUP) + threadIdx.x / NBR_GROUP; k < nkeys; k += groups) {
        uint64_t o = nbr_start[k];
        for (int j = -1; j < weight; j++) {
            if (j >= 0 && !((tmask >> j) & 1u)) continue;
            const uint32_t kk = j < 0 ? k : (k ^ (2u << (2 * j)));
            const uint32_t b = bucket_start[kk], n = bucket_start[kk + 1] - b;
            for (uint32_t i = gl; i < n; i += NBR_GROUP) nbr_pos[o + i] = pos_table[b + i];
            o += n;
        }
    }
}

// ---- context records (class filter, extend.hip 1d): kernels.h CtxRec --------------------------------------------------------
// reverse the sixteen 2-bit fields of a dword (bit reversal also swaps the two bits of every field: swap them back)
__device__ __forceinline__ uint32_t fieldrev16(uint32_t v) {
    const uint32_t r = __builtin_bitreverse32(v);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}
// the 54 bases from the anchor on lie in 16 bytes of copy `anchor & 3` of the 2-bit target, the 58 bases in front of the seed start in 16
// bytes of copy `pos & 3` (encode.hip: a window that starts or ends at ANY base is byte aligned in the copy of its phase; a span of
// up to 32 bytes lies inside the overlapped line chosen for its first byte)
__device__ __forceinline__ void cut_ctx(const uint8_t* __restrict__ ref2, size_t ref2_stride, uint32_t pos, uint32_t seed_size, uint32_t left_skip, uint4& c0, uint4& c1) {
    const uint32_t anchor = pos + seed_size;                    // :220
    const uint32_t jj0 = (anchor >> 2) + (uint32_t)PACK2_BIAS;  // logical byte of the anchor's 4-base group in copy anchor & 3
    const uint8_t* tp = ref2 + (size_t)(anchor & 3u) * ref2_stride + (jj0 + 32u * (jj0 / (uint32_t)PACK2_PAYLOAD));
    const uint4 rw = load16u(tp);
    const uint32_t lend = anchor - left_skip;  // the left context ends in front of this base: the seed start (left_skip = seed_size), or the anchor (0)
    const uint32_t jjl = (lend >> 2) + (uint32_t)PACK2_BIAS - 16u;  // first of the 16 bytes below its group in copy lend & 3
    const uint8_t* lp = ref2 + (size_t)(lend & 3u) * ref2_stride + (jjl + 32u * (jjl / (uint32_t)PACK2_PAYLOAD));
    const uint4 lw = load16u(lp);
    // bases lend-1 .. lend-64 in walking order, complemented; the record takes the first CTX_L_BASES of them behind the CTX_R_BASES
    // right bases: one 224-bit string (CtxRec)
    const uint32_t l0 = ~fieldrev16(lw.w), l1 = ~fieldrev16(lw.z), l2 = ~fieldrev16(lw.y), l3 = ~fieldrev16(lw.x);
    static_assert(CTX_R_BASES == 54 && CTX_L_BASES == 58, "the record's cut: 108 + 116 bits");
    c0 = make_uint4(pos, rw.x, rw.y, rw.z);
    c1 = make_uint4((rw.w & 0xFFFu) | (l0 << 12), __builtin_amdgcn_alignbit(l1, l0, 20), __builtin_amdgcn_alignbit(l2, l1, 20),
                    __builtin_amdgcn_alignbit(l3, l2, 20));
}
// stage 1: the context of every seed position ONCE, in pos_table order (one random target line per position)
__global__ __launch_bounds__(256) void ctx_by_index_kernel(const uint32_t* __restrict__ pos_table, uint32_t num_index,
                                                           const uint8_t* __restrict__ ref2, size_t ref2_stride, uint32_t seed_size, uint32_t left_skip,
                                                           uint4* __restrict__ out /* 2 x uint4 per position */) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < num_index; id += gridDim.x * blockDim.x) {
        uint4 c0, c1;
        cut_ctx(ref2, ref2_stride, pos_table[id], seed_size, left_skip, c0, c1);
        out[2 * (size_t)id] = c0;
        out[2 * (size_t)id + 1] = c1;
    }
}
// stage 2: one WAVE per key, one lane per run ENTRY.  The lanes 0..weight hold the key's source buckets {start, length}, a wave
// prefix gives every piece its offset inside the run, and entry t of the run finds its piece from scalar broadcasts; it then moves
// its whole 32-byte record (two aligned 16-byte loads / stores).  Consecutive lanes read consecutive records of a piece and write
// consecutive records of the run: both sides are coalesced, 64 entries per wave instruction (the 16-lane-per-key, dword-per-lane
// form of the first version ran at 1.3 TB/s of the ~68 GB this kernel moves per 100 Mbp block).
__global__ __launch_bounds__(256) void nbr_copy_ctx_kernel(const uint32_t* __restrict__ bucket_start, uint32_t nkeys, uint32_t tmask, int weight,
                                                           const uint64_t* __restrict__ nbr_start, const uint4* __restrict__ by_index,
                                                           uint4* __restrict__ ctx) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < nkeys; k += waves) {
        // piece j of the run (seed word order, seeder.cpp:60-69): j = 0 the key itself, then one per transition position
        uint32_t pb = 0, pn = 0;
        if (lane <= weight) {
            const int j = lane - 1;
            if (j < 0 || ((tmask >> j) & 1u)) {
                const uint32_t kk = j < 0 ? k : (k ^ (2u << (2 * j)));
                pb = bucket_start[kk];
                pn = bucket_start[kk + 1] - pb;
            }
        }
        uint32_t pre = pn;  // inclusive prefix of the piece lengths over lanes 0..16
#pragma unroll
        for (int off = 1; off <= 16; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)pre, off, 64);
            if (lane >= off) pre += v;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pre, MAX_CARE);  // (lanes past `weight` add nothing)
        const uint32_t excl = pre - pn;
        const uint64_t o = nbr_start[k];
        for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            // the piece that holds entry t: the last lane p <= weight with excl[p] <= t and a non-empty piece
            uint32_t src = 0;
#pragma unroll
            for (int p = 0; p <= MAX_CARE; p++) {
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)excl, p), n = (uint32_t)__builtin_amdgcn_readlane((int)pn, p),
                               bb = (uint32_t)__builtin_amdgcn_readlane((int)pb, p);  // (wave-uniform: scalar registers, no LDS traffic)
                if (p <= weight && n && t >= e) src = bb + (t - e);
            }
            if (t < total) {
                const uint4 a = by_index[2 * (size_t)src], b = by_index[2 * (size_t)src + 1];
                ctx[2 * (o + t)] = a;
                ctx[2 * (o + t) + 1] = b;
            }
        }
    }
}
// stage 2 for SPARSE tables -- fewer seed positions than keys: 14of22 (2^28 keys, 0.37 positions per bucket on a 100 Mbp block, a
// run of ~6 entries) and every small target.  A wave per key spent 119 ms on 14of22's 268 M keys with 6 of its 64 lanes at work;
// here a LANE takes a key: the extents of consecutive keys are read coalesced (piece j of key k is bucket k ^ const), the lane
// copies its few records one after the other, and the runs of a wave's 64 keys are one contiguous stretch of the table.
__global__ __launch_bounds__(256) void nbr_copy_ctx_sparse_kernel(const uint32_t* __restrict__ bucket_start, uint32_t nkeys, uint32_t tmask, int weight,
                                                                  const uint64_t* __restrict__ nbr_start, const uint4* __restrict__ by_index,
                                                                  uint4* __restrict__ ctx) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nkeys; k += stride) {
        uint64_t o = nbr_start[k];
        if (nbr_start[k + 1] == o) continue;
        for (int j = -1; j < weight; j++) {  // (seed word order, seeder.cpp:60-69: the key itself, then one piece per transition position)
            if (j >= 0 && !((tmask >> j) & 1u)) continue;
            const uint32_t kk = j < 0 ? k : (k ^ (2u << (2 * j)));
            const uint32_t pe = bucket_start[kk + 1];
            for (uint32_t e = bucket_start[kk]; e < pe; e++, o++) {
                const uint4 a = by_index[2 * (size_t)e], b = by_index[2 * (size_t)e + 1];
                ctx[2 * o] = a;
                ctx[2 * o + 1] = b;
            }
        }
    }
}
// one-stage form (no scratch): every table entry cuts its own context out of the target
__global__ __launch_bounds__(256) void nbr_fill_ctx_kernel(const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ pos_table,
                                                           uint32_t nkeys, uint32_t tmask, int weight, const uint64_t* __restrict__ nbr_start,
                                                           const uint8_t* __restrict__ ref2, size_t ref2_stride, uint32_t seed_size, uint32_t left_skip,
                                                           uint4* __restrict__ ctx) {
    const uint32_t gl = threadIdx.x & (NBR_GROUP - 1);
    const uint32_t groups = gridDim.x * (blockDim.x / NBR_GROUP);
    for (uint32_t k = blockIdx.x * (blockDim.x / NBR_GROUP) + threadIdx.x / NBR_GROUP; k < nkeys; k += groups) {
        uint64_t o = nbr_start[k];
        for (int j = -1; j < weight; j++) {
            if (j >= 0 && !((tmask >> j) & 1u)) continue;
            const uint32_t kk = j < 0 ? k : (k ^ (2u << (2 * j)));
            const uint32_t b = bucket_start[kk], n = bucket_start[kk + 1] - b;
            for (uint32_t i = gl; i < n; i += NBR_GROUP) {
                uint4 c0, c1;
                cut_ctx(ref2, ref2_stride, pos_table[b + i], seed_size, left_skip, c0, c1);
                ctx[2 * (o + i)] = c0;
                ctx[2 * (o + i) + 1] = c1;
            }
            o += n;
        }
    }
}

void launch_nbr_fill_ctx(const uint32_t* bucket_start, const uint32_t* pos_table, uint32_t nkeys, uint32_t tmask, int weight,
                         const uint64_t* nbr_start, const uint8_t* ref2, size_t ref2_stride, uint32_t seed_size, uint32_t left_skip, CtxRec* ctx,
                         CtxRec* scratch, uint32_t num_index, hipStream_t s) {
    if (scratch) {
        hipLaunchKernelGGL(ctx_by_index_kernel, dim3(8192), dim3(256), 0, s, pos_table, num_index, ref2, ref2_stride, seed_size, left_skip,
                           reinterpret_cast<uint4*>(scratch));
        if (num_index < nkeys)  // (less than one position per bucket: a lane per key)
            hipLaunchKernelGGL(nbr_copy_ctx_sparse_kernel, dim3(8192), dim3(256), 0, s, bucket_start, nkeys, tmask, weight, nbr_start,
                               reinterpret_cast<const uint4*>(scratch), reinterpret_cast<uint4*>(ctx));
        else
            hipLaunchKernelGGL(nbr_copy_ctx_kernel, dim3(8192), dim3(256), 0, s, bucket_start, nkeys, tmask, weight, nbr_start,
                               reinterpret_cast<const uint4*>(scratch), reinterpret_cast<uint4*>(ctx));
        return;
    }
    hipLaunchKernelGGL(nbr_fill_ctx_kernel, dim3(8192), dim3(256), 0, s, bucket_start, pos_table, nkeys, tmask, weight, nbr_start, ref2,
                       ref2_stride, seed_size, left_skip, reinterpret_cast<uint4*>(ctx));
}

void launch_nbr_count(const uint32_t* bucket_start, uint32_t nkeys, uint32_t tmask, int weight, uint32_t* cnt, uint32_t* overflow,
                      hipStream_t s) {
    hipLaunchKernelGGL(nbr_count_kernel, dim3(4096), dim3(256), 0, s, bucket_start, nkeys, tmask, weight, cnt, overflow);
}
void launch_nbr_fill(const uint32_t* bucket_start, const uint32_t* pos_table, uint32_t nkeys, uint32_t tmask, int weight,
                     const uint64_t* nbr_start, uint32_t* nbr_pos, hipStream_t s) {
    hipLaunchKernelGGL(nbr_fill_kernel, dim3(8192), dim3(256), 0, s, bucket_start, pos_table, nkeys, tmask, weight, nbr_start, nbr_pos);
}

// ---------------------------------------------------------------------------------------------------------------------
// probe: positions -> runs
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PR_THREADS = 256;
constexpr int PR_ITEMS = 4;                       // positions per thread: one ALIGNED group of four (kmer4_at)
constexpr int PR_TILE = PR_THREADS * PR_ITEMS;    // positions per workgroup
constexpr int HB_LDS_WORDS = 4096;                // head-bit words a workgroup of the compaction gathers in LDS (16 KB: 131072 hits)

struct Tri { uint64_t hits; uint32_t ne, valid; };  // (hits, non-empty positions, valid positions)

__device__ __forceinline__ Tri tri_add(Tri a, Tri b) { return {a.hits + b.hits, a.ne + b.ne, a.valid + b.valid}; }

__device__ __forceinline__ Tri wave_incl_scan(Tri v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t h = __shfl_up(v.hits, off, 64);
        const uint32_t a = __shfl_up(v.ne, off, 64), b = __shfl_up(v.valid, off, 64);
        if (lane >= off) { v.hits += h; v.ne += a; v.valid += b; }
    }
    return v;
}
// exclusive prefix over the workgroup + workgroup total
__device__ __forceinline__ Tri block_excl_scan(Tri v, Tri& total) {
    __shared__ Tri s_wave[PR_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Tri inc = wave_incl_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    Tri base = {0, 0, 0}, tot = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < PR_THREADS / 64; w++) {
        const Tri s = s_wave[w];
        if (w < wave) base = tri_add(base, s);
        tot = tri_add(tot, s);
    }
    __syncthreads();
    total = tot;
    return {base.hits + inc.hits - v.hits, base.ne + inc.ne - v.ne, base.valid + inc.valid - v.valid};
}

constexpr uint64_t PR_VALID = 1ull << 63;  // flag inside t_off: the window at this position is a valid k-mer

// The position grid of probe_kernel / probe_compact_kernel is aligned to multiples of four: element e of the grid is query
// position (start & ~3) + e, i.e. index i = e - (start & 3) of the call (elements outside [0, n) are skipped).
__global__ __launch_bounds__(PR_THREADS) void probe_kernel(const uint8_t* __restrict__ query, uint32_t start, uint32_t n, SeedShape sh,
                                                           const uint64_t* __restrict__ nbr_start, uint32_t nkeys,
                                                           uint64_t* __restrict__ t_off, uint32_t* __restrict__ t_cnt,
                                                           Tri* __restrict__ partial) {
    const uint32_t skew = start & 3u;
    const uint32_t e0 = blockIdx.x * PR_TILE + threadIdx.x * PR_ITEMS;  // first grid element of the thread (multiple of 4)
    Tri mine = {0, 0, 0};
    if (e0 < n + skew) {
        uint32_t key[4];
        const uint32_t valid = kmer4_at(query, (start - skew) + e0, sh, key);
        uint64_t off[PR_ITEMS];
        uint32_t cnt[PR_ITEMS];
#pragma unroll
        for (int j = 0; j < PR_ITEMS; j++) {
            const uint32_t e = e0 + j;
            off[j] = 0;
            cnt[j] = 0;
            if (e < skew || e - skew >= n) continue;
            if (((valid >> j) & 1u) && key[j] < nkeys) {
                const uint64_t b = nbr_start[key[j]], en = nbr_start[key[j] + 1];  // adjacent: one 16-byte extent per POSITION
                off[j] = b | PR_VALID;
                cnt[j] = (uint32_t)(en - b);
                mine.valid++;
                mine.ne += cnt[j] ? 1u : 0u;
                mine.hits += cnt[j];
            }
        }
        if (skew == 0u && e0 + (PR_ITEMS - 1) < n) {  // (calls start at multiples of four as a rule: three aligned 16-byte stores)
            static_assert(PR_ITEMS == 4, "vector stores of four positions");
            *reinterpret_cast<uint4*>(t_off + e0) = make_uint4((uint32_t)off[0], (uint32_t)(off[0] >> 32), (uint32_t)off[1], (uint32_t)(off[1] >> 32));
            *reinterpret_cast<uint4*>(t_off + e0 + 2) = make_uint4((uint32_t)off[2], (uint32_t)(off[2] >> 32), (uint32_t)off[3], (uint32_t)(off[3] >> 32));
            *reinterpret_cast<uint4*>(t_cnt + e0) = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
        } else {
#pragma unroll
            for (int j = 0; j < PR_ITEMS; j++) {
                const uint32_t e = e0 + j;
                if (e < skew || e - skew >= n) continue;
                t_off[e - skew] = off[j];
                t_cnt[e - skew] = cnt[j];
            }
        }
    }
    Tri total;
    block_excl_scan(mine, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// Exclusive prefixes of the workgroup sums, OUT of place (`scanned`; the sums stay, so a repeated compaction needs no second scan).
// One workgroup per 1024 sums: it adds up everything in front of its tile by itself (at most ~0.4 MB out of the L2), then scans its
// tile; the last workgroup also leaves the total.  (Rounds 2-4: ONE workgroup walked all sums serially, 75 us per 22 M positions.)
constexpr int PP_ITEMS = 4;
__global__ __launch_bounds__(PR_THREADS) void probe_partials_kernel(const Tri* __restrict__ partial, uint32_t nblocks, Tri* __restrict__ scanned,
                                                                    Tri* __restrict__ total_out) {
    const uint32_t t0 = blockIdx.x * (PR_THREADS * PP_ITEMS);
    Tri front = {0, 0, 0};
    for (uint32_t i = threadIdx.x; i < t0; i += PR_THREADS) front = tri_add(front, partial[i]);
    const uint32_t lo = t0 + threadIdx.x * PP_ITEMS;
    Tri v[PP_ITEMS];
    Tri mine = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < PP_ITEMS; j++) {
        v[j] = lo + j < nblocks ? partial[lo + j] : Tri{0, 0, 0};
        mine = tri_add(mine, v[j]);
    }
    Tri front_total, tile_total;
    block_excl_scan(front, front_total);
    Tri run = tri_add(block_excl_scan(mine, tile_total), front_total);
#pragma unroll
    for (int j = 0; j < PP_ITEMS; j++) {
        if (lo + j < nblocks) scanned[lo + j] = run;
        run = tri_add(run, v[j]);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = tri_add(front_total, tile_total);
}

// bounds[c] = exclusive prefix (hits, non-empty, valid) at position bpos[c] - start, c = 0..nb-1 (a bound == n takes the total)
__global__ __launch_bounds__(PR_THREADS) void probe_compact_kernel(uint32_t start, uint32_t n, const uint64_t* __restrict__ t_off,
                                                                   const uint32_t* __restrict__ t_cnt, const Tri* __restrict__ partial,
                                                                   const Tri* __restrict__ total, TdRec* __restrict__ c_rec,
                                                                   uint32_t* __restrict__ chunk_rec, uint32_t chunk_cap,
                                                                   uint32_t* __restrict__ head_bits, uint32_t head_words,
                                                                   TdBounds bpos, Tri* __restrict__ bounds) {
    const uint32_t skew = start & 3u;
    const uint32_t i0 = blockIdx.x * PR_TILE + threadIdx.x * PR_ITEMS - skew;  // (wraps below 0 for the first elements: i >= n then)
    uint64_t off[PR_ITEMS];
    uint32_t cnt[PR_ITEMS];
    Tri mine = {0, 0, 0};
    if (skew == 0u && i0 + (PR_ITEMS - 1) < n) {
        // (the usual case -- calls start at chunk borders, multiples of four: the thread's four extents and counts are three aligned
        //  16-byte loads instead of eight narrow ones, each of which made the wave touch four times the lines it used)
        static_assert(PR_ITEMS == 4, "vector loads of four positions");
        const uint4 o01 = *reinterpret_cast<const uint4*>(t_off + i0), o23 = *reinterpret_cast<const uint4*>(t_off + i0 + 2);
        const uint4 c4 = *reinterpret_cast<const uint4*>(t_cnt + i0);
        off[0] = ((uint64_t)o01.y << 32) | o01.x; off[1] = ((uint64_t)o01.w << 32) | o01.z;
        off[2] = ((uint64_t)o23.y << 32) | o23.x; off[3] = ((uint64_t)o23.w << 32) | o23.z;
        cnt[0] = c4.x; cnt[1] = c4.y; cnt[2] = c4.z; cnt[3] = c4.w;
    } else {
#pragma unroll
        for (int j = 0; j < PR_ITEMS; j++) {
            const uint32_t i = i0 + j;
            off[j] = 0;
            cnt[j] = 0;
            if (i < n) {
                off[j] = t_off[i];
                cnt[j] = t_cnt[i];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) {
        mine.valid += (off[j] & PR_VALID) ? 1u : 0u;
        mine.ne += cnt[j] ? 1u : 0u;
        mine.hits += cnt[j];
    }
    Tri tot;
    const Tri block_base = partial[blockIdx.x];
    Tri run = tri_add(block_excl_scan(mine, tot), block_base);
    // The workgroup's records and head bits are put together in LDS and leave in whole lines: its <= 1024 records are consecutive
    // in c_rec (16-byte stores at a 64-byte stride per lane before: every line was touched by four store instructions), and its
    // head bits span the words [w0, w0 + hb_words) of the map -- set with LDS atomics, the inner words stored plainly, only the
    // first and the last one OR-ed into memory (they are shared with the neighbouring workgroups).  Before: one global atomic per
    // position on dense input, 10 M per call.  A workgroup whose hits span more words than the LDS holds (> 128 hits per position
    // on average) keeps the per-thread atomics.
    __shared__ TdRec s_rec[PR_TILE];
    __shared__ uint32_t s_hb[HB_LDS_WORDS];
    const uint32_t w0 = (uint32_t)(block_base.hits >> 5);
    const uint32_t hb_words = tot.hits ? (uint32_t)((block_base.hits + tot.hits - 1) >> 5) - w0 + 1u : 0u;
    const bool hb_lds = head_bits && hb_words <= (uint32_t)HB_LDS_WORDS;
    if (hb_lds) {
        for (uint32_t t = threadIdx.x; t < hb_words; t += PR_THREADS) s_hb[t] = 0u;
        __syncthreads();
    }
    // the first chunk boundary at or behind this thread's first position (boundaries sit at multiples of the chunk size: at most one
    // falls into the thread's PR_ITEMS consecutive positions, PR_ITEMS <= chunk)
    const uint32_t i_first = i0 + ((int32_t)i0 < 0 ? skew : 0u);  // (the first thread of the grid starts below position 0)
    uint32_t cb = (i_first + bpos.chunk - 1u) / bpos.chunk;
    uint32_t ib = cb * bpos.chunk;
    // head bits of the thread's positions are gathered per 32-bit word before they go out: with a few hits per position (sparse-hit
    // calls) the four positions of a thread -- and those of its neighbours -- set bits of the SAME word, and one atomic per record
    // queued up on it
    uint32_t hb_word = 0xFFFFFFFFu, hb_mask = 0;
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) {
        const uint32_t i = i0 + j;
        if (i >= n) continue;
        if (i == ib) {
            if (cb < (uint32_t)bpos.nb) bounds[cb] = run;
            cb++;
            ib += bpos.chunk;  // (a chunk size below PR_ITEMS puts several boundaries into one thread)
        }
        if (cnt[j]) {
            TdRec r;
            r.prefix = (uint32_t)run.hits;  // (a call with >= 2^32 hits is sent down the general path, engine.hip td_front)
            r.qpos = start + i;
            r.off = off[j] & ~PR_VALID;
            s_rec[run.ne - block_base.ne] = r;
            // head bit of the position's first hit: bit g of the call-wide bitmap <=> a record starts at hit g.  The class filter
            // finds the record of every hit of a 64-hit buffer from ONE 64-bit word of it (extend.hip 1d)
            if (hb_lds) {
                atomicOr(&s_hb[(uint32_t)(run.hits >> 5) - w0], 1u << (run.hits & 31u));
            } else if (head_bits && (uint32_t)(run.hits >> 5) < head_words) {
                const uint32_t w = (uint32_t)(run.hits >> 5);
                if (w != hb_word) {
                    if (hb_mask) atomicOr(&head_bits[hb_word], hb_mask);
                    hb_word = w;
                    hb_mask = 0;
                }
                hb_mask |= 1u << (run.hits & 31u);
            }
            // which record holds hit k * TD_CHUNK_HITS?  (the context filter's waves start there without searching)
            for (uint64_t k = (run.hits + TD_CHUNK_HITS - 1) / TD_CHUNK_HITS; k * TD_CHUNK_HITS < run.hits + cnt[j] && k < chunk_cap; k++)
                chunk_rec[k] = run.ne;
        }
        run.hits += cnt[j];
        run.ne += cnt[j] ? 1u : 0u;
        run.valid += (off[j] & PR_VALID) ? 1u : 0u;
    }
    if (hb_mask) atomicOr(&head_bits[hb_word], hb_mask);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < tot.ne; t += PR_THREADS) c_rec[block_base.ne + t] = s_rec[t];
    if (hb_lds)
        for (uint32_t t = threadIdx.x; t < hb_words; t += PR_THREADS) {
            const uint32_t w = w0 + t, m = s_hb[t];
            if (w >= head_words) break;  // (a map that is too small is regrown by the host and the compaction repeated)
            if (t == 0u || t + 1u == hb_words) {
                if (m) atomicOr(&head_bits[w], m);
            } else {
                head_bits[w] = m;
            }
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const Tri t = *total;
        TdRec r;
        r.prefix = (uint32_t)t.hits;  // sentinel: one past the last non-empty position
        r.qpos = 0;
        r.off = 0;
        c_rec[t.ne] = r;
        for (int c = 0; c < bpos.nb; c++)
            if ((uint64_t)c * bpos.chunk >= n) bounds[c] = t;  // (boundaries at or beyond the end of the call take the total)
    }
}

// one lane per chunk: the reference's iteration plan (src/seed_filter.cu:718-745) in terms of HIT offsets.  The reference plans over the
// inclusive prefix of hits per SEED WORD: iteration i ends behind the last word whose prefix is below limit_i (lower_bound - 1), with
// limit_0 = min(num_hits, MAX_HITS) and limit_{i+1} = min(num_hits, prefix at that word + MAX_HITS); the last iteration takes the rest.
// Only the VALUES matter (an iteration is the hit range between two ends): iteration i ends at the largest word boundary below limit_i.
// Here the word boundaries are: first hit of non-empty position m (c_rec[m].prefix) + the plain bucket sizes of its words in emission
// order (seeder.cpp:60-69) -- a binary search over the chunk's positions, then a walk over <= 16 buckets.  num_hits < MAX_HITS gives the
// two iterations "everything in front of the last hit-bearing word / that word's hits"; a chunk at or above MAX_HITS gets the greedy
// groups, up to TD_MAX_ITER of them (round 6: such a chunk used to leave the table-direct path for the general one, 3.5 x slower at
// human-block density under an 8 GiB GPU's MAX_HITS -- profiles/r06/bench_line_human_maxhits_m60*.json).
__device__ __forceinline__ uint64_t plan_boundary_below(const uint8_t* __restrict__ query, const SeedShape& sh, uint32_t tmask, const uint32_t* __restrict__ bucket_start,
                                                        const TdRec* __restrict__ c_rec, uint32_t m_lo, uint32_t m_hi, uint64_t hit_base, uint64_t limit) {
    // largest word boundary (chunk-relative inclusive hit prefix) strictly below `limit` (>= 1); 0 when there is none (the reference's
    // wrapped index of hazard H5, read as "no hits")
    uint32_t lo = m_lo, hi = m_hi;  // last non-empty position whose FIRST hit lies below the limit: position m_lo does (its first hit is 0)
    while (lo + 1 < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)c_rec[mid].prefix - hit_base < limit) lo = mid; else hi = mid;
    }
    const uint64_t at = (uint64_t)c_rec[lo].prefix - hit_base;  // = the boundary behind the previous position's last word (0 for the first)
    uint32_t key = 0;
    kmer_at(query, c_rec[lo].qpos, sh, key);
    uint64_t best = at, cum = at + bucket_len(bucket_start, key);
    if (cum < limit) best = cum;
    for (int t = 0; t < sh.weight; t++)
        if ((tmask >> t) & 1u) {
            cum += bucket_len(bucket_start, key ^ (2u << (2 * t)));
            if (cum < limit) best = cum;
        }
    return best;
}

__global__ void probe_plan_kernel(const uint8_t* __restrict__ query, SeedShape sh, uint32_t tmask, const uint32_t* __restrict__ bucket_start,
                                  const Tri* __restrict__ bounds, int nchunks, const TdRec* __restrict__ c_rec, uint64_t max_hits, int wrap32,
                                  TdPlan* __restrict__ plan, uint64_t* __restrict__ seg_end) {
    __shared__ uint64_t s_upto[TD_MAX_BOUNDS * TD_MAX_ITER];
    __shared__ uint32_t s_iter[TD_MAX_BOUNDS];
    const int c = threadIdx.x;
    if (c < nchunks) {
        const Tri lo = bounds[c], hi = bounds[c + 1];
        TdPlan p;
        p.hit_base = lo.hits;
        p.num_hits = hi.hits - lo.hits;
        p.num_valid = hi.valid - lo.valid;
        p.m_lo = lo.ne;
        p.m_hi = hi.ne;
        p.n_iter = 0;
        for (int i = 0; i < TD_MAX_ITER; i++) p.upto[i] = lo.hits;
        if (p.num_hits > 0) {
            uint32_t n_iter;
            uint64_t limit;
            if (p.num_hits < max_hits) { n_iter = 2; limit = p.num_hits; }                        // :721-724
            else { n_iter = (uint32_t)std::min<uint64_t>(p.num_hits / max_hits + 2, 0xFFFFu); limit = max_hits; }  // :725-728
            // (the src/ binary counts hits in uint32: a chunk whose counts could wrap there is left to the general path, which restates the wrap)
            if (n_iter > (uint32_t)TD_MAX_ITER || (wrap32 && p.num_hits + max_hits > 0xFFFFFFFFull)) {
                p.n_iter = TD_PLAN_OVERFLOW;
            } else {
                for (uint32_t i = 0; i + 1 < n_iter; i++) {                                          // :732-739
                    const uint64_t at = plan_boundary_below(query, sh, tmask, bucket_start, c_rec, lo.ne, hi.ne, lo.hits, limit);
                    p.upto[i] = lo.hits + at;
                    limit = std::min<uint64_t>(at + max_hits, p.num_hits);
                }
                p.upto[n_iter - 1] = lo.hits + p.num_hits;                                           // :741 (the last one takes what is left: H16)
                p.n_iter = n_iter;                                                                   // (:743 never fires: limit <= num_hits, so the
            }                                                                                        //  planned ends lie in front of the last word)
        }
        plan[c] = p;
        s_iter[c] = p.n_iter;
        for (int i = 0; i < TD_MAX_ITER; i++) s_upto[c * TD_MAX_ITER + i] = p.upto[i];
    }
    __syncthreads();
    // the segment (reference iteration) ends of the call, in hit order -- what the host derives from the same plans for its own
    // bookkeeping (core.hip); the candidate-stage kernels search this array (extend.hip seg_of).  (A call whose chunks need more than
    // MAX_SEGS iterations, or hold an overflowed plan, is not run table-direct: td_front halves it.)
    if (threadIdx.x == 0 && seg_end) {
        int n = 0;
        for (int k = 0; k < nchunks; k++) {
            const uint32_t it = s_iter[k];
            if (it == TD_PLAN_OVERFLOW) continue;
            for (uint32_t i = 0; i < it && n < MAX_SEGS; i++) seg_end[n++] = s_upto[k * TD_MAX_ITER + i];
        }
    }
}

// ONE clearing kernel per table-direct call: the words of the head-bit map the call will use (its hit total is only known on the
// device at this point) and the small device-side state the later stages expect zeroed -- counters, the sub-list counters of the
// second-level list, the chain buckets, the per-segment info of the dedup stage (five hipMemsetAsync launches before).
__global__ __launch_bounds__(256) void call_clear_kernel(const Tri* __restrict__ total, uint32_t* __restrict__ head_bits, uint32_t head_words,
                                                         ZeroList z) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    if (head_bits) {
        const uint64_t need = min((uint64_t)head_words, ((total->hits + 63) >> 6) * 2 + 16);  // (+ the words the filter reads ahead)
        for (uint64_t i = tid; i < need; i += nth) head_bits[i] = 0u;
    }
#pragma unroll
    for (int r = 0; r < ZeroList::N; r++)
        for (uint64_t i = tid; i < z.n[r]; i += nth) z.p[r][i] = 0u;
}

// workgroup sums [blocks] | total, pad | their exclusive prefixes [blocks]
size_t probe_partial_bytes(uint32_t n) { return (2 * ((size_t)(n + 3 + PR_TILE - 1) / PR_TILE) + 2) * sizeof(Tri); }
size_t probe_bounds_bytes() { return (size_t)TD_MAX_BOUNDS * sizeof(Tri); }

static inline uint32_t probe_blocks(uint32_t start, uint32_t n) { return (n + (start & 3u) + PR_TILE - 1) / PR_TILE; }

void launch_probe_lookup(const uint8_t* query, uint32_t start, uint32_t n, SeedShape sh, const uint64_t* nbr_start, uint32_t nkeys,
                         uint64_t* t_off, uint32_t* t_cnt, void* partial_buf, hipStream_t s) {
    hipLaunchKernelGGL(probe_kernel, dim3(probe_blocks(start, n)), dim3(PR_THREADS), 0, s, query, start, n, sh, nbr_start, nkeys, t_off, t_cnt,
                       reinterpret_cast<Tri*>(partial_buf));
}
void launch_probe_compact(uint32_t start, uint32_t n, const uint64_t* t_off, const uint32_t* t_cnt, void* partial_buf, void* bounds_buf,
                          TdRec* c_rec, uint32_t* chunk_rec, uint32_t chunk_cap, uint32_t* head_bits, uint32_t head_words,
                          const ZeroList& zero, const TdBounds& bpos, bool first_pass, hipStream_t s) {
    Tri* partial = reinterpret_cast<Tri*>(partial_buf);
    const uint32_t nblocks = probe_blocks(start, n);
    Tri* total = partial + nblocks;
    Tri* scanned = partial + nblocks + 2;
    // (only once per probe: a repeated compaction, after the head-bit map was regrown, finds the prefixes in place)
    if (first_pass)
        hipLaunchKernelGGL(probe_partials_kernel, dim3((nblocks + PR_THREADS * PP_ITEMS - 1) / (PR_THREADS * PP_ITEMS)), dim3(PR_THREADS), 0, s, partial,
                           nblocks, scanned, total);
    hipLaunchKernelGGL(call_clear_kernel, dim3(head_bits ? 1024 : 64), dim3(256), 0, s, total, head_bits, head_words, zero);
    hipLaunchKernelGGL(probe_compact_kernel, dim3(nblocks), dim3(PR_THREADS), 0, s, start, n, t_off, t_cnt, scanned, total, c_rec, chunk_rec,
                       chunk_cap, head_bits, head_words, bpos, reinterpret_cast<Tri*>(bounds_buf));
}
void launch_call_clear(const ZeroList& zero, hipStream_t s) {  // the clearing kernel alone (key-ordered calls: no head-bit map)
    hipLaunchKernelGGL(call_clear_kernel, dim3(64), dim3(256), 0, s, (const Tri*)nullptr, (uint32_t*)nullptr, 0u, zero);
}
void launch_probe_plan(const uint8_t* query, SeedShape sh, uint32_t tmask, const uint32_t* bucket_start, const void* bounds_buf, int nchunks,
                       const TdRec* c_rec, uint64_t max_hits, int wrap32, TdPlan* plan, uint64_t* seg_end, hipStream_t s) {
    hipLaunchKernelGGL(probe_plan_kernel, dim3(1), dim3((nchunks + 63) / 64 * 64), 0, s, query, sh, tmask, bucket_start, reinterpret_cast<const Tri*>(bounds_buf),
                       nchunks, c_rec, max_hits, wrap32, plan, seg_end);
}

}  // namespace sa
