// join.hip -- preparation of a KEY-ORDERED table-direct call (join.h; the filter itself is extend.hip 1e).
//
// Reference shape (src/seeder.cpp:57-74 + src/seed_filter.cu:157-230): per query position 13 seed words, per word a bucket, per
// bucket entry a hit, hits in query order.  The streamed class filter (extend.hip 1d) keeps that order: every hit moves its 32-byte
// context record through the memory system once, and a key's run is fetched again for every position that carries the key.
// Here the positions of a call are SORTED BY KEY first -- with the table build's own LDS-staged radix partition (table.hip), the
// query strand in place of the target -- so that the hits of a key are a rectangle (run entries x positions):
//   join_stats    per chunk of the call: seed hits, valid positions, last non-empty position      (what the iteration plan needs)
//   join_plan     per chunk: the reference's iteration split (src/seed_filter.cu:718-745 for num_hits < MAX_HITS), as
//                 (p_last, e_thr): a hit is in the chunk's second iteration iff it sits at p_last at or behind run entry e_thr
//   join_count / join_layout / join_scatter   one ENTRY per (key, up to 16 of its positions), grouped by the number of positions c
//                 ("class"): all lanes of a wave then run the same number of steps
//   join_finish   class bases in the virtual record index space, work units per class, the dynamic work cursor
//   join_qx       per position (in key order): the field words of its two query windows (QRecX) -- the query half of every LDS
//                 address the filter will need, so that an address is ONE xor in the filter
#include "join.h"
#include "kmer_dev.h"

namespace sa {

constexpr int JS_THREADS = 1024;
constexpr int JS_KEYS = 4;        // keys per thread
constexpr int JS_MAXCH = (int)JOIN_SEG_FIRST;  // chunks of a call (= SA_MAX_CHUNKS, asserted in engine_internal.h)

// ---- per-chunk statistics, from the key-sorted list: a key's run length is read once per key (sequentially), not once per position ----
__global__ __launch_bounds__(JS_THREADS) void join_stats_kernel(const uint32_t* __restrict__ qk_start, const uint32_t* __restrict__ qpos,
                                                                const uint64_t* __restrict__ nbr_start, uint32_t nkeys, uint32_t start, uint32_t chunk, int K,
                                                                unsigned long long* __restrict__ hits, uint32_t* __restrict__ valid, uint32_t* __restrict__ last1) {
    __shared__ unsigned long long s_hits[JS_MAXCH];
    __shared__ uint32_t s_valid[JS_MAXCH], s_last[JS_MAXCH];
    for (int i = threadIdx.x; i < K; i += JS_THREADS) { s_hits[i] = 0; s_valid[i] = 0; s_last[i] = 0; }
    __syncthreads();
    const uint32_t k0 = (blockIdx.x * JS_THREADS + threadIdx.x) * JS_KEYS;
#pragma unroll
    for (int j = 0; j < JS_KEYS; j++) {
        const uint32_t k = k0 + j;
        if (k >= nkeys) break;
        const uint32_t qa = qk_start[k], qb = qk_start[k + 1];
        if (qa == qb) continue;
        const uint32_t n_t = (uint32_t)(nbr_start[k + 1] - nbr_start[k]);
        for (uint32_t i = qa; i < qb; i++) {
            const uint32_t p = qpos[i];
            const uint32_t c = (p - start) / chunk;
            if (c >= (uint32_t)K) continue;  // (cannot happen: the front sorts exactly the positions of the call's K chunks)
            atomicAdd(&s_valid[c], 1u);
            if (n_t) {
                atomicAdd(&s_hits[c], (unsigned long long)n_t);
                atomicMax(&s_last[c], p + 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += JS_THREADS) {
        if (s_valid[i]) atomicAdd(&valid[i], s_valid[i]);
        if (s_hits[i]) { atomicAdd(&hits[i], s_hits[i]); atomicMax(&last1[i], s_last[i]); }
    }
}

__device__ __forceinline__ uint32_t jbucket_len(const uint32_t* __restrict__ bucket_start, uint32_t key) { return bucket_start[key + 1] - bucket_start[key]; }

// one lane per chunk: the split of the chunk's hits into the reference's two iterations, and the segment numbering
__global__ __launch_bounds__(JS_MAXCH) void join_plan_kernel(const uint8_t* __restrict__ query, SeedShape sh, uint32_t tmask, const uint32_t* __restrict__ bucket_start,
                                                             const unsigned long long* __restrict__ hits, const uint32_t* __restrict__ valid,
                                                             const uint32_t* __restrict__ last1, int K, JoinChunk* __restrict__ plan, uint64_t* __restrict__ seg_table,
                                                             JoinHead* __restrict__ head) {
    __shared__ uint32_t s_has[JS_MAXCH];
    const int c = threadIdx.x;
    JoinChunk p;
    p.hits = 0; p.valid = 0; p.p_last = 0; p.e_thr = 0; p.seg0 = 0;
    if (c < K) {
        p.hits = hits[c];
        p.valid = valid[c];
        if (p.hits > 0) {
            // the last hit-bearing seed word of the chunk (:732-739 with limit = num_hits) lives in the last non-empty position: walk
            // that position's words in emission order (seeder.cpp:60-69) over the PLAIN buckets
            p.p_last = last1[c] - 1u;
            uint32_t key = 0;
            kmer_at(query, p.p_last, sh, key);
            uint32_t before = jbucket_len(bucket_start, key), before_last = 0;
            for (int t = 0; t < sh.weight; t++)
                if ((tmask >> t) & 1u) {
                    const uint32_t nt = jbucket_len(bucket_start, key ^ (2u << (2 * t)));
                    if (nt) before_last = before;
                    before += nt;
                }
            p.e_thr = before_last;
        }
    }
    s_has[c] = (c < K && p.hits > 0) ? 1u : 0u;
    __syncthreads();
    uint32_t seg0 = 0;
    for (int i = 0; i < c; i++) seg0 += 2u * s_has[i];  // (K <= 256: a serial prefix per lane is nothing)
    p.seg0 = seg0;
    if (c < K) plan[c] = p;
    // what the candidate stages stage in LDS (extend.hip seg_of): {p_last : e_thr} per chunk, then seg0 per chunk
    seg_table[c] = (uint64_t)p.p_last | ((uint64_t)p.e_thr << 32);
    seg_table[JS_MAXCH + c] = p.seg0;
    if (c == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < K; i++) tot += hits[i];
        head->total_hits = tot;
    }
}

// ---- entries by class ----
// entries of a key with n_q positions and a non-empty run: n_q / CMAX of class CMAX, one of class n_q % CMAX
template <bool SCATTER>
__global__ __launch_bounds__(JS_THREADS) void join_entries_kernel(const uint32_t* __restrict__ qk_start, const uint64_t* __restrict__ nbr_start, uint32_t nkeys,
                                                                  JoinHead* __restrict__ head, uint4* __restrict__ ent, uint32_t* __restrict__ ent_nt, uint32_t ent_cap) {
    __shared__ uint32_t s_cnt[JOIN_CMAX + 1], s_base[JOIN_CMAX + 1];
    if (threadIdx.x <= JOIN_CMAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t k0 = (blockIdx.x * JS_THREADS + threadIdx.x) * JS_KEYS;
    uint32_t qa[JS_KEYS], nq[JS_KEYS], nt[JS_KEYS];
    uint64_t rb[JS_KEYS];
#pragma unroll
    for (int j = 0; j < JS_KEYS; j++) {
        const uint32_t k = k0 + j;
        qa[j] = nq[j] = nt[j] = 0;
        rb[j] = 0;
        if (k >= nkeys) continue;
        qa[j] = qk_start[k];
        nq[j] = qk_start[k + 1] - qa[j];
        if (!nq[j]) continue;
        rb[j] = nbr_start[k];
        nt[j] = (uint32_t)(nbr_start[k + 1] - rb[j]);
        if (!nt[j]) { nq[j] = 0; continue; }
        const uint32_t full = nq[j] / JOIN_CMAX, rest = nq[j] % JOIN_CMAX;
        if (full) atomicAdd(&s_cnt[JOIN_CMAX], full);
        if (rest) atomicAdd(&s_cnt[rest], 1u);
    }
    __syncthreads();
    if (!SCATTER) {
        if (threadIdx.x >= 1 && threadIdx.x <= JOIN_CMAX && s_cnt[threadIdx.x]) atomicAdd(&head->cls_count[threadIdx.x], s_cnt[threadIdx.x]);
        return;
    }
    // reserve this workgroup's stretch of every class, then hand out slots inside it
    if (threadIdx.x >= 1 && threadIdx.x <= JOIN_CMAX) {
        s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&head->cls_cursor[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JS_KEYS; j++) {
        if (!nq[j]) continue;
        uint32_t left = nq[j], qf = qa[j];
        while (left) {
            const uint32_t c = left < (uint32_t)JOIN_CMAX ? left : (uint32_t)JOIN_CMAX;
            const uint32_t slot = s_base[c] + atomicAdd(&s_cnt[c], 1u);
            if (slot < ent_cap) {
                ent[slot] = make_uint4((uint32_t)rb[j], (uint32_t)(rb[j] >> 32), qf, nt[j]);
                ent_nt[slot] = nt[j];
            }
            qf += c;
            left -= c;
        }
    }
}

__global__ void join_layout_kernel(JoinHead* __restrict__ head) {
    if (threadIdx.x != 0) return;
    uint32_t run = 0;
    head->cls_first[0] = 0;
    for (int c = 1; c <= JOIN_CMAX; c++) {
        head->cls_first[c] = run;
        head->cls_cursor[c] = run;
        run += head->cls_count[c];
    }
    head->cls_first[JOIN_CMAX + 1] = run;
    head->n_entries = run;
}

__global__ void join_finish_kernel(JoinHead* __restrict__ head, const unsigned long long* __restrict__ vstart) {
    if (threadIdx.x != 0) return;
    for (int c = 1; c <= JOIN_CMAX + 1; c++) head->vbase[c] = vstart[head->cls_first[c]];
    unsigned long long wb = 0;
    for (int c = JOIN_CMAX; c >= 1; c--) {
        head->work_base[c] = wb;
        const unsigned long long tot = head->vbase[c + 1] - head->vbase[c];
        wb += ((tot + 63ull) >> 6) * (unsigned long long)c;
    }
    head->work_total = wb;
    head->work_next = 0;
}

// ---- QRecX: the query half of the filter's LDS addresses ----
// field word of the six-base field that starts at bit `bit` of the little-endian dword string w[0..nw): its byte offset in the table
__device__ __forceinline__ uint32_t jfield6(const uint32_t* w, int nw, int bit) {
    const int d = bit >> 5, o = bit & 31;
    const uint64_t two = (uint64_t)w[d] | ((uint64_t)(d + 1 < nw ? w[d + 1] : 0u) << 32);
    return (uint32_t)((two >> o) << 2) & 0x3FFCu;
}
__device__ __forceinline__ uint4 jload16(const uint8_t* p) {
    uint4 t;
    __builtin_memcpy(&t, __builtin_assume_aligned(p, 4), 16);
    return t;
}

__global__ __launch_bounds__(256) void join_qx_kernel(const uint32_t* __restrict__ qk_start, uint32_t nkeys, const uint32_t* __restrict__ qpos,
                                                      const uint8_t* __restrict__ q2_own, const uint8_t* __restrict__ q2_other, uint32_t query_len,
                                                      uint32_t seed_size, uint32_t left_skip, uint32_t* __restrict__ qx) {
    const uint32_t n = qk_start[nkeys];  // valid positions of the call
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = qpos[i];
        const uint32_t query_loc = p + seed_size;  // :204
        // the windows out of the unshifted 2-bit copies (extend.hip 1d, ONE_COPY): 54 bases from query_loc on this strand, 58 bases from
        // len - query_loc + seed_size on the other strand (= the bases left of the SEED in walking order, complemented: CtxRec)
        const uint8_t* rpp = q2_own + ((query_loc >> 4) << 2);
        const uint4 t = jload16(rpp);
        uint32_t t4;
        __builtin_memcpy(&t4, __builtin_assume_aligned(rpp + 16, 4), 4);
        const uint32_t sr = (query_loc & 15u) << 1;
        const uint32_t r0 = __builtin_amdgcn_alignbit(t.y, t.x, sr), r1 = __builtin_amdgcn_alignbit(t.z, t.y, sr), r2 = __builtin_amdgcn_alignbit(t.w, t.z, sr),
                       r3 = __builtin_amdgcn_alignbit(t4, t.w, sr);
        const uint32_t lp = query_len - query_loc + left_skip;  // (the left context starts in front of the seed: CtxRec)
        const uint8_t* lpp = q2_other + ((lp >> 4) << 2);
        const uint4 u = jload16(lpp);
        uint32_t u4;
        __builtin_memcpy(&u4, __builtin_assume_aligned(lpp + 16, 4), 4);
        const uint32_t sl = (lp & 15u) << 1;
        const uint32_t l0 = __builtin_amdgcn_alignbit(u.y, u.x, sl), l1 = __builtin_amdgcn_alignbit(u.z, u.y, sl), l2 = __builtin_amdgcn_alignbit(u.w, u.z, sl),
                       l3 = __builtin_amdgcn_alignbit(u4, u.w, sl);
        // the query's side of the record's 224-bit string (CtxRec: 54 bases of this strand, then 58 of the other), and its field words
        const uint32_t w[7] = {r0, r1, r2, (r3 & 0xFFFu) | (l0 << 12), __builtin_amdgcn_alignbit(l1, l0, 20), __builtin_amdgcn_alignbit(l2, l1, 20),
                               __builtin_amdgcn_alignbit(l3, l2, 20)};
        uint32_t o[JOIN_QX_DW];
        o[0] = p;
#pragma unroll
        for (int k = 0; k < 18; k++) o[1 + k] = jfield6(w, 7, 12 * k);
        o[19] = (w[6] >> 22) & 0x3FCu;  // the four-base tail field (its table offset is on the record side)
        uint4* dst = reinterpret_cast<uint4*>(qx + (size_t)i * JOIN_QX_DW);
#pragma unroll
        for (int k = 0; k < JOIN_QX_DW / 4; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
}

// ---- launchers ----
static inline uint32_t key_blocks(uint32_t nkeys) { return (nkeys + JS_THREADS * JS_KEYS - 1) / (JS_THREADS * JS_KEYS); }

void launch_join_stats(const uint32_t* qk_start, const uint32_t* qpos, const uint64_t* nbr_start, uint32_t nkeys, uint32_t start, uint32_t chunk, int K,
                       unsigned long long* hits, uint32_t* valid, uint32_t* last1, hipStream_t s) {
    hipLaunchKernelGGL(join_stats_kernel, dim3(key_blocks(nkeys)), dim3(JS_THREADS), 0, s, qk_start, qpos, nbr_start, nkeys, start, chunk, K, hits, valid, last1);
}
void launch_join_plan(const uint8_t* query, SeedShape sh, uint32_t tmask, const uint32_t* bucket_start, const unsigned long long* hits, const uint32_t* valid,
                      const uint32_t* last1, int K, JoinChunk* plan, uint64_t* seg_table, JoinHead* head, hipStream_t s) {
    hipLaunchKernelGGL(join_plan_kernel, dim3(1), dim3(JS_MAXCH), 0, s, query, sh, tmask, bucket_start, hits, valid, last1, K, plan, seg_table, head);
}
void launch_join_entries(const uint32_t* qk_start, const uint64_t* nbr_start, uint32_t nkeys, JoinHead* head, uint4* ent, uint32_t* ent_nt, uint32_t ent_cap,
                         hipStream_t s) {
    hipLaunchKernelGGL(join_entries_kernel<false>, dim3(key_blocks(nkeys)), dim3(JS_THREADS), 0, s, qk_start, nbr_start, nkeys, head, ent, ent_nt, ent_cap);
    hipLaunchKernelGGL(join_layout_kernel, dim3(1), dim3(64), 0, s, head);
    hipLaunchKernelGGL(join_entries_kernel<true>, dim3(key_blocks(nkeys)), dim3(JS_THREADS), 0, s, qk_start, nbr_start, nkeys, head, ent, ent_nt, ent_cap);
}
void launch_join_finish(JoinHead* head, const unsigned long long* vstart, hipStream_t s) {
    hipLaunchKernelGGL(join_finish_kernel, dim3(1), dim3(64), 0, s, head, vstart);
}
void launch_join_qx(const uint32_t* qk_start, uint32_t nkeys, const uint32_t* qpos, const uint8_t* q2_own, const uint8_t* q2_other, uint32_t query_len,
                    uint32_t seed_size, uint32_t left_skip, uint32_t* qx, hipStream_t s) {
    hipLaunchKernelGGL(join_qx_kernel, dim3(8192), dim3(256), 0, s, qk_start, nkeys, qpos, q2_own, q2_other, query_len, seed_size, left_skip, qx);
}

}  // namespace sa
