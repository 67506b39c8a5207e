// join_proto.hip -- prototype of the KEY-ORDERED class filter: what does the X-drop class filter cost when
// the hits of a call are enumerated as a JOIN per seed key -- (the key's run of context records) x (the query positions that carry the
// key) -- instead of as a stream of records in query order?
//   * a lane holds ONE context record in registers and scores it against the c query positions of its key, one per loop step: the
//     record is fetched once for c hits (HBM bytes per hit: 32 / c instead of 32), the query windows come pre-shifted (QRec);
//   * keys are grouped by c (entries of class c), so every lane of a wave runs the same number of steps;
//   * the class table is replicated per LDS bank: a lane only ever reads "its" bank, no conflicts (today: 70 % of the LDS cycles).
// Synthetic data (hashed records, Poisson run lengths); the forwarded records are checked against a host emulation on a small case.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o join_proto join_proto.hip
// Run:   join_proto [log2 keys=24] [mean run=73] [mean q=3] [check=0]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct CtxRec { uint32_t pos, r[3], l[4]; };
struct QRec { uint32_t qpos, r[3], l[4]; };
struct L2Rec { uint32_t ref_loc, query_loc, hidx, state, meta; };
struct JEnt { uint32_t run_lo, run_hi, q_first, n_t; };

constexpr int CMAX = 16;
constexpr int L2_NSUB = 256, L2_CNT_STRIDE = 32;
constexpr int STAGE_FLUSH = 32, STAGE_CAP = STAGE_FLUSH - 1 + 64 + 1;

struct JoinArgs {
    const uint4* ctx;
    const uint4* qrec;
    const unsigned long long* p_end;  // [entries] class-major: exclusive end of the entry's run inside its class's virtual index space
    const uint4* ent;                 // [entries] JEnt
    uint32_t cls_first[CMAX + 2];     // first entry of class c (c = 1..CMAX), cls_first[CMAX + 1] = number of entries
    unsigned long long cls_total[CMAX + 1];  // virtual indices (context records) of class c
    unsigned long long work_base[CMAX + 2];  // work units (tile steps) in front of class c, classes taken from CMAX down to 1
    unsigned long long work_total;
    int cls[4];
    int xdrop, hspthresh;
    uint32_t seed_size;
    L2Rec* l2_list;
    uint32_t* l2_count;
    uint32_t l2_cap;
    unsigned long long* n_hits;  // (check) hits scored
    unsigned long long* work_next;  // dynamic distribution: next unclaimed work unit (variant 2 with GRAIN > 0)
    uint32_t grain;
};

__host__ __device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// synthetic records: the 19 bases left of the anchor (the seed) agree between a key's target and query records
__host__ __device__ inline void make_rec(uint32_t key, uint64_t idx, uint32_t salt, uint32_t out[8]) {
    const uint32_t h = hash32((uint32_t)idx * 2654435761u + salt) ^ hash32((uint32_t)(idx >> 32) + 77u * salt);
    out[0] = (uint32_t)idx;
    for (int i = 1; i < 8; i++) out[i] = hash32(h + (uint32_t)i * 0x9E3779B9u);
    const uint32_t s0 = hash32(key * 0x85EBCA6Bu + 1u), s1 = hash32(key * 0xC2B2AE35u + 2u);
    out[4] = s0;                                         // l[0]: bases 1..16 left of the anchor
    out[5] = (out[5] & ~0x3Fu) | (s1 & 0x3Fu);           // l[1]: bases 17..19
}

__global__ void fill_kernel(uint4* out, const uint32_t* key_of_idx_dummy, uint64_t n, uint32_t salt, const uint64_t* start, uint32_t nkeys) {
    // one thread per record; the key of record idx is found by binary search in start[]
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nkeys;  // largest k with start[k] <= idx
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (start[mid] <= idx) lo = mid; else hi = mid; }
        uint32_t r[8];
        make_rec(lo, idx, salt, r);
        out[2 * idx] = make_uint4(r[0], r[1], r[2], r[3]);
        out[2 * idx + 1] = make_uint4(r[4], r[5], r[6], r[7]);
    }
}

// ---- the class table: entry {sum : sum - mx} of a field of FB bases, REP replicas: entry i of replica r at dword i * REP + r ----
__host__ __device__ inline uint32_t cls_entry(int f, int nb, const int* cls) {
    int sum = 0, mx = 0;
    for (int k = 0; k < nb; k++) {
        const int x = (f >> (2 * k)) & 3;
        sum += cls[x];
        mx = sum > mx ? sum : mx;
    }
    return ((uint32_t)sum << 16) | ((uint32_t)(sum - mx) & 0xFFFFu);
}

__device__ __forceinline__ void cls_step(const uint32_t* __restrict__ s_tab, uint32_t addr, uint32_t& P, uint32_t& W) {
    const uint32_t e = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_tab) + addr);
    asm("v_pk_add_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(P) : "v"(P), "v"(e));
    asm("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(P) : "v"(e));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    W = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, W), __builtin_bit_cast(s16x2, P)));
}

// byte address of the field of FB bases that starts at bit O of w0 | w1 << 32, in a table whose entries are 4 << SH bytes apart
// (SH = log2 REP), plus the lane's replica offset
template <int O, int FB, int SH>
__device__ __forceinline__ uint32_t field_addr(uint32_t w0, uint32_t w1, uint32_t lane_off) {
    constexpr uint32_t MASK = ((1u << (2 * FB)) - 1u) << (2 + SH);
    constexpr int S = O - (2 + SH);  // right shift that puts the field at bit 2 + SH
    uint32_t v;
    if (O + 2 * FB <= 32) v = S >= 0 ? (w0 >> (S >= 0 ? S : 0)) : (w0 << (S < 0 ? -S : 0));
    else if (S >= 0) v = __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(S >= 0 ? S : 0));
    else v = (w0 >> O) << (2 + SH) | (w1 << (32 - O + 2 + SH));  // (not reached with the field sizes used here)
    // (v & MASK) | lane_off in ONE instruction; the mask sits in a scalar register (gfx9 VOP3 takes no literal)
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(MASK), "v"(lane_off));
    return r;
}

template <int FB, int SH, int K, int NF>
struct Walk {  // fields K .. NF-1 of a string of NW dwords w[]
    template <int NW>
    static __device__ __forceinline__ void run(const uint32_t* __restrict__ s_tab, const uint32_t (&w)[NW], uint32_t lane_off, uint32_t& P, uint32_t& W) {
        if constexpr (K < NF) {
            constexpr int bit = K * 2 * FB, d = bit / 32, o = bit % 32;
            const uint32_t w0 = w[d], w1 = (d + 1 < NW) ? w[d + 1 < NW ? d + 1 : d] : 0u;
            cls_step(s_tab, field_addr<o, FB, SH>(w0, w1, lane_off), P, W);
            Walk<FB, SH, K + 1, NF>::run(s_tab, w, lane_off, P, W);
        }
    }
};

template <int FB, int REP>
__global__ __launch_bounds__(1024) void join_filter_kernel(JoinArgs a) {
    constexpr int SH = REP == 1 ? 0 : REP == 2 ? 1 : REP == 4 ? 2 : REP == 8 ? 3 : REP == 16 ? 4 : 5;
    constexpr int NTAB = 1 << (2 * FB);
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_tab = s_dyn;                                                   // NTAB * REP dwords
    L2Rec* s_stage = reinterpret_cast<L2Rec*>(s_dyn + NTAB * REP);             // [waves][STAGE_CAP]
    for (int i = threadIdx.x; i < NTAB * REP; i += blockDim.x) s_tab[i] = cls_entry(i >> SH, FB, a.cls);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t lane_off = (uint32_t)(lane & (REP - 1)) << 2;
    L2Rec* stage = s_stage + (threadIdx.x >> 6) * STAGE_CAP;
    int n_stage = 0;
    const uint64_t NWV = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wid = (uint64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint32_t my_sub = (uint32_t)wid & (L2_NSUB - 1);
    L2Rec* __restrict__ my_list = a.l2_list + (size_t)my_sub * a.l2_cap;
    uint32_t* __restrict__ my_count = a.l2_count + my_sub * L2_CNT_STRIDE;
    const uint64_t w_lo = wid * a.work_total / NWV, w_hi = (wid + 1) * a.work_total / NWV;
    const int xdrop = a.xdrop;
    unsigned long long scored = 0;

    for (int c = CMAX; c >= 1; c--) {
        const uint64_t B = a.work_base[c], tot = a.cls_total[c];
        const uint64_t ntiles = (tot + 63) >> 6;
        if (ntiles == 0 || w_hi <= B || w_lo >= B + ntiles * (uint64_t)c) continue;
        const uint64_t m_lo = w_lo <= B ? 0 : std::min<uint64_t>((w_lo - B + c - 1) / c, ntiles);
        const uint64_t m_hi = std::min<uint64_t>((w_hi - B + c - 1) / c, ntiles);
        if (m_lo >= m_hi) continue;
        const uint32_t e_first = a.cls_first[c], e_last = a.cls_first[c + 1];  // entries of the class (class-major list: class c + 1 follows c)
        // the entry that holds virtual index 64 * m_lo: first e with p_end[e] > v (uniform binary search, once per class and wave)
        uint32_t e_cur;
        {
            const uint64_t v0 = m_lo << 6;
            uint32_t lo = e_first, hi = e_last - 1;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.p_end[mid] > v0) hi = mid; else lo = mid + 1; }
            e_cur = lo;
        }
        for (uint64_t m = m_lo; m < m_hi; m++) {
            const uint64_t v = (m << 6) + (uint64_t)lane;
            const bool valid = v < tot;
            // ---- which entry? ----
            uint32_t e = e_cur;
            for (;;) {
                const uint32_t ej = min(e_cur + (uint32_t)(lane & 15), e_last - 1);
                const uint64_t pe = a.p_end[ej];
                bool more = false;
#pragma unroll 1
                for (int j = 0; j < 16; j++) {
                    const uint64_t pj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(pe >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pe, j);
                    if (pj > (m << 6) + 63u || e_cur + (uint32_t)j >= e_last - 1) break;  // (uniform) entry j reaches past the tile: nobody moves on
                    e += (v >= pj) ? 1u : 0u;
                    more = j == 15;
                }
                if (!more) break;
                e_cur += 16;
            }
            e = min(e, e_last - 1);
            const uint4 en = a.ent[e];
            const uint64_t p_start = a.p_end[e] - en.w;
            const uint64_t rec = (((uint64_t)en.y << 32) | en.x) + (valid ? v - p_start : 0);
            const uint32_t tidx = (uint32_t)(v - p_start);
            e_cur = (uint32_t)__builtin_amdgcn_readlane((int)e, 63);
            const uint4 c0 = a.ctx[2 * rec], tl = a.ctx[2 * rec + 1];
            const uint32_t ref_loc = c0.x + a.seed_size;
            uint32_t qoff = en.z;  // index of the key's first QRec
            uint4 qa = a.qrec[2 * (size_t)qoff], qb = a.qrec[2 * (size_t)qoff + 1];
            for (int qi = 0; qi < c; qi++) {
                const uint4 q0 = qa, q1 = qb;
                if (qi + 1 < c) {
                    qoff++;
                    qa = a.qrec[2 * (size_t)qoff];
                    qb = a.qrec[2 * (size_t)qoff + 1];
                }
                const uint32_t xr[3] = {c0.y ^ q0.y, c0.z ^ q0.z, c0.w ^ q0.w};
                const uint32_t xl[4] = {tl.x ^ q1.x, tl.y ^ q1.y, tl.z ^ q1.z, tl.w ^ q1.w};
                uint32_t P = 0, Wd = 0;
                Walk<FB, SH, 0, (48 + FB - 1) / FB>::run(s_tab, xr, lane_off, P, Wd);
                const bool r_alive = (int)(short)(Wd & 0xFFFFu) >= -xdrop;
                const int bestR = ((int)P >> 16) - (int)(short)(P & 0xFFFFu);
                P = 0; Wd = 0;
                Walk<FB, SH, 0, (64 + FB - 1) / FB>::run(s_tab, xl, lane_off, P, Wd);
                const bool l_alive = (int)(short)(Wd & 0xFFFFu) >= -xdrop;
                const int bestL = ((int)P >> 16) - (int)(short)(P & 0xFFFFu);
                const bool fwd = valid && (r_alive || l_alive || bestR + bestL >= a.hspthresh);
                if (a.n_hits) scored += valid ? 1 : 0;
                const unsigned long long fm = __ballot(fwd);
                if (fm) {
                    L2Rec cr;
                    cr.ref_loc = ref_loc;
                    cr.query_loc = q0.x + a.seed_size;
                    cr.hidx = tidx;
                    cr.state = (P & 0xFFFF0000u) | ((0u - P) & 0xFFFFu);
                    const uint32_t fl = (r_alive ? 1u : 0u) | (l_alive ? 2u : 0u);
                    cr.meta = (uint32_t)(r_alive ? bestL : bestR) | ((fl ? fl : 3u) << 16);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                    if (fwd) stage[n_stage + (int)rank] = cr;
                    n_stage += __popcll(fm);
                    __builtin_amdgcn_wave_barrier();
                    if (n_stage >= STAGE_FLUSH) {
                        const int k = n_stage < 64 ? n_stage : 64;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(my_count, (uint32_t)k);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (lane < k && base + (uint32_t)lane < a.l2_cap) my_list[base + (uint32_t)lane] = stage[lane];
                        const int rest = n_stage - k;
                        L2Rec tmp = cr;
                        if (lane < rest) tmp = stage[k + lane];
                        __builtin_amdgcn_wave_barrier();
                        if (lane < rest) stage[lane] = tmp;
                        __builtin_amdgcn_wave_barrier();
                        n_stage = rest;
                    }
                }
            }
        }
    }
    if (n_stage > 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(my_count, (uint32_t)n_stage);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < n_stage && base + (uint32_t)lane < a.l2_cap) my_list[base + (uint32_t)lane] = stage[lane];
    }
    if (a.n_hits && scored) atomicAdd(a.n_hits, scored);
}

// =====================================================================================================================
// variant 2: FIELD WORDS.  shift and mask are linear over xor, so the LDS address of field k of (t ^ q) is
//     fw_k(t) ^ fw_k(q),   fw_k(w) = (w >> (bit_k - 2 - SH)) & MASK
// The lane computes fw_k(t) | lane_off of its record ONCE per tile (registers), the query side stores fw_k(q) with the position
// (QRecX: 1 + NF dwords, built once per call): ONE v_xor per field address instead of shift + and + or.  Right and left walks are
// interleaved (two independent dependency chains).
// =====================================================================================================================
template <int FB, int SH, int NBASES, int NW>
__device__ __forceinline__ void field_words(const uint32_t (&w)[NW], uint32_t lane_off, uint32_t* out) {
    constexpr int NF = (NBASES + FB - 1) / FB;
    constexpr uint32_t MASK = ((1u << (2 * FB)) - 1u) << (2 + SH);
#pragma unroll
    for (int k = 0; k < NF; k++) {
        const int bit = k * 2 * FB, d = bit / 32, o = bit % 32;
        const uint64_t two = (uint64_t)w[d] | ((uint64_t)(d + 1 < NW ? w[d + 1 < NW ? d + 1 : d] : 0u) << 32);
        const int S = o - (2 + SH);
        const uint32_t v = S >= 0 ? (uint32_t)(two >> (S >= 0 ? S : 0)) : (uint32_t)(two << (S < 0 ? -S : 0));
        out[k] = (v & MASK) | lane_off;
    }
}

template <int FB>
struct QX {  // layout of a QRecX
    static constexpr int NFR = (48 + FB - 1) / FB, NFL = (64 + FB - 1) / FB;
    static constexpr int DW = ((1 + NFR + NFL) + 3) & ~3;  // dwords, a multiple of four
};

template <int FB, int REP>
__global__ void qx_build_kernel(const uint4* __restrict__ qrec, uint64_t n, uint32_t* __restrict__ qx) {
    constexpr int SH = REP == 1 ? 0 : REP == 2 ? 1 : REP == 4 ? 2 : REP == 8 ? 3 : REP == 16 ? 4 : 5;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 a = qrec[2 * i], b = qrec[2 * i + 1];
        const uint32_t r[3] = {a.y, a.z, a.w}, l[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[QX<FB>::DW];
        for (int k = 0; k < QX<FB>::DW; k++) o[k] = 0;
        o[0] = a.x;
        field_words<FB, SH, 48>(r, 0u, o + 1);
        field_words<FB, SH, 64>(l, 0u, o + 1 + QX<FB>::NFR);
        uint4* dst = reinterpret_cast<uint4*>(qx + i * QX<FB>::DW);
#pragma unroll
        for (int k = 0; k < QX<FB>::DW / 4; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
}

template <bool TRACKW>
__device__ __forceinline__ void cls_step2(const uint32_t* __restrict__ s_tab, uint32_t addr, uint32_t& P, uint32_t& W) {
    const uint32_t e = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_tab) + addr);
    asm("v_pk_add_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(P) : "v"(P), "v"(e));
    asm("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(P) : "v"(e));
    if (TRACKW) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        W = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, W), __builtin_bit_cast(s16x2, P)));
    }
}

template <int FB, int REP, bool TRACKW, int THREADS>
__global__ __launch_bounds__(THREADS) void join_filter2_kernel(JoinArgs a, const uint32_t* __restrict__ qx) {
    constexpr int SH = REP == 1 ? 0 : REP == 2 ? 1 : REP == 4 ? 2 : REP == 8 ? 3 : REP == 16 ? 4 : 5;
    constexpr int NTAB = 1 << (2 * FB);
    constexpr int NFR = QX<FB>::NFR, NFL = QX<FB>::NFL, DW = QX<FB>::DW;
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_tab = s_dyn;
    L2Rec* s_stage = reinterpret_cast<L2Rec*>(s_dyn + NTAB * REP);
    for (int i = threadIdx.x; i < NTAB * REP; i += blockDim.x) s_tab[i] = cls_entry(i >> SH, FB, a.cls);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t lane_off = (uint32_t)(lane & (REP - 1)) << 2;
    L2Rec* stage = s_stage + (threadIdx.x >> 6) * STAGE_CAP;
    int n_stage = 0;
    const uint64_t NWV = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wid = (uint64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint32_t my_sub = (uint32_t)wid & (L2_NSUB - 1);
    L2Rec* __restrict__ my_list = a.l2_list + (size_t)my_sub * a.l2_cap;
    uint32_t* __restrict__ my_count = a.l2_count + my_sub * L2_CNT_STRIDE;
    uint64_t w_lo = wid * a.work_total / NWV, w_hi = (wid + 1) * a.work_total / NWV;
    const int xdrop = a.xdrop;
    unsigned long long scored = 0;

    for (bool first = true;; first = false) {
    if (a.grain) {  // dynamic: a wave claims `grain` work units at a time (the per-tile overhead differs by class: static shares finish unevenly)
        unsigned long long g = 0;
        if (lane == 0) g = atomicAdd(a.work_next, (unsigned long long)a.grain);
        g = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g);
        if (g >= a.work_total) break;
        w_lo = g;
        w_hi = std::min<uint64_t>(g + a.grain, a.work_total);
    } else if (!first) break;
    for (int c = CMAX; c >= 1; c--) {
        const uint64_t B = a.work_base[c], tot = a.cls_total[c];
        const uint64_t ntiles = (tot + 63) >> 6;
        if (ntiles == 0 || w_hi <= B || w_lo >= B + ntiles * (uint64_t)c) continue;
        const uint64_t m_lo = w_lo <= B ? 0 : std::min<uint64_t>((w_lo - B + c - 1) / c, ntiles);
        const uint64_t m_hi = std::min<uint64_t>((w_hi - B + c - 1) / c, ntiles);
        if (m_lo >= m_hi) continue;
        const uint32_t e_first = a.cls_first[c], e_last = a.cls_first[c + 1];
        uint32_t e_cur;
        {
            const uint64_t v0 = m_lo << 6;
            uint32_t lo = e_first, hi = e_last - 1;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.p_end[mid] > v0) hi = mid; else lo = mid + 1; }
            e_cur = lo;
        }
        for (uint64_t m = m_lo; m < m_hi; m++) {
            const uint64_t v = (m << 6) + (uint64_t)lane;
            const bool valid = v < tot;
            uint32_t e = e_cur;
            for (;;) {
                const uint32_t ej = min(e_cur + (uint32_t)(lane & 15), e_last - 1);
                const uint64_t pe = a.p_end[ej];
                bool more = false;
#pragma unroll 1
                for (int j = 0; j < 16; j++) {
                    const uint64_t pj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(pe >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pe, j);
                    if (pj > (m << 6) + 63u || e_cur + (uint32_t)j >= e_last - 1) break;
                    e += (v >= pj) ? 1u : 0u;
                    more = j == 15;
                }
                if (!more) break;
                e_cur += 16;
            }
            e = min(e, e_last - 1);
            const uint4 en = a.ent[e];
            const uint64_t p_start = a.p_end[e] - en.w;
            const uint64_t rec = (((uint64_t)en.y << 32) | en.x) + (valid ? v - p_start : 0);
            const uint32_t tidx = (uint32_t)(v - p_start);
            e_cur = (uint32_t)__builtin_amdgcn_readlane((int)e, 63);
            const uint4 c0 = a.ctx[2 * rec], tl = a.ctx[2 * rec + 1];
            const uint32_t ref_loc = c0.x + a.seed_size;
            // the record's field words, once per tile
            uint32_t tf[NFR + NFL];
            {
                const uint32_t r[3] = {c0.y, c0.z, c0.w}, l[4] = {tl.x, tl.y, tl.z, tl.w};
                field_words<FB, SH, 48>(r, lane_off, tf);
                field_words<FB, SH, 64>(l, lane_off, tf + NFR);
            }
            const uint32_t* qp = qx + (size_t)en.z * DW;
            for (int qi = 0; qi < c; qi++, qp += DW) {
                uint32_t q[DW];
#pragma unroll
                for (int k = 0; k < DW / 4; k++) {
                    const uint4 t4 = reinterpret_cast<const uint4*>(qp)[k];
                    q[4 * k] = t4.x; q[4 * k + 1] = t4.y; q[4 * k + 2] = t4.z; q[4 * k + 3] = t4.w;
                }
                uint32_t PR = 0, WR = 0, PL = 0, WL = 0;
#pragma unroll
                for (int k = 0; k < NFL; k++) {
                    if (k < NFR) cls_step2<TRACKW>(s_tab, tf[k] ^ q[1 + k], PR, WR);
                    cls_step2<TRACKW>(s_tab, tf[NFR + k] ^ q[1 + NFR + k], PL, WL);
                }
                const bool r_alive = (int)(short)((TRACKW ? WR : PR) & 0xFFFFu) >= -xdrop;
                const int bestR = ((int)PR >> 16) - (int)(short)(PR & 0xFFFFu);
                const bool l_alive = (int)(short)((TRACKW ? WL : PL) & 0xFFFFu) >= -xdrop;
                const int bestL = ((int)PL >> 16) - (int)(short)(PL & 0xFFFFu);
                const bool fwd = valid && (r_alive || l_alive || bestR + bestL >= a.hspthresh);
                if (a.n_hits) scored += valid ? 1 : 0;
                const unsigned long long fm = __ballot(fwd);
                if (fm) {
                    L2Rec cr;
                    cr.ref_loc = ref_loc;
                    cr.query_loc = q[0] + a.seed_size;
                    cr.hidx = tidx;
                    cr.state = (PL & 0xFFFF0000u) | ((0u - PL) & 0xFFFFu);
                    const uint32_t fl = (r_alive ? 1u : 0u) | (l_alive ? 2u : 0u);
                    cr.meta = (uint32_t)(r_alive ? bestL : bestR) | ((fl ? fl : 3u) << 16);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                    if (fwd) stage[n_stage + (int)rank] = cr;
                    n_stage += __popcll(fm);
                    __builtin_amdgcn_wave_barrier();
                    if (n_stage >= STAGE_FLUSH) {
                        const int k = n_stage < 64 ? n_stage : 64;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(my_count, (uint32_t)k);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (lane < k && base + (uint32_t)lane < a.l2_cap) my_list[base + (uint32_t)lane] = stage[lane];
                        const int rest = n_stage - k;
                        L2Rec tmp = cr;
                        if (lane < rest) tmp = stage[k + lane];
                        __builtin_amdgcn_wave_barrier();
                        if (lane < rest) stage[lane] = tmp;
                        __builtin_amdgcn_wave_barrier();
                        n_stage = rest;
                    }
                }
            }
        }
    }
    }
    if (n_stage > 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(my_count, (uint32_t)n_stage);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < n_stage && base + (uint32_t)lane < a.l2_cap) my_list[base + (uint32_t)lane] = stage[lane];
    }
    if (a.n_hits && scored) atomicAdd(a.n_hits, scored);
}

// ---- host emulation of one hit (same field cutting, same padding) ----
static bool emu_hit(int FB, bool trackw, const int* cls, int xdrop, int hspthresh, const uint32_t t[8], const uint32_t q[8], uint32_t seed, L2Rec& out, uint32_t tidx) {
    auto walk = [&](const uint32_t* w, int nw, int nbases, int& T, int& N, int& Wm) {
        T = 0; N = 0; Wm = 0;
        const int nf = (nbases + FB - 1) / FB;
        for (int k = 0; k < nf; k++) {
            int f = 0;
            for (int b = 0; b < FB; b++) {
                const int bit = (k * FB + b) * 2;
                const uint32_t x = bit / 32 < nw ? (w[bit / 32] >> (bit % 32)) & 3u : 0u;
                f |= (int)x << (2 * b);
            }
            const uint32_t e = cls_entry(f, FB, cls);
            const int sum = (short)(e >> 16), m = (short)(e & 0xFFFF);
            T = (short)(T + sum);
            N = std::min((int)(short)(N + sum), m);
            Wm = std::min(Wm, N);
        }
        if (!trackw) Wm = N;
    };
    uint32_t xr[3] = {t[1] ^ q[1], t[2] ^ q[2], t[3] ^ q[3]}, xl[4] = {t[4] ^ q[4], t[5] ^ q[5], t[6] ^ q[6], t[7] ^ q[7]};
    int T, N, Wm;
    walk(xr, 3, 48, T, N, Wm);
    const bool r_alive = Wm >= -xdrop;
    const int bestR = T - N;
    walk(xl, 4, 64, T, N, Wm);
    const bool l_alive = Wm >= -xdrop;
    const int bestL = T - N;
    if (!(r_alive || l_alive || bestR + bestL >= hspthresh)) return false;
    out.ref_loc = t[0] + seed;
    out.query_loc = q[0] + seed;
    out.hidx = tidx;
    out.state = ((uint32_t)(uint16_t)(short)T << 16) | (uint32_t)(uint16_t)(short)(-N);
    const uint32_t fl = (r_alive ? 1u : 0u) | (l_alive ? 2u : 0u);
    out.meta = (uint32_t)(r_alive ? bestL : bestR) | ((fl ? fl : 3u) << 16);
    return true;
}

template <int FB, int REP>
static double run_variant(const char* name, JoinArgs a, int blocks, uint64_t hits, uint64_t ctx_bytes, std::vector<L2Rec>* got) {
    const size_t lds = ((size_t)(1 << (2 * FB)) * REP) * 4 + (size_t)16 * STAGE_CAP * sizeof(L2Rec);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&join_filter_kernel<FB, REP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        CK(hipMemset(a.l2_count, 0, L2_NSUB * L2_CNT_STRIDE * 4));
        if (a.n_hits) CK(hipMemset(a.n_hits, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((join_filter_kernel<FB, REP>), dim3(blocks), dim3(1024), lds, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<uint32_t> cnt(L2_NSUB * L2_CNT_STRIDE);
    CK(hipMemcpy(cnt.data(), a.l2_count, cnt.size() * 4, hipMemcpyDeviceToHost));
    uint64_t fwd = 0, mx = 0;
    for (int s = 0; s < L2_NSUB; s++) { fwd += cnt[s * L2_CNT_STRIDE]; mx = std::max<uint64_t>(mx, cnt[s * L2_CNT_STRIDE]); }
    unsigned long long scored = 0;
    if (a.n_hits) CK(hipMemcpy(&scored, a.n_hits, 8, hipMemcpyDeviceToHost));
    printf("%-28s lds %6zu B  %8.3f ms  %7.1f G hits/s  fwd %.2f %% (max sub-list %lu of %u)  ctx %.2f TB/s%s\n", name, lds, best, hits / (best * 1e6), 100.0 * fwd / hits,
           (unsigned long)mx, a.l2_cap, ctx_bytes / (best * 1e9), a.n_hits ? (scored == hits ? "  [hit count ok]" : "  [HIT COUNT WRONG]") : "");
    if (got) {
        got->clear();
        std::vector<L2Rec> sub(a.l2_cap);
        for (int s = 0; s < L2_NSUB; s++) {
            const uint32_t n = std::min(cnt[s * L2_CNT_STRIDE], a.l2_cap);
            if (!n) continue;
            CK(hipMemcpy(sub.data(), a.l2_list + (size_t)s * a.l2_cap, (size_t)n * sizeof(L2Rec), hipMemcpyDeviceToHost));
            got->insert(got->end(), sub.begin(), sub.begin() + n);
        }
    }
    return best;
}

template <int FB, int REP, bool TRACKW, int THREADS>
static double run_variant2(const char* name, JoinArgs a, int blocks, uint64_t hits, uint64_t ctx_bytes, uint64_t total_q, std::vector<L2Rec>* got, uint32_t grain = 0) {
    a.grain = grain;
    uint32_t* d_qx;
    CK(hipMalloc(&d_qx, (total_q + 64) * QX<FB>::DW * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((qx_build_kernel<FB, REP>), dim3(4096), dim3(256), 0, 0, a.qrec, total_q, d_qx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float bms = 0;
    CK(hipEventElapsedTime(&bms, e0, e1));
    const size_t lds = ((size_t)(1 << (2 * FB)) * REP) * 4 + (size_t)(THREADS / 64) * STAGE_CAP * sizeof(L2Rec);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&join_filter2_kernel<FB, REP, TRACKW, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        CK(hipMemset(a.l2_count, 0, L2_NSUB * L2_CNT_STRIDE * 4));
        if (a.n_hits) CK(hipMemset(a.n_hits, 0, 8));
        CK(hipMemset(a.work_next, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((join_filter2_kernel<FB, REP, TRACKW, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, a, d_qx);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<uint32_t> cnt(L2_NSUB * L2_CNT_STRIDE);
    CK(hipMemcpy(cnt.data(), a.l2_count, cnt.size() * 4, hipMemcpyDeviceToHost));
    uint64_t fwd = 0, mx = 0;
    for (int s = 0; s < L2_NSUB; s++) { fwd += cnt[s * L2_CNT_STRIDE]; mx = std::max<uint64_t>(mx, cnt[s * L2_CNT_STRIDE]); }
    unsigned long long scored = 0;
    if (a.n_hits) CK(hipMemcpy(&scored, a.n_hits, 8, hipMemcpyDeviceToHost));
    printf("%-34s lds %6zu B x %4d thr  %8.3f ms  %7.1f G hits/s  fwd %.2f %%  (QRecX %d B, built in %.3f ms)%s\n", name, lds, THREADS, best, hits / (best * 1e6), 100.0 * fwd / hits,
           QX<FB>::DW * 4, bms, a.n_hits ? (scored == hits ? "  [hit count ok]" : "  [HIT COUNT WRONG]") : "");
    if (got) {
        got->clear();
        std::vector<L2Rec> sub(a.l2_cap);
        for (int s = 0; s < L2_NSUB; s++) {
            const uint32_t n = std::min(cnt[s * L2_CNT_STRIDE], a.l2_cap);
            if (!n) continue;
            CK(hipMemcpy(sub.data(), a.l2_list + (size_t)s * a.l2_cap, (size_t)n * sizeof(L2Rec), hipMemcpyDeviceToHost));
            got->insert(got->end(), sub.begin(), sub.begin() + n);
        }
    }
    CK(hipFree(d_qx));
    return best;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const double mean_t = argc > 2 ? atof(argv[2]) : 73.0, mean_q = argc > 3 ? atof(argv[3]) : 3.0;
    const int check = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned long sel = argc > 5 ? strtoul(argv[5], nullptr, 0) : ~0ul;  // bit i: run variant i (in the order below)
    int vi = 0;
#define SEL() ((sel >> (vi++)) & 1ul)
    const uint32_t nkeys = 1u << lg;
    std::mt19937_64 rng(12345);
    std::poisson_distribution<int> pt(mean_t), pq(mean_q);
    std::vector<uint32_t> nt(nkeys), nq(nkeys);
    std::vector<uint64_t> tstart(nkeys + 1), qstart(nkeys + 1);
    uint64_t hits = 0;
    for (uint32_t k = 0; k < nkeys; k++) {
        nt[k] = (uint32_t)pt(rng);
        nq[k] = (uint32_t)pq(rng);
        if (check && (k % 97) == 5) nq[k] = 40 + k % 7;    // (a few keys with more than CMAX positions)
        if (check && (k % 101) == 7) nt[k] = 1 + k % 3;    // (... and tiny runs: tiles that span many entries)
        tstart[k + 1] = tstart[k] + nt[k];
        qstart[k + 1] = qstart[k] + nq[k];
        hits += (uint64_t)nt[k] * nq[k];
    }
    const uint64_t total_t = tstart[nkeys], total_q = qstart[nkeys];
    printf("keys %u, context records %.1f M (%.2f GB), query positions %.1f M, hits %.3f G\n", nkeys, total_t / 1e6, total_t * 32 / 1e9, total_q / 1e6, hits / 1e9);
    // entries by class
    std::vector<std::vector<JEnt>> byc(CMAX + 1);
    for (uint32_t k = 0; k < nkeys; k++) {
        if (!nt[k]) continue;
        uint32_t left = nq[k], qf = (uint32_t)qstart[k];
        while (left) {
            const uint32_t c = std::min<uint32_t>(left, CMAX);
            byc[c].push_back({(uint32_t)tstart[k], (uint32_t)(tstart[k] >> 32), qf, nt[k]});
            qf += c;
            left -= c;
        }
    }
    JoinArgs a;
    memset(&a, 0, sizeof(a));
    std::vector<JEnt> ent;
    std::vector<unsigned long long> p_end;
    for (int c = 1; c <= CMAX; c++) {
        a.cls_first[c] = (uint32_t)ent.size();
        unsigned long long p = 0;
        for (const JEnt& e : byc[c]) { p += e.n_t; ent.push_back(e); p_end.push_back(p); }
        a.cls_total[c] = p;
    }
    a.cls_first[CMAX + 1] = (uint32_t)ent.size();
    unsigned long long wb = 0;
    for (int c = CMAX; c >= 1; c--) { a.work_base[c] = wb; wb += ((a.cls_total[c] + 63) >> 6) * (unsigned long long)c; }
    a.work_total = wb;
    printf("entries %zu, tile steps %.2f M (lane utilisation %.3f)\n", ent.size(), wb / 1e6, (double)hits / (64.0 * wb));

    uint4 *d_ctx, *d_q, *d_ent;
    unsigned long long* d_pend;
    uint64_t *d_ts, *d_qs;
    CK(hipMalloc(&d_ctx, (total_t + 256) * 32));
    CK(hipMalloc(&d_q, (total_q + 256) * 32));
    CK(hipMalloc(&d_ts, (nkeys + 1) * 8));
    CK(hipMalloc(&d_qs, (nkeys + 1) * 8));
    CK(hipMemcpy(d_ts, tstart.data(), (nkeys + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_qs, qstart.data(), (nkeys + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, d_ctx, nullptr, total_t, 1u, d_ts, nkeys);
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, d_q, nullptr, total_q, 2u, d_qs, nkeys);
    CK(hipDeviceSynchronize());
    CK(hipMalloc(&d_ent, ent.size() * 16 + 64));
    CK(hipMalloc(&d_pend, p_end.size() * 8 + 64));
    CK(hipMemcpy(d_ent, ent.data(), ent.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pend, p_end.data(), p_end.size() * 8, hipMemcpyHostToDevice));
    a.ctx = d_ctx; a.qrec = d_q; a.ent = d_ent; a.p_end = d_pend;
    const int cls[4] = {100, -114, -31, -123};
    memcpy(a.cls, cls, sizeof(cls));
    a.xdrop = 910; a.hspthresh = 3000; a.seed_size = 19;
    a.l2_cap = (uint32_t)std::max<uint64_t>(hits / 8 / L2_NSUB, 1u << 14);
    CK(hipMalloc(&a.l2_list, (size_t)a.l2_cap * L2_NSUB * sizeof(L2Rec)));
    CK(hipMalloc(&a.l2_count, L2_NSUB * L2_CNT_STRIDE * 4));
    if (check) CK(hipMalloc(&a.n_hits, 8));
    CK(hipMalloc(&a.work_next, 8));
    const uint64_t ctx_bytes = total_t * 32;

    std::vector<L2Rec> got;
    auto verify = [&](int FB, const char* nm, bool trackw = true) {
        if (!check) return;
        std::vector<L2Rec> want;
        for (uint32_t k = 0; k < nkeys; k++)
            for (uint32_t t = 0; t < nt[k]; t++) {
                uint32_t tr[8];
                make_rec(k, tstart[k] + t, 1u, tr);
                for (uint32_t q = 0; q < nq[k]; q++) {
                    uint32_t qr[8];
                    make_rec(k, qstart[k] + q, 2u, qr);
                    L2Rec r;
                    if (emu_hit(FB, trackw, cls, a.xdrop, a.hspthresh, tr, qr, a.seed_size, r, t)) want.push_back(r);
                }
            }
        auto less = [](const L2Rec& x, const L2Rec& y) { return memcmp(&x, &y, sizeof(L2Rec)) < 0; };
        std::sort(want.begin(), want.end(), less);
        std::sort(got.begin(), got.end(), less);
        const bool ok = want.size() == got.size() && (want.empty() || memcmp(want.data(), got.data(), want.size() * sizeof(L2Rec)) == 0);
        printf("   check %s: %zu forwarded records expected, %zu found: %s\n", nm, want.size(), got.size(), ok ? "IDENTICAL" : "DIFFERENT");
    };
    if (SEL()) { run_variant<6, 1>("6-base, 1 replica", a, 512, hits, ctx_bytes, check ? &got : nullptr);   verify(6, "6/1"); }
    if (SEL()) { run_variant<6, 8>("6-base, 8 replicas", a, 256, hits, ctx_bytes, check ? &got : nullptr);  verify(6, "6/8"); }
    if (SEL()) { run_variant<5, 16>("5-base, 16 replicas", a, 512, hits, ctx_bytes, check ? &got : nullptr); verify(5, "5/16"); }
    if (SEL()) { run_variant<5, 32>("5-base, 32 replicas", a, 256, hits, ctx_bytes, check ? &got : nullptr); verify(5, "5/32"); }
    if (SEL()) { run_variant<4, 32>("4-base, 32 replicas", a, 512, hits, ctx_bytes, check ? &got : nullptr); verify(4, "4/32"); }
    std::vector<L2Rec>* g = check ? &got : nullptr;
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep", a, 512, hits, ctx_bytes, total_q, g);            verify(6, "fw 6/1"); }
    if (SEL()) { run_variant2<6, 1, true, 512>("fw 6-base 1 rep", a, 1024, hits, ctx_bytes, total_q, g);            verify(6, "fw 6/1 512"); }
    if (SEL()) { run_variant2<6, 4, true, 1024>("fw 6-base 4 rep", a, 256, hits, ctx_bytes, total_q, g);            verify(6, "fw 6/4"); }
    if (SEL()) { run_variant2<6, 4, true, 512>("fw 6-base 4 rep", a, 512, hits, ctx_bytes, total_q, g);             verify(6, "fw 6/4 512"); }
    if (SEL()) { run_variant2<6, 2, true, 1024>("fw 6-base 2 rep", a, 512, hits, ctx_bytes, total_q, g);            verify(6, "fw 6/2"); }
    if (SEL()) { run_variant2<5, 8, true, 1024>("fw 5-base 8 rep", a, 512, hits, ctx_bytes, total_q, g);            verify(5, "fw 5/8"); }
    if (SEL()) { run_variant2<5, 16, true, 512>("fw 5-base 16 rep", a, 512, hits, ctx_bytes, total_q, g);           verify(5, "fw 5/16 512"); }
    if (SEL()) { run_variant2<5, 16, true, 1024>("fw 5-base 16 rep", a, 256, hits, ctx_bytes, total_q, g);          verify(5, "fw 5/16"); }
    if (SEL()) { run_variant2<5, 32, true, 1024>("fw 5-base 32 rep", a, 256, hits, ctx_bytes, total_q, g);          verify(5, "fw 5/32"); }
    if (SEL()) { run_variant2<6, 1, false, 1024>("fw 6-base 1 rep, no W", a, 512, hits, ctx_bytes, total_q, g);     verify(6, "fw 6/1 noW", false); }
    if (SEL()) { run_variant2<6, 4, false, 512>("fw 6-base 4 rep, no W", a, 512, hits, ctx_bytes, total_q, g);      verify(6, "fw 6/4 noW", false); }
    if (SEL()) { run_variant2<5, 16, false, 512>("fw 5-base 16 rep, no W", a, 512, hits, ctx_bytes, total_q, g);    verify(5, "fw 5/16 noW", false); }
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep, dynamic 4096", a, 512, hits, ctx_bytes, total_q, g, 4096);   verify(6, "fw 6/1 dyn"); }
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep, dynamic 1024", a, 512, hits, ctx_bytes, total_q, g, 1024);   verify(6, "fw 6/1 dyn"); }
    if (SEL()) { run_variant2<6, 1, false, 1024>("fw 6-base 1 rep, no W, dynamic 4096", a, 512, hits, ctx_bytes, total_q, g, 4096);   verify(6, "fw 6/1 noW dyn", false); }
    if (SEL()) { run_variant2<6, 4, true, 512>("fw 6-base 4 rep, dynamic 4096", a, 512, hits, ctx_bytes, total_q, g, 4096);   verify(6, "fw 6/4 dyn"); }
    if (SEL()) { run_variant2<5, 8, true, 1024>("fw 5-base 8 rep, dynamic 4096", a, 512, hits, ctx_bytes, total_q, g, 4096);   verify(5, "fw 5/8 dyn"); }
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep, dynamic 512", a, 512, hits, ctx_bytes, total_q, g, 512);   verify(6, "fw 6/1 dyn"); }
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep, dynamic 256", a, 512, hits, ctx_bytes, total_q, g, 256);   verify(6, "fw 6/1 dyn"); }
    if (SEL()) { run_variant2<6, 1, true, 1024>("fw 6-base 1 rep, dynamic 128", a, 512, hits, ctx_bytes, total_q, g, 128);   verify(6, "fw 6/1 dyn"); }
    if (SEL()) { run_variant2<6, 1, false, 1024>("fw 6-base 1 rep, no W, dynamic 256", a, 512, hits, ctx_bytes, total_q, g, 256);   verify(6, "fw 6/1 noW dyn", false); }
    if (SEL()) { run_variant2<6, 2, true, 1024>("fw 6-base 2 rep, dynamic 256", a, 512, hits, ctx_bytes, total_q, g, 256);   verify(6, "fw 6/2 dyn"); }
    if (SEL()) { run_variant2<6, 4, true, 512>("fw 6-base 4 rep, dynamic 256", a, 512, hits, ctx_bytes, total_q, g, 256);   verify(6, "fw 6/4 dyn"); }
    if (SEL()) { run_variant2<5, 8, true, 1024>("fw 5-base 8 rep, dynamic 256", a, 512, hits, ctx_bytes, total_q, g, 256);   verify(5, "fw 5/8 dyn"); }
    if (SEL()) { run_variant2<6, 1, true, 512>("fw 6-base 1 rep 512 thr, dynamic 256", a, 1024, hits, ctx_bytes, total_q, g, 256);   verify(6, "fw 6/1 dyn"); }
    return 0;
}
