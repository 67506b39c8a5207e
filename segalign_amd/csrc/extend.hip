// extend.hip -- ungapped X-drop extension + entropy filter + wavefront-ballot compaction of the survivors.
// Replaces find_hsps (src/seed_filter.cu:232-652), the done-flag scan (:769) and compress_output (:654-680).
//
// Scalar recurrence per side (k = 0,1,.. right of the anchor; k = 1,2,.. left of it), which is what the reference's
// 32-lane tile loop computes (:326-453 right, :478-604 left) independently of its tile width:
//     score += M[r][q];  if (max(best,score) - score > xdrop) stop;  if (score > best) { best = score; bestpos = k; }
// stop also at the first position outside either sequence.  A hit survives iff
// (int)((float)(bestR + bestL) * entropy) >= hspthresh (:633).
//
// Design (wave64, CDNA4) -- NOT the reference's shape (one 32-lane warp per hit, four shuffle scans + ~10 syncs per
// 32 bases, although ~98 % of all hits are random and die after ~20-60 bases below the threshold).  Three kernels:
//
//  1. extend_filter_kernel -- the X-DROP FILTER.  ONE LANE OWNS ONE HIT, lanes are PERSISTENT.  Each lane runs a
//     small state machine (right side -> left side -> finished) that tracks only the running score and the best score
//     of the side -- no positions: 4.5 VALU + 1 ds_read_b32 per base.  Every trip of the wave loop advances every
//     live lane by 8 bases; finished lanes are handled in batches and REFILLED from the wave's queue (a register-held,
//     double-buffered 64-hit buffer; buffers are dealt round-robin to the waves of the grid, no atomics on the fetch
//     side).  A hit whose bestR + bestL can pass the threshold, or whose side is still alive after `long_cap` bases
//     (real homology, may run for kilobases), becomes a CANDIDATE (12-byte record, wave-aggregated append).
//  2. extend_exact_kernel -- ONE WAVE OWNS ONE CANDIDATE and extends it exactly, 512 bases per step, with a segmented
//     scan: lane l scores bases [8l, 8l+8) of the window, a DPP wave sum-scan gives every lane its entry score, a DPP
//     max-scan (ties -> earlier position) its entry best, then each lane REPLAYS its 8 bases with the exact entry
//     state; the first lane that drops holds the final (best, bestpos).  ~0.25 wave-instructions per base.
//  3. extend_entropy_kernel -- the few hits with hspthresh <= score <= 3*hspthresh (:608) get their fp64 entropy
//     factor here, one lane per hit, so the hot kernels carry no fp64 code or registers.
//
// Shared machinery:
//   * The target is kept in HBM a second time "row coded" (r<<3, one byte per base) so that `rw | qw` of two
//     8-byte windows IS the 8 matrix indices r*8+q; one unaligned global_load_dwordx2 per sequence per 8 bases.
//     The left side reverses its window with the same two v_perm_b32 (per-lane selector) the right side uses.
//   * The 8x8 matrix sits in LDS as one 128-entry table: entries 64..127 hold a large negative "terminator" that
//     out-of-range positions are mapped to (bit 6 OR-ed into their index byte), which folds the sequence-edge
//     test (:332,:482) into the X-drop test.  ACGTxACGT pairs occupy 16 distinct banks: conflict-free.
//     Address = one SDWA byte-select shift.
//   * Integer DP only (no MFMA).  Survivors are appended with wave-aggregated atomics.  Append order is arbitrary;
//     the dedup stage sorts on a total order, so the output is deterministic.
#include <atomic>
#include <cstdlib>

#include "kernels.h"
#include "join.h"
#include "kmer_dev.h"  // load8u

namespace sa {

constexpr int EXT_THREADS = 256;
constexpr int NEG = -(1 << 26);   // terminator score: 8 of them still fit an int, one forces the drop test (|xdrop| < 2^25)
constexpr int DEAD = -(1 << 29);  // sticky running score of a side that has dropped (exact / counting paths)
constexpr uint64_t TERM_ALL = 0x4040404040404040ull;
constexpr uint32_t BIAS = SEQ_PAD;  // offsets are kept unsigned: base pointers point at the start of the front pad

__device__ __forceinline__ int f64_to_i32(double x) { return (int)x; }  // v_cvt_i32_f64: NaN -> 0, saturating (as on CUDA)

enum : int { PH_RIGHT = 0, PH_LEFT = 1, PH_FIN = 2, PH_IDLE = 3 };

// ---- DPP helpers (row shifts inside 16-lane rows + row broadcasts: the classic 6-step wave64 scan) -----------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)dpp_mov<0x111>(0, (int)v);        // row_shr:1
    v += (uint32_t)dpp_mov<0x112>(0, (int)v);        // row_shr:2
    v += (uint32_t)dpp_mov<0x114>(0, (int)v);        // row_shr:4
    v += (uint32_t)dpp_mov<0x118>(0, (int)v);        // row_shr:8
    v += (uint32_t)dpp_mov<0x142, 0xa>(0, (int)v);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)dpp_mov<0x143, 0xc>(0, (int)v);   // row_bcast:31 -> rows 2,3
    return v;
}
// inclusive max-scan of (value, position); ties keep the EARLIER lane's pair (:350 strict ">" + :361-372 ">=")
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void max_step(int& mv, int& mp) {
    const int tv = dpp_mov<CTRL, ROW_MASK>(INT32_MIN, mv);
    const int tp = dpp_mov<CTRL, ROW_MASK>(0, mp);
    const bool take = tv >= mv;
    mp = take ? tp : mp;
    mv = take ? tv : mv;
}
__device__ __forceinline__ void wave_inclusive_max(int& mv, int& mp) {
    max_step<0x111, 0xf>(mv, mp);
    max_step<0x112, 0xf>(mv, mp);
    max_step<0x114, 0xf>(mv, mp);
    max_step<0x118, 0xf>(mv, mp);
    max_step<0x142, 0xa>(mv, mp);
    max_step<0x143, 0xc>(mv, mp);
}

// ---- pieces shared by the kernels -------------------------------------------------------------------------------------
// exact per-base recurrence with positions on the packed index word x (byte j = matrix index of offset k+j)
template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__device__ __forceinline__ void chunk8_exact(const int* __restrict__ s_tab, uint64_t x, uint32_t k, int xdrop, int& score,
                                             int& best, int& bpos, uint32_t& examined) {
    const uint32_t xlo = (uint32_t)x, xhi = (uint32_t)(x >> 32);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = j < 4 ? xlo : xhi;
        const uint32_t idx = (w >> (8 * (j & 3))) & 0xffu;  // SDWA byte select
        if (COUNT_EXAMINED) examined += (score > (DEAD >> 1) && idx < 64u) ? 1u : 0u;
        const int t = score + s_tab[idx];
        const int nb = max(best, t);
        const bool drop = (nb - t) > xdrop;  // :374 / :523 (also fires on a terminator = sequence edge :332/:482)
        if (XDROP_NONNEG) {
            bpos = (t > best) ? (int)(k + j) : bpos;  // :350 strict: first position attaining the max
            best = nb;
        } else {
            const bool up = !drop && (t > best);
            bpos = up ? (int)(k + j) : bpos;
            best = up ? t : best;
        }
        score = drop ? DEAD : t;
    }
}

// segment (reference iteration) of a hit: the first s with g < seg_end[s] -- a binary search over the batch's <= MAX_SEGS ascending
// ends.  Every candidate-stage workgroup stages the ends in LDS first (SEG_TABLE below): the search is 6-9 dependent reads, and from
// the device array itself it cost the five candidate-stage kernels ~100 us per call together (a 63-step compare loop over kernel
// arguments, which is what a 64-segment limit allowed, cost about the same).
__device__ __forceinline__ uint32_t seg_of(const ExtendArgs& a, const uint64_t* __restrict__ s_seg, uint64_t local_idx, uint32_t query_loc) {
    if (a.join) {
        // key-ordered call (join.h): no hit indices.  `local_idx` is the hit's entry index inside its key's run, s_seg holds per chunk
        // {p_last : e_thr} and, JOIN_SEG_FIRST (= SA_MAX_CHUNKS) entries further on, the chunk's first segment: second iteration of the chunk iff the hit sits at the
        // chunk's last non-empty position at or behind the last hit-bearing seed word (src/seed_filter.cu:732-741)
        const uint32_t p = query_loc - a.seed_size;
        const uint32_t c = (p - a.join_q_lo) / a.join_chunk;
        const uint64_t w = s_seg[c];
        return (uint32_t)s_seg[(uint32_t)JOIN_SEG_FIRST + c] + ((p == (uint32_t)w && (uint32_t)local_idx >= (uint32_t)(w >> 32)) ? 1u : 0u);
    }
    const uint64_t g = a.hit_base + local_idx;
    uint32_t lo = 0, hi = (uint32_t)a.num_segs - 1u;  // answer in [lo, hi]; a hit beyond the last end belongs to the last segment
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g >= s_seg[mid]) lo = mid + 1; else hi = mid;
    }
    return a.seg_base + lo;
}
#define SEG_TABLE()                                                                                                      \
    __shared__ uint64_t s_seg[MAX_SEGS];                                                                                 \
    for (uint32_t t_ = threadIdx.x; t_ < (a.join ? (uint32_t)MAX_SEGS : (uint32_t)a.num_segs); t_ += blockDim.x) s_seg[t_] = a.seg_end[t_]; \
    __syncthreads();

// What to do with a finished hit: 0 = reject, 1 = survivor with entropy 1, 2 = needs the entropy factor (:608,:633)
__device__ __forceinline__ int classify(const ExtendArgs& a, int total) {
    if (!a.noentropy && total >= a.hspthresh && total <= 3 * a.hspthresh) return 2;
    return (f64_to_i32((double)(float)total) >= a.hspthresh) ? 1 : 0;  // entropy stays 1.0
}

// "may this bound pass?" for the packed filter levels, whose totals are sums of two int16 walks: classify(total) != 0 without its
// conversions.  classify returns 2 inside the entropy band (total >= hspthresh there), else 1 iff (int)(double)(float)total >= hspthresh
// (:633 with entropy 1.0) -- and below 2^24 the float round trip is exact, so both branches say total >= hspthresh.
__device__ __forceinline__ bool bound_passes(const ExtendArgs& a, int total) { return total >= a.hspthresh; }

__device__ __forceinline__ HspRec make_rec(const ExtendArgs& a, uint32_t ref_loc, uint32_t query_loc, int boffL, int extent,
                                           int score, uint32_t seg) {
    HspRec rec;
    rec.ref_start = ref_loc - (uint32_t)boffL;      // :634
    rec.query_start = query_loc - (uint32_t)boffL;  // :635
    rec.len = (uint32_t)extent;                     // :636
    rec.score = score;
    if (a.rm && a.rm_rev)  // rc coordinate flip of the repeat masker's compress_output (rm :705-708)
        rec.query_start = a.ref_len - 1u - (rec.query_start + rec.len);
    rec.seg = seg;
    return rec;
}

// wave-aggregated append of one record per flagged lane (overflowing writes are dropped, the counter keeps counting
// so that the host can grow the list and rerun the batch)
template <typename T>
__device__ __forceinline__ void wave_append(bool flag, const T& rec, T* __restrict__ list, uint32_t* __restrict__ counter,
                                            uint32_t cap, int lane, unsigned long long lane_lt) {
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t wbase = 0;
        if (lane == leader) wbase = atomicAdd(counter, (uint32_t)__popcll(m));
        wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, leader);
        const uint32_t slot = wbase + (uint32_t)__popcll(m & lane_lt);
        if (flag && slot < cap) list[slot] = rec;
    }
}


// ---- per-wave LDS staging: records are collected 64 at a time so that one atomicAdd (and one coalesced store) serves
// 64 appends -- single-address atomics cost ~11 ns each and would otherwise bound these kernels -----------------------
constexpr int STAGE_CAP = 128;  // records per wave: < 64 pending + <= 64 new

template <typename T>
__device__ __forceinline__ void stage_flush(const T* __restrict__ stage, int cnt, T* __restrict__ list,
                                            uint32_t* __restrict__ counter, uint32_t cap, int lane) {
    if (cnt <= 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, (uint32_t)cnt);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    __builtin_amdgcn_wave_barrier();
    if (lane < cnt && base + (uint32_t)lane < cap) list[base + (uint32_t)lane] = stage[lane];
}

// append one record per flagged lane to the wave's stage; flush up to 64 of them once FLUSH are pending (stage capacity:
// FLUSH - 1 pending + 64 new).  n is wave-uniform.
template <int FLUSH = 64, typename T>
__device__ __forceinline__ void stage_append(T* __restrict__ stage, int& n, bool flag, const T& rec, T* __restrict__ list,
                                             uint32_t* __restrict__ counter, uint32_t cap, int lane, unsigned long long lane_lt) {
    const unsigned long long m = __ballot(flag);
    if (m == 0ull) return;
    if (flag) stage[n + __popcll(m & lane_lt)] = rec;
    n += __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (n >= FLUSH) {
        const int k = n < 64 ? n : 64;
        stage_flush(stage, k, list, counter, cap, lane);
        const int rest = n - k;
        T tmp = rec;
        if (lane < rest) tmp = stage[k + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < rest) stage[lane] = tmp;
        __builtin_amdgcn_wave_barrier();
        n = rest;
    }
}

// =====================================================================================================================
// 1. the X-drop filter: persistent lanes, best scores only
// =====================================================================================================================
// FAST: xdrop >= 0 and 7*max(M) <= xdrop, so after a drop the remaining <= 7 bases of the chunk can never lift the
// score above the best again: no sticky select is needed, "dropped" = max over the chunk of (runmax - score) > xdrop.
template <bool COUNT_EXAMINED, bool FAST>
__global__ __launch_bounds__(EXT_THREADS) void extend_filter_kernel(ExtendArgs a) {
    __shared__ int s_tab[128];
    __shared__ CandRec s_cand[EXT_THREADS / 64][STAGE_CAP];
    if (threadIdx.x < 128) s_tab[threadIdx.x] = threadIdx.x < 64 ? a.sub_mat[threadIdx.x] : NEG;
    __syncthreads();
    CandRec* stage = s_cand[threadIdx.x >> 6];
    int n_stage = 0;

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint8_t* __restrict__ R8b = a.ref8 - BIAS;  // row-coded target: byte = r << 3
    const uint8_t* __restrict__ Qb = a.query - BIAS;  // plain codes
    const int xdrop = a.xdrop;
    const int fin_batch = a.fin_batch;
    const uint32_t long_cap = a.long_cap;

    // ---- the wave's queue: 64-hit buffers, round-robin over all waves of the grid ----
    const uint64_t num_buf = (a.num_hits + 63) >> 6;
    const uint64_t G = (uint64_t)gridDim.x * (EXT_THREADS / 64);
    uint64_t cur_buf = (uint64_t)blockIdx.x * (EXT_THREADS / 64) + (threadIdx.x >> 6);
    uint64_t nxt_buf = cur_buf + G;
    auto buf_count = [&](uint64_t b) -> int {
        if (b >= num_buf) return 0;
        uint64_t rem = a.num_hits - (b << 6);
        return rem >= 64 ? 64 : (int)rem;
    };
    int buf_cnt = buf_count(cur_buf), nxt_cnt = buf_count(nxt_buf), consumed = 0;
    Hit buf = {0u, 0u}, nxt = {0u, 0u};
    if (lane < buf_cnt) buf = a.hits[(cur_buf << 6) + lane];
    if (lane < nxt_cnt) nxt = a.hits[(nxt_buf << 6) + lane];

    // ---- per-lane state ----
    int phase = PH_FIN;  // "finished" with nothing to emit: the first trip refills every lane
    bool has_hit = false, forward = false;
    uint32_t ref_loc = 0, query_loc = 0, hidx = 0;
    uint32_t roff = 0, qoff = 0;            // byte offsets (from the padded bases) of the next 16-byte window
    int dstep = 16;                         // +16 on the right side, -16 on the left
    uint32_t bsel = 0x03020100u;            // v_perm selector: identity (right) / byte reversal inside a dword (left)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // the window: 16 matrix indices in walking order; two trips consume it
    bool half = false;                      // false: fetch a new window and use (w0,w1); true: use the held (w2,w3)
    uint32_t walked = 0;                    // bases walked on this side
    int score = 0, best = 0, bestR = 0;
    uint32_t ex_hit = 0;
    unsigned long long examined = 0, examined_all = 0;

    for (;;) {
        // ================= 1. advance every live lane by one 8-base chunk =================
        if (phase < PH_FIN) {
            // One 16-byte window per sequence feeds two trips: the 8-byte-per-trip form made every lane load a
            // separate L1 miss (the L1 is thrashed by 16 waves x 64 random lines), i.e. twice the L2 requests.
            if (!half) {
                const uint4 rw = load16u(R8b + roff), qw = load16u(Qb + qoff);
                const uint32_t o0 = rw.x | qw.x, o1 = rw.y | qw.y, o2 = rw.z | qw.z, o3 = rw.w | qw.w;  // 16 indices r<<3|q
                const bool left = dstep < 0;  // walking order: the left side reverses dwords and bytes
                w0 = __builtin_amdgcn_perm(0u, left ? o3 : o0, bsel);
                w1 = __builtin_amdgcn_perm(0u, left ? o2 : o1, bsel);
                w2 = __builtin_amdgcn_perm(0u, left ? o1 : o2, bsel);
                w3 = __builtin_amdgcn_perm(0u, left ? o0 : o3, bsel);
                roff += (uint32_t)dstep;
                qoff += (uint32_t)dstep;
            }
            // positions outside the block read guard bytes (0x40): their index selects a terminator entry (:332,:482)
            const uint32_t xlo = half ? w2 : w0, xhi = half ? w3 : w1;
            half = !half;
            bool dropped;
            if (FAST && !COUNT_EXAMINED) {
                int t = score, m = best, dmax = 0;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t w = j < 4 ? xlo : xhi;
                    t += s_tab[(w >> (8 * (j & 3))) & 0xffu];
                    m = max(m, t);
                    const int d0 = m - t;
                    t += s_tab[(w >> (8 * ((j + 1) & 3))) & 0xffu];
                    m = max(m, t);
                    const int d1 = m - t;
                    dmax = max(dmax, max(d0, d1));  // v_max3_i32
                }
                dropped = dmax > xdrop;  // :374 / :523 ; a terminator (edge :332/:482) always trips it
                score = t;
                best = m;  // == best before the drop (see FAST)
            } else {
                int dummy = 0;
                uint32_t ex = 0;
                const uint64_t x = ((uint64_t)xhi << 32) | xlo;
                if (xdrop >= 0) chunk8_exact<COUNT_EXAMINED, true>(s_tab, x, 0u, xdrop, score, best, dummy, ex);
                else chunk8_exact<COUNT_EXAMINED, false>(s_tab, x, 0u, xdrop, score, best, dummy, ex);
                if (COUNT_EXAMINED) ex_hit += ex;
                dropped = score < (DEAD >> 1);
            }
            walked += 8;
            if (dropped) {
                if (phase == PH_RIGHT) {  // -> left side (:457-476): anchor-1, anchor-2, ...
                    bestR = best;
                    phase = PH_LEFT;
                    roff = ref_loc + BIAS - 16u;  // bytes loc-16 .. loc-1 ; reversed: byte 0 <-> offset 1
                    qoff = query_loc + BIAS - 16u;
                    dstep = -16;
                    bsel = 0x00010203u;
                    half = false;
                    walked = 0;
                    score = 0;
                    best = 0;
                } else {
                    phase = PH_FIN;
                }
            } else if (walked >= long_cap) {  // still alive after long_cap bases: real homology -> exact kernel
                forward = true;
                phase = PH_FIN;
            }
        }

        // ================= 2. finalise + refill in batches =================
        const unsigned long long fin = __ballot(phase == PH_FIN);
        const unsigned long long live = __ballot(phase < PH_FIN);
        if (fin != 0ull && (__popcll(fin) >= fin_batch || live == 0ull)) {
            // ---- candidates: capped walks, and finished hits whose bestR + bestL can survive (:608-633) ----
            bool cand = false;
            if (phase == PH_FIN && has_hit) {
                cand = forward || classify(a, bestR + best) != 0;
                if (COUNT_EXAMINED) {
                    examined_all += ex_hit;             // every base the filter scored (its own algorithmic bytes)
                    if (!cand) examined += ex_hit;      // candidates are re-extended (and counted) by the exact kernel
                }
            }
            {
                CandRec cr;
                cr.ref_loc = ref_loc; cr.query_loc = query_loc; cr.hidx = hidx;
                stage_append(stage, n_stage, cand, cr, a.cand_list, a.cand_count, a.cand_cap_recs, lane, lane_lt);
            }
            // ---- refill the finished lanes from the wave's queue (wave-uniform control flow) ----
            unsigned long long need = fin;
            bool got = false;
            Hit mine = {0u, 0u};
            uint32_t mine_idx = 0;
            while (need != 0ull) {
                const int avail = buf_cnt - consumed;
                if (avail <= 0) {
                    if (nxt_cnt == 0) break;  // queue exhausted
                    buf = nxt;
                    buf_cnt = nxt_cnt;
                    cur_buf = nxt_buf;
                    consumed = 0;
                    nxt_buf += G;
                    nxt_cnt = buf_count(nxt_buf);
                    if (lane < nxt_cnt) nxt = a.hits[(nxt_buf << 6) + lane];
                    continue;
                }
                const int rank = __popcll(need & lane_lt);
                const bool take = ((need >> lane) & 1ull) && rank < avail;
                const int src = (consumed + rank) & 63;
                const uint32_t hr = (uint32_t)__shfl((int)buf.ref_loc, src, 64);
                const uint32_t hq = (uint32_t)__shfl((int)buf.query_loc, src, 64);
                if (take) {
                    mine.ref_loc = hr;
                    mine.query_loc = hq;
                    mine_idx = (uint32_t)(cur_buf << 6) + (uint32_t)src;
                    got = true;
                }
                consumed += min(__popcll(need), avail);
                need &= ~__ballot(take);
            }
            if (phase == PH_FIN) {
                forward = false;
                ex_hit = 0;
                if (got) {
                    has_hit = true;
                    ref_loc = mine.ref_loc;
                    query_loc = mine.query_loc;
                    hidx = mine_idx;
                    bool skip = false;
                    if (a.rm)  // repeat masker: hits outside [ref_start, ref_end] are not extended (rm :239-244,:305-333)
                        skip = !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);
                    bestR = 0;
                    best = 0;
                    // anchors beyond the block (only possible with hand-made seed words) are outside the guard bytes'
                    // reach: never extended (the reference would read out of bounds on the left side there)
                    if (ref_loc > a.ref_len || query_loc > a.query_len) skip = true;
                    if (skip) {  // both loops skipped: total 0
                        phase = PH_FIN;
                    } else {
                        phase = PH_RIGHT;  // :299-324
                        roff = ref_loc + BIAS;
                        qoff = query_loc + BIAS;
                        dstep = 16;
                        bsel = 0x03020100u;
                        half = false;
                        walked = 0;
                        score = 0;
                    }
                } else {
                    has_hit = false;
                    phase = PH_IDLE;
                }
            }
        }
        if (__ballot(phase != PH_IDLE) == 0ull) break;
    }
    stage_flush(stage, n_stage, a.cand_list, a.cand_count, a.cand_cap_recs, lane);

    if (COUNT_EXAMINED) {
        unsigned long long v = examined, w = examined_all;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); w += __shfl_down(w, off, 64); }
        if (lane == 0 && v) atomicAdd(a.examined, v);
        if (lane == 0 && w) atomicAdd(a.examined + 1, w);
    }
}

// =====================================================================================================================
// 1b. the X-drop filter, PACKED form: 2-bit target, 4-bit query, an upper bound priced in lines and load instructions
// =====================================================================================================================
// Measured on MI355X (tools/micro/gather_bw.hip, gather_l1.hip): the memory system delivers ~57 G random 128-byte lines
// per second, and a DIVERGENT 16-byte load (every lane its own line) occupies a CU's L1 for ~140 clocks even when it
// hits -- so for anchors scattered over a 100 MB target the filter is priced in lines and in target load instructions
// per hit, not in bytes or VALU.  The byte-coded kernels above fetch ~2.1 lines and issue ~3.4 target loads per hit.
// Here the target is read from a 2-bit copy: ONE 16-byte load covers 64 bases, so a hit needs one load per side and
// its whole neighbourhood [loc-64, loc+64) spans 32 bytes (1.25 lines).
// The filter only has to be CONSERVATIVE: every hit it forwards is re-extended exactly by the exact kernels (from the
// anchor, nothing is carried over), so it may over-estimate a side's best score but must never under-estimate it:
//   * target codes >= 4 (soft-masked, N, other IUPAC, separators) are stored as code 0, and the table row 0 holds, per
//     query code, the MAXIMUM over the rows {A, L, N, X, E} -- every score the filter adds is >= the reference's;
//   * the drop test (:374/:523) is evaluated once per 16 bases instead of per base.  A walk therefore never stops before
//     the reference's walk does (at the reference's stopping position the test is either not looked at, or it is looked
//     at a window end beyond it), and the best over a longer walk is >= the best over its prefix;
//   * two bases are scored per LDS access: the pair index selects an entry holding {s0, s0+s1} as two int16; with the
//     running score broadcast to both halves, ONE v_pk_add_i16 yields the scores after base 0 and after base 1 and ONE
//     v_pk_max_i16 folds both into the running maxima.  The adds saturate (clamp) and matrix entries below -16383 are
//     raised to -16383: both only raise scores;
//   * beyond a block edge the walk reads pad codes: arbitrary scores, but the reference has already stopped there.
// Eligibility (engine.hip): 0 <= xdrop <= 16383 and max(M) * long_cap <= 16383 (int16 scores with room for the drop test).
// Phase copies (encode.hip) make every window byte aligned: copy k = position & 3 (target) / & 1 (query).
// Left walks: the 128-bit target window is bit-reversed (v_bfrev_b32 + dword order), which also swaps the two bits of
// every code; the query bytes are byte-reversed and get bit 3 of both nibbles set; table entries with those bits set
// decode the swapped code bits and the swapped pair order, so both directions share one inner loop.
// Index of a pair = 4 target bits | query byte (2 nibbles) << 4 : 4096 entries x {s0, s0+s1} int16 = 16 KB of LDS.
// The target bits are the low (bank-selecting) part on purpose: the lanes of a wave sit at nearly the same query
// position (hits are generated query-major) but at unrelated target positions; with the query byte in the low bits the
// table reads were 16-way bank conflicted (SQ_LDS_BANK_CONFLICT = 87 % of SQ_LDS_IDX_ACTIVE, the kernel's bottleneck).
typedef short s16x2 __attribute__((ext_vector_type(2)));
constexpr int PK_THREADS = 512;
constexpr int PK_TAB = 4096;

// ---- pair table: entry (query byte << 4) | 4 target bits = {s0, s0 + s1} as two int16 (see above); reversed walk = query byte | 0x88
__device__ __forceinline__ void pk_table_init(uint32_t* __restrict__ s_pk, const int* __restrict__ sub_mat, int nthreads) {
    // the 64 matrix entries go through LDS first (the 4096 pair entries read ~10 of them each): the row-0 maximum over
    // the rows {A, L, N, X, E} is folded in there, scores are raised to -16383
    __shared__ int s_m[64];
    if (threadIdx.x < 64) {
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        int v = sub_mat[threadIdx.x];
        if (r == 0)  // row 0 also stands for every target code >= 4
            for (int rr = 4; rr < 8; rr++) v = max(v, sub_mat[rr * 8 + q]);
        s_m[threadIdx.x] = max(v, -16383);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PK_TAB; i += nthreads) {
        const int rp = i & 15, qb = i >> 4;  // target bits select the LDS bank: they differ between lanes, query bytes rarely do
        const bool rev = (qb & 0x88) == 0x88;
        int q0 = qb & 7, q1 = (qb >> 4) & 7, r0 = rp & 3, r1 = rp >> 2;
        if (rev) {  // walking order = high nibble first; code bits swapped by the bit reversal
            const int t = q0; q0 = q1; q1 = t;
            r0 = ((r0 & 1) << 1) | (r0 >> 1);
            r1 = ((r1 & 1) << 1) | (r1 >> 1);
        }
        const int s0 = s_m[r0 * 8 + q0], s1 = s_m[r1 * 8 + q1];
        s_pk[i] = ((uint32_t)s0 & 0xffffu) | ((uint32_t)(s0 + s1) << 16);
    }
}

// ---- table-direct cursor (probe.hip): which run entry is hit g of the call? ------------------------------------------
// A wave owns a CONTIGUOUS range of the call's hit indices and walks it in 64-hit buffers.  Invariant:
// td_rec[m0].prefix <= g0 < td_rec[m0 + 1].prefix for the next buffer start g0.  The window -- records m0 .. m0 + 63, one
// per lane, plus the prefix of record m0 + 64 in lane 0 -- is loaded ONE BUFFER AHEAD, so locate() never waits for memory:
// six shuffles find every lane's record (every record holds >= 1 hit, so 64 records always cover 64 hits), three more
// fetch its run offset and query position.
struct TdCursor {
    uint32_t m0;
    TdRec win;
    uint32_t win64;
    __device__ __forceinline__ void load_window(const ExtendArgs& a, int lane) {
        const uint32_t mi = m0 + (uint32_t)lane;
        win = a.td_rec[mi < a.td_m ? mi : a.td_m];
        if (mi > a.td_m) win.prefix = 0xFFFFFFFFu;  // past the sentinel
        win64 = 0xFFFFFFFFu;
        if (lane == 0 && m0 + 64u <= a.td_m) win64 = a.td_rec[m0 + 64u].prefix;
    }
    __device__ __forceinline__ void seek(const ExtendArgs& a, int lane, uint32_t g0) {
        uint32_t lo = 0, hi = a.td_m;  // td_rec[0].prefix = 0 <= g0 < td_rec[td_m].prefix = num_hits
        while (lo + 1 < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (a.td_rec[mid].prefix <= g0) lo = mid; else hi = mid;
        }
        m0 = lo;
        load_window(a, lane);
    }
    // hit g0 + lane: entry = index of its run entry in the neighbourhood table, qpos = its query position (seed start);
    // then advances the cursor to g0 + 64 and issues the load of the next window
    __device__ __forceinline__ void locate(const ExtendArgs& a, int lane, uint32_t g0, uint64_t& entry, uint32_t& qpos) {
        // R: first hit of the lane's record relative to g0; lanes >= 1 hold values >= 1 (invariant), lane 0 counts as 0;
        // records past the sentinel as infinity
        const uint32_t R = lane == 0 ? 0u : (win.prefix == 0xFFFFFFFFu ? 0xFFFFFFFFu : win.prefix - g0);
        const uint32_t d0 = g0 - (uint32_t)__builtin_amdgcn_readfirstlane((int)win.prefix);  // hits of record m0 in earlier buffers
        uint32_t lo = 0, rv = 0;  // largest window entry whose first hit is <= this lane's hit
#pragma unroll
        for (uint32_t step = 32; step >= 1; step >>= 1) {
            const uint32_t cand = lo + step;
            const uint32_t v = (uint32_t)__shfl((int)R, (int)(cand & 63u), 64);
            if (cand < 64u && v <= (uint32_t)lane) { lo = cand; rv = v; }
        }
        const uint32_t o_lo = (uint32_t)__shfl((int)(uint32_t)win.off, (int)lo, 64);
        const uint32_t o_hi = (uint32_t)__shfl((int)(uint32_t)(win.off >> 32), (int)lo, 64);
        qpos = (uint32_t)__shfl((int)win.qpos, (int)lo, 64);
        const uint32_t k = lo == 0u ? d0 + (uint32_t)lane : (uint32_t)lane - rv;
        entry = (((uint64_t)o_hi << 32) | o_lo) + k;
        // the record that holds hit g0 + 64: entries 1..63 by their lanes, entry 64 by lane 0
        const bool adv = lane == 0 ? (win64 != 0xFFFFFFFFu && win64 - g0 <= 64u) : (R <= 64u);
        m0 += (uint32_t)__popcll(__ballot(adv));
        load_window(a, lane);
    }
};

// SRC: where the anchors come from.  SRC_HITS: the hit list of the general path (64-hit buffers dealt round-robin to the
// waves).  SRC_TD: straight out of the neighbourhood table runs -- no hit list is written or read (16*S + 4*H instead of
// 16*S + 12*H).  SRC_CAND: the {ref_loc, query_loc, hidx} records the context filter (1c) could not decide.
enum : int { SRC_HITS = 0, SRC_TD = 1, SRC_CAND = 2 };

template <int SRC>
__global__ __launch_bounds__(PK_THREADS) void extend_filter_packed_kernel(ExtendArgs a) {
    constexpr bool TD = SRC == SRC_TD;
    __shared__ uint32_t s_pk[PK_TAB];
    __shared__ CandRec s_cand[PK_THREADS / 64][STAGE_CAP];
    __shared__ uint32_t s_l2pre[SRC == SRC_CAND ? L2_NSUB + 1 : 1];  // SRC_CAND: prefix of the sub-lists of the second-level list
    pk_table_init(s_pk, a.sub_mat, PK_THREADS);
    if (SRC == SRC_CAND) {
        // prefix of the sub-list counts (clamped to their capacity), computed by every workgroup for itself -- 256 values, a
        // microsecond -- instead of by a kernel of its own; workgroup 0 also reports the total and the largest raw count
        __shared__ uint32_t s_l2cnt[SRC == SRC_CAND ? L2_NSUB : 1];
        uint32_t raw = 0;
        if (threadIdx.x < L2_NSUB) {
            raw = a.l2_count[threadIdx.x * L2_CNT_STRIDE];
            s_l2cnt[threadIdx.x & (SRC == SRC_CAND ? 0xFFFF : 0)] = min(raw, a.l2_cap);
        }
        __syncthreads();
        if (threadIdx.x <= L2_NSUB) {
            uint32_t pre = 0;
            for (int t = 0; t < (int)threadIdx.x; t++) pre += s_l2cnt[t & (SRC == SRC_CAND ? 0xFFFF : 0)];
            s_l2pre[threadIdx.x & (SRC == SRC_CAND ? 0xFFFF : 0)] = pre;
            if (blockIdx.x == 0 && threadIdx.x == L2_NSUB) *a.l2_total = pre;
        }
        if (blockIdx.x == 0 && threadIdx.x < L2_NSUB) {
            uint32_t mx = raw;
            for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(a.l2_max, mx);
        }
    }
    __syncthreads();
    CandRec* stage = s_cand[threadIdx.x >> 6];
    int n_stage = 0;

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int xdrop = a.xdrop;
    const int fin_batch = a.fin_batch;
    const uint32_t long_cap = a.long_cap;

    // ---- the wave's queue of 64-hit buffers: round-robin over all waves of the grid (hit list), or one contiguous range
    //      per wave (TD) ----
    const uint64_t total_hits = SRC == SRC_CAND ? (uint64_t)s_l2pre[SRC == SRC_CAND ? L2_NSUB : 0] : a.num_hits;
    const uint64_t num_buf = (total_hits + 63) >> 6;
    const uint64_t W = (uint64_t)gridDim.x * (PK_THREADS / 64);
    const uint64_t wid = (uint64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (PK_THREADS / 64) + (threadIdx.x >> 6)));
    const uint64_t G = TD ? 1ull : W;
    const uint64_t buf_end = TD ? ((wid + 1) * num_buf) / W : num_buf;
    uint64_t cur_buf = TD ? (wid * num_buf) / W : wid;
    uint64_t nxt_buf = cur_buf + G;
    auto buf_count = [&](uint64_t b) -> int {
        if (b >= buf_end) return 0;
        uint64_t rem = total_hits - (b << 6);
        return rem >= 64 ? 64 : (int)rem;
    };
    TdCursor cursor = {0u, {0u, 0u, 0ull}, 0u};
    if (TD && cur_buf < buf_end) cursor.seek(a, lane, (uint32_t)(cur_buf << 6));
    uint32_t f_idx = 0;  // SRC_CAND: the fetched record's own hit index ...
    uint32_t f_known = 0, f_tm = 0, f_flags = 3;  // ... and what level 1 knows about it (L2Rec)
    auto fetch = [&](uint64_t b, int cnt) -> Hit {
        Hit h = {0u, 0u};
        f_idx = (uint32_t)(b << 6) + (uint32_t)lane;
        if (SRC == SRC_HITS) {
            if (lane < cnt) h = a.hits[(b << 6) + lane];
        } else if (SRC == SRC_CAND) {
            if (lane < cnt) {
                const uint32_t g = (uint32_t)(b << 6) + (uint32_t)lane;  // record g of the concatenated sub-lists
                uint32_t sl = 0;
#pragma unroll
                for (uint32_t step = L2_NSUB / 2; step >= 1; step >>= 1)
                    if (s_l2pre[(sl + step) & (SRC == SRC_CAND ? 0xFFFFu : 0u)] <= g) sl += step;  // largest sl with prefix[sl] <= g
                const L2Rec c = a.l2_list[(size_t)sl * a.l2_cap + (g - s_l2pre[sl & (SRC == SRC_CAND ? 0xFFFFu : 0u)])];
                h.ref_loc = c.ref_loc;
                h.query_loc = c.query_loc;
                f_idx = c.hidx;
                f_known = c.meta & 0xFFFFu;
                f_flags = (c.meta >> 16) & 3u;
                // level 1's packed left-walk state {T : D} -> {running score (low 16) | best score = T + D (high 16)}
                const int t16 = (int)c.state >> 16;
                f_tm = ((uint32_t)t16 & 0xFFFFu) | ((uint32_t)(t16 + (int)(c.state & 0xFFFFu)) << 16);
            }
        } else if (cnt > 0) {  // (wave-uniform)
            uint64_t entry;
            uint32_t qp;
            cursor.locate(a, lane, (uint32_t)(b << 6), entry, qp);
            if (lane < cnt) {
                h.ref_loc = a.td_pos[entry] + a.seed_size;  // :220
                h.query_loc = qp + a.seed_size;             // :204
            }
        }
        return h;
    };
    int buf_cnt = buf_count(cur_buf), nxt_cnt = buf_count(nxt_buf), consumed = 0;
    Hit buf = fetch(cur_buf, buf_cnt);
    uint32_t buf_idx = f_idx, buf_known = f_known, buf_tm = f_tm, buf_flags = f_flags;
    Hit nxt = fetch(nxt_buf, nxt_cnt);
    uint32_t nxt_idx = f_idx, nxt_known = f_known, nxt_tm = f_tm, nxt_flags = f_flags;

    // ---- per-lane state ----
    int phase = PH_FIN;
    bool has_hit = false, forward = false;
    uint32_t ref_loc = 0, query_loc = 0, hidx = 0;
    uint32_t walked = 0;  // bases walked on this side (multiple of 64)
    uint4 h_tw = {0u, 0u, 0u, 0u}, h_qlo = {0u, 0u, 0u, 0u}, h_qhi = {0u, 0u, 0u, 0u};  // first LEFT window, fetched with the hit
    s16x2 T = {0, 0}, M = {0, 0};
    int bestR = 0, best = 0;
    bool left_known = false;  // SRC_CAND: level 1 settled the left side, `left_best` is its best score
    int left_best = 0;

    for (;;) {
        // ================= 1. advance every live lane by one 64-base window of its current side =================
        // Every lane enters with a fresh window, so the four 16-base steps below read fixed registers (no per-lane
        // selection of the held dwords) and a lane that drops simply sits out the remaining steps.
        if (phase < PH_FIN) {
            const bool left = phase == PH_LEFT;
            // A new hit (right side, nothing walked) also fetches the first window of its LEFT side: both lie in the
            // same 32 bytes of the 2-bit copy, which would otherwise be fetched again after the L2 has turned over (a
            // few microseconds at this miss rate): 1.41 -> 1.30 lines per hit, one wait fewer; with the overlapped-line
            // layout those 32 bytes are always inside one line: ~1.07 lines per hit.
            uint4 qlo, qhi, tw;
            if (left && walked == 0u) {
                qlo = h_qlo; qhi = h_qhi; tw = h_tw;
            } else {
                // signed positions: a long left walk near the block start may reach below 0 (pad bytes)
                const int32_t qpos = left ? (int32_t)query_loc - (int32_t)walked : (int32_t)(query_loc + walked);
                const int32_t qbyte = left ? (qpos >> 1) - 32 : (qpos >> 1);
                const uint8_t* qp = a.query4 + (size_t)(qpos & 1) * a.query4_stride + qbyte;
                qlo = load16u(qp);
                qhi = load16u(qp + 16);
                const int32_t tpos = left ? (int32_t)ref_loc - (int32_t)walked : (int32_t)(ref_loc + walked);
                const int32_t tbyte = left ? (tpos >> 2) - 16 : (tpos >> 2);
                // overlapped-line layout of the 2-bit copies (encode.hip): the line is chosen for the first byte that is
                // needed -- the LEFT window of a new hit, else this window -- so that a hit's [-64, +64) bases sit in ONE line
                const bool fresh = !left && walked == 0u;
                const uint32_t jj0 = (uint32_t)(tbyte + PACK2_BIAS);
                const uint32_t line = (fresh ? jj0 - 16u : jj0) / (uint32_t)PACK2_PAYLOAD;
                const uint8_t* tp = a.ref2 + (size_t)(tpos & 3) * a.ref2_stride + (jj0 + 32u * line);
                tw = load16u(tp);
                if (fresh) {  // same phase copies (qpos, tpos are the anchor itself)
                    h_tw = load16u(tp - 16);
                    h_qlo = load16u(qp - 32);
                    h_qhi = load16u(qp - 16);
                }
            }
            uint32_t tw0, tw1, tw2, tw3, qw0, qw1, qw2, qw3, qw4, qw5, qw6, qw7;  // 4 x 16 bases in walking order
            if (left) {
                const uint32_t R = 0x00010203u, D = 0x88888888u;
                qw0 = __builtin_amdgcn_perm(0u, qhi.w, R) | D; qw1 = __builtin_amdgcn_perm(0u, qhi.z, R) | D;
                qw2 = __builtin_amdgcn_perm(0u, qhi.y, R) | D; qw3 = __builtin_amdgcn_perm(0u, qhi.x, R) | D;
                qw4 = __builtin_amdgcn_perm(0u, qlo.w, R) | D; qw5 = __builtin_amdgcn_perm(0u, qlo.z, R) | D;
                qw6 = __builtin_amdgcn_perm(0u, qlo.y, R) | D; qw7 = __builtin_amdgcn_perm(0u, qlo.x, R) | D;
                tw0 = __builtin_bitreverse32(tw.w); tw1 = __builtin_bitreverse32(tw.z);
                tw2 = __builtin_bitreverse32(tw.y); tw3 = __builtin_bitreverse32(tw.x);
            } else {
                qw0 = qlo.x; qw1 = qlo.y; qw2 = qlo.z; qw3 = qlo.w; qw4 = qhi.x; qw5 = qhi.y; qw6 = qhi.z; qw7 = qhi.w;
                tw0 = tw.x; tw1 = tw.y; tw2 = tw.z; tw3 = tw.w;
            }
            bool alive = true;
            int m = 0;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                if (alive) {
                    const uint32_t td = st == 0 ? tw0 : st == 1 ? tw1 : st == 2 ? tw2 : tw3;
                    const uint32_t q0 = st == 0 ? qw0 : st == 1 ? qw2 : st == 2 ? qw4 : qw6;
                    const uint32_t q1 = st == 0 ? qw1 : st == 1 ? qw3 : st == 2 ? qw5 : qw7;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t qd = j < 4 ? q0 : q1;
                        uint32_t qaddr;  // (query byte j) << 6 in one SDWA shift
                        switch (j & 3) {
                            case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(qaddr) : "v"(6), "v"(qd)); break;
                            case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(qaddr) : "v"(6), "v"(qd)); break;
                            case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(qaddr) : "v"(6), "v"(qd)); break;
                            default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(qaddr) : "v"(6), "v"(qd)); break;
                        }
                        const uint32_t rp = (td >> (4 * j)) & 15u;
                        const uint32_t addr = (rp << 2) | qaddr;  // byte address of entry rp | qbyte << 4
                        const s16x2 e = __builtin_bit_cast(s16x2, *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_pk) + addr));
                        const s16x2 Tb = T.yy;
                        T = __builtin_elementwise_add_sat(Tb, e);  // {t + s0, t + s0 + s1}
                        M = __builtin_elementwise_max(M, T);
                    }
                    m = max((int)M.x, (int)M.y);
                    alive = (m - (int)T.y) <= xdrop;  // :374 / :523, looked at once per 16 bases
                }
            }
            walked += 64;
            if (!alive) {
                if (!left) {  // -> left side (:457-476): anchor-1, anchor-2, ...
                    bestR = m;
                    if (SRC == SRC_CAND && left_known) {
                        best = left_best;
                        phase = PH_FIN;
                    } else {
                        phase = PH_LEFT;
                        walked = 0;
                        T = (s16x2){0, 0};
                        M = (s16x2){0, 0};
                    }
                } else {
                    best = m;
                    phase = PH_FIN;
                }
            } else if (walked + (uint32_t)(64 - CTX_R_BASES) >= long_cap) {  // still alive after long_cap bases: real homology -> exact kernel
                // (walks from the anchor count 64, 128, ...; a left walk resumed behind level 1's context 77 + 64; a right walk resumed there
                //  54 + 64 = 118, which counts as the 128 it replaces: ONE window behind the context instead of two from the anchor)
                forward = true;
                phase = PH_FIN;
            }
        }

        // ================= 2. finalise + refill in batches =================
        const unsigned long long fin = __ballot(phase == PH_FIN);
        const unsigned long long live = __ballot(phase < PH_FIN);
        if (fin != 0ull && (__popcll(fin) >= fin_batch || live == 0ull)) {
            bool cand = false;
            if (phase == PH_FIN && has_hit) cand = forward || bound_passes(a, bestR + best);  // (|bestR + best| < 2^16)
            {
                CandRec cr;
                cr.ref_loc = ref_loc; cr.query_loc = query_loc; cr.hidx = hidx;
                stage_append(stage, n_stage, cand, cr, a.cand_list, a.cand_count, a.cand_cap_recs, lane, lane_lt);
            }
            if (a.audit_list) {  // (tests) the hits this level rejects
                uint2 ar;
                ar.x = ref_loc; ar.y = query_loc;
                wave_append(phase == PH_FIN && has_hit && !cand, ar, a.audit_list, a.audit_count, a.audit_cap, lane, lane_lt);
            }
            unsigned long long need = fin;
            bool got = false;
            Hit mine = {0u, 0u};
            uint32_t mine_idx = 0, mine_known = 0, mine_tm = 0, mine_flags = 3;
            while (need != 0ull) {
                const int avail = buf_cnt - consumed;
                if (avail <= 0) {
                    if (nxt_cnt == 0) break;  // queue exhausted
                    buf = nxt;
                    buf_idx = nxt_idx;
                    buf_known = nxt_known; buf_tm = nxt_tm; buf_flags = nxt_flags;
                    buf_cnt = nxt_cnt;
                    cur_buf = nxt_buf;
                    consumed = 0;
                    nxt_buf += G;
                    nxt_cnt = buf_count(nxt_buf);
                    nxt = fetch(nxt_buf, nxt_cnt);
                    nxt_idx = f_idx;
                    nxt_known = f_known; nxt_tm = f_tm; nxt_flags = f_flags;
                    continue;
                }
                const int rank = __popcll(need & lane_lt);
                const bool take = ((need >> lane) & 1ull) && rank < avail;
                const int src = (consumed + rank) & 63;
                const uint32_t hr = (uint32_t)__shfl((int)buf.ref_loc, src, 64);
                const uint32_t hq = (uint32_t)__shfl((int)buf.query_loc, src, 64);
                const uint32_t hi = SRC == SRC_CAND ? (uint32_t)__shfl((int)buf_idx, src, 64) : (uint32_t)(cur_buf << 6) + (uint32_t)src;
                uint32_t hk = 0, ht = 0, hf = 3;
                if (SRC == SRC_CAND) {
                    hk = (uint32_t)__shfl((int)buf_known, src, 64);
                    ht = (uint32_t)__shfl((int)buf_tm, src, 64);
                    hf = (uint32_t)__shfl((int)buf_flags, src, 64);
                }
                if (take) {
                    mine.ref_loc = hr;
                    mine.query_loc = hq;
                    mine_idx = hi;
                    mine_known = hk; mine_tm = ht; mine_flags = hf;
                    got = true;
                }
                consumed += min(__popcll(need), avail);
                need &= ~__ballot(take);
            }
            if (phase == PH_FIN) {
                forward = false;
                if (got) {
                    has_hit = true;
                    ref_loc = mine.ref_loc;
                    query_loc = mine.query_loc;
                    hidx = mine_idx;
                    bestR = 0;
                    best = 0;
                    bool skip = false;
                    if (a.rm) skip = !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);  // rm :239-244,:305-333
                    // anchors beyond the block (hand-made seed words only) are never extended, see extend_filter_kernel
                    left_known = false;
                    if (skip || ref_loc > a.ref_len || query_loc > a.query_len) {
                        phase = PH_FIN;
                    } else if (SRC == SRC_CAND && mine_flags == 0u) {  // both sides settled by level 1: only the verdict is left
                        bestR = (int)mine_known;
                        best = (int)mine_tm;
                        phase = PH_FIN;
                    } else if (SRC == SRC_CAND && mine_flags == 2u) {  // right side settled; the left walk continues behind level 1's context:
                        bestR = (int)mine_known;                          // the seed window (bounded there) + the 58 bases in front of it
                        phase = PH_LEFT;
                        walked = (uint32_t)CTX_L_BASES + a.left_skip;
                        const short t16 = (short)(mine_tm & 0xFFFFu), m16 = (short)(mine_tm >> 16);
                        T = (s16x2){t16, t16};
                        M = (s16x2){m16, m16};
                    } else {
                        phase = PH_RIGHT;  // :299-324
                        walked = 0;
                        T = (s16x2){0, 0};
                        M = (s16x2){0, 0};
                        if (SRC == SRC_CAND && mine_flags == 1u) {  // left side settled by level 1
                            left_known = true;
                            left_best = (int)mine_known;
                            if (a.l2_right_state) {  // ... and the right walk resumes behind level 1's context from its packed state: an upper
                                walked = (uint32_t)CTX_R_BASES;  // bound of the exact walk's (running score, best) there, like the left one's
                                const short t16 = (short)(mine_tm & 0xFFFFu), m16 = (short)(mine_tm >> 16);
                                T = (s16x2){t16, t16};
                                M = (s16x2){m16, m16};
                            }
                        }
                    }
                } else {
                    has_hit = false;
                    phase = PH_IDLE;
                }
            }
        }
        if (__ballot(phase != PH_IDLE) == 0ull) break;
    }
    stage_flush(stage, n_stage, a.cand_list, a.cand_count, a.cand_cap_recs, lane);
}

// =====================================================================================================================
// 1d. the X-drop filter on the CONTEXT table with CLASS scoring: the target bases travel with the seed table entry (no random
//     target access), 6 bases per table lookup
// =====================================================================================================================
// A filter that fetches the target per hit is priced in random 128-byte lines: one per hit, ~57 G lines/s on the whole chip
// (tools/micro/gather_bw.hip) -- 173 M hits of a call cannot take less than ~3 ms however little arithmetic they need.  With
// 288 GB of HBM the neighbourhood table (probe.hip) carries, next to every seed position, the 2-bit target bases the filter looks
// at: 54 from the anchor on and the 58 in front of the seed window (CtxRec, 32 bytes; rounds 2-4: 48 + the 64 left of the anchor).  The hits of a call are then ONE SEQUENTIAL STREAM of
// records, the query windows of a wave's 64 hits are a couple of L1-resident lines, and the filter is an arithmetic kernel: every
// hit costs the same, a wave walks its contiguous range of 64-hit buffers in lockstep, straight-line code.
// The round-2 form of this kernel scored two bases per LDS lookup with exact pair scores: ~39 VALU + 8 LDS per 16 bases, 365 VALU
// + 68 LDS wave-instructions per 64 hits, bound by instruction issue (profiles/r02).  The filter only needs an UPPER bound of
// every walk, and the pair score is bounded by a function of (target code XOR query code) alone:
//     x = t ^ q :  0 = same base,  2 = transition (A<->G, C<->T),  1 / 3 = the two transversion classes,
//     cls[x] = max of the matrix entries of that class (engine.hip class_scores(): for HOXD70 100 / -114 / -31 / -123 against
//     the exact 91..100 / -114 / -31 / -123..-125 -- the drift of a random walk is -42.0 instead of -43.4 per base).
// With target AND query at 2 bits per base one XOR per 16 bases yields the class string, and a 12-bit field of it -- SIX bases
// -- indexes a 4096-entry table: 19 lookups per hit instead of 56, no byte permutes.
// The walk is kept as (T, N) in ONE register, two int16: T = running score, N = T - best <= 0 (minus the current drop).  A field
// with score `sum` and largest prefix score `mx` does  T' = T + sum,  N' = min(N + sum, sum - mx)  -- two VALU ops on a 4-byte
// table entry {sum : sum - mx} (cls_step below; |T| and |N| stay far below 2^15: class scores are clamped to >= -255).  A side is
// NOT stopped or frozen when it drops (:374 / :523): it walks on to the end of its context, which can only raise its best score
// (still an upper bound), so all 19 table reads of a buffer are independent of the scores and the code is straight-line.  W keeps
// the lowest N seen at a field end: "alive" = never more than xdrop below the best at a field end.  (Asking only at the end of the
// context -- N_end >= -xdrop, implied by the former, so it errs on the forwarding side -- saves one op per field and forwards 6.3 %
// instead of 4.6 % of the hits: the second level then costs 0.08 ms more per call than the filter saves.)
// What bounds the kernel (profiles/r03, tools/pmc_mem.sh): the L1's miss path.  A CU has ~57 lines of 128 bytes in
// flight at an L1->L2 read latency of ~840 cycles under this load (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ;
// TCP_PENDING_STALL_CYCLES is 61 % of the kernel's cycles), i.e. ~9 bytes per cycle and CU = 5.2 TB/s for the chip by the TCC
// counters (0.63-0.68 of the 8 TB/s peak), against 6.2 TB/s for a pure stream of the same 32-byte records with no arithmetic at all
// (tools/micro/stream_rec2.hip) and 5.8-6.0 TB/s for a bare read of the same random 2.5 KB pieces (tools/micro/piece_order.hip).
// Measured on this kernel, each changing NOTHING in its duration: 12 % fewer VALU instructions (no W), half the LDS bytes (these
// 4-byte entries instead of 8-byte ones; the LDS pipe stays ~67 % busy: 64 random reads conflict in the banks whatever their
// width), the next buffer's records requested one iteration ahead.  They are kept because the kernel then
// leaves more of the CU to the kernels of the other calls in flight (0.96 -> 1.04 Gbp/s for the whole pass).  Only fewer LINES
// per hit would make it faster: reading half of every record did not (same lines, -9 % from the smaller loads alone).
// Why the bound holds: with u_i >= s_i pointwise the bounded walk's drop max_i<=k(Q_i) - Q_k never exceeds the exact walk's, so it
// cannot stop earlier, and its best is taken over a superset of positions.  Codes >= 4 (soft-masked, N, separators, other IUPAC
// letters) are stored as code 0 on both sides; cls[] covers every matrix entry such a pair could have had, for the codes that
// occur in the two blocks (engine.hip).
// Verdicts (all conservative):
//   both sides dropped inside the context and bestR + bestL cannot pass (:608-633)   -> rejected here (~96 % of all hits)
//   a side still alive at the end of its context (0.8 % right, 0.5 % left on random hits; rounds 2-4: 1.8 % / 2.5 %), or the bound passes
//        -> L2Rec (kernels.h) to the second level: kernel 1b on that list (SRC_CAND), which walks only what is still open with exact
//           pair scores and decides between reject and the exact kernels.
constexpr int CTX_THREADS = 1024;      // two workgroups per CU = 8 waves per SIMD; the class table is built once per 16 waves
constexpr int CTX_THREADS_MAX = 1024;  // (ExtendArgs::ctx_threads, option ctx_threads, overrides the default per launch)
constexpr int CTX_STAGE_FLUSH = 32;                    // forwards are rare (~4 % of the hits): flush early, keep the stage small
constexpr int CTX_STAGE_CAP = CTX_STAGE_FLUSH - 1 + 64 + 1;  // 96 records of 20 bytes per wave
constexpr int CLS_TAB = 4096;                       // 12-bit fields: six bases
constexpr int CLS_TAIL = 256;                       // the left context ends with a four-base field (64 = 10 x 6 + 4)
constexpr int CLS_LDS_DWORDS = CLS_TAB + CLS_TAIL;  // 4-byte entries: 16 KB + 1 KB
constexpr bool CLS_TRACK_DROP = true;               // keep the lowest N seen at a field end (see cls_step)

// entry of a field: {sum : sum - mx} as two int16 (sum = score of the field, mx = its largest prefix score, >= 0)
__device__ __forceinline__ void cls_table_init(uint32_t* __restrict__ s_cls, const int* cls, int nthreads) {
    const int c0 = cls[0], c1 = cls[1], c2 = cls[2], c3 = cls[3];
    for (int i = threadIdx.x; i < CLS_TAB + CLS_TAIL; i += nthreads) {
        const int nb = i < CLS_TAB ? 6 : 4;
        const int f = i < CLS_TAB ? i : i - CLS_TAB;
        int sum = 0, mx = 0;  // (a negative prefix never raises the best score)
        for (int k = 0; k < nb; k++) {
            const int x = (f >> (2 * k)) & 3;
            sum += x == 0 ? c0 : x == 1 ? c1 : x == 2 ? c2 : c3;
            mx = max(mx, sum);
        }
        s_cls[i] = ((uint32_t)sum << 16) | ((uint32_t)(sum - mx) & 0xFFFFu);
    }
}

// byte address (inside a table of 4-byte entries) of the 12-bit field that starts at bit O of the dword string w0 | w1 << 32
template <int O>
__device__ __forceinline__ uint32_t cls_field_addr(uint32_t w0, uint32_t w1) {
    constexpr uint32_t MASK = 0x3FFCu;
    if (O + 12 <= 32) return O >= 2 ? ((w0 >> (O - 2)) & MASK) : ((w0 << (2 - O)) & MASK);
    return __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(O - 2)) & MASK;  // the field straddles the two dwords
}

__device__ __forceinline__ uint32_t mul24(uint32_t x, uint32_t s_uniform) {  // full-rate 24-bit multiply (v_mul_lo_u32 is quarter rate)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(s_uniform), "v"(x));
    return r;
}

// One field.  P = {T : N} as two int16: T = running score, N = T - best <= 0 (minus the current drop).  With the field's
// (sum, mx):  T' = T + sum,  N' = min(N + sum, sum - mx)   [= -(max(D, mx) - sum) for D = -N]
//   v_pk_add_i16 with op_sel adds the entry's HIGH half (sum) to both halves of P;
//   v_min_i16 (SDWA, destination's other half preserved) takes the low half against the entry's low half (sum - mx).
// Two VALU and one 4-byte LDS read per six bases.  (Round 3 started with 8-byte entries {sum * 65535, INT16_MIN : mx} and
// P = pk_max(P, mxw) + add: the same two ops, but a wave's 64 random 8-byte reads keep the LDS pipe busy ~8.6 cycles each --
// SQ_LDS_IDX_ACTIVE was 65 % of the kernel's cycles, 70 % of them bank conflicts -- and the LDS, not the VALU or the record
// stream, set the pace: 12 % fewer VALU instructions changed nothing, half the record bytes 9 %.)
// W (CLS_TRACK_DROP): the lowest N seen at a field end, one more v_pk_min per field.
__device__ __forceinline__ void cls_step(const uint32_t* __restrict__ s_tab, uint32_t addr, uint32_t& P, uint32_t& W) {
    const uint32_t e = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_tab) + addr);
    asm("v_pk_add_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(P) : "v"(P), "v"(e));
    asm("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(P) : "v"(e));
    if (CLS_TRACK_DROP) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        W = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, W), __builtin_bit_cast(s16x2, P)));
    }
}

// The left walk crosses the seed window first (anchor - 1 is its last base).  Every base there is bounded by the largest class score
// -- exact for the care positions of the k-mer's own seed word, an upper bound for a transition or a don't-care position --, so the
// walk enters the record's left context with T = seed_size x that score and N = 0 (a run of non-negative steps ends at its own
// best): the same pointwise-upper-bound argument as for the class scores themselves, at no lookup.  What it costs: the bound on
// bestL is looser by what the seed window really scores below that (about 1000 for 12of19), which only matters for the verdict
// bestR + bestL >= hspthresh of hits with both sides dropped: 1e-4 of the hits.
__device__ __forceinline__ uint32_t cls_seed_state(const ExtendArgs& a) {
    const int cmax = max(max(a.cls[0], a.cls[1]), max(a.cls[2], a.cls[3]));
    return (uint32_t)((int)a.left_skip * max(cmax, 0)) << 16;
}

// ONE_COPY: the query windows come out of the UNSHIFTED 2-bit copy of either strand (copy 0 of the sixteen) with funnel shifts, instead
// of aligned dwords of the copy that starts at the position's phase.  With ~78 hits per query position a 64-hit buffer holds about
// one position and the sixteen copies cost nothing; with ~5 hits per position (--notransition, small targets) it holds 13 positions
// whose windows lie in 13 different copies x 2 strands = 26 lines per buffer that miss the L1 every time and the L2 half of the time
// (the record stream flushes both): 3.1 HBM lines per 160-byte run where a bare walk of the same runs needs 2.1
// (tools/micro/hit_shaped.hip).  From copy 0 the windows of consecutive positions share their lines.
template <bool ONE_COPY>
__global__ __launch_bounds__(CTX_THREADS_MAX, 8) void extend_filter_cls_kernel(ExtendArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_cls[CLS_LDS_DWORDS];  // 4-byte entry per 6-base class field (16 KB) + the 4-base tail fields (1 KB)
    extern __shared__ L2Rec s_l2_dyn[];          // [waves of the workgroup][CTX_STAGE_CAP]
    cls_table_init(s_cls, a.cls, (int)blockDim.x);
    __syncthreads();
    const uint32_t* __restrict__ s_tail = s_cls + CLS_TAB;
    L2Rec* stage = s_l2_dyn + (threadIdx.x >> 6) * CTX_STAGE_CAP;
    int n_stage = 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int xdrop = a.xdrop;
    const uint4* __restrict__ ctx = reinterpret_cast<const uint4*>(a.td_ctx);
    const uint32_t seed_state = cls_seed_state(a);  // the left walk's {T : N} behind the seed window

    // a wave takes a contiguous range of TD_CHUNK_HITS-hit chunks (64 buffers each); the record that holds a chunk's first hit
    // was noted by the probe (td_chunk), so no wave has to search for its starting point
    const uint64_t num_buf = (a.num_hits + 63) >> 6;
    const uint64_t n_chunks = (a.num_hits + TD_CHUNK_HITS - 1) / TD_CHUNK_HITS;
    const uint64_t W = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wid = (uint64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint64_t c_lo = (wid * n_chunks) / W, c_hi = ((wid + 1) * n_chunks) / W;
    const uint64_t b_lo = c_lo * (TD_CHUNK_HITS / 64);
    const uint64_t b_hi = min(c_hi * (TD_CHUNK_HITS / 64), num_buf);
    if (b_lo >= b_hi) return;
    const uint32_t my_sub = (uint32_t)wid & (uint32_t)(L2_NSUB - 1);  // this wave's sub-list of the second-level list
    L2Rec* __restrict__ my_list = a.l2_list + (size_t)my_sub * a.l2_cap;
    uint32_t* __restrict__ my_count = a.l2_count + my_sub * L2_CNT_STRIDE;
    // Which record (query position) does hit g belong to?  The probe left a HEAD-BIT map of the call's hits (bit g set <=> a
    // record starts at hit g), so the record index of hit g is (number of head bits in [0, g]) - 1: per 64-hit buffer ONE 64-bit
    // word, a running scalar count and two v_mbcnt -- no search, no shuffles (the TdCursor of 1b spends six ds_bpermute + ~40
    // VALU per buffer on the same question).  td_chunk gives the count at the wave's first hit.  The map is read through the
    // constant address space: a wave-uniform address there is a SCALAR load (s_load_dwordx2), no VALU or vector-memory slot.
    // Latency: the head word is fetched four buffers ahead, the TdRec gather two and the records one buffer ahead (see the loop).
    // Requests past the wave's range read valid memory (the next wave's buffers, the map's zero words, 16 KB of slack behind the
    // table) and are never used -- conditional loads would turn every s_waitcnt of the loop into a drain.
    typedef const uint64_t __attribute__((address_space(4))) * HeadPtr;
    HeadPtr head = (HeadPtr)a.td_bits;
    const uint32_t stride16 = (uint32_t)(a.q2_stride >> 4) & 0xFFFFFFu;  // (the sixteen copies of a strand stay below 4 GB, engine.hip)
    uint32_t cbefore;  // head bits in [0, first hit of the next buffer to be located)
    auto locate = [&](uint64_t B) -> uint32_t {  // record index of hit (buffer start + lane), advances cbefore
        const uint64_t Bs = B >> 1;              // bits 1..lane of B = the bits below `lane` of Bs
        const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(Bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bs, cbefore - 1u + (uint32_t)(B & 1ull)));
        cbefore += (uint32_t)__builtin_popcountll(B);
        return idx;
    };
    struct Stage {
        uint4 c0, tl;      // the record (CtxRec): c0.x = seed position, c0.y .. tl.w = its 224-bit context string
        uint32_t q[7];     // the query's side of that string: 54 bases of this strand from the anchor on, then 58 of the other strand
        uint32_t query_loc;
    };
    // the query string out of the two windows: qr = 4 dwords from the anchor on (this strand), ql = 4 dwords of the other strand's window
    auto merge_q = [&](Stage& S, uint32_t qr0, uint32_t qr1, uint32_t qr2, uint32_t qr3, uint32_t ql0, uint32_t ql1, uint32_t ql2, uint32_t ql3) {
        S.q[0] = qr0; S.q[1] = qr1; S.q[2] = qr2;
        S.q[3] = (qr3 & 0xFFFu) | (ql0 << 12);  // bits 96..107: right bases 48..53; from bit 108 on: the left window
        S.q[4] = __builtin_amdgcn_alignbit(ql1, ql0, 20);
        S.q[5] = __builtin_amdgcn_alignbit(ql2, ql1, 20);
        S.q[6] = __builtin_amdgcn_alignbit(ql3, ql2, 20);
    };
    uint64_t w0, w1, w2;  // head words of the buffers one, two and three ahead of the one being requested
    TdRec hnext;          // TdRec of this lane's hit in the NEXT buffer to be requested
    auto advance_map = [&](uint64_t b_req) {  // after the request of buffer b_req: gather for b_req + 1, map word for b_req + 4
        hnext = a.td_rec[locate(w0)];
        w0 = w1;
        w1 = w2;
        w2 = head[b_req + 4];
    };
    auto request = [&](uint64_t b, Stage& S) {  // uses hnext = record of buffer b
        // (lanes past the call's last hit stay on the last record and run up to 63 entries past its run: still inside the table
        //  allocation -- the engine keeps a page of slack behind it -- and their verdict is discarded)
        const uint64_t entry = hnext.off + (uint64_t)((uint32_t)(b << 6) + (uint32_t)lane - hnext.prefix);  // run offset + index inside the run
        S.c0 = ctx[2 * entry];  // two aligned 16-byte loads: the stream of the kernel
        S.tl = ctx[2 * entry + 1];  // (28-byte records were measured in round 5 and lost: profiles/r05/exp_rec28.txt)
        const uint32_t query_loc = hnext.qpos + a.seed_size;  // :204
        S.query_loc = query_loc;
        // copy (pos & 3, (pos >> 2) & 3) = copy number pos & 15, dword pos >> 4 (encode.hip); strides are multiples of 16 bytes, so
        // the offset is ONE 32-bit value next to a scalar base
        // the left context starts in front of the seed (CtxRec): the other strand's forward window at len - anchor + seed_size
        const uint32_t lp = a.query_len - query_loc + a.left_skip;
        if (ONE_COPY) {
            // copy 0: dword j holds bases [16 j, 16 j + 16), base 16 j + k in bits 2k, 2k + 1; the window at position p is the bit string
            // from bit 2 (p & 15) of dword p >> 4 on: one more dword per side and a funnel shift per dword
            uint4 t;
            uint32_t t4;
            const uint8_t* rpp = a.q2_own + ((query_loc >> 4) << 2);
            __builtin_memcpy(&t, __builtin_assume_aligned(rpp, 4), 16);
            __builtin_memcpy(&t4, __builtin_assume_aligned(rpp + 16, 4), 4);
            const uint32_t sr = (query_loc & 15u) << 1;
            uint4 u;
            uint32_t u4;
            const uint8_t* lpp = a.q2_other + ((lp >> 4) << 2);
            __builtin_memcpy(&u, __builtin_assume_aligned(lpp, 4), 16);
            __builtin_memcpy(&u4, __builtin_assume_aligned(lpp + 16, 4), 4);
            const uint32_t sl = (lp & 15u) << 1;
            merge_q(S, __builtin_amdgcn_alignbit(t.y, t.x, sr), __builtin_amdgcn_alignbit(t.z, t.y, sr), __builtin_amdgcn_alignbit(t.w, t.z, sr),
                    __builtin_amdgcn_alignbit(t4, t.w, sr), __builtin_amdgcn_alignbit(u.y, u.x, sl), __builtin_amdgcn_alignbit(u.z, u.y, sl),
                    __builtin_amdgcn_alignbit(u.w, u.z, sl), __builtin_amdgcn_alignbit(u4, u.w, sl));
        } else {
            const uint32_t po = (mul24(query_loc & 15u, stride16) << 4) + ((query_loc >> 4) << 2);
            uint4 t, u;
            __builtin_memcpy(&t, __builtin_assume_aligned(a.q2_own + po, 4), 16);
            const uint32_t qo = (mul24(lp & 15u, stride16) << 4) + ((lp >> 4) << 2);
            __builtin_memcpy(&u, __builtin_assume_aligned(a.q2_other + qo, 4), 16);
            merge_q(S, t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w);
        }
    };
    auto score = [&](uint64_t b, const Stage& S) {
        const bool valid = (b << 6) + (uint64_t)lane < a.num_hits;
        const uint32_t query_loc = S.query_loc;
        const uint32_t ref_loc = S.c0.x + a.seed_size;  // :220
        bool skip = !valid;
        if (a.rm) skip = skip || !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);  // rm :239-244,:305-333: total stays 0
        // ---- the class string: 18 six-base fields (9 right, 9 left) and the left side's four-base tail ----
        const uint32_t x0 = S.c0.y ^ S.q[0], x1 = S.c0.z ^ S.q[1], x2 = S.c0.w ^ S.q[2], x3 = S.tl.x ^ S.q[3], x4 = S.tl.y ^ S.q[4],
                       x5 = S.tl.z ^ S.q[5], x6 = S.tl.w ^ S.q[6];
        // ---- right side (:326-453): 54 bases = 9 fields ----
        uint32_t P = 0, Wd = 0;
        cls_step(s_cls, cls_field_addr<0>(x0, x1), P, Wd);
        cls_step(s_cls, cls_field_addr<12>(x0, x1), P, Wd);
        cls_step(s_cls, cls_field_addr<24>(x0, x1), P, Wd);
        cls_step(s_cls, cls_field_addr<4>(x1, x2), P, Wd);
        cls_step(s_cls, cls_field_addr<16>(x1, x2), P, Wd);
        cls_step(s_cls, cls_field_addr<28>(x1, x2), P, Wd);
        cls_step(s_cls, cls_field_addr<8>(x2, x3), P, Wd);
        cls_step(s_cls, cls_field_addr<20>(x2, x3), P, Wd);
        cls_step(s_cls, cls_field_addr<0>(x3, x4), P, Wd);
        // alive: never more than xdrop below its best at a field end (:374; without W: at the end of the context)
        const bool r_alive = (int)(short)((CLS_TRACK_DROP ? Wd : P) & 0xFFFFu) >= -xdrop;
        const int bestR = ((int)P >> 16) - (int)(short)(P & 0xFFFFu);  // best = T - N
        const uint32_t PR = P;  // the right walk's register at the end of its context: level 2 resumes from it (L2Rec, flags 1)
        // ---- left side (:478-604): the seed window bounded by seed_state (no lookup), then 58 bases = 9 fields + a four-base tail ----
        P = seed_state; Wd = 0;
        cls_step(s_cls, cls_field_addr<12>(x3, x4), P, Wd);
        cls_step(s_cls, cls_field_addr<24>(x3, x4), P, Wd);
        cls_step(s_cls, cls_field_addr<4>(x4, x5), P, Wd);
        cls_step(s_cls, cls_field_addr<16>(x4, x5), P, Wd);
        cls_step(s_cls, cls_field_addr<28>(x4, x5), P, Wd);
        cls_step(s_cls, cls_field_addr<8>(x5, x6), P, Wd);
        cls_step(s_cls, cls_field_addr<20>(x5, x6), P, Wd);
        cls_step(s_cls, cls_field_addr<0>(x6, 0u), P, Wd);
        cls_step(s_cls, cls_field_addr<12>(x6, 0u), P, Wd);
        cls_step(s_tail, (x6 >> 22) & 0x3FCu, P, Wd);
        const bool l_alive = (int)(short)((CLS_TRACK_DROP ? Wd : P) & 0xFFFFu) >= -xdrop;  // (:523)
        const int bestL = ((int)P >> 16) - (int)(short)(P & 0xFFFFu);
        const bool fwd = !skip && (r_alive || l_alive || bound_passes(a, bestR + bestL));
        // what is known travels with the anchor (kernels.h L2Rec): level 2 walks only what is still open
        const unsigned long long fm = __ballot(fwd);
        if (fm) {
            L2Rec cr;
            cr.ref_loc = ref_loc;
            cr.query_loc = query_loc;
            cr.hidx = (uint32_t)(b << 6) + (uint32_t)lane;
            const uint32_t fl = (r_alive ? 1u : 0u) | (l_alive ? 2u : 0u);  // 0 (both settled, the bound passes) -> 3: level 2 re-walks both
            const uint32_t Ps = (fl == 1u && a.l2_right_state) ? PR : P;    // the walk level 2 resumes: the left one, or the right one when it alone is open
            cr.state = (Ps & 0xFFFF0000u) | ((0u - Ps) & 0xFFFFu);          // as {T : D = -N}
            cr.meta = (uint32_t)(r_alive ? bestL : bestR) | ((fl ? fl : 3u) << 16);
            // -> the wave's LDS stage (rank by v_mbcnt on the ballot), flushed 64 records at a time
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            if (fwd) stage[n_stage + (int)rank] = cr;
            n_stage += __popcll(fm);
            __builtin_amdgcn_wave_barrier();
            if (n_stage >= CTX_STAGE_FLUSH) {
                const int k = n_stage < 64 ? n_stage : 64;
                stage_flush(stage, k, my_list, my_count, a.l2_cap, lane);
                const int rest = n_stage - k;
                L2Rec tmp = cr;
                if (lane < rest) tmp = stage[k + lane];
                __builtin_amdgcn_wave_barrier();
                if (lane < rest) stage[lane] = tmp;
                __builtin_amdgcn_wave_barrier();
                n_stage = rest;
            }
        }
        if (a.audit_list) {  // (tests) the hits this level rejects
            uint2 ar;
            ar.x = ref_loc;
            ar.y = query_loc;
            wave_append(valid && !skip && !fwd, ar, a.audit_list, a.audit_count, a.audit_cap, lane, lane_lt);
        }
    };
    // Software pipeline, two register sets: while buffer b is scored the records and query windows of buffer b + 1 are in
    // flight, the position-record gather of b + 2 and the map word of b + 5.  A wave alone keeps only one buffer's loads in flight;
    // at 8 waves per SIMD that left the kernel waiting for memory latency (half the record bytes: -9 %; 12 % fewer VALU
    // instructions: -0 %), not for bandwidth or issue slots.
    Stage SA, SB;
    {
        const uint64_t h0 = head[b_lo];
        w0 = head[b_lo + 1];
        w1 = head[b_lo + 2];
        w2 = head[b_lo + 3];
        cbefore = a.td_chunk[c_lo] + 1u - (uint32_t)(h0 & 1ull);
        hnext = a.td_rec[locate(h0)];
        request(b_lo, SA);
        advance_map(b_lo);  // hnext = record of b_lo + 1
    }
    for (uint64_t b = b_lo; b < b_hi; b += 2) {
        request(b + 1, SB);
        advance_map(b + 1);
        score(b, SA);
        if (b + 1 >= b_hi) break;
        request(b + 2, SA);
        advance_map(b + 2);
        score(b + 1, SB);
    }
    stage_flush(stage, n_stage, my_list, my_count, a.l2_cap, lane);
}

// =====================================================================================================================
// 1e. the class filter of a KEY-ORDERED call (join.h, join.hip): the same verdicts as 1d on the same (record, position) pairs, the
//     pairs enumerated per seed key
// =====================================================================================================================
// 1d streams the hits in query order: every hit brings its own 32-byte record through the memory system, a buffer's 64 lanes hold 64
// different records and one or two query positions, and a field's LDS address costs shift + mask (+ funnel shift).  Here a wave takes a
// TILE of 64 consecutive records of the entry list (one or two keys' runs, all of class c = c query positions each) and walks the c
// positions in lockstep:
//   * the record is fetched once for c hits and lives in registers as its 19 FIELD WORDS -- fw_k(t) = the byte offset of field k of the
//     record's class string in the table -- computed once per tile;
//   * the query side of every address was computed once per position (QRecX, join_qx_kernel): shift and mask are linear over xor, so
//     the address of field k of (t ^ q) is fw_k(t) ^ fw_k(q): ONE v_xor per lookup;
//   * right and left walk are interleaved (two independent dependency chains);
//   * work is claimed dynamically, JOIN_GRAIN steps at a time: the per-tile cost differs by class, static shares finished unevenly
//     (142 -> 215 G hits/s in the prototype, tools/micro/join_proto.hip).
// Table, walk state, verdicts and the L2Rec hand-over are those of 1d (the four-base tail table included), so both forms forward the
// same hits with the same state; L2Rec::hidx is the hit's entry index inside its run (ExtendArgs::join).
constexpr int JOIN_THREADS = 1024;
constexpr int JOIN_NFR = 9, JOIN_NFL = 10;  // lookups of the right / left walk (the left one ends with the tail field): CtxRec's cut

__global__ __launch_bounds__(JOIN_THREADS, 8) void join_filter_kernel(ExtendArgs a, JoinArgs jn) {
    __shared__ __attribute__((aligned(16))) uint32_t s_cls[CLS_LDS_DWORDS];
    extern __shared__ L2Rec s_l2_dyn[];
    cls_table_init(s_cls, a.cls, (int)blockDim.x);
    __syncthreads();
    L2Rec* stage = s_l2_dyn + (threadIdx.x >> 6) * CTX_STAGE_CAP;
    int n_stage = 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int xdrop = a.xdrop;
    const uint4* __restrict__ ctx = reinterpret_cast<const uint4*>(a.td_ctx);
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint32_t my_sub = wid & (uint32_t)(L2_NSUB - 1);
    L2Rec* __restrict__ my_list = a.l2_list + (size_t)my_sub * a.l2_cap;
    uint32_t* __restrict__ my_count = a.l2_count + my_sub * L2_CNT_STRIDE;
    const JoinHead* __restrict__ H = jn.head;
    const unsigned long long work_total = H->work_total;
    const uint32_t seed_state = cls_seed_state(a);  // (1d: the left walk enters its context behind the seed window)

    for (;;) {
        // claim JOIN_GRAIN work units (a unit = one 64-hit step of one tile)
        unsigned long long g = 0;
        if (lane == 0) g = atomicAdd(&jn.head_rw->work_next, (unsigned long long)JOIN_GRAIN);
        g = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g);
        if (g >= work_total) break;
        const unsigned long long w_lo = g, w_hi = min(g + (unsigned long long)JOIN_GRAIN, work_total);
        for (int c = JOIN_CMAX; c >= 1; c--) {
            const unsigned long long B = H->work_base[c], vb = H->vbase[c], tot = H->vbase[c + 1] - vb;
            const unsigned long long ntiles = (tot + 63ull) >> 6;
            if (ntiles == 0 || w_hi <= B || w_lo >= B + ntiles * (unsigned long long)c) continue;
            const unsigned long long m_lo = w_lo <= B ? 0ull : min((w_lo - B + (unsigned long long)c - 1ull) / (unsigned long long)c, ntiles);
            const unsigned long long m_hi = min((w_hi - B + (unsigned long long)c - 1ull) / (unsigned long long)c, ntiles);
            if (m_lo >= m_hi) continue;
            const uint32_t e_first = H->cls_first[c], e_last = H->cls_first[c + 1];  // entries of the class (e_last > e_first: tot > 0)
            // the entry that holds the first record of tile m_lo: the last e with vstart[e] <= v (uniform binary search, once per claim and class)
            uint32_t e_cur;
            {
                const unsigned long long v0 = vb + (m_lo << 6);
                uint32_t lo = e_first, hi = e_last - 1u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (jn.vstart[mid] <= v0) lo = mid; else hi = mid - 1u;
                }
                e_cur = lo;
            }
            for (unsigned long long m = m_lo; m < m_hi; m++) {
                const unsigned long long v = vb + (m << 6) + (unsigned long long)lane;
                const unsigned long long v_end = vb + (m << 6) + 63ull;
                const bool valid = v < vb + tot;
                // ---- which entry?  the starts of the next sixteen entries, one per lane group; a tile spans one to three entries as a rule ----
                uint32_t e = e_cur;
                for (;;) {
                    const uint32_t ej = min(e_cur + 1u + (uint32_t)(lane & 15), e_last);
                    const unsigned long long ps = jn.vstart[ej];
                    bool more = false;
#pragma unroll 1
                    for (int j = 0; j < 16; j++) {
                        const unsigned long long pj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(ps >> 32), j) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ps, j);
                        if (e_cur + 1u + (uint32_t)j >= e_last || pj > v_end) break;  // (uniform) the class ends, or entry j starts behind the tile
                        e += (v >= pj) ? 1u : 0u;
                        more = j == 15;
                    }
                    if (!more) break;
                    e_cur += 16u;
                }
                const uint4 en = jn.ent[e];
                const unsigned long long p_start = jn.vstart[e];
                const uint32_t tidx = valid ? (uint32_t)(v - p_start) : 0u;  // entry index inside the run (lanes past the class stay on entry 0 of the last run)
                const unsigned long long rec = (((unsigned long long)en.y << 32) | en.x) + tidx;
                e_cur = (uint32_t)__builtin_amdgcn_readlane((int)e, 63);
                const uint4 c0 = ctx[2 * rec], tl = ctx[2 * rec + 1];
                const uint32_t ref_loc = c0.x + a.seed_size;  // :220
                // the record's field words, once per tile
                uint32_t tf[JOIN_NFR + JOIN_NFL];
                {
                    const uint32_t w[7] = {c0.y, c0.z, c0.w, tl.x, tl.y, tl.z, tl.w};  // the 224-bit context string (CtxRec)
#pragma unroll
                    for (int k = 0; k < JOIN_NFR + JOIN_NFL - 1; k++) {
                        const int bit = 12 * k, d = bit >> 5, o = bit & 31;
                        const uint64_t two = (uint64_t)w[d] | ((uint64_t)(d + 1 < 7 ? w[d + 1 < 7 ? d + 1 : d] : 0u) << 32);
                        tf[k] = (uint32_t)((two >> o) << 2) & 0x3FFCu;
                    }
                    tf[JOIN_NFR + JOIN_NFL - 1] = ((w[6] >> 22) & 0x3FCu) | JOIN_TAIL_OFF;
                }
                const uint32_t* __restrict__ qp = jn.qx + (size_t)en.z * JOIN_QX_DW;
                for (int qi = 0; qi < c; qi++, qp += JOIN_QX_DW) {
                    uint32_t q[JOIN_QX_DW];
#pragma unroll
                    for (int k = 0; k < JOIN_QX_DW / 4; k++) {
                        const uint4 t4 = reinterpret_cast<const uint4*>(qp)[k];
                        q[4 * k] = t4.x; q[4 * k + 1] = t4.y; q[4 * k + 2] = t4.z; q[4 * k + 3] = t4.w;
                    }
                    uint32_t PR = 0, WR = 0, PL = seed_state, WL = 0;
#pragma unroll
                    for (int k = 0; k < JOIN_NFL; k++) {  // (the two walks interleaved: independent dependency chains)
                        if (k < JOIN_NFR) cls_step(s_cls, tf[k] ^ q[1 + k], PR, WR);
                        cls_step(s_cls, tf[JOIN_NFR + k] ^ q[1 + JOIN_NFR + k], PL, WL);
                    }
                    const bool r_alive = (int)(short)((CLS_TRACK_DROP ? WR : PR) & 0xFFFFu) >= -xdrop;  // (:374)
                    const int bestR = ((int)PR >> 16) - (int)(short)(PR & 0xFFFFu);
                    const bool l_alive = (int)(short)((CLS_TRACK_DROP ? WL : PL) & 0xFFFFu) >= -xdrop;  // (:523)
                    const int bestL = ((int)PL >> 16) - (int)(short)(PL & 0xFFFFu);
                    const bool fwd = valid && (r_alive || l_alive || bound_passes(a, bestR + bestL));
                    const uint32_t query_loc = q[0] + a.seed_size;  // :204
                    const unsigned long long fm = __ballot(fwd);
                    if (fm) {
                        L2Rec cr;
                        cr.ref_loc = ref_loc;
                        cr.query_loc = query_loc;
                        cr.hidx = tidx;
                        const uint32_t fl = (r_alive ? 1u : 0u) | (l_alive ? 2u : 0u);
                        const uint32_t Ps = (fl == 1u && a.l2_right_state) ? PR : PL;  // (1d: the walk level 2 resumes)
                        cr.state = (Ps & 0xFFFF0000u) | ((0u - Ps) & 0xFFFFu);
                        cr.meta = (uint32_t)(r_alive ? bestL : bestR) | ((fl ? fl : 3u) << 16);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        if (fwd) stage[n_stage + (int)rank] = cr;
                        n_stage += __popcll(fm);
                        __builtin_amdgcn_wave_barrier();
                        if (n_stage >= CTX_STAGE_FLUSH) {
                            const int k = n_stage < 64 ? n_stage : 64;
                            stage_flush(stage, k, my_list, my_count, a.l2_cap, lane);
                            const int rest = n_stage - k;
                            L2Rec tmp = cr;
                            if (lane < rest) tmp = stage[k + lane];
                            __builtin_amdgcn_wave_barrier();
                            if (lane < rest) stage[lane] = tmp;
                            __builtin_amdgcn_wave_barrier();
                            n_stage = rest;
                        }
                    }
                    if (a.audit_list) {  // (tests) the hits this level rejects
                        uint2 ar;
                        ar.x = ref_loc;
                        ar.y = query_loc;
                        wave_append(valid && !fwd, ar, a.audit_list, a.audit_count, a.audit_cap, lane, lane_lt);
                    }
                }
            }
        }
    }
    stage_flush(stage, n_stage, my_list, my_count, a.l2_cap, lane);
}

// =====================================================================================================================
// 2. exact extension of the candidates: one wave per hit, 512 bases per step
// =====================================================================================================================
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Exact two-sided extension of one anchor by a whole wave.  All results are wave-uniform.
template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__device__ __forceinline__ void wave_extend_exact(const ExtendArgs& a, const int* __restrict__ s_tab,
                                                  const uint8_t* __restrict__ R8b, const uint8_t* __restrict__ Qb, int lane,
                                                  uint32_t ref_loc, uint32_t query_loc, int& bestR, int& bposR, int& bestL,
                                                  int& boffL, unsigned long long& examined) {
    const int xdrop = a.xdrop;
    bestR = 0; bposR = -1; bestL = 0; boffL = 0;
    bool skip = false;
    if (a.rm) skip = !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);  // rm :239-244,:305-333
    if (skip) bposR = 0;  // extent stays 0 (:311)
    for (int side = skip ? 2 : 0; side < 2; side++) {
        const bool left = side == 1;
        const uint32_t lim = left ? min(ref_loc, query_loc)
                                  : ((ref_loc < a.ref_len && query_loc < a.query_len)
                                         ? min(a.ref_len - ref_loc, a.query_len - query_loc) : 0u);
        uint32_t k0 = left ? 1u : 0u;                            // :327 / :479
        int score_in = 0, best_in = 0, bpos_in = left ? 0 : -1;  // :308-310 / :465-467
        for (;;) {  // 512-base windows
            const uint32_t k = k0 + 8u * (uint32_t)lane;
            // in-range positions from k on, clamped to [0, 8]
            const int64_t rem64 = left ? (int64_t)lim - (int64_t)k + 1 : (int64_t)lim - (int64_t)k;
            const int remaining = rem64 > 8 ? 8 : (rem64 < 0 ? 0 : (int)rem64);
            uint64_t x = 0;
            if (remaining > 0) {
                const uint32_t roff = left ? ref_loc + BIAS - k - 7u : ref_loc + BIAS + k;
                const uint32_t qoff = left ? query_loc + BIAS - k - 7u : query_loc + BIAS + k;
                x = load8u(R8b + roff) | load8u(Qb + qoff);
                if (left) x = __builtin_bswap64(x);  // byte j <-> offset k+j on both sides
            }
            if (remaining < 8) x |= (remaining <= 0) ? TERM_ALL : (TERM_ALL << (8 * remaining));
            // ---- local prefix sums of the 8 scores, local maximum prefix (first position attaining it) ----
            // (sums are formed in uint32: lanes past a sequence edge accumulate terminators and may wrap; they lie
            //  after the first dropping lane and are discarded)
            uint32_t run = 0;
            int mx = INT32_MIN, amx = 0;
            {
                const uint32_t xlo = (uint32_t)x, xhi = (uint32_t)(x >> 32);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t w = j < 4 ? xlo : xhi;
                    run += (uint32_t)s_tab[(w >> (8 * (j & 3))) & 0xffu];
                    const bool up = (int)run > mx;
                    amx = up ? j : amx;
                    mx = up ? (int)run : mx;
                }
            }
            // ---- entry score of every lane: exclusive wave sum-scan (DPP) ----
            const uint32_t inc = wave_inclusive_sum(run);
            const int base = (int)((uint32_t)score_in + inc - run);
            // ---- entry best of every lane: exclusive max-scan, ties keep the EARLIER position ----
            int mv = (int)((uint32_t)base + (uint32_t)mx), mp = (int)(k + (uint32_t)amx);
            wave_inclusive_max(mv, mp);
            int ev = dpp_mov<0x138>(INT32_MIN, mv);  // wave_shr:1 -> best over earlier lanes
            int ep = dpp_mov<0x138>(0, mp);
            if (best_in >= ev) { ev = best_in; ep = bpos_in; }  // the carried-in best is the earliest of all
            // ---- exact replay of the lane's 8 bases ----
            int score = base, best = ev, bpos = ep;
            uint32_t ex_step = 0;
            chunk8_exact<COUNT_EXAMINED, XDROP_NONNEG>(s_tab, x, k, xdrop, score, best, bpos, ex_step);
            const unsigned long long dm = __ballot(score < (DEAD >> 1));
            if (dm) {
                const int f = __ffsll((long long)dm) - 1;  // first lane that dropped holds the final state
                best_in = __builtin_amdgcn_readlane(best, f);
                bpos_in = __builtin_amdgcn_readlane(bpos, f);
                if (COUNT_EXAMINED && lane <= f) examined += ex_step;  // lanes after f never happened
                break;
            }
            if (COUNT_EXAMINED) examined += ex_step;
            score_in = __builtin_amdgcn_readlane(score, 63);
            best_in = __builtin_amdgcn_readlane(best, 63);
            bpos_in = __builtin_amdgcn_readlane(bpos, 63);
            k0 += 512u;
        }
        if (!left) { bestR = best_in; bposR = bpos_in; }
        else { bestL = best_in; boffL = bpos_in; }
    }
}

// per-wave LDS stages of the exact kernels: survivors and entropy records, flushed 64 at a time
struct ExactStage {
    HspRec* st_out;
    EntRec* st_ent;
    int n_out, n_ent;
};

// classify + stage one finished extension (all arguments wave-uniform)
__device__ __forceinline__ void wave_finalize(const ExtendArgs& a, ExactStage& st, int lane, uint32_t ref_loc, uint32_t query_loc,
                                              uint32_t seg, int bestR, int bposR, int bestL, int boffL) {
    const int total = bestR + bestL, extent = bposR + boffL;  // :414-421, :563-574
    const int cls = classify(a, total);
    if (!cls) return;
    if (cls == 1) {
        if (lane == 0) st.st_out[st.n_out] = make_rec(a, ref_loc, query_loc, boffL, extent, total, seg);  // :638 with entropy 1
        st.n_out++;
        __builtin_amdgcn_wave_barrier();
        if (st.n_out == 64) { stage_flush(st.st_out, 64, a.out, a.out_count, a.out_cap, lane); st.n_out = 0; __builtin_amdgcn_wave_barrier(); }
    } else {
        if (lane == 0) {
            EntRec er;
            er.ref_loc = ref_loc; er.query_loc = query_loc; er.bposR = bposR; er.boffL = boffL; er.total = total; er.seg = seg;
            st.st_ent[st.n_ent] = er;
        }
        st.n_ent++;
        __builtin_amdgcn_wave_barrier();
        if (st.n_ent == 64) { stage_flush(st.st_ent, 64, a.ent_list, a.ent_count, a.ent_cap_recs, lane); st.n_ent = 0; __builtin_amdgcn_wave_barrier(); }
    }
}

#define EXACT_KERNEL_PROLOGUE()                                                                            \
    __shared__ int s_tab[128];                                                                             \
    __shared__ HspRec s_out[EXT_THREADS / 64][64];                                                         \
    __shared__ EntRec s_ent[EXT_THREADS / 64][64];                                                         \
    if (threadIdx.x < 128) s_tab[threadIdx.x] = threadIdx.x < 64 ? a.sub_mat[threadIdx.x] : NEG;           \
    __syncthreads();                                                                                       \
    ExactStage st = {s_out[threadIdx.x >> 6], s_ent[threadIdx.x >> 6], 0, 0};                             \
    const int lane = threadIdx.x & 63;                                                                     \
    const uint8_t* __restrict__ R8b = a.ref8 - BIAS;                                                       \
    const uint8_t* __restrict__ Qb = a.query - BIAS;                                                       \
    const uint32_t n_cand = min(*a.cand_count, a.cand_cap_recs);                                           \
    const uint32_t G = gridDim.x * (EXT_THREADS / 64);                                                     \
    SEG_TABLE()                                                                                            \
    unsigned long long examined = 0;

#define EXACT_KERNEL_EPILOGUE()                                                                            \
    stage_flush(st.st_out, st.n_out, a.out, a.out_count, a.out_cap, lane);                                 \
    stage_flush(st.st_ent, st.n_ent, a.ent_list, a.ent_count, a.ent_cap_recs, lane);                       \
    if (COUNT_EXAMINED) {                                                                                  \
        unsigned long long v = examined;                                                                   \
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);                               \
        if (lane == 0 && v) atomicAdd(a.examined, v);                                                      \
    }

// every candidate extended on its own (used when the chain shortcut below is off or the candidate list overflows it)
template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__global__ __launch_bounds__(EXT_THREADS) void extend_exact_kernel(ExtendArgs a) {
    EXACT_KERNEL_PROLOGUE()
    if (a.chain_cap && n_cand <= a.chain_cap) return;  // the chain kernels handle this batch
    for (uint32_t i = blockIdx.x * (EXT_THREADS / 64) + (threadIdx.x >> 6); i < n_cand; i += G) {
        const CandRec cr = a.cand_list[i];  // same address in every lane: one broadcast load
        const uint32_t ref_loc = (uint32_t)rfl((int)cr.ref_loc), query_loc = (uint32_t)rfl((int)cr.query_loc);
        const uint32_t hidx = (uint32_t)rfl((int)cr.hidx);
        int bestR, bposR, bestL, boffL;
        wave_extend_exact<COUNT_EXAMINED, XDROP_NONNEG>(a, s_tab, R8b, Qb, lane, ref_loc, query_loc, bestR, bposR, bestL, boffL, examined);
        wave_finalize(a, st, lane, ref_loc, query_loc, seg_of(a, s_seg, hidx, query_loc), bestR, bposR, bestL, boffL);
    }
    EXACT_KERNEL_EPILOGUE()
}

// =====================================================================================================================
// 2b. the CHAIN shortcut: most candidates are hits inside one and the same HSP and would all extend to the very same
//     record.  For the recurrence above (ties included) two facts hold for anchors a < b on one diagonal:
//       (R) if b <= E_R(a) (b not beyond a's best right end) then E_R(b) = E_R(a);
//       (L) if b's left walk reaches a STRICT new best at a position < a before it terminates, then S_L(b) = S_L(a).
//     (Sketch: the later-starting walk has point-wise smaller drops, so it cannot stop earlier; at the earlier walk's
//     stop both see the same running maximum, so it stops there too; the earliest arg-max is shared.)  Same interval
//     => same score, same entropy factor, same record.  So: candidates are sorted by (iteration, diagonal, position);
//     one lane per candidate tests (L) against its predecessor with a SHORT walk (gap + a few bases); candidates whose
//     test fails start a run.  One wave per run extends the run head exactly and skips every member whose anchor is
//     <= the head's right end -- the member's record would be an exact duplicate, which the dedup stage (adjacent-pair
//     containment, :47-52,:778) would drop anyway.  A member beyond the head's right end becomes the next head.
// =====================================================================================================================
constexpr uint32_t CHAIN_GAP_MAX = 256;   // predecessor further away than this: no test, the candidate starts a run
constexpr uint32_t CHAIN_WALK_EXTRA = 96; // bases walked past the predecessor's anchor looking for the new best

// Grouping without a general sort: candidates are dealt into hash buckets of (iteration, diagonal, 512-position window) --
// counting pass, scan (a workgroup per 4096 counters), scatter --, then workgroups take eight buckets at a time and rank-sort every
// bucket's entries in LDS by a 32-bit key (chain_key32: hash of the bucket's coordinates | position inside the window | index).  A
// bucket holds ~32 candidates by construction (its count is picked on the device from the candidate count) and up to 512 where
// one HSP fills a window, so the quadratic rank sort stays small; all buckets together are the candidate list with every
// (diagonal, window)'s candidates contiguous and ordered by position -- exactly what the link test needs.  No host involvement.
constexpr uint32_t CHAIN_BUCKETS_MIN = 16384, CHAIN_BUCKETS_MAX = 262144;  // what chain_buckets_of picks between (option chain_buckets forces any power of two >= 64)
constexpr uint32_t CHAIN_SORT_MAX = 4096;  // most entries a bucket may hold and still be sorted (option chain_group_max <= this); larger: left unsorted,
                                           // which only makes link tests fail, i.e. costs extensions, never correctness

// A bucket = hash of (iteration, diagonal, 512-position window): one diagonal can carry every candidate of a call (a
// collinear query), so the window keeps a group at <= 512 entries and spreads the counting atomics; chains simply
// restart at window borders (one extra extension per 512 bases of HSP).
constexpr uint32_t CHAIN_QSHIFT = 9;
__device__ __forceinline__ uint32_t chain_bucket_of(uint32_t buckets, uint32_t seg, const CandRec& c) {
    const uint32_t diag = c.ref_loc - c.query_loc;
    return (((diag * 2654435761u) ^ ((c.query_loc >> CHAIN_QSHIFT) * 0x85EBCA6Bu) ^ (seg * 0x9E3779B1u)) >> 14) & (buckets - 1u);
}
// Buckets of this launch: a power of two that leaves ~chain_bucket_target candidates per bucket, from the candidate count ON THE
// DEVICE (the host has no count before its sync, and the share of the hits that become candidates runs from 0.6 % on ordinary
// sequence to 3 % on sparse-hit calls and 8 % on repeat-rich ones: sized by hits -- rounds 2-4 -- a bucket held 50 or 300, and the
// rank sort is quadratic in that).  Every chain kernel of a launch derives the same number from the same count.
__device__ __forceinline__ uint32_t chain_buckets_of(const ExtendArgs& a, uint32_t n) {
    if (a.chain_buckets) return a.chain_buckets;  // (forced: option chain_buckets)
    const uint32_t want = n / a.chain_bucket_target;
    uint32_t b = CHAIN_BUCKETS_MIN;
    while (b < CHAIN_BUCKETS_MAX && b < want) b <<= 1;
    return b;
}
// The candidates the chain stages of this launch work on: all of the batch when they fit the chain buffers; a slice of the list when
// the host runs an oversized batch slice by slice; nothing (false) for an oversized batch that is not sliced yet -- the counts live
// on the device, so the first attempt finds out here and the host follows up after its sync.
__device__ __forceinline__ bool chain_range(const ExtendArgs& a, uint32_t& first, uint32_t& n) {
    const uint32_t total = min(*a.cand_count, a.cand_cap_recs);
    if (a.cand_sliced) {
        first = a.cand_first;
        n = total > first ? min(total - first, a.chain_cap) : 0u;
        return n > 0;
    }
    first = 0;
    n = total;
    return total <= a.chain_cap;
}

// One atomic per DISTINCT bucket of a wave, not per candidate: the candidates of one HSP follow each other in the list and share a
// bucket (same diagonal, same 512-base window), so 64 lanes used to queue up 20-60 deep on one L2 address -- with 3 % of the hits
// candidates (sparse-hit calls) counting and scattering took 0.32 ms per call.  The lanes agree on the buckets among themselves:
// the first active lane's bucket is broadcast, its holders are counted by a ballot and retired, until no lane is left.
// Returns this lane's rank among the wave's lanes with the same bucket and, in `base`, the first slot of the wave's lanes: what the
// leader's atomic returned (counting up), or that minus the wave's k lanes (DOWN: the counter is taken down by k).
template <bool DOWN>
__device__ __forceinline__ uint32_t wave_bucket_add(uint32_t* __restrict__ counters, bool active, uint32_t b, uint32_t& base) {
    const int lane = threadIdx.x & 63;
    uint32_t rank = 0, k = 0;
    int my_leader = lane;
    // 1. who shares a bucket with whom: registers only.  (First form: the leader's atomic sat INSIDE this loop and every round
    //    waited for its return -- fine for a wave of one HSP's candidates, two or three rounds; on ordinary input a wave holds ~64
    //    different buckets and paid 64 atomic round trips one after the other.)
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
        const unsigned long long same = __ballot(active && b == b0) & todo;
        if ((same >> lane) & 1ull) {
            my_leader = leader;
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            k = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
    // 2. all leaders' atomics in one instruction, the answers handed to their lanes by one shuffle
    uint32_t got = 0;
    if (active && lane == my_leader) got = DOWN ? atomicSub(&counters[b], k) - k : atomicAdd(&counters[b], k);
    base = (uint32_t)__shfl((int)got, my_leader, 64);
    return rank;
}

__global__ __launch_bounds__(256) void chain_count_kernel(ExtendArgs a) {
    uint32_t first, n;
    if (!chain_range(a, first, n)) return;
    SEG_TABLE()
    const uint32_t B = chain_buckets_of(a, n);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;  // wave-uniform trip count (ballots inside)
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t b = 0;
        if (i < n) {
            const CandRec c = a.cand_list[first + i];
            b = chain_bucket_of(B, seg_of(a, s_seg, c.hidx, c.query_loc), c);
        }
        uint32_t base;
        wave_bucket_add<false>(a.chain_bucket_cnt, i < n, b, base);
    }
}

// Exclusive scan of the bucket counters into chain_bucket_start[0..buckets].  One workgroup per tile of 4096 counters (every thread
// takes four consecutive ones as ONE 16-byte load); a workgroup first adds up everything in front of its tile by itself -- at most
// 1 MB out of the L2, no second kernel and no waiting for another workgroup -- then scans the tile (wave scan + the totals of the
// waves before).  The counters stay as they are: the scatter counts them DOWN to zero.  (One workgroup walking all tiles, rounds
// 3-4: 7 us for 16384 buckets, 37 us for 131072.)
constexpr uint32_t CHAIN_SCAN_THREADS = 1024, CHAIN_SCAN_TILE = CHAIN_SCAN_THREADS * 4;
__global__ __launch_bounds__(CHAIN_SCAN_THREADS) void chain_scan_kernel(ExtendArgs a) {
    __shared__ uint32_t s_tile[CHAIN_SCAN_THREADS / 64], s_front[CHAIN_SCAN_THREADS / 64];
    uint32_t first, n;
    if (!chain_range(a, first, n)) return;
    const uint32_t B = chain_buckets_of(a, n);
    const uint32_t t0 = blockIdx.x * CHAIN_SCAN_TILE;  // (the grid is sized for CHAIN_BUCKETS_MAX)
    if (t0 >= B) return;
    const uint4* cnt4 = reinterpret_cast<const uint4*>(a.chain_bucket_cnt);
    uint4* start4 = reinterpret_cast<uint4*>(a.chain_bucket_start);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t front = 0;  // this thread's share of the counters in front of the tile
    for (uint32_t q = threadIdx.x; q < (t0 >> 2); q += CHAIN_SCAN_THREADS) {
        const uint4 f = cnt4[q];
        front += f.x + f.y + f.z + f.w;
    }
    const uint4 v = cnt4[(t0 >> 2) + threadIdx.x];
    const uint32_t sum = v.x + v.y + v.z + v.w;
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) front += __shfl_xor(front, off, 64);
    if (lane == 63) {
        s_tile[wave] = inc;
        s_front[wave] = front;
    }
    __syncthreads();
    uint32_t base = inc - sum;
    for (int w = 0; w < (int)(CHAIN_SCAN_THREADS / 64); w++) base += s_front[w] + (w < wave ? s_tile[w] : 0u);
    start4[(t0 >> 2) + threadIdx.x] = make_uint4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
    if (t0 + CHAIN_SCAN_TILE >= B && threadIdx.x == CHAIN_SCAN_THREADS - 1) a.chain_bucket_start[B] = base + sum;  // (= n)
}

__global__ __launch_bounds__(256) void chain_scatter_kernel(ExtendArgs a) {
    uint32_t first, n;
    if (!chain_range(a, first, n)) return;
    SEG_TABLE()
    const uint32_t B = chain_buckets_of(a, n);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t b = 0;
        CandRec c = {0u, 0u, 0u};
        if (i < n) {
            c = a.cand_list[first + i];
            b = chain_bucket_of(B, seg_of(a, s_seg, c.hidx, c.query_loc), c);
        }
        // the counters of the counting pass are the cursors, counted down to zero: a wave takes the k slots below what it finds
        uint32_t base;
        const uint32_t rank = wave_bucket_add<true>(a.chain_bucket_cnt, i < n, b, base);  // (one cursor atomic per distinct bucket of the wave)
        if (i < n) a.chain_tmp[a.chain_bucket_start[b] + base + rank] = c;
    }
}

// CHAIN_SORT_GROUP consecutive buckets per workgroup step (a bucket holds a few dozen candidates: one workgroup per bucket was
// thousands of tiny workgroups whose dispatch cost more than their work): rank sort in LDS, every entry ranked inside its own
// bucket; then, in the SAME workgroup, the link test of every entry against its predecessor in the bucket (the separate
// chain_link_kernel of rounds 1-3: one launch and one pass over the sorted list less).  The records are read once from the
// scatter's output into LDS, the sorted order is a permutation in LDS, and the records leave in order, coalesced.
constexpr uint32_t CHAIN_SORT_GROUP = 8;
// Entries the workgroup's LDS holds at a time = ExtendArgs.chain_group_max (dynamic LDS, 18 bytes per entry; <= CHAIN_SORT_MAX): a
// whole group usually.  It sets the kernel's occupancy -- round 4 started with 4096 entries = 45 KB = 3 workgroups of 4 waves per CU,
// and the link test is a chain of dependent random reads that wants every wave slot (4096 -> 1024: -25 % on the kernel).

// does candidate c start a run?  (test (L) of DESIGN.md 4.5' against its predecessor pc on the same diagonal, bounded walk)
template <bool XDROP_NONNEG>
__device__ __forceinline__ bool chain_is_run_head(const ExtendArgs& a, const uint64_t* __restrict__ s_seg, const int* __restrict__ s_tab, const uint8_t* __restrict__ R8b,
                                                  const uint8_t* __restrict__ Qb, const CandRec& c, const CandRec& pc) {
    // same iteration, same diagonal, predecessor strictly before this anchor
    if (!((c.ref_loc - c.query_loc) == (pc.ref_loc - pc.query_loc) && c.query_loc > pc.query_loc && seg_of(a, s_seg, c.hidx, c.query_loc) == seg_of(a, s_seg, pc.hidx, pc.query_loc)))
        return true;
    const uint32_t g = c.query_loc - pc.query_loc;  // anchor gap (> 0)
    if (g > CHAIN_GAP_MAX) return true;
    const uint32_t lim = min(c.ref_loc, c.query_loc);
    int score = 0, best = 0, bpos = 0;
    uint32_t ex = 0;
    for (uint32_t k = 1; k <= g + CHAIN_WALK_EXTRA; k += 8) {
        const int64_t rem64 = (int64_t)lim - (int64_t)k + 1;
        const int remaining = rem64 > 8 ? 8 : (rem64 < 0 ? 0 : (int)rem64);
        uint64_t x = 0;
        if (remaining > 0)
            x = __builtin_bswap64(load8u(R8b + (c.ref_loc + BIAS - k - 7u)) | load8u(Qb + (c.query_loc + BIAS - k - 7u)));
        if (remaining < 8) x |= (remaining <= 0) ? TERM_ALL : (TERM_ALL << (8 * remaining));
        chunk8_exact<false, XDROP_NONNEG>(s_tab, x, k, a.xdrop, score, best, bpos, ex);
        if ((uint32_t)bpos > g) return false;  // strict new best at a position < predecessor's anchor: (L)
        if (score < (DEAD >> 1)) break;        // walk ended first: extend this one on its own
    }
    return true;
}

// Sort key of an entry, 32 bits, unique inside a piece: 11 bits of a hash of its diagonal and 512-position window (and iteration) |
// its position inside the window | its index in the piece.  Entries of one diagonal AND window always share a bucket
// (chain_bucket_of) and these hash bits, so the low position bits order them; two (diagonal, window) pairs of a bucket that share the
// 11 bits (1 in 2048) interleave, which only makes link tests fail.  The window must be in the hash: a collinear query puts most
// candidates on ONE diagonal, a bucket then holds several windows of it, and ordered by (diagonal, position mod 512) they interleave
// completely (first cut of this key: 8 x the extensions).  The multipliers are not the bucket hash's: a bucket's entries agree in
// bits 14.. of THAT product.  (Rounds 1-4 ranked 64-bit keys {diagonal, position}: twice the LDS bytes, 64-bit compares, one key
// per LDS read.)
__device__ __forceinline__ uint32_t chain_key32(const ExtendArgs& a, const uint64_t* __restrict__ s_seg, const CandRec& c, uint32_t idx) {
    uint32_t h = ((c.ref_loc - c.query_loc) * 0xC2B2AE35u) ^ ((c.query_loc >> CHAIN_QSHIFT) * 0x27D4EB2Fu);
    if (a.chain_q_bits != 32u) h ^= (seg_of(a, s_seg, c.hidx, c.query_loc) - a.seg_base) * 0x165667B1u;  // (several iterations per batch: general path)
    return (h & 0xFFE00000u) | ((c.query_loc & ((1u << CHAIN_QSHIFT) - 1u)) << 12) | idx;
}
static_assert(CHAIN_QSHIFT == 9 && CHAIN_SORT_MAX <= 4096, "chain_key32: 11 + 9 + 12 bits");

template <bool XDROP_NONNEG>
__global__ __launch_bounds__(512) void chain_sort_link_kernel(ExtendArgs a) {
    // LDS of a piece (chain_group_max = GM entries, 18 bytes each): keys -- every bucket's keys start at a multiple of four and are
    // padded with 0xFFFFFFFF, so the rank loop reads four per instruction --, the records themselves (read once from the scatter's
    // output, then ranked, linked and written in order from here), the permutation sorted position -> entry
    extern __shared__ __attribute__((aligned(16))) uint32_t s_chain_dyn[];
    const uint32_t CHAIN_GROUP_MAX = a.chain_group_max;
    uint32_t* s_key = s_chain_dyn;                                                          // [GM + 4 * CHAIN_SORT_GROUP]
    CandRec* s_rec = reinterpret_cast<CandRec*>(s_chain_dyn + CHAIN_GROUP_MAX + 4 * CHAIN_SORT_GROUP);  // [GM]
    uint16_t* s_perm = reinterpret_cast<uint16_t*>(s_rec + CHAIN_GROUP_MAX);               // [GM]
    __shared__ uint32_t s_b[CHAIN_SORT_GROUP + 1];
    __shared__ int s_tab[128];
    uint32_t first, n;
    if (!chain_range(a, first, n)) return;
    SEG_TABLE()
    for (uint32_t t = threadIdx.x; t < 128u; t += blockDim.x) s_tab[t] = t < 64u ? a.sub_mat[t] : NEG;  // (any workgroup size: option chain_sort_threads goes down to 64)
    const uint32_t groups = chain_buckets_of(a, n) / CHAIN_SORT_GROUP;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint8_t* __restrict__ R8b = a.ref8 - BIAS;
    const uint8_t* __restrict__ Qb = a.query - BIAS;
    // The group is handled in pieces of consecutive buckets that fit the LDS together (usually the whole group at once; a group of
    // crowded buckets -- sparse-hit calls carry 3 % candidates -- goes bucket by bucket).  Only a single bucket above the LDS
    // capacity is left unsorted, every entry of it a run head (costs extensions, never results).
    for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {  // (the grid is the host's, the group count the device's)
    __syncthreads();  // (s_b of the group before is done with; first round: s_tab is written)
    if (threadIdx.x <= CHAIN_SORT_GROUP) s_b[threadIdx.x] = a.chain_bucket_start[grp * CHAIN_SORT_GROUP + threadIdx.x];
    __syncthreads();
    if (s_b[CHAIN_SORT_GROUP] == s_b[0]) continue;
    for (uint32_t ja = 0; ja < CHAIN_SORT_GROUP;) {
        uint32_t jb = ja + 1;
        while (jb < CHAIN_SORT_GROUP && s_b[jb + 1] - s_b[ja] <= CHAIN_GROUP_MAX) jb++;
        const uint32_t g0 = s_b[ja], m_all = s_b[jb] - g0;
        const bool big = m_all > CHAIN_GROUP_MAX;  // (then jb == ja + 1: one bucket that does not fit)
        if (m_all == 0) { ja = jb; continue; }
        // entry i of the piece -> its bucket [lo, hi) (piece-relative) and where the bucket's padded keys start
        auto bucket_of = [&](uint32_t i, uint32_t& lo, uint32_t& hi, uint32_t& kb) {
            uint32_t j = ja;
            kb = 0;
            while (g0 + i >= s_b[j + 1]) {
                kb += (s_b[j + 1] - s_b[j] + 3u) & ~3u;
                j++;
            }
            lo = s_b[j] - g0;
            hi = s_b[j + 1] - g0;
        };
        if (!big) {
            for (uint32_t i = threadIdx.x; i < m_all; i += blockDim.x) {
                const CandRec c = a.chain_tmp[g0 + i];
                s_rec[i] = c;
                uint32_t lo, hi, kb;
                bucket_of(i, lo, hi, kb);
                s_key[kb + (i - lo)] = chain_key32(a, s_seg, c, i);
                if (i + 1 == hi)
                    for (uint32_t t = kb + (hi - lo); t < kb + ((hi - lo + 3u) & ~3u); t++) s_key[t] = 0xFFFFFFFFu;
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < m_all; i += blockDim.x) {
                uint32_t lo, hi, kb;
                bucket_of(i, lo, hi, kb);
                const uint32_t k = s_key[kb + (i - lo)];
                uint32_t rank = 0;  // entries of the bucket below this one (keys are unique: the index is part of them)
                const uint32_t ke = kb + ((hi - lo + 3u) & ~3u);
#pragma unroll 2
                for (uint32_t t = kb; t < ke; t += 4) {
                    const uint4 kk = *reinterpret_cast<const uint4*>(s_key + t);
                    rank += (kk.x < k ? 1u : 0u) + (kk.y < k ? 1u : 0u) + (kk.z < k ? 1u : 0u) + (kk.w < k ? 1u : 0u);
                }
                s_perm[lo + rank] = (uint16_t)i;
            }
            __syncthreads();
        } else if (threadIdx.x == 0 && a.chain_big) {
            atomicAdd(a.chain_big, 1u);
        }
        const uint32_t rounds = (m_all + blockDim.x - 1) / blockDim.x;  // wave-uniform trip count (ballot below)
        for (uint32_t r = 0; r < rounds; r++) {
            const uint32_t p = r * blockDim.x + threadIdx.x;  // sorted position inside the piece
            bool head = false;
            if (p < m_all) {
                const CandRec c = big ? a.chain_tmp[g0 + p] : s_rec[s_perm[p]];
                a.chain_sorted[g0 + p] = c;
                head = true;
                if (!big) {
                    uint32_t lo, hi, kb;
                    bucket_of(p, lo, hi, kb);
                    if (p > lo && !a.chain_no_link) head = chain_is_run_head<XDROP_NONNEG>(a, s_seg, s_tab, R8b, Qb, c, s_rec[s_perm[p - 1]]);
                }
                a.chain_is_head[g0 + p] = head ? 1u : 0u;
            }
            // run heads -> head list (order irrelevant)
            const unsigned long long m = __ballot(head);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t wbase = 0;
                if (lane == leader) wbase = atomicAdd(a.chain_head_count, (uint32_t)__popcll(m));
                wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, leader);
                if (head) a.chain_heads[wbase + (uint32_t)__popcll(m & lane_lt)] = g0 + p;
            }
        }
        __syncthreads();  // (the next piece reuses the LDS)
        ja = jb;
    }
    }
}

// one wave per run: extend the head, skip members covered by (R), promote the first member beyond the right end
template <bool XDROP_NONNEG>
__global__ __launch_bounds__(EXT_THREADS) void extend_exact_chain_kernel(ExtendArgs a) {
    constexpr bool COUNT_EXAMINED = false;
    EXACT_KERNEL_PROLOGUE()
    uint32_t first_cand, n_chain;
    if (!chain_range(a, first_cand, n_chain)) return;  // oversized batch: the host reruns the chain stages slice by slice
    const uint32_t n_heads = *a.chain_head_count;
    for (uint32_t j = blockIdx.x * (EXT_THREADS / 64) + (threadIdx.x >> 6); j < n_heads; j += G) {
        uint32_t cur = (uint32_t)rfl((int)a.chain_heads[j]);
        for (;;) {
            const CandRec cr = a.chain_sorted[cur];
            const uint32_t ref_loc = (uint32_t)rfl((int)cr.ref_loc), query_loc = (uint32_t)rfl((int)cr.query_loc);
            const uint32_t hidx = (uint32_t)rfl((int)cr.hidx);
            int bestR, bposR, bestL, boffL;
            wave_extend_exact<false, XDROP_NONNEG>(a, s_tab, R8b, Qb, lane, ref_loc, query_loc, bestR, bposR, bestL, boffL, examined);
            wave_finalize(a, st, lane, ref_loc, query_loc, seg_of(a, s_seg, hidx, query_loc), bestR, bposR, bestL, boffL);
            // members of the run follow in the sorted list until the next run head
            const int64_t right_end = (int64_t)ref_loc + (int64_t)bposR;  // E_R(head) as a target position
            uint32_t nxt = 0xFFFFFFFFu;
            for (uint32_t m0 = cur + 1; m0 < n_chain; m0 += 64) {
                const uint32_t m = m0 + (uint32_t)lane;
                bool stop_run = false, beyond = false;
                if (m < n_chain) {
                    stop_run = a.chain_is_head[m] != 0u;  // (also set wherever the iteration or the diagonal changes)
                    beyond = (int64_t)a.chain_sorted[m].ref_loc > right_end;  // (R) does not cover it
                } else {
                    stop_run = true;
                }
                const unsigned long long sm = __ballot(stop_run), bm = __ballot(beyond);
                const int fs = sm ? __ffsll((long long)sm) - 1 : 64, fb = bm ? __ffsll((long long)bm) - 1 : 64;
                if (fb < fs) { nxt = m0 + (uint32_t)fb; break; }  // a member beyond the right end: it becomes the next head
                if (fs < 64) break;                                // run ended
            }
            if (nxt == 0xFFFFFFFFu) break;
            cur = nxt;
        }
    }
    EXACT_KERNEL_EPILOGUE()
}

// =====================================================================================================================
// 3. entropy kernel: lane per hit (:608-647)
// =====================================================================================================================
__global__ __launch_bounds__(EXT_THREADS) void extend_entropy_kernel(ExtendArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint8_t* __restrict__ R8 = a.ref8;
    const uint8_t* __restrict__ Q = a.query;
    const uint32_t n = min(*a.ent_count, a.ent_cap_recs);
    const uint32_t stride = gridDim.x * EXT_THREADS;
    const uint32_t rounds = (n + stride - 1) / stride;  // wave-uniform trip count (wave_append needs whole waves)
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = r * stride + blockIdx.x * EXT_THREADS + threadIdx.x;
        bool pass = false;
        HspRec rec;
        rec.ref_start = rec.query_start = rec.len = rec.seg = 0; rec.score = 0;
        if (i < n) {
            const EntRec e = a.ent_list[i];
            const int extent = e.bposR + e.boffL;
            // matches r==q<4 over the final interval [loc-boffL, loc+bposR]: equals the reference kernel's running
            // count[] (:444-451,:595-602); r>=4 would be its out-of-bounds counter write (hazard H1), not counted
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int d = -e.boffL; d <= e.bposR; d++) {
                const uint32_t rr = (uint32_t)R8[(int64_t)e.ref_loc + d] >> 3, qq = Q[(int64_t)e.query_loc + d];
                if (rr == qq) { c0 += (rr == 0); c1 += (rr == 1); c2 += (rr == 2); c3 += (rr == 3); }
            }
            const short s0 = (short)c0, s1 = (short)c1, s2 = (short)c2, s3 = (short)c3;  // `short` counters :263
            double entropy = 1.0;                                                       // :307
            if ((s0 + s1 + s2 + s3) >= 20) {                                            // :617
                const double len1 = (double)(extent + 1);
                double h = 0.0;  // :620-622, same evaluation order
                h += ((double)s0) / len1 * ((s0 != 0) ? log(((double)s0) / len1) : 0.0);
                h += ((double)s1) / len1 * ((s1 != 0) ? log(((double)s1) / len1) : 0.0);
                h += ((double)s2) / len1 * ((s2 != 0) ? log(((double)s2) / len1) : 0.0);
                h += ((double)s3) / len1 * ((s3 != 0) ? log(((double)s3) / len1) : 0.0);
                // :623 divides by log(4.0f): the FLOAT overload, i.e. (double)0x3FB17218 (hazard H2); option log4_double: log(4.0)
                entropy = -h / (a.log4_double ? 1.3862943611198906 : (double)1.38629436492919921875f);
                // (tests, hazard H13: device log() against the host's -- how far is any verdict or score from flipping?)
                for (int u = 0; u < a.entropy_ulps; u++) entropy = nextafter(entropy, 2.0);
                for (int u = 0; u > a.entropy_ulps; u--) entropy = nextafter(entropy, -1.0);
            }
            pass = f64_to_i32(((double)(float)e.total) * entropy) >= a.hspthresh;  // :633
            int sc = 0;
            if (entropy > 0) sc = f64_to_i32((double)e.total * entropy);           // :637-638
            rec = make_rec(a, e.ref_loc, e.query_loc, e.boffL, extent, sc, e.seg);
        }
        wave_append(pass, rec, a.out, a.out_count, a.out_cap, lane, lane_lt);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
void launch_extend_filter(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    const uint64_t num_buf = (a.num_hits + 63) / 64;
    // filter waves: ~4 per SIMD saturate instruction issue (more only add contention); at least `bufs_per_wave` buffers
    // per wave so the drain phase of a wave (bounded by long_cap) is amortised
    uint64_t waves = num_buf / (uint64_t)(a.bufs_per_wave > 0 ? a.bufs_per_wave : 8);
    const uint64_t max_waves = a.max_waves ? a.max_waves : 4096u;
    if (waves > max_waves) waves = max_waves;
    if (waves < 4) waves = 4;
    const uint32_t blocks = (uint32_t)((waves + 3) / 4);
    if (!a.examined && a.fast_filter == 3) {
        const uint32_t pblocks = (uint32_t)((waves + PK_THREADS / 64 - 1) / (PK_THREADS / 64));
        if (a.src_cand) {  // second level behind the context filter: the count lives on the device, the grid is fixed
            hipLaunchKernelGGL(extend_filter_packed_kernel<SRC_CAND>, dim3(a.l2_blocks ? a.l2_blocks : 256), dim3(PK_THREADS), 0, s, a);
        } else if (a.td) hipLaunchKernelGGL(extend_filter_packed_kernel<SRC_TD>, dim3(pblocks), dim3(PK_THREADS), 0, s, a);
        else hipLaunchKernelGGL(extend_filter_packed_kernel<SRC_HITS>, dim3(pblocks), dim3(PK_THREADS), 0, s, a);
        return;
    }
    if (a.examined) hipLaunchKernelGGL((extend_filter_kernel<true, false>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
    else if (a.fast_filter) hipLaunchKernelGGL((extend_filter_kernel<false, true>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((extend_filter_kernel<false, false>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
}

// class filter (1d): table-direct calls whose neighbourhood table carries 32-byte context records; fills a.l2_list
void launch_extend_filter_cls(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    uint64_t waves = (a.num_hits + TD_CHUNK_HITS - 1) / TD_CHUNK_HITS;  // at most one wave per chunk
    if (a.ctx_waves && waves > a.ctx_waves) waves = a.ctx_waves;
    uint32_t threads = a.ctx_threads ? a.ctx_threads : (uint32_t)CTX_THREADS;
    threads = std::min<uint32_t>(CTX_THREADS_MAX, std::max<uint32_t>(64, threads & ~63u));
    const uint32_t wpb = threads / 64;
    const uint32_t blocks = (uint32_t)((waves + wpb - 1) / wpb);
    const size_t lds = wpb * CTX_STAGE_CAP * sizeof(L2Rec);
    // the query windows out of ONE copy (see the kernel): -24 % on the kernel with ~5 hits per position, no loss with ~78 (the default
    // workload measures the same either way, same box); option cls_one_copy = 2 keeps the sixteen shifted copies (A/B)
    const bool one_copy = a.cls_one_copy != 2;
    if (one_copy) hipLaunchKernelGGL(extend_filter_cls_kernel<true>, dim3(blocks), dim3(threads), lds, s, a);
    else hipLaunchKernelGGL(extend_filter_cls_kernel<false>, dim3(blocks), dim3(threads), lds, s, a);
}

// class filter of a key-ordered call (1e): persistent waves that claim their work on the device
void launch_join_filter(const ExtendArgs& a, const JoinArgs& j, hipStream_t s) {
    if (a.num_hits == 0) return;
    const uint32_t wpb = JOIN_THREADS / 64;
    const size_t lds = wpb * CTX_STAGE_CAP * sizeof(L2Rec);
    hipLaunchKernelGGL(join_filter_kernel, dim3(512), dim3(JOIN_THREADS), lds, s, a, j);
}

void launch_chain_group(const ExtendArgs& a, hipStream_t s) {  // chain_bucket_cnt must be zero on entry
    if (a.num_hits == 0 || !a.chain_cap) return;
    hipLaunchKernelGGL(chain_count_kernel, dim3(1024), dim3(256), 0, s, a);
    hipLaunchKernelGGL(chain_scan_kernel, dim3(CHAIN_BUCKETS_MAX / CHAIN_SCAN_TILE), dim3(CHAIN_SCAN_THREADS), 0, s, a);
    hipLaunchKernelGGL(chain_scatter_kernel, dim3(1024), dim3(256), 0, s, a);
    // (chain_group_max = 4096 asks for 72 KB of dynamic LDS next to ~4.6 KB static: above the 64 KB a launch gets without being asked)
    const size_t sort_lds = (size_t)a.chain_group_max * (sizeof(uint32_t) + sizeof(CandRec) + sizeof(uint16_t)) + 4 * CHAIN_SORT_GROUP * sizeof(uint32_t);
    static std::atomic<size_t> sort_lds_set[64];  // per device ordinal (the attribute is per device); slot threads of a device race here:
    int dev = 0;                                  // a stale read only repeats an idempotent hipFuncSetAttribute
    (void)hipGetDevice(&dev);
    if (sort_lds > 48 * 1024 && dev >= 0 && dev < 64 && sort_lds_set[dev].load(std::memory_order_acquire) < sort_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_sort_link_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds);
        size_t seen = sort_lds_set[dev].load(std::memory_order_relaxed);
        while (seen < sort_lds && !sort_lds_set[dev].compare_exchange_weak(seen, sort_lds, std::memory_order_release)) {}
    }
    hipLaunchKernelGGL((chain_sort_link_kernel<true>), dim3(a.chain_sort_blocks ? a.chain_sort_blocks : 4096), dim3(a.chain_sort_threads ? a.chain_sort_threads : 256),
                       sort_lds, s, a);
}
uint32_t chain_num_buckets() { return CHAIN_BUCKETS_MAX; }  // (what the bucket arrays are sized for)
void launch_extend_exact_chain(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0 || !a.chain_cap) return;
    hipLaunchKernelGGL((extend_exact_chain_kernel<true>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
}

void launch_extend_exact(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    const bool nonneg = a.xdrop >= 0;
    if (a.examined) {
        if (nonneg) hipLaunchKernelGGL((extend_exact_kernel<true, true>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
        else hipLaunchKernelGGL((extend_exact_kernel<true, false>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
    } else {
        if (nonneg) hipLaunchKernelGGL((extend_exact_kernel<false, true>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
        else hipLaunchKernelGGL((extend_exact_kernel<false, false>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
    }
}

void launch_extend_entropy(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    hipLaunchKernelGGL(extend_entropy_kernel, dim3(a.ent_blocks), dim3(EXT_THREADS), 0, s, a);
}

}  // namespace sa
