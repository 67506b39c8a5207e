// extend.hip -- ungapped X-drop extension + entropy filter + wavefront-ballot compaction of the survivors.
// Replaces find_hsps (src/seed_filter.cu:232-652), the done-flag scan (:769) and compress_output (:654-680).
//
// Scalar recurrence per side (k = 0,1,.. right of the anchor; k = 1,2,.. left of it), which is what the reference's
// 32-lane tile loop computes (:326-453 right, :478-604 left) independently of its tile width:
//     score += M[r][q];  if (max(best,score) - score > xdrop) stop;  if (score > best) { best = score; bestpos = k; }
// stop also at the first position outside either sequence.
//
// Design (wave64, CDNA4) -- NOT the reference's shape (one 32-lane warp per hit, four shuffle scans + ~10 syncs
// per 32 bases, although a random hit dies after ~20-60 bases).  Three kernels:
//
//  1. extend_main_kernel: ONE LANE OWNS ONE HIT, lanes are PERSISTENT.  Each lane runs a small state machine
//     (right side -> left side -> finished); every trip of the wave loop advances every live lane by 8 bases.
//     Finished lanes are finalised in batches and REFILLED from the wave's queue (a register-held, double-buffered
//     64-hit buffer; buffers are dealt round-robin to the waves of the grid -- no atomics on the fetch side).
//     A side that is still alive after `long_cap` bases is almost surely real homology and may run for kilobases:
//     the lane PARKS the hit with its state in the long list and takes new work, so no wave ever waits on one lane.
//  2. extend_long_kernel: ONE WAVE OWNS ONE PARKED HIT and advances it 512 bases per step with an exact segmented
//     scan: lane l scores bases [8l, 8l+8) of the window, a wave sum-scan gives every lane its entry score, a wave
//     max-scan (ties -> earlier position) its entry best, then each lane REPLAYS its 8 bases with the exact entry
//     state; the first lane that drops holds the final (best, bestpos).  ~0.25 wave-instructions per base.
//  3. extend_entropy_kernel: the few hits with hspthresh <= score <= 3*hspthresh (:608) get their fp64 entropy
//     factor here, one lane per candidate, so the hot kernels carry no fp64 code or registers.
//
// Shared machinery:
//   * The target is kept in HBM a second time "row coded" (r<<3, one byte per base) so that `rw | qw` of two
//     8-byte windows IS the 8 matrix indices r*8+q; one unaligned global_load_dwordx2 per sequence per 8 bases.
//     The left side byte-swaps its window so both directions share the same straight-line code.
//   * The 8x8 matrix sits in LDS as one 128-entry table: entries 64..127 hold a large negative "terminator" that
//     out-of-range positions are mapped to (bit 6 OR-ed into their index byte), which folds the sequence-edge
//     test (:332,:482) into the X-drop test.  ACGTxACGT pairs occupy 16 distinct banks: conflict-free.
//     Address = one SDWA byte-select shift; per base: 1 ds_read_b32 + 7 VALU.
//   * Once a side has dropped its running score is pinned to DEAD, which makes every later base of the chunk a
//     no-op without per-base predication; "side finished" is read off the score after the chunk.
//   * Integer DP only (no MFMA).  Survivors are appended with one atomicAdd per wave per batch (ballot + popcount
//     prefix).  Append order is arbitrary; the dedup stage sorts on a total order, so the output is deterministic.
#include "kernels.h"
#include "kmer_dev.h"  // load8u

namespace sa {

constexpr int EXT_THREADS = 256;
constexpr int NEG = -(1 << 28);   // score of a terminator pair: forces the drop test for any |xdrop| < 2^27
constexpr int DEAD = -(1 << 29);  // sticky running score of a side that has dropped
constexpr uint64_t TERM_ALL = 0x4040404040404040ull;
constexpr uint32_t BIAS = SEQ_PAD;  // offsets are kept unsigned: base pointers point at the start of the front pad

__device__ __forceinline__ int f64_to_i32(double x) { return (int)x; }  // v_cvt_i32_f64: NaN -> 0, saturating (as on CUDA)

enum : int { PH_RIGHT = 0, PH_LEFT = 1, PH_FIN = 2, PH_IDLE = 3 };

// One 8-base chunk of the recurrence on the packed index word x (byte j = matrix index of offset k+j).
template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__device__ __forceinline__ void chunk8(const int* __restrict__ s_tab, uint64_t x, uint32_t k, int xdrop, int& score,
                                       int& best, int& bpos, unsigned long long& examined) {
    const uint32_t xlo = (uint32_t)x, xhi = (uint32_t)(x >> 32);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = j < 4 ? xlo : xhi;
        const uint32_t idx = (w >> (8 * (j & 3))) & 0xffu;  // SDWA byte select
        if (COUNT_EXAMINED) examined += (score > (DEAD >> 1) && idx < 64u) ? 1ull : 0ull;
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

// The 8 matrix indices of offsets k..k+7 of one side.  `remaining` = in-range positions from k on (<= 0: none).
__device__ __forceinline__ uint64_t fetch_indices(const uint8_t* __restrict__ R8b, const uint8_t* __restrict__ Qb,
                                                  uint32_t ref_loc, uint32_t query_loc, bool left, uint32_t k, int remaining) {
    uint64_t x = 0;
    if (remaining > 0) {
        const uint32_t roff = left ? ref_loc + BIAS - k - 7u : ref_loc + BIAS + k;
        const uint32_t qoff = left ? query_loc + BIAS - k - 7u : query_loc + BIAS + k;
        x = load8u(R8b + roff) | load8u(Qb + qoff);
        if (left) x = __builtin_bswap64(x);  // byte j <-> offset k+j on both sides
    }
    if (remaining < 8) x |= (remaining <= 0) ? TERM_ALL : (TERM_ALL << (8 * remaining));
    return x;
}

__device__ __forceinline__ uint32_t seg_of(const ExtendArgs& a, uint64_t local_idx) {
    uint32_t seg = 0;
    const uint64_t g = a.hit_base + local_idx;
#pragma unroll
    for (int s = 0; s < MAX_SEGS - 1; s++)
        if (s < a.num_segs - 1 && g >= a.seg_end[s]) seg = s + 1;
    return a.seg_base + seg;
}

// What to do with a finished hit: 0 = reject, 1 = survivor with entropy 1, 2 = entropy candidate (:608,:633)
__device__ __forceinline__ int classify(const ExtendArgs& a, int total) {
    if (!a.noentropy && total >= a.hspthresh && total <= 3 * a.hspthresh) return 2;
    // entropy stays 1.0: (int)((float)total * 1.0) >= hspthresh
    return (f64_to_i32((double)(float)total) >= a.hspthresh) ? 1 : 0;
}

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

// wave-aggregated append of one record per flagged lane; returns nothing (overflowing writes are dropped, the
// counter keeps counting so that the host can grow the list and rerun the batch)
template <typename T>
__device__ __forceinline__ void wave_append(bool flag, const T& rec, T* __restrict__ list, uint32_t* __restrict__ counter,
                                            uint32_t cap, int lane, unsigned long long lane_lt) {
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t wbase = 0;
        if (lane == leader) wbase = atomicAdd(counter, (uint32_t)__popcll(m));
        wbase = __shfl(wbase, leader, 64);
        const uint32_t slot = wbase + (uint32_t)__popcll(m & lane_lt);
        if (flag && slot < cap) list[slot] = rec;
    }
}

// =====================================================================================================================
// 1. main kernel: persistent lanes
// =====================================================================================================================
template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__global__ __launch_bounds__(EXT_THREADS) void extend_main_kernel(ExtendArgs a) {
    __shared__ int s_tab[128];
    if (threadIdx.x < 128) s_tab[threadIdx.x] = threadIdx.x < 64 ? a.sub_mat[threadIdx.x] : NEG;
    __syncthreads();

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
    bool has_hit = false;
    uint32_t ref_loc = 0, query_loc = 0, hidx = 0;
    uint32_t k = 0, lim = 0;
    int score = 0, best = 0, bpos = 0, bestR = 0, bposR = 0;
    unsigned long long examined = 0;

    for (;;) {
        // ================= 1. advance every live lane by one 8-base chunk =================
        if (phase < PH_FIN) {
            const bool left = phase == PH_LEFT;
            // in-range positions from k on; a live lane always has k <= lim (+1 on the left), so no underflow
            const uint32_t rem_u = left ? lim - k + 1u : lim - k;
            const int remaining = (int)min(rem_u, 8u);
            const uint64_t x = fetch_indices(R8b, Qb, ref_loc, query_loc, left, k, remaining);
            chunk8<COUNT_EXAMINED, XDROP_NONNEG>(s_tab, x, k, xdrop, score, best, bpos, examined);
            k += 8;
            const bool dead = score < (DEAD >> 1);
            if (dead) {
                if (!left) {  // -> left side (:457-476): anchor-1, anchor-2, ...
                    bestR = best;
                    bposR = bpos;
                    phase = PH_LEFT;
                    k = 1;
                    lim = min(ref_loc, query_loc);  // offsets 1..lim are in range (:482)
                    score = 0;
                    best = 0;
                    bpos = 0;
                } else {
                    phase = PH_FIN;
                }
            }
        }
        // ---- park sides that outlived long_cap: hand the hit (with its state) to the long kernel ----
        {
            const bool park = phase < PH_FIN && (k - (uint32_t)phase) >= long_cap;
            LongRec lr;
            lr.ref_loc = ref_loc; lr.query_loc = query_loc; lr.hidx = hidx; lr.side = (uint32_t)phase; lr.k = k;
            lr.score = score; lr.best = best; lr.bpos = bpos; lr.bestR = bestR; lr.bposR = bposR;
            wave_append(park, lr, a.long_list, a.long_count, a.long_cap_recs, lane, lane_lt);
            if (park) {
                has_hit = false;  // nothing to finalise here
                phase = PH_FIN;
            }
        }

        // ================= 2. finalise + refill in batches =================
        const unsigned long long fin = __ballot(phase == PH_FIN);
        const unsigned long long live = __ballot(phase < PH_FIN);
        if (fin != 0ull && (__popcll(fin) >= fin_batch || live == 0ull)) {
            // ---- score + filter (:608-647) for lanes that hold a finished hit ----
            int cls = 0;
            int total = 0, extent = 0;
            uint32_t seg = 0;
            if (phase == PH_FIN && has_hit) {
                total = bestR + best;   // best/bpos hold the left side now
                extent = bposR + bpos;
                cls = classify(a, total);
                if (cls) seg = seg_of(a, hidx);
            }
            {
                const HspRec rec = make_rec(a, ref_loc, query_loc, bpos, extent, total, seg);  // entropy 1: score = total (:638)
                wave_append(cls == 1, rec, a.out, a.out_count, a.out_cap, lane, lane_lt);
            }
            {
                EntRec er;
                er.ref_loc = ref_loc; er.query_loc = query_loc; er.bposR = bposR; er.boffL = bpos; er.total = total; er.seg = seg;
                wave_append(cls == 2, er, a.ent_list, a.ent_count, a.ent_cap_recs, lane, lane_lt);
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
                if (got) {
                    has_hit = true;
                    ref_loc = mine.ref_loc;
                    query_loc = mine.query_loc;
                    hidx = mine_idx;
                    bool skip = false;
                    if (a.rm)  // repeat masker: hits outside [ref_start, ref_end] are not extended (rm :239-244,:305-333)
                        skip = !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);
                    if (skip) {  // both loops skipped: total 0, extent 0 (:311)
                        bestR = 0; bposR = 0; best = 0; bpos = 0;
                        phase = PH_FIN;
                    } else {
                        phase = PH_RIGHT;  // :299-324
                        k = 0;
                        lim = (ref_loc < a.ref_len && query_loc < a.query_len)
                                  ? min(a.ref_len - ref_loc, a.query_len - query_loc) : 0u;
                        score = 0;
                        best = 0;
                        bpos = -1;
                    }
                } else {
                    has_hit = false;
                    phase = PH_IDLE;
                }
            }
        }
        if (__ballot(phase != PH_IDLE) == 0ull) break;
    }

    if (COUNT_EXAMINED) {
        unsigned long long v = examined;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && v) atomicAdd(a.examined, v);
    }
}

// =====================================================================================================================
// 2. long kernel: one wave per parked hit, 512 bases per step
// =====================================================================================================================
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__global__ __launch_bounds__(EXT_THREADS) void extend_long_kernel(ExtendArgs a) {
    __shared__ int s_tab[128];
    if (threadIdx.x < 128) s_tab[threadIdx.x] = threadIdx.x < 64 ? a.sub_mat[threadIdx.x] : NEG;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint8_t* __restrict__ R8b = a.ref8 - BIAS;
    const uint8_t* __restrict__ Qb = a.query - BIAS;
    const int xdrop = a.xdrop;
    const uint32_t n_long = min(*a.long_count, a.long_cap_recs);
    const uint32_t G = gridDim.x * (EXT_THREADS / 64);
    unsigned long long examined = 0;

    for (uint32_t i = blockIdx.x * (EXT_THREADS / 64) + (threadIdx.x >> 6); i < n_long; i += G) {
        const LongRec lr = a.long_list[i];  // same address in every lane: one broadcast load
        const uint32_t ref_loc = (uint32_t)rfl((int)lr.ref_loc), query_loc = (uint32_t)rfl((int)lr.query_loc);
        int side = rfl((int)lr.side);
        uint32_t k0 = (uint32_t)rfl((int)lr.k);
        int score_in = rfl(lr.score), best_in = rfl(lr.best), bpos_in = rfl(lr.bpos);
        int bestR = rfl(lr.bestR), bposR = rfl(lr.bposR);

        for (;;) {  // sides
            const bool left = side == PH_LEFT;
            const uint32_t lim = left ? min(ref_loc, query_loc)
                                      : ((ref_loc < a.ref_len && query_loc < a.query_len)
                                             ? min(a.ref_len - ref_loc, a.query_len - query_loc) : 0u);
            for (;;) {  // 512-base windows
                const uint32_t k = k0 + 8u * (uint32_t)lane;
                // in-range positions from k on; clamp far-away lanes so the int cast cannot wrap
                const int64_t rem64 = left ? (int64_t)lim - (int64_t)k + 1 : (int64_t)lim - (int64_t)k;
                const int remaining = rem64 > 8 ? 8 : (rem64 < 0 ? 0 : (int)rem64);
                const uint64_t x = fetch_indices(R8b, Qb, ref_loc, query_loc, left, k, remaining);
                // ---- local prefix sums of the 8 scores; local maximum prefix ----
                int s[8];
                {
                    const uint32_t xlo = (uint32_t)x, xhi = (uint32_t)(x >> 32);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t w = j < 4 ? xlo : xhi;
                        s[j] = s_tab[(w >> (8 * (j & 3))) & 0xffu];
                    }
                }
                // (sums are formed in uint32: lanes past a sequence edge accumulate terminators and may wrap; they lie
                //  after the first dropping lane and are discarded)
                uint32_t run = 0;
                int mx = INT32_MIN, amx = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    run += (uint32_t)s[j];
                    if ((int)run > mx) { mx = (int)run; amx = j; }  // first position attaining the local maximum
                }
                // ---- entry score of every lane: exclusive wave sum-scan ----
                uint32_t inc = run;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
                    if (lane >= off) inc += t;
                }
                const int base = (int)((uint32_t)score_in + inc - run);
                // ---- entry best of every lane: exclusive wave max-scan, ties keep the EARLIER position (:350,:361-372) ----
                int mv = (int)((uint32_t)base + (uint32_t)mx), mp = (int)(k + (uint32_t)amx);
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int tv = __shfl_up(mv, off, 64);
                    const int tp = __shfl_up(mp, off, 64);
                    if (lane >= off && tv >= mv) { mv = tv; mp = tp; }
                }
                int ev = __shfl_up(mv, 1, 64), ep = __shfl_up(mp, 1, 64);  // exclusive: best over earlier lanes
                if (lane == 0 || best_in >= ev) { ev = best_in; ep = bpos_in; }  // the carried-in best is the earliest of all
                // ---- exact replay of the lane's 8 bases ----
                int score = base, best = ev, bpos = ep;
                unsigned long long ex_step = 0;
                chunk8<COUNT_EXAMINED, XDROP_NONNEG>(s_tab, x, k, xdrop, score, best, bpos, ex_step);
                const bool dropped = score < (DEAD >> 1);
                const unsigned long long dm = __ballot(dropped);
                if (dm) {
                    const int f = __ffsll((long long)dm) - 1;  // first lane that dropped holds the final state
                    best_in = rfl(__shfl(best, f, 64));
                    bpos_in = rfl(__shfl(bpos, f, 64));
                    if (COUNT_EXAMINED && lane <= f) examined += ex_step;  // lanes after f never happened
                    break;
                }
                if (COUNT_EXAMINED) examined += ex_step;
                score_in = rfl(__shfl(score, 63, 64));
                best_in = rfl(__shfl(best, 63, 64));
                bpos_in = rfl(__shfl(bpos, 63, 64));
                k0 += 512u;
            }
            if (!left) {  // right side done -> left side from scratch (:457-476)
                bestR = best_in;
                bposR = bpos_in;
                side = PH_LEFT;
                k0 = 1;
                score_in = 0;
                best_in = 0;
                bpos_in = 0;
            } else {
                break;
            }
        }
        // ---- finalise (lane 0) ----
        if (lane == 0) {
            const int total = bestR + best_in, extent = bposR + bpos_in;
            const int cls = classify(a, total);
            if (cls) {
                const uint32_t seg = seg_of(a, lr.hidx);
                if (cls == 1) {
                    const uint32_t slot = atomicAdd(a.out_count, 1u);
                    if (slot < a.out_cap) a.out[slot] = make_rec(a, ref_loc, query_loc, bpos_in, extent, total, seg);
                } else {
                    EntRec er;
                    er.ref_loc = ref_loc; er.query_loc = query_loc; er.bposR = bposR; er.boffL = bpos_in; er.total = total; er.seg = seg;
                    const uint32_t slot = atomicAdd(a.ent_count, 1u);
                    if (slot < a.ent_cap_recs) a.ent_list[slot] = er;
                }
            }
        }
    }
    if (COUNT_EXAMINED) {
        unsigned long long v = examined;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && v) atomicAdd(a.examined, v);
    }
}

// =====================================================================================================================
// 3. entropy kernel: lane per candidate (:608-647)
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
                // :623 divides by log(4.0f): the FLOAT overload, i.e. (double)0x3FB17218 (hazard H2)
                entropy = -h / (double)1.38629436492919921875f;
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
template <bool C, bool X>
static void launch_variants(const ExtendArgs& a, uint32_t main_blocks, hipStream_t s) {
    hipLaunchKernelGGL((extend_main_kernel<C, X>), dim3(main_blocks), dim3(EXT_THREADS), 0, s, a);
    hipLaunchKernelGGL((extend_long_kernel<C, X>), dim3(a.long_blocks), dim3(EXT_THREADS), 0, s, a);
}

void launch_extend(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    const uint64_t num_buf = (a.num_hits + 63) / 64;
    // waves: enough to fill the chip (256 CUs x up to 32 waves); at least `bufs_per_wave` buffers per wave so the
    // drain phase of a wave (bounded by long_cap) is amortised
    uint64_t waves = num_buf / (uint64_t)(a.bufs_per_wave > 0 ? a.bufs_per_wave : 8);
    const uint64_t max_waves = 256ull * 32ull;
    if (waves > max_waves) waves = max_waves;
    if (waves < 4) waves = 4;
    const uint32_t blocks = (uint32_t)((waves + 3) / 4);
    const bool nonneg = a.xdrop >= 0;
    if (a.examined) {
        if (nonneg) launch_variants<true, true>(a, blocks, s);
        else launch_variants<true, false>(a, blocks, s);
    } else {
        if (nonneg) launch_variants<false, true>(a, blocks, s);
        else launch_variants<false, false>(a, blocks, s);
    }
    hipLaunchKernelGGL(extend_entropy_kernel, dim3(a.ent_blocks), dim3(EXT_THREADS), 0, s, a);
}

}  // namespace sa
