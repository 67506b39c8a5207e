// extend.hip -- ungapped X-drop extension + entropy filter + wavefront-ballot compaction of the survivors.
// Replaces find_hsps (src/seed_filter.cu:232-652), the done-flag scan (:769) and compress_output (:654-680).
//
// Design (wave64, CDNA4) -- NOT the reference's shape.  The reference gives every hit a 32-lane warp and pays four
// shuffle scans + ~10 warp syncs per 32-base tile although a random hit dies after ~20-60 bases.  The outcome of
// that tile loop does not depend on the tile width: it is the scalar recurrence below (tests/ check it against a
// 32-lane restatement).  So here ONE LANE OWNS ONE HIT and lanes are PERSISTENT:
//
//   * Every lane runs a small state machine (right side -> left side -> finished).  Each trip of the wave loop
//     advances every live lane by 8 bases.  A lane whose hit is finished does not wait for the slowest lane of
//     the wave: finished lanes are finalised in batches and REFILLED with the next hits of the wave's queue, so a
//     single 5 kb homologous extension no longer idles 63 lanes.
//   * The wave's queue is a register-held buffer of 64 hits (one coalesced 512 B load), double buffered; waves
//     take 64-hit buffers round-robin (buffer b belongs to wave b mod #waves): no atomics on the fetch side.
//   * The target is kept in HBM a second time "row coded" (r<<3, one byte per base) so that `rw | qw` of two
//     8-byte windows IS the 8 table indices r*8+q; one unaligned global_load_dwordx2 per sequence per 8 bases.
//     The left side byte-swaps its window so both directions share the same straight-line code.
//   * The 8x8 matrix sits in LDS as one 128-entry table: entries 64..127 hold a large negative "terminator" that
//     out-of-range positions are mapped to (bit 6 OR-ed into their index byte), which folds the sequence-edge
//     test into the X-drop test.  ACGTxACGT pairs (the common case) occupy 16 distinct banks: conflict-free.
//     Address = one SDWA byte-select shift; per base: 1 ds_read_b32 + 7 VALU.
//   * Once a side has dropped, its running score is pinned to DEAD, which makes every later base of the chunk a
//     no-op without per-base predication; "side finished" is read off the score after the chunk.
//   * Integer DP only (no MFMA).  The fp64 entropy term runs only for hits with hspthresh <= score <= 3*hspthresh
//     and recounts matches over the final interval (equal to the reference kernel's running counters).
//   * Survivors are appended with one atomicAdd per wave per batch (ballot + popcount prefix).  Append order is
//     arbitrary; the dedup stage sorts on a total order, so the output is deterministic.
//
// Scalar recurrence per side (k = 0,1,.. right of the anchor; k = 1,2,.. left of it):
//     score += M[r][q];  if (max(best,score) - score > xdrop) stop;  if (score > best) { best = score; bestpos = k; }
// stop also at the first position outside either sequence.  (:326-453 right, :478-604 left.)
#include "kernels.h"
#include "kmer_dev.h"  // load8u

namespace sa {

constexpr int EXT_THREADS = 256;
constexpr int NEG = -(1 << 28);   // score of a terminator pair: forces the drop test for any sane xdrop
constexpr int DEAD = -(1 << 29);  // sticky running score of a side that has dropped
constexpr uint64_t TERM_ALL = 0x4040404040404040ull;

__device__ __forceinline__ int f64_to_i32(double x) { return (int)x; }  // v_cvt_i32_f64: NaN -> 0, saturating (as on CUDA)

enum : int { PH_RIGHT = 0, PH_LEFT = 1, PH_FIN = 2, PH_IDLE = 3 };

template <bool COUNT_EXAMINED, bool XDROP_NONNEG>
__global__ __launch_bounds__(EXT_THREADS) void extend_kernel(ExtendArgs a) {
    __shared__ int s_tab[128];
    if (threadIdx.x < 128) s_tab[threadIdx.x] = threadIdx.x < 64 ? a.sub_mat[threadIdx.x] : NEG;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint8_t* __restrict__ R8 = a.ref8;   // row-coded target: byte = r << 3
    const uint8_t* __restrict__ Q = a.query;   // plain codes
    const int xdrop = a.xdrop;
    const int fin_batch = a.fin_batch;

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
    uint32_t ref_loc = 0, query_loc = 0;
    uint64_t hidx = 0;
    uint32_t k = 0, lim = 0;
    int score = 0, best = 0, bpos = 0, bestR = 0, bposR = 0;
    unsigned long long examined = 0;

    for (;;) {
        // ================= 1. advance every live lane by one 8-base chunk =================
        if (phase < PH_FIN) {
            const int remaining = (phase == PH_RIGHT) ? (int)(lim - k) : (int)(lim - k + 1u);  // in-range positions left
            uint64_t rw = 0, qw = 0;
            if (remaining > 0) {
                const int64_t roff = (phase == PH_RIGHT) ? (int64_t)ref_loc + (int64_t)k : (int64_t)ref_loc - (int64_t)k - 7;
                const int64_t qoff = (phase == PH_RIGHT) ? (int64_t)query_loc + (int64_t)k : (int64_t)query_loc - (int64_t)k - 7;
                rw = load8u(R8 + roff);
                qw = load8u(Q + qoff);
                if (phase == PH_LEFT) {  // byte j <-> offset k+j on both sides
                    rw = __builtin_bswap64(rw);
                    qw = __builtin_bswap64(qw);
                }
            }
            uint64_t x = rw | qw;  // 8 table indices (r<<3 | q), one per byte
            if (remaining < 8) x |= (remaining <= 0) ? TERM_ALL : (TERM_ALL << (8 * remaining));
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
            k += 8;
            const bool dead = score < (DEAD >> 1);
            if (dead) {
                if (phase == PH_RIGHT) {  // -> left side (:457-476): anchor-1, anchor-2, ...
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

        // ================= 2. finalise + refill in batches =================
        const unsigned long long fin = __ballot(phase == PH_FIN);
        const unsigned long long live = __ballot(phase < PH_FIN);
        if (fin != 0ull && (__popcll(fin) >= fin_batch || live == 0ull)) {
            // ---- score, entropy, filter (:608-647) for lanes that hold a finished hit ----
            bool pass = false;
            HspRec rec;
            rec.ref_start = rec.query_start = rec.len = 0; rec.score = 0; rec.seg = 0;
            if (phase == PH_FIN && has_hit) {
                const int total = bestR + best;     // best/bpos hold the left side now
                const int extent = bposR + bpos;
                double entropy = 1.0;
                if (total >= a.hspthresh && total <= 3 * a.hspthresh && !a.noentropy) {
                    // matches r==q<4 over the final interval [loc-boff, loc+bposR] (== the kernel's count[] :444-451;
                    // r>=4 would be the out-of-bounds counter write H1 and is not counted)
                    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                    for (int d = -bpos; d <= bposR; d++) {
                        const uint32_t r = (uint32_t)R8[(int64_t)ref_loc + d] >> 3, q = Q[(int64_t)query_loc + d];
                        if (r == q) { c0 += (r == 0); c1 += (r == 1); c2 += (r == 2); c3 += (r == 3); }
                    }
                    const short s0 = (short)c0, s1 = (short)c1, s2 = (short)c2, s3 = (short)c3;  // `short` counters :263
                    if ((s0 + s1 + s2 + s3) >= 20) {                                              // :617
                        const double len1 = (double)(extent + 1);
                        double e = 0.0;  // :620-622, same evaluation order
                        e += ((double)s0) / len1 * ((s0 != 0) ? log(((double)s0) / len1) : 0.0);
                        e += ((double)s1) / len1 * ((s1 != 0) ? log(((double)s1) / len1) : 0.0);
                        e += ((double)s2) / len1 * ((s2 != 0) ? log(((double)s2) / len1) : 0.0);
                        e += ((double)s3) / len1 * ((s3 != 0) ? log(((double)s3) / len1) : 0.0);
                        // :623 divides by log(4.0f): the FLOAT overload, i.e. (double)0x3FB17218 (hazard H2)
                        entropy = -e / (double)1.38629436492919921875f;
                    }
                }
                pass = f64_to_i32(((double)(float)total) * entropy) >= a.hspthresh;  // :633
                rec.ref_start = ref_loc - (uint32_t)bpos;      // :634
                rec.query_start = query_loc - (uint32_t)bpos;  // :635
                rec.len = (uint32_t)extent;                    // :636
                if (entropy > 0) rec.score = f64_to_i32((double)total * entropy);  // :637-638
                if (a.rm && a.rm_rev)  // rc coordinate flip of the repeat masker's compress_output (rm :705-708)
                    rec.query_start = a.ref_len - 1u - (rec.query_start + rec.len);
                uint32_t seg = 0;
                const uint64_t g = a.hit_base + hidx;
#pragma unroll
                for (int s = 0; s < MAX_SEGS - 1; s++)
                    if (s < a.num_segs - 1 && g >= a.seg_end[s]) seg = s + 1;
                rec.seg = a.seg_base + seg;
            }
            // ---- wave-level compaction: one atomic per wave per batch ----
            const unsigned long long m = __ballot(pass);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t wbase = 0;
                if (lane == leader) wbase = atomicAdd(a.out_count, (uint32_t)__popcll(m));
                wbase = __shfl(wbase, leader, 64);
                const uint32_t slot = wbase + (uint32_t)__popcll(m & lane_lt);
                if (pass && slot < a.out_cap) a.out[slot] = rec;  // overflow: host grows the buffer and reruns the batch
            }
            // ---- refill the finished lanes from the wave's queue (wave-uniform control flow) ----
            unsigned long long need = fin;
            bool got = false;
            Hit mine = {0u, 0u};
            uint64_t mine_idx = 0;
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
                    mine_idx = (cur_buf << 6) + (uint64_t)src;
                    got = true;
                }
                const int ntake = min(__popcll(need), avail);
                consumed += ntake;
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

void launch_extend(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    const uint64_t num_buf = (a.num_hits + 63) / 64;
    // waves: enough to fill the chip (256 CUs x up to 32 waves) but at least `bufs_per_wave` buffers per wave so
    // the drain phase of a wave (lanes finishing their last, possibly long, hits) is amortised
    uint64_t waves = num_buf / (uint64_t)(a.bufs_per_wave > 0 ? a.bufs_per_wave : 8);
    const uint64_t max_waves = 256ull * 32ull;
    if (waves > max_waves) waves = max_waves;
    if (waves < 4) waves = 4;
    const uint32_t blocks = (uint32_t)((waves + 3) / 4);
    const bool nonneg = a.xdrop >= 0;
    if (a.examined) {
        if (nonneg) hipLaunchKernelGGL((extend_kernel<true, true>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
        else hipLaunchKernelGGL((extend_kernel<true, false>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
    } else {
        if (nonneg) hipLaunchKernelGGL((extend_kernel<false, true>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
        else hipLaunchKernelGGL((extend_kernel<false, false>), dim3(blocks), dim3(EXT_THREADS), 0, s, a);
    }
}

}  // namespace sa
