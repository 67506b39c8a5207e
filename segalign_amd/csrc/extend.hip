// extend.hip -- ungapped X-drop extension + entropy filter + wavefront-ballot compaction of the survivors.
// Replaces find_hsps (src/seed_filter.cu:232-652), the done-flag scan (:769) and compress_output (:654-680).
//
// Design (wave64, CDNA4) -- NOT the reference's shape.  The reference gives every hit a 32-lane warp and pays four
// shuffle scans + ~10 warp syncs per 32-base tile, although a random hit dies after ~20-60 bases.  The result of
// the tile loop does not depend on the tile width (it is the scalar recurrence below; the tests show it against a
// 32-lane restatement), so here ONE LANE OWNS ONE HIT:
//   * sequence bytes are fetched 8 at a time (one unaligned global_load_dwordx2 per sequence per 8 bases) and
//     walked from registers;
//   * the 8x8 substitution matrix sits in LDS replicated 32x (8 KB) so that lane l always reads bank l%32:
//     every ds_read_b32 is conflict-free regardless of the (r,q) pairs the 64 lanes look up;
//   * integer DP only (no MFMA); the fp64 entropy term runs only for the few hits with
//     hspthresh <= score <= 3*hspthresh and recounts the matches over the final interval;
//   * survivors are appended with one atomicAdd per wave (ballot + popcount prefix).  Append order is arbitrary;
//     the dedup stage sorts on a total order, so the output is deterministic.
//
// Scalar recurrence per side (k = 0,1,.. right of the anchor; k = 1,2,.. left of it):
//     score += M[r][q];  if (max(best,score) - score > xdrop) stop;  if (score > best) { best = score; bestpos = k; }
// stop also at the first position outside either sequence.  (:326-453 right, :478-604 left.)
#include "kernels.h"
#include "kmer_dev.h"  // load8u

namespace sa {

constexpr int EXT_THREADS = 256;

__device__ __forceinline__ int f64_to_i32(double x) { return (int)x; }  // v_cvt_i32_f64: NaN -> 0, saturating (as on CUDA)

template <bool COUNT_EXAMINED>
__global__ __launch_bounds__(EXT_THREADS) void extend_kernel(ExtendArgs a) {
    __shared__ int s_mat[64 * 32];  // s_mat[idx*32 + (lane&31)] == sub_mat[idx]
    for (int i = threadIdx.x; i < 64 * 32; i += EXT_THREADS) s_mat[i] = a.sub_mat[i >> 5];
    __syncthreads();
    const int* mat = s_mat + (threadIdx.x & 31);
    const int lane = threadIdx.x & 63;
    const uint8_t* __restrict__ R = a.ref;
    const uint8_t* __restrict__ Q = a.query;
    const int xdrop = a.xdrop;

    const uint64_t stride = (uint64_t)gridDim.x * EXT_THREADS;
    // wave-uniform trip count: every lane of a wave leaves the loop together (ballots below need all lanes)
    for (uint64_t base = (uint64_t)blockIdx.x * EXT_THREADS; base < a.num_hits; base += stride) {
        const uint64_t hid = base + threadIdx.x;
        const bool active = hid < a.num_hits;
        Hit h = {0u, 0u};
        if (active) h = a.hits[hid];
        const uint32_t ref_loc = h.ref_loc, query_loc = h.query_loc;
        bool skip = !active;
        if (a.rm && active)  // repeat masker: hits outside [ref_start, ref_end] are not extended (rm :239-244,:305-333)
            skip = !(ref_loc >= a.rm_win_start && ref_loc <= a.rm_win_end);

        unsigned long long examined = 0;

        // ---------------- right extension (:299-453) ----------------
        int bestR = 0, bposR = skip && a.rm ? 0 : -1;
        {
            uint32_t lim = 0;  // number of in-range positions to the right
            if (!skip && ref_loc < a.ref_len && query_loc < a.query_len)
                lim = min(a.ref_len - ref_loc, a.query_len - query_loc);
            int score = 0;
            uint32_t k = 0;
            bool done = (lim == 0);
            while (!done) {
                const uint64_t rw = load8u(R + ref_loc + k);
                const uint64_t qw = load8u(Q + query_loc + k);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (!done) {
                        if (k + j >= lim) {
                            done = true;
                        } else {
                            const uint32_t r = (uint32_t)(rw >> (8 * j)) & 7u;
                            const uint32_t q = (uint32_t)(qw >> (8 * j)) & 7u;
                            score += mat[((r << 3) | q) << 5];
                            if (COUNT_EXAMINED) examined++;
                            const int nb = max(bestR, score);
                            if (nb - score > xdrop) done = true;
                            else if (score > bestR) { bestR = score; bposR = (int)(k + j); }
                        }
                    }
                }
                k += 8;
            }
        }
        // ---------------- left extension (:457-604) ----------------
        int bestL = 0, boffL = 0;
        {
            const uint32_t lim = skip ? 0u : min(ref_loc, query_loc);  // positions k = 1..lim are in range (:482)
            int score = 0;
            uint32_t k = 1;
            bool done = (lim == 0);
            while (!done) {
                // bytes at positions loc-k-7 .. loc-k ; byte (7-j) <-> offset k+j
                const uint64_t rw = load8u(R + ref_loc - k - 7);
                const uint64_t qw = load8u(Q + query_loc - k - 7);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (!done) {
                        if (k + j > lim) {
                            done = true;
                        } else {
                            const uint32_t r = (uint32_t)(rw >> (8 * (7 - j))) & 7u;
                            const uint32_t q = (uint32_t)(qw >> (8 * (7 - j))) & 7u;
                            score += mat[((r << 3) | q) << 5];
                            if (COUNT_EXAMINED) examined++;
                            const int nb = max(bestL, score);
                            if (nb - score > xdrop) done = true;
                            else if (score > bestL) { bestL = score; boffL = (int)(k + j); }
                        }
                    }
                }
                k += 8;
            }
        }
        // ---------------- score, entropy, filter (:608-647) ----------------
        const int total = bestR + bestL;
        const int extent = bposR + boffL;
        double entropy = 1.0;
        if (active && total >= a.hspthresh && total <= 3 * a.hspthresh && !a.noentropy) {
            // matches r==q<4 over the final interval [loc-boffL, loc+bposR]  (== the kernel's count[] at :444-451;
            // r>=4 would be the out-of-bounds counter write H1 and is not counted)
            int cnt[4] = {0, 0, 0, 0};
            for (int k = -boffL; k <= bposR; k++) {
                const uint32_t r = R[ref_loc + k], q = Q[query_loc + k];
                if (r == q && r < 4) cnt[r]++;
            }
            short c0 = (short)cnt[0], c1 = (short)cnt[1], c2 = (short)cnt[2], c3 = (short)cnt[3];  // `short` counters :263
            if ((c0 + c1 + c2 + c3) >= 20) {                                                         // :617
                const double len1 = (double)(extent + 1);
                double e = 0.0;
                e += ((double)c0) / len1 * ((c0 != 0) ? log(((double)c0) / len1) : 0.0);  // :620-622, same order
                e += ((double)c1) / len1 * ((c1 != 0) ? log(((double)c1) / len1) : 0.0);
                e += ((double)c2) / len1 * ((c2 != 0) ? log(((double)c2) / len1) : 0.0);
                e += ((double)c3) / len1 * ((c3 != 0) ? log(((double)c3) / len1) : 0.0);
                // :623 divides by log(4.0f): the FLOAT overload, i.e. (double)0x3FB17218 (hazard H2)
                entropy = -e / (double)1.38629436492919921875f;
            }
        }
        bool pass = active && (f64_to_i32(((double)(float)total) * entropy) >= a.hspthresh);  // :633

        HspRec rec;
        rec.ref_start = ref_loc - (uint32_t)boffL;    // :634
        rec.query_start = query_loc - (uint32_t)boffL;  // :635
        rec.len = (uint32_t)extent;                   // :636
        rec.score = 0;
        if (entropy > 0) rec.score = f64_to_i32((double)total * entropy);  // :637-638
        if (a.rm && a.rm_rev)  // rc coordinate flip of the repeat masker's compress_output (rm :705-708)
            rec.query_start = a.ref_len - 1u - (rec.query_start + rec.len);
        // segment (reference iteration) of this hit
        uint32_t seg = 0;
        {
            const uint64_t g = a.hit_base + hid;
#pragma unroll
            for (int s = 0; s < MAX_SEGS - 1; s++)
                if (s < a.num_segs - 1 && g >= a.seg_end[s]) seg = s + 1;
        }
        rec.seg = a.seg_base + seg;

        // ---------------- wave-level compaction: one atomic per wave ----------------
        const unsigned long long m = __ballot(pass);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t wbase = 0;
            if (lane == leader) wbase = atomicAdd(a.out_count, (uint32_t)__popcll(m));
            wbase = __shfl(wbase, leader, 64);
            const uint32_t slot = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pass && slot < a.out_cap) a.out[slot] = rec;  // overflow: host grows the buffer and reruns the batch
        }
        if (COUNT_EXAMINED) {
            // wave reduction then one atomic
            unsigned long long v = examined;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(a.examined, v);
        }
    }
}

void launch_extend(const ExtendArgs& a, hipStream_t s) {
    if (a.num_hits == 0) return;
    uint64_t blocks = (a.num_hits + EXT_THREADS - 1) / EXT_THREADS;
    const uint64_t max_blocks = 256ull * 8ull * 4ull;  // 256 CUs x 8 resident blocks x 4 rounds, grid-stride beyond
    if (blocks > max_blocks) blocks = max_blocks;
    if (a.examined)
        hipLaunchKernelGGL(extend_kernel<true>, dim3((uint32_t)blocks), dim3(EXT_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL(extend_kernel<false>, dim3((uint32_t)blocks), dim3(EXT_THREADS), 0, s, a);
}

}  // namespace sa
