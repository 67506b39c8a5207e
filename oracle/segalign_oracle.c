/*
 * segalign_oracle.c -- CPU restatement of SegAlign's seed -> filter -> ungapped-extend path.
 * TEST INFRASTRUCTURE ONLY (see segalign_oracle.h for the rules and the pinning status).
 * Plain C11 + optional OpenMP; no code is taken from the reference, every function cites the lines it follows.
 */
#include "segalign_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =====================================================================================================
 * Scoring matrix -- src/main.cpp:187-268
 * ===================================================================================================== */
void orc_build_sub_mat(int* m, int xdrop, int ambiguous_mode, int ambiguous_reward, int ambiguous_penalty) {
    const int fill_score = -100, bad_score = -1000; /* main.cpp:189-190 */
    /* HOXD70 core, main.cpp:208-211 */
    static const int core[4][4] = {
        {91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
    memset(m, 0, 64 * sizeof(int));
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m[i * 8 + j] = core[i][j];
    /* lower case (soft-masked), main.cpp:220-224 */
    for (int i = 0; i < ORC_L; i++) {
        m[i * 8 + ORC_L] = bad_score;
        m[ORC_L * 8 + i] = bad_score;
    }
    m[ORC_L * 8 + ORC_L] = bad_score;
    /* N, main.cpp:227-240 */
    if (ambiguous_mode == 1 || ambiguous_mode == 2) {
        for (int i = 0; i < ORC_N; i++) {
            m[i * 8 + ORC_N] = ambiguous_penalty;
            m[ORC_N * 8 + i] = ambiguous_penalty;
        }
        m[ORC_N * 8 + ORC_N] = ambiguous_reward;
    } else {
        for (int i = 0; i < ORC_N; i++) {
            m[i * 8 + ORC_N] = bad_score;
            m[ORC_N * 8 + i] = bad_score;
        }
        m[ORC_N * 8 + ORC_N] = bad_score;
    }
    /* other IUPAC -> X, main.cpp:243-261 */
    if (ambiguous_mode == 2) {
        for (int i = 0; i < ORC_X; i++) {
            m[i * 8 + ORC_X] = ambiguous_penalty;
            m[ORC_X * 8 + i] = ambiguous_penalty;
        }
        m[ORC_X * 8 + ORC_X] = ambiguous_reward;
    } else {
        for (int i = 0; i < ORC_L; i++) {
            m[i * 8 + ORC_X] = fill_score;
            m[ORC_X * 8 + i] = fill_score;
        }
        for (int i = ORC_L; i < ORC_X; i++) {
            m[i * 8 + ORC_X] = bad_score;
            m[ORC_X * 8 + i] = bad_score;
        }
        m[ORC_X * 8 + ORC_X] = fill_score;
    }
    /* sequence separator '&' -> E, main.cpp:263-267 */
    for (int i = 0; i < ORC_E; i++) {
        m[i * 8 + ORC_E] = -10 * xdrop;
        m[ORC_E * 8 + i] = -10 * xdrop;
    }
    m[ORC_E * 8 + ORC_E] = -10 * xdrop;
}

/* =====================================================================================================
 * Encoding -- common/seed_filter_interface.cu:18-47 (target), src/seed_filter.cu:110-155 (query + rc)
 * ===================================================================================================== */
static inline uint8_t enc(char ch) {
    switch (ch) {
        case 'A': return ORC_A;
        case 'C': return ORC_C;
        case 'G': return ORC_G;
        case 'T': return ORC_T;
        case 'a': case 'c': case 'g': case 't': return ORC_L;
        case 'n': case 'N': return ORC_N;
        case '&': return ORC_E;
        default: return ORC_X;
    }
}
/* team size of a parallel loop: never more threads than 4096-item blocks (a 256-thread fork/join per tiny test chunk costs
 * more than the chunk) */
static inline int orc_team(int want, uint64_t items) {
    uint64_t blocks = (items + 4095) / 4096;
    if (want < 1) want = 1;
    if (blocks < 1) blocks = 1;
    return blocks < (uint64_t)want ? (int)blocks : want;
}
static inline uint8_t comp_code(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : c; } /* :124-151, L/N/E/X keep */

void orc_encode(const char* src, size_t len, uint8_t* dst) {
    for (size_t i = 0; i < len; i++) dst[i] = enc(src[i]);
}
void orc_encode_rev_comp(const char* src, size_t len, uint8_t* dst, uint8_t* dst_rc) {
    for (size_t i = 0; i < len; i++) {
        uint8_t c = enc(src[i]);
        dst[i] = c;
        dst_rc[len - 1 - i] = comp_code(c); /* seed_filter.cu:153 */
    }
}
void orc_rev_comp_codes(const uint8_t* src, size_t len, uint8_t* dst_rc) {
    for (size_t i = 0; i < len; i++) dst_rc[len - 1 - i] = comp_code(src[i]); /* rm seed_filter.cu:150-165 */
}

/* common/ntcoding.cpp:63-105.  Characters outside {acgtnACGTN&} print a warning there and are SKIPPED
 * (r is not advanced); restated identically, minus the printf. */
void orc_rev_comp_ascii(char* dst, const char* src, size_t rc_start, size_t start, size_t len) {
    size_t r = rc_start;
    for (size_t i = start + len; i > start; i--) {
        char o = 0;
        switch (src[i - 1]) {
            case 'a': o = 't'; break;
            case 'A': o = 'T'; break;
            case 'c': o = 'g'; break;
            case 'C': o = 'G'; break;
            case 'g': o = 'c'; break;
            case 'G': o = 'C'; break;
            case 't': o = 'a'; break;
            case 'T': o = 'A'; break;
            case 'n': o = 'n'; break;
            case 'N': o = 'N'; break;
            case '&': o = '&'; break;
            default: o = 0;
        }
        if (o) dst[r++] = o;
    }
}

/* =====================================================================================================
 * Seed shape + k-mer -- common/ntcoding.cpp:6-61
 * ===================================================================================================== */
static int g_shape_pos[32];
static int g_shape_size;
static int g_transition_pos[32];

int orc_generate_shape_pos(const char* shape) { /* ntcoding.cpp:21-37 */
    g_shape_size = 0;
    int j = 0;
    for (int i = 0; shape[i] != '\0'; i++) {
        if (shape[i] == '1' || shape[i] == 'T') {
            g_shape_pos[g_shape_size++] = i;
            g_transition_pos[j] = (shape[i] == 'T') ? 1 : 0;
            j++;
        }
    }
    return g_shape_size;
}
int orc_is_transition_at_pos(int t) { return g_transition_pos[t]; } /* ntcoding.cpp:39-41 */

uint32_t orc_kmer_index_at_pos(const char* seq, size_t pos, uint32_t seed_size) { /* ntcoding.cpp:43-61 */
    uint32_t nt[64];
    for (uint32_t i = 0; i < seed_size; i++) {
        switch (seq[pos + i]) { /* ntcoding.cpp:10-19: anything but upper-case ACGT is N -> invalid */
            case 'A': nt[i] = 0; break;
            case 'C': nt[i] = 1; break;
            case 'G': nt[i] = 2; break;
            case 'T': nt[i] = 3; break;
            default: return ORC_INVALID_KMER;
        }
    }
    uint32_t kmer = 0;
    for (int i = 0; i < g_shape_size; i++) kmer = (kmer << 2) + nt[g_shape_pos[i]]; /* first care pos = MSBs */
    return kmer;
}

/* =====================================================================================================
 * Seed position table -- common/seed_pos_table.cu:49-109
 * ===================================================================================================== */
uint32_t orc_generate_seed_pos_table(const char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step,
                                     int shape_size, int kmer_size, uint32_t* index_out, uint32_t* pos_out) {
    uint32_t offset = (uint32_t)(shape_size + 1) % step;                  /* :58 */
    uint32_t start_offset = step - offset;                                /* :59 */
    uint32_t nkeys = (uint32_t)1 << (2 * kmer_size);                      /* :61 (without the +1 slot) */
    uint32_t num_steps = (ref_length - (uint32_t)shape_size + offset) / step; /* :64 */
    uint32_t* keys = (uint32_t*)malloc((size_t)num_steps * sizeof(uint32_t));
    uint32_t* start = (uint32_t*)calloc((size_t)nkeys + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < num_steps; i++) { /* pass 1, :69-81 */
        uint32_t k = orc_kmer_index_at_pos(ref_str, start_addr + start_offset + (size_t)i * step, (uint32_t)shape_size);
        keys[i] = k;
        if (k != ORC_INVALID_KMER) start[k + 1]++;
    }
    for (uint32_t k = 0; k < nkeys; k++) start[k + 1] += start[k]; /* InclusivePrefixScan, :8-31,:83 */
    uint32_t num_index = start[nkeys];                             /* :85 */
    for (uint32_t k = 0; k < nkeys; k++) index_out[k] = start[k + 1]; /* device view = index_table+1, :103 */
    for (uint32_t i = 0; i < num_steps; i++) {                        /* pass 2, :89-101 (ascending = stable) */
        uint32_t k = keys[i];
        if (k != ORC_INVALID_KMER) pos_out[start[k]++] = start_offset + i * step;
    }
    free(keys);
    free(start);
    return num_index;
}

/* =====================================================================================================
 * Host seeding loop -- src/seeder.cpp:57-74 (plus strand) == :94-109 (minus strand, on the rc buffer)
 * ===================================================================================================== */
size_t orc_make_seeds(const char* qbuf, size_t q_block_start, uint32_t i, uint32_t e, uint32_t seed_size,
                      int kmer_size, int transition, uint64_t* out) {
    size_t n = 0;
    for (uint32_t j = i; j < e; j++) {
        uint64_t kmer = orc_kmer_index_at_pos(qbuf, q_block_start + j, seed_size);
        if (kmer != ((uint32_t)1 << 31)) {
            out[n++] = (kmer << 32) + j; /* :60-61 */
            if (transition) {
                for (int t = 0; t < kmer_size; t++) {
                    if (orc_is_transition_at_pos(t) == 1) {
                        uint64_t tr = kmer ^ ((uint64_t)2 << (2 * t)); /* TRANSITION_MASK, :66 */
                        out[n++] = (tr << 32) + j;
                    }
                }
            }
        }
    }
    return n;
}

/* =====================================================================================================
 * Ungapped X-drop extension -- src/seed_filter.cu:232-652, scalar form.
 *
 * One side of the kernel's tile loop, position by position.  Equivalence with the tiled form:
 *   - prefix-sum of tile scores + prev_score (:339-349)            == running `score`
 *   - "new max only if strictly greater" (:350) + max-scan whose ties keep the LOWER lane (:361-372)
 *                                                                  == best updated on `score > best`, first
 *                                                                     position attaining it is kept
 *   - xdrop flag computed against the INCLUSIVE prefix max (:374), OR-scanned (:377-384), lanes at/after
 *     the first flagged lane fall back to the previous max (:386-389) == stop at the first k where
 *     max(best,score) - score > xdrop, best taken over positions < k
 *   - lanes outside either sequence contribute 0 (:330-336) and the loop ends when the LAST lane of a tile
 *     is outside (:420)                                            == stop at the first out-of-range position
 * The match counters (:436-451) end up as "# positions <= best position with r==q (and r<4, see H1)".
 * ===================================================================================================== */
typedef struct {
    int best;
    int bestpos; /* right: index k of the best position, -1 if none; left: offset (>=1) of best, 0 if none */
} side_result;

static side_result extend_right(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, uint64_t* ex) {
    side_result s = {0, -1}; /* prev_max_score = 0, prev_max_pos = -1, :308-310 */
    int score = 0;
    for (int k = 0;; k++) {
        uint32_t rp = ref_loc + (uint32_t)k, qp = query_loc + (uint32_t)k; /* :328-329 */
        if (!(rp < p->ref_len && qp < p->query_len)) break;                /* :332, :420 */
        score += p->sub_mat[p->ref[rp] * 8 + p->query[qp]];
        if (ex) (*ex)++;
        int nb = score > s.best ? score : s.best;
        if (nb - score > p->xdrop) break; /* :374 */
        if (score > s.best) {             /* :350 */
            s.best = score;
            s.bestpos = k;
        }
    }
    return s;
}

static side_result extend_left(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, uint64_t* ex) {
    side_result s = {0, 0}; /* prev_max_pos = 0, :465-467 */
    int score = 0;
    for (uint32_t k = 1;; k++) {                          /* pos_offset = lane+1+tile, :479 */
        if (!(ref_loc >= k && query_loc >= k)) break;     /* :482, :570 */
        uint32_t rp = ref_loc - k, qp = query_loc - k;
        score += p->sub_mat[p->ref[rp] * 8 + p->query[qp]];
        if (ex) (*ex)++;
        int nb = score > s.best ? score : s.best;
        if (nb - score > p->xdrop) break; /* :523 */
        if (score > s.best) {             /* :500 */
            s.best = score;
            s.bestpos = (int)k;
        }
    }
    return s;
}

/* float->int conversion as the GPU does it (NaN -> 0, saturating); only differs from C on NaN/overflow */
static int f64_to_i32_gpu(double x) {
    if (isnan(x)) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return (int)x;
}

/* :608-647 given both sides; counts are recomputed over the final interval (see equivalence note above). */
static int finish_hit(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, side_result R, side_result L,
                      orc_segment* out) {
    int total = R.best + L.best;   /* :414/:421 + :563/:571 */
    int extent = R.bestpos + L.bestpos; /* :416 then :566 */
    double entropy = 1.0;          /* :307 (1.0f stored to a double) */
    if (total >= p->hspthresh && total <= 3 * p->hspthresh && !p->noentropy) { /* :608 */
        long cnt[4] = {0, 0, 0, 0};
        for (int k = 0; k <= R.bestpos; k++) {
            uint8_t r = p->ref[ref_loc + (uint32_t)k], q = p->query[query_loc + (uint32_t)k];
            if (r == q && r < 4) cnt[r]++; /* :444-447; r>=4 is the out-of-bounds write H1: not counted */
        }
        for (int k = 1; k <= L.bestpos; k++) {
            uint8_t r = p->ref[ref_loc - (uint32_t)k], q = p->query[query_loc - (uint32_t)k];
            if (r == q && r < 4) cnt[r]++; /* :595-598 */
        }
        short c[4]; /* per-lane `short` counters summed in `short` (:263, :609-614): wraps mod 2^16 */
        for (int i = 0; i < 4; i++) c[i] = (short)(cnt[i] & 0xFFFF);
        if ((c[0] + c[1] + c[2] + c[3]) >= 20) { /* :617 */
            entropy = 0.f;
            for (int i = 0; i < 4; i++) { /* :620-622, evaluation order kept */
                entropy += ((double)c[i]) / ((double)(extent + 1)) *
                           ((c[i] != 0) ? log(((double)c[i]) / ((double)(extent + 1))) : 0.f);
            }
            double div = p->log4_is_float ? (double)logf(4.0f) : log(4.0); /* :623, hazard H2 */
            entropy = -entropy / div;
            for (int u = 0; u < p->entropy_ulps; u++) entropy = nextafter(entropy, 2.0);   /* H13 probe, 0 by default */
            for (int u = 0; u > p->entropy_ulps; u--) entropy = nextafter(entropy, -1.0);
        }
    }
    if (f64_to_i32_gpu(((float)total) * entropy) >= p->hspthresh) { /* :633 */
        out->ref_start = ref_loc - (uint32_t)L.bestpos;               /* :634 */
        out->query_start = query_loc - (uint32_t)L.bestpos;           /* :635 */
        out->len = (uint32_t)extent;                                  /* :636 */
        out->score = 0;                                               /* hit record was created with 0, :226 */
        if (entropy > 0) out->score = f64_to_i32_gpu(total * entropy); /* :637-638 */
        return 1;
    }
    out->ref_start = ref_loc; /* :642-646 */
    out->query_start = query_loc;
    out->len = 0;
    out->score = 0;
    return 0;
}

int orc_extend_hit(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, orc_segment* out,
                   uint64_t* examined) {
    side_result R = extend_right(p, ref_loc, query_loc, examined);
    side_result L = extend_left(p, ref_loc, query_loc, examined);
    return finish_hit(p, ref_loc, query_loc, R, L, out);
}

/* -----------------------------------------------------------------------------------------------------
 * Independent tile-by-tile restatement of the same kernel (src/seed_filter.cu:299-647) with W lanes.
 * Lane-private variables are arrays; warp-shared scalars are plain variables; each __shfl_up scan is
 * written as the inclusive scan it implements.  H1: the counter index is guarded with r<4; out-of-range
 * lanes keep their stale r_chr/q_chr exactly like the kernel (they can only reach count_del, which is
 * discarded).  Used by the tests to show the scalar form above is tile-width independent.
 * ----------------------------------------------------------------------------------------------------- */
int orc_extend_hit_tiled(const orc_extend_params* p, uint32_t ref_loc, uint32_t query_loc, int W,
                         orc_segment* out) {
    int thread_score[64], max_score[64], max_pos[64], xd[64];
    short count[64][4], count_del[64][4];
    uint8_t r_chr[64], q_chr[64];
    int total_score = 0, prev_score, prev_max_score, prev_max_pos, extent = 0;
    uint32_t left_extent = 0, tile;
    int xdrop_found, edge_found, new_max_found;
    double entropy = 1.0f;
    memset(count, 0, sizeof(count));
    memset(count_del, 0, sizeof(count_del));
    for (int l = 0; l < W; l++) { r_chr[l] = 0; q_chr[l] = 1; } /* uninitialised in the kernel; neutral here */

    for (int side = 0; side < 2; side++) {
        tile = 0; xdrop_found = 0; edge_found = 0; new_max_found = 0;
        prev_score = 0; prev_max_score = 0;
        prev_max_pos = side == 0 ? -1 : 0; /* :310 / :467 */
        for (int l = 0; l < W; l++) memset(count_del[l], 0, sizeof(count_del[l])); /* :318-321 / :471-474 */
        while (!xdrop_found && !edge_found) {
            int last_in_range = 1;
            int pos_offset[64];
            for (int l = 0; l < W; l++) {
                thread_score[l] = 0;
                if (side == 0) {
                    pos_offset[l] = l + (int)tile; /* :327 */
                    uint32_t rp = ref_loc + (uint32_t)pos_offset[l], qp = query_loc + (uint32_t)pos_offset[l];
                    int in = (rp < p->ref_len && qp < p->query_len);
                    if (in) { r_chr[l] = p->ref[rp]; q_chr[l] = p->query[qp]; thread_score[l] = p->sub_mat[r_chr[l] * 8 + q_chr[l]]; }
                    if (l == W - 1) last_in_range = in;
                } else {
                    pos_offset[l] = l + 1 + (int)tile; /* :479 */
                    int in = (ref_loc >= (uint32_t)pos_offset[l] && query_loc >= (uint32_t)pos_offset[l]);
                    if (in) {
                        r_chr[l] = p->ref[ref_loc - (uint32_t)pos_offset[l]];
                        q_chr[l] = p->query[query_loc - (uint32_t)pos_offset[l]];
                        thread_score[l] = p->sub_mat[r_chr[l] * 8 + q_chr[l]];
                    }
                    if (l == W - 1) last_in_range = in;
                }
            }
            for (int l = 1; l < W; l++) thread_score[l] += thread_score[l - 1]; /* sum scan :339-346 */
            for (int l = 0; l < W; l++) {
                thread_score[l] += prev_score; /* :349 */
                if (thread_score[l] > prev_max_score) { max_score[l] = thread_score[l]; max_pos[l] = pos_offset[l]; }
                else { max_score[l] = prev_max_score; max_pos[l] = prev_max_pos; }
            }
            for (int l = 1; l < W; l++) /* max scan, ties keep the lower lane :361-372 */
                if (max_score[l - 1] >= max_score[l]) { max_score[l] = max_score[l - 1]; max_pos[l] = max_pos[l - 1]; }
            for (int l = 0; l < W; l++) xd[l] = (max_score[l] - thread_score[l]) > p->xdrop; /* :374 */
            for (int l = 1; l < W; l++) xd[l] |= xd[l - 1];                                   /* :377-384 */
            for (int l = 0; l < W; l++)
                if (xd[l]) { max_score[l] = prev_max_score; max_pos[l] = prev_max_pos; }     /* :386-389 */
            for (int l = 1; l < W; l++) /* second max scan :392-403 */
                if (max_score[l - 1] >= max_score[l]) { max_score[l] = max_score[l - 1]; max_pos[l] = max_pos[l - 1]; }
            { /* last lane :406-433 / :555-584 */
                int l = W - 1;
                new_max_found = max_pos[l] > prev_max_pos;
                if (xd[l] || !last_in_range) {
                    total_score += max_score[l];
                    if (xd[l]) xdrop_found = 1; else edge_found = 1;
                    if (side == 0) extent = max_pos[l];
                    else { left_extent = (uint32_t)max_pos[l]; extent += (int)left_extent; }
                    prev_max_pos = max_pos[l];
                    tile = (uint32_t)max_pos[l];
                } else {
                    prev_score = thread_score[l];
                    prev_max_score = max_score[l];
                    prev_max_pos = max_pos[l];
                    tile += (uint32_t)W;
                }
            }
            for (int l = 0; l < W; l++) {
                if (new_max_found) /* :436-441 */
                    for (int i = 0; i < 4; i++) { count[l][i] = (short)(count[l][i] + count_del[l][i]); count_del[l][i] = 0; }
                if (r_chr[l] == q_chr[l] && r_chr[l] < 4) { /* :444-451, guarded (H1) */
                    if (pos_offset[l] <= prev_max_pos) count[l][r_chr[l]] = (short)(count[l][r_chr[l]] + 1);
                    else count_del[l][r_chr[l]] = (short)(count_del[l][r_chr[l]] + 1);
                }
            }
        }
    }
    if (total_score >= p->hspthresh && total_score <= 3 * p->hspthresh && !p->noentropy) { /* :608 */
        short c[4];
        for (int i = 0; i < 4; i++) { /* :609-614: the value read is the last lane's = sum over lanes, in short */
            short s = 0;
            for (int l = 0; l < W; l++) s = (short)(s + count[l][i]);
            c[i] = s;
        }
        if ((c[0] + c[1] + c[2] + c[3]) >= 20) {
            entropy = 0.f;
            for (int i = 0; i < 4; i++)
                entropy += ((double)c[i]) / ((double)(extent + 1)) *
                           ((c[i] != 0) ? log(((double)c[i]) / ((double)(extent + 1))) : 0.f);
            double div = p->log4_is_float ? (double)logf(4.0f) : log(4.0);
            entropy = -entropy / div;
        }
    }
    if (f64_to_i32_gpu(((float)total_score) * entropy) >= p->hspthresh) {
        out->ref_start = ref_loc - left_extent;
        out->query_start = query_loc - left_extent;
        out->len = (uint32_t)extent;
        out->score = 0;
        if (entropy > 0) out->score = f64_to_i32_gpu(total_score * entropy);
        return 1;
    }
    out->ref_start = ref_loc; out->query_start = query_loc; out->len = 0; out->score = 0;
    return 0;
}

/* =====================================================================================================
 * Comparators -- src/seed_filter.cu:47-108 ; repeat_masker_src/seed_filter.cu:45-135
 * ===================================================================================================== */
static int hsp_equal(const orc_segment* x, const orc_segment* y) { /* :47-52 (u32 arithmetic, H8) */
    return ((uint32_t)(x->ref_start - x->query_start) == (uint32_t)(y->ref_start - y->query_start)) &&
           (((x->ref_start >= y->ref_start) && ((uint32_t)(x->ref_start + x->len) <= (uint32_t)(y->ref_start + y->len))) ||
            ((y->ref_start >= x->ref_start) && ((uint32_t)(y->ref_start + y->len) <= (uint32_t)(x->ref_start + x->len))));
}
static int hsp_comp(const orc_segment* x, const orc_segment* y) { /* :54-80: (diag u32, ref_start, len, score desc) */
    uint32_t dx = x->ref_start - x->query_start, dy = y->ref_start - y->query_start;
    if (dx != dy) return dx < dy;
    if (x->ref_start != y->ref_start) return x->ref_start < y->ref_start;
    if (x->len != y->len) return x->len < y->len;
    return x->score > y->score;
}
static int hsp_comp_lastz(const orc_segment* x, const orc_segment* y) { /* :82-108 */
    if (x->query_start != y->query_start) return x->query_start < y->query_start;
    if (x->ref_start != y->ref_start) return x->ref_start < y->ref_start;
    if (x->len != y->len) return x->len < y->len;
    return x->score > y->score;
}
/* repeat masker */
static int rm_hsp_comp(const orc_segment* x, const orc_segment* y) { /* rm :109-135: (q, len desc, ref, score desc) */
    if (x->query_start != y->query_start) return x->query_start < y->query_start;
    if (x->len != y->len) return x->len > y->len;
    if (x->ref_start != y->ref_start) return x->ref_start < y->ref_start;
    return x->score > y->score;
}
static int rm_hsp_equal(const orc_segment* x, const orc_segment* y) { /* rm :80-85 */
    return x->ref_start == y->ref_start && x->query_start == y->query_start && x->len == y->len && x->score == y->score;
}
static int rm_diag_comp(const orc_segment* x, const orc_segment* y) { /* rm :52-78: (diag, ref, query, score desc) */
    uint32_t dx = x->ref_start - x->query_start, dy = y->ref_start - y->query_start;
    if (dx != dy) return dx < dy;
    if (x->ref_start != y->ref_start) return x->ref_start < y->ref_start;
    if (x->query_start != y->query_start) return x->query_start < y->query_start;
    return x->score > y->score;
}
static int rm_final_comp(const orc_segment* x, const orc_segment* y) { /* rm :87-107: (q, score desc, ref desc) */
    if (x->query_start != y->query_start) return x->query_start < y->query_start;
    if (x->score != y->score) return x->score > y->score;
    return x->ref_start > y->ref_start;
}

typedef int (*less_fn)(const orc_segment*, const orc_segment*);
typedef int (*eq_fn)(const orc_segment*, const orc_segment*);

/* stable merge sort == thrust::stable_sort semantics (:776, :782) */
static void stable_sort_seg(orc_segment* a, size_t n, less_fn less) {
    if (n < 2) return;
    orc_segment* tmp = (orc_segment*)malloc(n * sizeof(orc_segment));
    orc_segment *src = a, *dst = tmp;
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = less(&src[j], &src[i]) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        orc_segment* t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(orc_segment));
    free(tmp);
}
/* thrust::unique_copy on the device backends = head flags on ADJACENT INPUT pairs (hazard H3) */
static size_t unique_adjacent(const orc_segment* in, size_t n, orc_segment* out, eq_fn eq) {
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (i == 0 || !eq(&in[i - 1], &in[i])) out[m++] = in[i];
    return m;
}

int orc_max_hits_for_mem(uint64_t total_global_mem) { /* src/seed_filter.cu:832-841 */
    float global_mem_gb = (float)(total_global_mem / 1073741824.0f);
    return (int)(4194304 * global_mem_gb);
}

void orc_free(void* p) { free(p); }

/* =====================================================================================================
 * SeedAndFilter -- src/seed_filter.cu:682-828 (rm = 0) / repeat_masker_src/seed_filter.cu:724-876 (rm = 1)
 * ===================================================================================================== */
/* Stage trace (tests only: tests/test_oracle_rm_golden.py holds the oracle's intermediate lists against the reference kernels'
 * own text executed under SIMT emulation): when a trace is attached, every iteration appends its hit list as find_hits leaves it
 * (score 0 / -1 = the repeat masker's window flag), the records and done flags as find_hsps leaves them, and the compacted
 * list as compress_output leaves it (rc flip included).  The concatenation over the iterations is what ONE launch over all the
 * seeds gives: slot order and per-hit results do not depend on the iteration split. */
static void trace_append(orc_segment** dst, size_t* n, size_t* cap, const orc_segment* src, size_t k) {
    if (*n + k > *cap) { *cap = (*n + k) * 2 + 16; *dst = (orc_segment*)realloc(*dst, *cap * sizeof(orc_segment)); }
    memcpy(*dst + *n, src, k * sizeof(orc_segment));
    *n += k;
}
static size_t saf_impl_traced(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds_sz, int rm, int rev,
                       uint32_t win_start, uint32_t win_end, orc_segment** out_vec, orc_saf_stats* stats, orc_stage_trace* tr);
static size_t saf_impl(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds_sz, int rm, int rev,
                       uint32_t win_start, uint32_t win_end, orc_segment** out_vec, orc_saf_stats* stats) {
    return saf_impl_traced(p, seeds, num_seeds_sz, rm, rev, win_start, win_end, out_vec, stats, NULL);
}
static size_t saf_impl_traced(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds_sz, int rm, int rev,
                       uint32_t win_start, uint32_t win_end, orc_segment** out_vec, orc_saf_stats* stats, orc_stage_trace* tr) {
    uint32_t num_seeds = (uint32_t)num_seeds_sz;
    uint64_t num_hits = 0, total_anchors = 0, examined = 0, survivors = 0;
    orc_segment* result = (orc_segment*)malloc(sizeof(orc_segment));
    size_t result_n = 1, result_cap = 1;
    uint32_t iters_run = 0;
    memset(&result[0], 0, sizeof(orc_segment));

    /* find_num_hits :157-182 + inclusive_scan :714 (u32 prefix in src/, u64 in the repeat masker) */
    uint64_t* prefix = (uint64_t*)malloc(((size_t)num_seeds + 1) * sizeof(uint64_t));
    uint64_t run = 0;
    for (uint32_t i = 0; i < num_seeds; i++) {
        uint32_t seed = (uint32_t)(seeds[i] >> 32);
        uint32_t n = p->index_table[seed];
        if (seed > 0) n -= p->index_table[seed - 1];
        run += n;
        if (!rm) run &= 0xFFFFFFFFu;
        prefix[i] = run;
    }
    if (num_seeds > 0) num_hits = prefix[num_seeds - 1]; /* :716 */

    if (num_seeds > 0 && num_hits > 0) {
        /* iteration plan :718-745 */
        const uint64_t MAXH = (uint64_t)(uint32_t)p->max_hits; /* int compared/added as unsigned */
        uint32_t num_iter;
        uint64_t iter_hit_limit;
        if (num_hits < MAXH) { num_iter = 2; iter_hit_limit = num_hits; }
        else { num_iter = (uint32_t)(num_hits / MAXH + 2); iter_hit_limit = MAXH; }
        int64_t* limit_pos = (int64_t*)malloc((size_t)num_iter * sizeof(int64_t)); /* -1 = the wrapped index of H5 */
        for (uint32_t i = 0; i + 1 < num_iter; i++) {
            uint32_t lo = 0, hi = num_seeds; /* lower_bound :733 */
            while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (prefix[mid] < iter_hit_limit) lo = mid + 1; else hi = mid; }
            int64_t pos = (int64_t)lo - 1;
            limit_pos[i] = pos;
            /* :736 reads prefix[pos]; pos == -1 is the reference's out-of-bounds read (H5): treated as 0 hits */
            iter_hit_limit = (pos >= 0 ? prefix[pos] : 0) + MAXH;
            if (!rm) iter_hit_limit &= 0xFFFFFFFFu;
            if (iter_hit_limit > num_hits) iter_hit_limit = num_hits;
        }
        limit_pos[num_iter - 1] = (int64_t)num_seeds - 1; /* :741 */
        if (limit_pos[num_iter - 1] == limit_pos[num_iter - 2]) num_iter--; /* :743 */

        int64_t start_seed_index = 0;
        uint64_t start_hit_val = 0;
        for (uint32_t it = 0; it < num_iter; it++) { /* :756-793 */
            int64_t lp = limit_pos[it];
            uint64_t upto = lp >= 0 ? prefix[lp] : 0;
            int64_t iter_num_seeds = lp + 1 - start_seed_index;
            uint64_t iter_num_hits = upto - start_hit_val;
            iters_run++;
            if (iter_num_hits > 0 && iter_num_seeds > 0) {
                orc_segment* hsp = (orc_segment*)malloc((size_t)iter_num_hits * sizeof(orc_segment));
                /* find_hits :184-230: k-th bucket entry of seed s -> slot prefix_incl[s]-1-k-start_hit */
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 4096) num_threads(orc_team(p->num_threads, (uint64_t)(lp - start_seed_index + 1)))
#endif
                for (int64_t s = start_seed_index; s <= lp; s++) {
                    uint32_t seed = (uint32_t)(seeds[s] >> 32);
                    uint32_t qloc = (uint32_t)(seeds[s] & 0xFFFFFFFFu) + p->seed_size; /* :204 */
                    uint32_t e = p->index_table[seed], b = seed > 0 ? p->index_table[seed - 1] : 0;
                    for (uint32_t id = b; id < e; id++) {
                        uint64_t slot = prefix[s] - (id - b) - 1 - start_hit_val; /* :221 */
                        hsp[slot].ref_start = p->pos_table[id] + p->seed_size;    /* :220 */
                        hsp[slot].query_start = qloc;
                        hsp[slot].len = 0;
                        hsp[slot].score = 0;
                        if (rm && !(hsp[slot].ref_start >= win_start && hsp[slot].ref_start <= win_end))
                            hsp[slot].score = -1; /* rm :239-244 */
                    }
                }
                if (tr) { size_t c0 = tr->cap_hits; trace_append(&tr->hits, &tr->n_hits, &c0, hsp, (size_t)iter_num_hits); tr->cap_hits = c0; }
                /* find_hsps :232-652 ; done flags */
                uint8_t* done = (uint8_t*)malloc((size_t)iter_num_hits);
                uint64_t ex_local = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : ex_local) num_threads(orc_team(p->num_threads, iter_num_hits))
#endif
                for (int64_t h = 0; h < (int64_t)iter_num_hits; h++) {
                    orc_segment o;
                    uint64_t ex = 0;
                    if (rm && hsp[h].score < 0) { /* rm :305-333: both loops skipped -> total 0, extent 0 (:311) */
                        side_result z0 = {0, 0}, z1 = {0, 0};
                        done[h] = (uint8_t)finish_hit(&p->ext, hsp[h].ref_start, hsp[h].query_start, z0, z1, &o);
                    } else {
                        done[h] = (uint8_t)orc_extend_hit(&p->ext, hsp[h].ref_start, hsp[h].query_start, &o, &ex);
                    }
                    hsp[h] = o;
                    ex_local += ex;
                }
                examined += ex_local;
                if (tr) {
                    size_t c0 = tr->cap_ext, n0 = tr->n_ext;
                    trace_append(&tr->ext, &tr->n_ext, &c0, hsp, (size_t)iter_num_hits);
                    tr->cap_ext = c0;
                    tr->done = (uint8_t*)realloc(tr->done, tr->n_ext + 1);
                    memcpy(tr->done + n0, done, (size_t)iter_num_hits);
                }
                /* inclusive_scan(done) :769 + compress_output :654-680 = order-preserving compaction */
                size_t na = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : na) num_threads(orc_team(p->num_threads, iter_num_hits >> 4))
#endif
                for (int64_t h = 0; h < (int64_t)iter_num_hits; h++) na += done[h];
                survivors += na;
                if (na > 0) {
                    orc_segment* red = (orc_segment*)malloc(na * sizeof(orc_segment));
                    orc_segment* uni = (orc_segment*)malloc(na * sizeof(orc_segment));
                    size_t k = 0;
                    for (uint64_t h = 0; h < iter_num_hits; h++)
                        if (done[h]) {
                            red[k] = hsp[h];
                            if (rm && rev) /* rm :705-708 */
                                red[k].query_start = p->ext.ref_len - 1 - (red[k].query_start + red[k].len);
                            k++;
                        }
                    if (tr) { size_t c0 = tr->cap_reduced; trace_append(&tr->reduced, &tr->n_reduced, &c0, red, na); tr->cap_reduced = c0; }
                    size_t nu;
                    orc_segment* fin;
                    if (!rm) {
                        stable_sort_seg(red, na, hsp_comp);              /* :776 */
                        nu = unique_adjacent(red, na, uni, hsp_equal);   /* :778 */
                        stable_sort_seg(uni, nu, hsp_comp_lastz);        /* :782 */
                        fin = uni;
                    } else {
                        stable_sort_seg(red, na, rm_hsp_comp);           /* rm :819 */
                        nu = unique_adjacent(red, na, uni, rm_hsp_equal);/* rm :821 */
                        stable_sort_seg(uni, nu, rm_diag_comp);          /* rm :825 */
                        nu = unique_adjacent(uni, nu, red, hsp_equal);   /* rm :827 (hspDiagEqual == hspEqual) */
                        stable_sort_seg(red, nu, rm_final_comp);         /* rm :831 */
                        fin = red;
                    }
                    if (result_n + nu > result_cap) {
                        result_cap = (result_n + nu) * 2;
                        result = (orc_segment*)realloc(result, result_cap * sizeof(orc_segment));
                    }
                    memcpy(result + result_n, fin, nu * sizeof(orc_segment)); /* :811-822 */
                    result_n += nu;
                    total_anchors += nu;
                    free(red);
                    free(uni);
                }
                free(done);
                free(hsp);
            }
            start_seed_index = lp + 1; /* :791 */
            start_hit_val = upto;      /* :792 */
        }
        free(limit_pos);
    }
    free(prefix);
    /* header :806-809 ; rm :857-861 */
    if (!rm) {
        result[0].len = (uint32_t)total_anchors;
        result[0].score = (int32_t)(uint32_t)num_hits;
    } else {
        result[0].ref_start = (uint32_t)(num_hits & 0xFFFFFFFFu);
        result[0].query_start = (uint32_t)(num_hits >> 32);
        result[0].len = (uint32_t)(total_anchors & 0xFFFFFFFFu);
        result[0].score = (int32_t)(total_anchors >> 32);
    }
    if (stats) {
        stats->num_hits = num_hits;
        stats->num_survivors = survivors;
        stats->num_examined = examined;
        stats->num_iter = iters_run;
    }
    *out_vec = result;
    return result_n;
}

size_t orc_seed_and_filter(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, orc_segment** out_vec,
                           orc_saf_stats* stats) {
    return saf_impl(p, seeds, num_seeds, 0, 0, 0, 0, out_vec, stats);
}
size_t orc_seed_and_filter_rm(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, int rev,
                              uint32_t ref_start, uint32_t ref_end, orc_segment** out_vec, orc_saf_stats* stats) {
    return saf_impl(p, seeds, num_seeds, 1, rev, ref_start, ref_end, out_vec, stats);
}

/* the ordering chain alone on one dedup scope (tests/test_gpu_thrust_order.py): :776-782, or rm :819-831 */
size_t orc_order_hsps(const orc_segment* in, size_t n, int rm, orc_segment** out) {
    orc_segment* red = (orc_segment*)malloc((n + 1) * sizeof(orc_segment));
    orc_segment* uni = (orc_segment*)malloc((n + 1) * sizeof(orc_segment));
    memcpy(red, in, n * sizeof(orc_segment));
    size_t nu;
    orc_segment* fin;
    if (!rm) {
        stable_sort_seg(red, n, hsp_comp);
        nu = unique_adjacent(red, n, uni, hsp_equal);
        stable_sort_seg(uni, nu, hsp_comp_lastz);
        fin = uni;
        free(red);
    } else {
        stable_sort_seg(red, n, rm_hsp_comp);
        nu = unique_adjacent(red, n, uni, rm_hsp_equal);
        stable_sort_seg(uni, nu, rm_diag_comp);
        nu = unique_adjacent(uni, nu, red, hsp_equal);
        stable_sort_seg(red, nu, rm_final_comp);
        fin = red;
        free(uni);
    }
    *out = fin;
    return nu;
}

size_t orc_seed_and_filter_traced(const orc_saf_params* p, const uint64_t* seeds, size_t num_seeds, int rm, int rev, uint32_t ref_start,
                                  uint32_t ref_end, orc_segment** out_vec, orc_saf_stats* stats, orc_stage_trace* tr) {
    memset(tr, 0, sizeof(*tr));
    return saf_impl_traced(p, seeds, num_seeds, rm, rev, ref_start, ref_end, out_vec, stats, tr);
}
void orc_stage_trace_free(orc_stage_trace* tr) {
    free(tr->hits); free(tr->ext); free(tr->done); free(tr->reduced);
    memset(tr, 0, sizeof(*tr));
}

/* ---- repeat masker: coverage counting and run extraction (repeat_masker_src/seeder.cpp:57-58,153-188) ------------- */
size_t orc_rm_coverage_intervals(const orc_segment* hsps, size_t num_hsps, uint32_t block_len, uint32_t M, orc_interval** out) {
    uint8_t* int_count = (uint8_t*)calloc(block_len ? block_len : 1, 1); /* :57-58 */
    for (size_t h = 0; h < num_hsps; h++) {                               /* :155-159 (the sort at :154 cannot matter) */
        uint64_t b = hsps[h].query_start, e = (uint64_t)hsps[h].query_start + hsps[h].len;
        for (uint64_t j = b; j < e && j < block_len; j++) int_count[j]++;
    }
    size_t cap = 16, n = 0;
    orc_interval* res = (orc_interval*)malloc(cap * sizeof(orc_interval));
    int run = 0;
    uint32_t query_start = 0, len = 0;
    for (uint32_t i = 0; i < block_len; i++) { /* :168-186 */
        if (int_count[i] >= M) {
            if (run == 0) {
                run = 1;
                query_start = i;
            }
            len++;
        } else {
            if (run == 1) {
                run = 0;
                if (n == cap) {
                    cap *= 2;
                    res = (orc_interval*)realloc(res, cap * sizeof(orc_interval));
                }
                res[n].query_start = query_start;
                res[n].len = len;
                n++;
            }
            query_start = 0;
            len = 0;
        }
    }
    free(int_count);
    *out = res;
    return n;
}

/* ---- repeat masker: block / interval plan (repeat_masker_src/main.cpp:316-436) ----------------------------------- */
size_t orc_rm_plan(uint64_t seq_len, uint32_t seq_block_size, uint32_t lastz_interval_size, float prop_neigh_interval,
                   uint32_t seed_size, orc_rm_task** out) {
    if (seq_block_size == 1000000000u) seq_block_size -= seq_block_size % lastz_interval_size; /* :255-258 */
    uint32_t total_query_intervals = (uint32_t)ceil((float)seq_len / lastz_interval_size);  /* :316 */
    uint32_t num_neigh_interval = (uint32_t)ceil((float)prop_neigh_interval * total_query_intervals);
    uint32_t left_intervals = (uint32_t)ceil((float)(num_neigh_interval - 1) / 2); /* :319 */
    uint32_t right_intervals = num_neigh_interval - 1 - left_intervals;
    uint32_t left_overlap = left_intervals * lastz_interval_size;
    uint32_t right_overlap = right_intervals * lastz_interval_size;
    uint32_t max_interval_seq_len = left_overlap + lastz_interval_size + right_overlap;
    size_t cap = 64, n = 0;
    orc_rm_task* res = (orc_rm_task*)malloc(cap * sizeof(orc_rm_task));
    uint32_t block_index = 0;
    for (uint64_t l = 0; l < seq_len; l += seq_block_size) { /* :341 */
        uint64_t seq_block_start = (l < left_overlap) ? l : l - left_overlap;
        uint32_t seq_block_len;
        if (l + seq_block_size + right_overlap > seq_len)
            seq_block_len = (uint32_t)(seq_len - seq_block_start);
        else
            seq_block_len = (uint32_t)(l - seq_block_start + seq_block_size) + right_overlap;
        uint32_t start_pos = (uint32_t)(l - seq_block_start), end_pos;
        if (seq_block_len < seq_block_size)
            end_pos = start_pos + seq_block_len - (uint32_t)(l - seq_block_start) - seed_size;
        else
            end_pos = start_pos + seq_block_size - seed_size;
        while (start_pos < end_pos) { /* :367 */
            orc_rm_task t;
            t.block_index = block_index;
            t.block_start = seq_block_start;
            t.block_len = seq_block_len;
            t.start = start_pos;
            t.end = end_pos < start_pos + lastz_interval_size ? end_pos : start_pos + lastz_interval_size;
            int left_limit = t.start < left_overlap;
            int right_limit = (t.end + right_overlap) > seq_block_len;
            if (left_limit) {
                t.ref_start = 0;
                if (right_limit) t.ref_end = seq_block_len;
                else t.ref_end = max_interval_seq_len > seq_block_len ? seq_block_len : max_interval_seq_len;
            } else if (right_limit) {
                t.ref_end = seq_block_len;
                t.ref_start = seq_block_len < max_interval_seq_len ? 0 : seq_block_len - max_interval_seq_len;
            } else {
                t.ref_start = t.start - left_overlap;
                t.ref_end = t.end + right_overlap;
            }
            if (n == cap) {
                cap *= 2;
                res = (orc_rm_task*)realloc(res, cap * sizeof(orc_rm_task));
            }
            res[n++] = t;
            start_pos += lastz_interval_size;
        }
        block_index++;
    }
    *out = res;
    return n;
}
