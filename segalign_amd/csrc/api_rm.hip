// api_rm.hip -- C-ABI, repeat masker (repeat_masker_src/seed_filter.cu, repeat_masker_src/seeder.cpp:28-195) and its device-side
// coverage post-processing (8f-4).
#include "engine_internal.h"

using namespace sa;

extern "C" {

// ---- repeat masker --------------------------------------------------------------------------------------------------
void sa_rm_send_query_write_request(void) {  // rm :951-961
    require_init("SendQueryWriteRequest");
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendQueryWriteRequest");
        dc->ref_rc.create(dc->ref.len, "seq_rc", dc->admin);
        launch_rev_comp_codes(dc->ref.codes, dc->ref_rc.codes, dc->ref.len, dc->admin);
        dc->ref4.create(dc->ref.codes, dc->ref.len, 4, "seq 4-bit", dc->admin);
        dc->ref4_rc.create(dc->ref_rc.codes, dc->ref.len, 4, "seq_rc 4-bit", dc->admin);
        dc->refq2.create_q2(dc->ref.codes, dc->ref.len, "seq 2-bit shifted", dc->admin);
        dc->refq2_rc.create_q2(dc->ref_rc.codes, dc->ref.len, "seq_rc 2-bit shifted", dc->admin);
        check_launch("rev_comp_string");
        check_sync(dc->admin, "SendQueryWriteRequest");
    }
}
void sa_rm_clear_query(void) {  // rm :964-972
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "ClearQuery");
        dc->ref_rc.release("d_seq_rc");
        dc->ref4.release("d_seq 4-bit");
        dc->ref4_rc.release("d_seq_rc 4-bit");
        dc->refq2.release("d_seq 2-bit shifted");
        dc->refq2_rc.release("d_seq_rc 2-bit shifted");
    }
}
size_t sa_rm_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t ref_start, uint32_t ref_end,
                             sa_segment_pair** out) {  // rm :724-876
    require_init("SeedAndFilter");
    if ((int64_t)num_seeds > g_max_seeds) {
        printf("MAX_SEEDS exceeded\n");
        fflush(stdout);
        fprintf(stderr, "Assertion `num_seeds <= MAX_SEEDS' failed.\n");
        abort();
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    upload_seeds(sl, seeds, num_seeds);
    CoreArgs ca = {rev ? dc->ref_rc.codes : dc->ref.codes, dc->ref.len, 1, rev ? 1 : 0, ref_start, ref_end, 0, 0, nullptr, 0,
                   rev ? &dc->ref4_rc : &dc->ref4};  // rm :805-810
    set_query2_rm(ca, dc, rev);
    uint32_t lo = 0, hi = 0, words = 0;
    if (dropin_td_front(dc, sl, ca.query, ca.query_len, seeds, num_seeds, ca.query4, ca.q2_own, ca.q2_other, 1, &lo, &hi, &words) != 0xFFFFFFFFu) {
        ca.td = 1;
        ca.td_words = words;
        ca.q_lo = lo;
        ca.q_hi = hi;
    }
    size_t n = saf_core(dc, sl, (uint32_t)num_seeds, ca, out);
    release_slot(sl);
    return n;
}

// ---- repeat-masker post-processing on the device (8f-4) -------------------------------------------------------------
namespace sa {

constexpr uint32_t COV_TILE = 1u << 26;  // positions per scan tile (bounds the scratch at 5 x 256 MiB)

// make the slot's difference array cover `block_len` positions, all zero, and reset the touched range
static void coverage_begin(Slot* sl, uint32_t block_len) {
    hipStream_t st = sl->stream;
    const size_t need = (size_t)block_len + 1;
    if (sl->cov_diff.cap < need) {  // a fresh allocation is cleared once; afterwards only the touched range is re-cleared
        sl->cov_diff.ensure(need, "coverage diff");
        check_memcpy(hipMemsetAsync(sl->cov_diff.p, 0, sl->cov_diff.cap * sizeof(uint32_t), st), "coverage diff");
    }
    launch_coverage_range_reset(sl->d_cov_range, st);
    check_launch("coverage begin");
}

// runs with (depth mod 256) >= M over the touched range -> malloc-ed sa_interval list; leaves the array zeroed again
static size_t coverage_finish(Slot* sl, uint32_t block_len, uint32_t M, uint64_t num_hsps, sa_interval** out) {
    hipStream_t st = sl->stream;
    *out = nullptr;
    check_memcpy(hipMemcpyAsync(sl->h_cov, sl->d_cov_range, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage range");
    check_sync(st, "coverage range");
    const uint32_t lo = sl->h_cov[0], hi = sl->h_cov[1];
    if (lo == 0xFFFFFFFFu) return 0;  // nothing was counted: depth 0 everywhere (M == 0: one unterminated run, see below)
    size_t n_out = 0;
    // M == 0 makes every position of the block "covered": one run that reaches the end of the block and is therefore
    // never written (seeder.cpp:168-186 has no flush after the loop)
    if (M > 0 && M <= 255) {
        // every run boundary sits on a position with a non-zero difference, so there are at most 2 per HSP
        const uint64_t cap64 = std::min<uint64_t>(2 * num_hsps + 2, (uint64_t)block_len + 1);
        const uint32_t cap = (uint32_t)std::min<uint64_t>(cap64, 0x7FFFFFFFull);
        sl->cov_pairs.ensure((size_t)cap * 2, "coverage intervals");
        const uint32_t span = hi - lo + 1;  // depth is 0 before lo and from hi on
        const uint32_t tile_cap = std::min(span, COV_TILE);
        sl->cov_pre.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->cov_is_start.ensure(tile_cap, "coverage scan");
        sl->cov_is_end.ensure(tile_cap, "coverage scan");
        sl->cov_sidx.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->cov_eidx.ensure((size_t)tile_cap + 1, "coverage scan");
        sl->scan_temp.ensure(scan_temp_bytes(tile_cap), "scan temp");
        uint32_t depth = 0, nstart = 0, nend = 0;
        for (uint64_t off = 0; off < span; off += COV_TILE) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(COV_TILE, span - off);
            const uint32_t pos0 = lo + (uint32_t)off;
            const uint32_t* d = sl->cov_diff.p + pos0;
            ProfScope p(sl, "coverage_runs");
            launch_exclusive_scan_u32(d, sl->cov_pre.p, n, sl->scan_temp.p, st);
            launch_coverage_flags(d, sl->cov_pre.p, n, depth, M, sl->cov_is_start.p, sl->cov_is_end.p, st);
            launch_exclusive_scan_u32(sl->cov_is_start.p, sl->cov_sidx.p, n, sl->scan_temp.p, st);
            launch_exclusive_scan_u32(sl->cov_is_end.p, sl->cov_eidx.p, n, sl->scan_temp.p, st);
            launch_coverage_emit(sl->cov_is_start.p, sl->cov_is_end.p, sl->cov_sidx.p, sl->cov_eidx.p, n, pos0, nstart, nend, cap,
                                 sl->cov_pairs.p, st);
            check_launch("coverage runs");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[2], sl->cov_pre.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[3], sl->cov_sidx.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_memcpy(hipMemcpyAsync(&sl->h_cov[4], sl->cov_eidx.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "coverage totals");
            check_sync(st, "coverage totals");
            depth += sl->h_cov[2];
            nstart += sl->h_cov[3];
            nend += sl->h_cov[4];
        }
        // the depth returns to 0 at `hi`, so every run that started has ended (nstart == nend)
        n_out = std::min(nend, cap);
        if (n_out > 0) {
            launch_coverage_finish(sl->cov_pairs.p, (uint32_t)n_out, st);
            check_launch("coverage finish");
            sa_interval* res = (sa_interval*)malloc(n_out * sizeof(sa_interval));
            if (!res) {
                fprintf(stderr, "Error: malloc for the interval list failed\n");
                exit(12);
            }
            check_memcpy(hipMemcpyAsync(res, sl->cov_pairs.p, n_out * sizeof(sa_interval), hipMemcpyDeviceToHost, st), "intervals");
            check_sync(st, "intervals");
            // A run is written when the first uncovered position AFTER it is seen, and the reference's loop ends at block_len - 1
            // (seeder.cpp:166-186, no flush behind it): a run that covers the block's last position is never written.  Real HSPs
            // cannot produce one (their last base is not counted); HSPs handed to sa_rm_coverage_intervals can
            // (tests/golden/rm_host_golden.json)
            if ((uint64_t)res[n_out - 1].query_start + res[n_out - 1].len >= block_len) n_out--;
            if (n_out == 0) { free(res); res = nullptr; }
            *out = res;
        }
    }
    // leave the difference array zeroed for the next interval
    check_memcpy(hipMemsetAsync(sl->cov_diff.p + lo, 0, ((size_t)hi - lo + 1) * sizeof(uint32_t), st), "coverage clear");
    check_sync(st, "coverage clear");
    return n_out;
}

}  // namespace sa

size_t sa_rm_mask_interval(uint32_t start_pos, uint32_t end_pos, uint32_t ref_start, uint32_t ref_end, int strands, uint32_t M,
                           sa_interval** out, uint64_t* totals) {  // repeat_masker_src/seeder.cpp:28-195
    require_init("MaskInterval");
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    const uint32_t block_len = dc->ref.len;
    if (!dc->ref_rc.codes && (strands & SA_STRAND_MINUS)) {
        fprintf(stderr, "Error: MaskInterval on the minus strand before SendQueryWriteRequest\n");
        exit(1);
    }
    coverage_begin(sl, block_len);
    // a seed window must lie inside the block (the reference reads its host arena past the block end for the last
    // positions of the minus strand of a block's first interval: undefined there, no seed here)
    const uint32_t lim = block_len >= g_seed_size ? block_len - g_seed_size + 1 : 0;
    const uint32_t end_pos_rc = block_len - 1 - start_pos;  // seeder.cpp:46-47
    uint64_t tot_seeds = 0, tot_hits = 0, tot_hsps = 0, tot_ex = 0, tot_exf = 0, tot_cand = 0;
    int path = 0;
    // The reference walks the plus-strand chunks and derives a minus-strand chunk from each (:73-150).  Coverage counting is
    // order independent and every chunk keeps its own iteration plan and dedup scope, so the chunks of a strand are grouped:
    // consecutive chunks that tile a range go through ONE table-direct pass (up to min(chunks_per_call, 20) of them), the rest --
    // the minus-strand chunk of a short last plus chunk overlaps its neighbour (:118-119) -- go on their own.
    const int rm_group = std::min(g_chunks_per_call, 20);  // (a self-alignment is hit-dense: twenty chunks are ~0.5 G hits)
    struct Range { uint32_t s0, s1; };
    for (int rev = 0; rev < 2; rev++) {
        if (!(strands & (rev ? SA_STRAND_MINUS : SA_STRAND_PLUS))) continue;
        std::vector<Range> rs;
        for (uint64_t i = start_pos; i < end_pos; i += g_wga_chunk) {  // :73
            const uint32_t start = (uint32_t)i;
            const uint32_t end = (uint32_t)std::min<uint64_t>(i + g_wga_chunk, end_pos);  // :76-77
            uint32_t s0 = start, s1 = end;
            if (rev) {  // :118-119: the minus-strand chunk is derived from the plus-strand chunk END
                s0 = block_len - 1 - end;
                s1 = (uint32_t)std::min<uint64_t>((uint64_t)s0 + g_wga_chunk, end_pos_rc);
            }
            if (s1 > lim) s1 = lim;
            if (s1 > s0) rs.push_back({s0, s1});
        }
        if (rev) std::reverse(rs.begin(), rs.end());  // ascending positions
        const uint8_t* q = rev ? dc->ref_rc.codes : dc->ref.codes;
        const PackedBuf* q4 = rev ? &dc->ref4_rc : &dc->ref4;
        const bool td_ok = td_eligible(dc, q4, rev ? &dc->refq2_rc : &dc->refq2, rev ? &dc->refq2 : &dc->refq2_rc);
        size_t a = 0;
        while (a < rs.size()) {
            size_t b = a + 1;
            while (td_ok && b < rs.size() && (int)(b - a) < rm_group && rs[b].s0 == rs[b - 1].s1) b++;
            int Kc = (int)(b - a);
            uint32_t bp[SA_MAX_CHUNKS + 1];
            for (int c = 0; c < Kc; c++) bp[c] = rs[a + c].s0;
            bp[Kc] = rs[b - 1].s1;
            uint32_t ns = 0xFFFFFFFFu, words = 0;
            if (td_ok) ns = td_front(dc, sl, q, Kc, bp, 1, &words);
            if (ns == 0xFFFFFFFFu && Kc > 1) {  // one of the chunks needs the general path (MAX_HITS): one chunk per call
                b = a + 1;
                Kc = 1;
                bp[1] = rs[a].s1;
                ns = td_front(dc, sl, q, 1, bp, 1, &words);
            }
            const bool td = ns != 0xFFFFFFFFu;
            if (!td) ns = device_seeds(sl, q, rs[a].s0, rs[a].s1);
            if (ns != 0) {  // :103,140
                CoreArgs ca = {q, block_len, 1, rev, ref_start, ref_end, bp[0], bp[Kc], sl->cov_diff.p, block_len + 1, q4};
                set_query2_rm(ca, dc, rev);
                ca.td = td ? 1 : 0;
                ca.td_words = words;
                if (Kc > 1) {
                    ca.nchunks = Kc;
                    for (int c = 0; c <= Kc; c++) ca.seed_bound[c] = 0u;  // (a table-direct call derives them from its chunk plans)
                }
                saf_core(dc, sl, ns, ca, nullptr);
                tot_seeds += ns;
                tot_hits += t_stats.num_hits;
                tot_hsps += t_stats.num_anchors;
                tot_ex += t_stats.num_examined;
                tot_exf += t_stats.num_examined_filter;
                tot_cand += t_stats.num_candidates;
                path = t_stats.lookup_path;
            }
            a = b;
        }
    }
    const size_t n = coverage_finish(sl, block_len, M, tot_hsps, out);
    prof_flush(sl);
    release_slot(sl);
    if (totals) { totals[0] = tot_seeds; totals[1] = tot_hits; totals[2] = tot_hsps; }
    // sa_get_last_call_stats after an interval call: the sums over its SeedAndFilter passes
    t_stats.num_seeds = tot_seeds;
    t_stats.num_hits = tot_hits;
    t_stats.num_anchors = tot_hsps;
    t_stats.num_examined = tot_ex;
    t_stats.num_examined_filter = tot_exf;
    t_stats.num_candidates = tot_cand;
    t_stats.lookup_path = path;
    return n;
}

size_t sa_rm_coverage_intervals(const sa_segment_pair* hsps, size_t num_hsps, uint32_t block_len, uint32_t M, sa_interval** out) {
    require_init("CoverageIntervals");  // repeat_masker_src/seeder.cpp:153-188
    Slot* sl = acquire_slot();
    hipStream_t st = sl->stream;
    coverage_begin(sl, block_len);
    const size_t BATCH = 1u << 24;
    for (size_t off = 0; off < num_hsps; off += BATCH) {
        const size_t n = std::min(BATCH, num_hsps - off);
        sl->out16.ensure(n, "out16");
        check_memcpy(hipMemcpyAsync(sl->out16.p, hsps + off, n * sizeof(sa_segment_pair), hipMemcpyHostToDevice, st), "hsps h2d");
        launch_coverage_add_pairs(reinterpret_cast<const SegPair16*>(sl->out16.p), (uint32_t)n, sl->cov_diff.p, block_len + 1,
                                  sl->d_cov_range, st);
        check_launch("coverage add");
        check_sync(st, "coverage add");
    }
    const size_t n = coverage_finish(sl, block_len, M, num_hsps, out);
    prof_flush(sl);
    release_slot(sl);
    return n;
}

void sa_free_intervals(sa_interval* p) { free(p); }

}  // extern "C"
