// front.hip -- the front of a call: seed-vector upload, the device-side seeder (8f-1), the neighbourhood table build and the
// table-direct position probe with its chunk plans (probe.hip), and the on-device check that lets g_SeedAndFilter take that path.
#include "engine_internal.h"

namespace sa {

void upload_seeds(Slot* sl, const uint64_t* seeds, size_t n) {
    sl->seeds.ensure(std::max<size_t>(n, (size_t)g_max_seeds), "seed_offsets");
    if (n == 0) return;
    if (g_seed_upload == 1) {  // the runtime stages the pageable vector itself (chunked, synchronous for the caller)
        ProfScope p(sl, "h2d_seeds");
        check_memcpy(hipMemcpyAsync(sl->seeds.p, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream), "seed_offsets");
        return;
    }
    if (g_seed_upload == 2) {  // pin the caller's pages for the duration of the copy
        const uintptr_t lo = (uintptr_t)seeds & ~(uintptr_t)4095, hi = ((uintptr_t)(seeds + n) + 4095) & ~(uintptr_t)4095;
        if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
            {
                ProfScope p(sl, "h2d_seeds");
                check_memcpy(hipMemcpyAsync(sl->seeds.p, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream), "seed_offsets");
            }
            check_sync(sl->stream, "seed_offsets");
            hipHostUnregister((void*)lo);
            return;
        }
        (void)hipGetLastError();
    }
    if (sl->h_seeds_cap < n) {
        if (sl->h_seeds) hipHostFree(sl->h_seeds);
        sl->h_seeds_cap = std::max<size_t>(n, (size_t)g_max_seeds);
        if (hipHostMalloc((void**)&sl->h_seeds, sl->h_seeds_cap * sizeof(uint64_t)) != hipSuccess) {
            fprintf(stderr, "Error: hipHostMalloc for seed_offsets failed\n");
            exit(12);
        }
    }
    memcpy(sl->h_seeds, seeds, n * sizeof(uint64_t));  // reference copies the vector too (:694-697)
    ProfScope p(sl, "h2d_seeds");
    check_memcpy(hipMemcpyAsync(sl->seeds.p, sl->h_seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice, sl->stream),
                 "seed_offsets");  // :710
}

// device-side seeder (8f-1): fills sl->seeds for query positions [start,end); returns number of seed words
// nb > 0: also reports, for nb positions bpos[] in [start, end], the number of seed words emitted before them
uint32_t device_seeds(Slot* sl, const uint8_t* qcodes, uint32_t start, uint32_t end, int nb, const uint32_t* bpos, uint32_t* bseed) {
    for (int b = 0; b < nb; b++) bseed[b] = 0;
    if (end <= start) return 0;
    hipStream_t st = sl->stream;
    const uint32_t n = end - start;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    const uint32_t tmask = g_transition ? (sh.transition_mask & ((1u << sh.weight) - 1u)) : 0u;
    const uint32_t per = 1u + (uint32_t)__builtin_popcount(tmask);
    sl->flags.ensure(n, "seed flags");
    sl->flag_prefix.ensure((size_t)n + 1, "seed flag prefix");
    sl->scan_temp.ensure(scan_temp_bytes(n), "scan temp");
    {
        ProfScope p(sl, "seed_flags");
        launch_seed_flags(qcodes, start, end, sh, sl->flags.p, st);
    }
    {
        ProfScope p(sl, "seed_flag_scan");
        launch_exclusive_scan_u32(sl->flags.p, sl->flag_prefix.p, n, sl->scan_temp.p, st);
    }
    check_launch("seed flags");
    uint32_t nvalid = 0;
    check_memcpy(hipMemcpyAsync(&sl->h_cnt->pad, sl->flag_prefix.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "nvalid");
    for (int b = 0; b < nb; b++)
        check_memcpy(hipMemcpyAsync(&sl->h_bounds[b], sl->flag_prefix.p + (std::min(std::max(bpos[b], start), end) - start), sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, st), "chunk bounds");
    check_sync(st, "seed flags");
    nvalid = sl->h_cnt->pad;
    for (int b = 0; b < nb; b++) bseed[b] = sl->h_bounds[b] * per;
    const uint64_t nseeds = (uint64_t)nvalid * per;
    if (nseeds == 0) return 0;
    sl->seeds.ensure(std::max<size_t>((size_t)nseeds, (size_t)g_max_seeds), "seed_offsets");
    {
        ProfScope p(sl, "seed_emit");
        launch_seed_emit(qcodes, start, end, sh, g_transition, sl->flag_prefix.p, sl->seeds.p, st);
    }
    check_launch("seed emit");
    return (uint32_t)nseeds;
}

// ---- table-direct lookup (probe.hip) -----------------------------------------------------------------------------------
uint32_t seed_tmask() {
    return g_transition ? (g_shape.transition_mask & ((1u << g_shape.weight) - 1u)) : 0u;
}

__global__ void widen_u32_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}

void nbr_release(DevCtx* dc) {
    if (!dc->nbr_alias) dev_free(dc->nbr_pos, "nbr_pos");
    dc->nbr_start = nullptr;  // (its memory stays: keep_nbr_start)
    dc->nbr_pos = nullptr;
    dc->nbr_ctx = nullptr;  // (the memory stays with the arena)
    dc->nbr_alias = false;
    dc->nbr_total = 0;
    dc->nbr_state = 0;
}

// Builds (once per table and transition mask) the neighbourhood table of the device; false when it is not available:
// no table yet, a merged run longer than 2^32 entries, or not enough free HBM for (words per position) x pos_table.
bool ensure_nbr(DevCtx* dc) {
    if (!g_td || !dc->bucket_start || !dc->pos_table) return false;
    const uint32_t tmask = seed_tmask();
    std::lock_guard<std::mutex> lk(dc->nbr_mu);
    if (dc->nbr_state != 0 && dc->nbr_tmask == tmask) return dc->nbr_state == 1;
    check_set_device(dc->dev, "neighbourhood table");
    hipStream_t st = dc->admin;
    nbr_release(dc);
    dc->nbr_tmask = tmask;
    dc->nbr_state = -1;
    const uint32_t nkeys = dc->nkeys;
    dc->keep_nbr_start.ensure((size_t)nkeys + 1, "nbr_start");
    dc->nbr_start = dc->keep_nbr_start.p;
    uint64_t total = dc->num_index;
    if (tmask == 0) {  // one word per position: the runs ARE the buckets
        hipLaunchKernelGGL(widen_u32_kernel, dim3(4096), dim3(256), 0, st, dc->bucket_start, dc->nbr_start, nkeys + 1);
        check_launch("nbr widen");
        check_sync(st, "nbr widen");
    } else {
        // (scratch from the base of the table arena when it is mapped: it holds nothing until the fill below, and both are done by then)
        const size_t cnt_b = (((size_t)nkeys + 1) * sizeof(uint32_t) + 255) & ~(size_t)255, scan_b = scan_temp_bytes(nkeys);
        const bool in_arena = g_table_scratch_arena && g_arena_gb != 0 && dc->arena.vmm && arena_wait(dc->arena, cnt_b + scan_b);
        uint32_t* cnt = in_arena ? reinterpret_cast<uint32_t*>(dc->arena.base) : (uint32_t*)dev_malloc(cnt_b, "nbr counts");
        void* scan_tmp = in_arena ? (void*)(dc->arena.base + cnt_b) : dev_malloc(scan_b, "scan temp");
        check_memcpy(hipMemsetAsync(cnt + nkeys, 0, sizeof(uint32_t), st), "nbr overflow flag");  // cnt[nkeys] doubles as the flag
        launch_nbr_count(dc->bucket_start, nkeys, tmask, g_shape.weight, cnt, cnt + nkeys, st);
        launch_exclusive_scan_u64(cnt, dc->nbr_start, nkeys, scan_tmp, st);
        check_launch("nbr count/scan");
        uint32_t overflow = 0;
        check_memcpy(hipMemcpyAsync(&total, dc->nbr_start + nkeys, sizeof(uint64_t), hipMemcpyDeviceToHost, st), "nbr total");
        check_memcpy(hipMemcpyAsync(&overflow, cnt + nkeys, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "nbr overflow");
        check_sync(st, "nbr count");
        if (!in_arena) { dev_free(cnt, "nbr counts"); dev_free(scan_tmp, "scan temp"); }
        if (overflow) {
            dc->nbr_start = nullptr;
            return false;
        }
    }
    const bool dbg = opt_value("debug") != 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    // (what the arena already holds does not count against the free memory.  Read BEFORE the free figure: a chunk the background
    //  worker maps in between is then missing from both, never counted twice)
    const size_t have = arena_mapped(dc->arena);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = total_b = 0;  // (then only the plain lookup modes are tried)
    // keep room for the slots' work buffers: at human-scale hit density a fourteen-chunk call (1 G hits, option call_hits_max) holds ~6 GB of lists per slot
    size_t reserve = ((size_t)8 << 30) + ((size_t)4 << 30) * (size_t)SLOTS_PER_DEVICE;
    reserve -= std::min(reserve / 2, arena_mapped(dc->work_arena));  // (what the work arena holds IS part of that room)
    const size_t need_pos = (size_t)std::max<uint64_t>(total, 1) * sizeof(uint32_t);
    // context records (class filter): 32 bytes per entry; the two-stage fill wants num_index records of scratch behind them, which is
    // given up (one-stage fill) when only the table itself fits.  (+ 16 KB of slack: the filter requests two buffers ahead, so
    // its lanes read up to 3 x 64 entries past the last run)
    const size_t rec_b = (size_t)std::max<uint64_t>(total, 1) * sizeof(CtxRec) + 16384;
    const size_t scratch_b = (size_t)dc->num_index * sizeof(CtxRec);
    bool built = false;
    if (g_ctx && dc->ref2.base && rec_b + reserve <= free_b + have) {
        const bool two_stage = g_nbr_two_stage && tmask != 0 && rec_b + scratch_b + reserve <= free_b + have;
        const size_t need = rec_b + (two_stage ? scratch_b : 0);
        if (arena_wait(dc->arena, need)) {
            arena_settle(dc->arena, need);
            uint8_t* arena = dc->arena.base;
            dc->nbr_ctx = reinterpret_cast<CtxRec*>(arena);
            if (dbg) fprintf(stderr, "neighbourhood table: %.1f M entries, waited %.1f ms for %.1f GB of arena\n", total / 1e6, now() - t_a, need / 1e9);
            const double t_b = now();
            dc->nbr_left_skip = g_ctx_skip_seed ? g_seed_size : 0u;
            launch_nbr_fill_ctx(dc->bucket_start, dc->pos_table, nkeys, tmask, g_shape.weight, dc->nbr_start, dc->ref2.base, dc->ref2.stride,
                                g_seed_size, dc->nbr_left_skip, dc->nbr_ctx, two_stage ? reinterpret_cast<CtxRec*>(arena + rec_b) : nullptr, (uint32_t)dc->num_index, st);
            check_launch("nbr fill ctx");
            check_sync(st, "nbr fill ctx");
            if (dbg) fprintf(stderr, "neighbourhood table: context fill (%s) took %.1f ms\n", two_stage ? "two-stage" : "one-stage", now() - t_b);
            built = true;
        }
    }
    if (built) {
        // (nothing else to do)
    } else if (tmask == 0) {
        dc->nbr_pos = dc->pos_table;
        dc->nbr_alias = true;
    } else if ([&] {  // positions only: an arena kept from an earlier (smaller) block gives its memory back first
                   if (arena_mapped(dc->arena)) {
                       arena_trim(dc->arena, 0);
                       hipDeviceSynchronize();
                       if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
                   }
                   if (dbg) fprintf(stderr, "neighbourhood table: no room for the context table (%.1f GB + %.1f GB reserve), positions only need %.1f GB, free %.1f GB\n",
                                    rec_b / 1e9, reserve / 1e9, need_pos / 1e9, free_b / 1e9);
                   return need_pos + reserve <= free_b;
               }()) {
        dc->nbr_pos = (uint32_t*)dev_malloc(need_pos, "nbr_pos");
        launch_nbr_fill(dc->bucket_start, dc->pos_table, nkeys, tmask, g_shape.weight, dc->nbr_start, dc->nbr_pos, st);
        check_launch("nbr fill");
        check_sync(st, "nbr fill");
    } else {
        dc->nbr_start = nullptr;
        return false;
    }
    dc->nbr_total = total;
    dc->nbr_state = 1;
    return true;
}

// may this call take the table-direct path?  (the anchors are only ever read by the packed filter's TD fetch)
// With the context table resident the only anchor source of a table-direct call is the class filter, which needs the 2-bit shifted
// copies of BOTH query strands, all sixteen of a strand below `q2_limit` bytes (one 32-bit offset next to a scalar base: blocks of up
// to ~1 Gbp).  A call that cannot have them takes the general path.
bool q2_usable(const PackedBuf* q2_own, const PackedBuf* q2_other) {
    return q2_own && q2_own->base && q2_other && q2_other->base && q2_own->stride * q2_own->copies < g_q2_limit &&
           q2_other->stride * q2_other->copies < g_q2_limit;
}
bool td_eligible(DevCtx* dc, const PackedBuf* query4, const PackedBuf* q2_own, const PackedBuf* q2_other) {
    if (!(g_td && g_packed_filter && !g_count_examined && query4 && query4->base && dc->ref2.base && ensure_nbr(dc))) return false;
    return !dc->nbr_ctx || q2_usable(q2_own, q2_other);
}

// Position probe + chunk plans for query positions [bpos[0], bpos[K]) (chunk c = [bpos[c], bpos[c+1])); one D2H, one sync.
// Returns the number of seed words the reference would have been handed (0: nothing to do), or UINT32_MAX when the call must
// take the general path (or be halved): a chunk that needs more than TD_MAX_ITER reference iterations (num_hits >= 6 x MAX_HITS), more
// than MAX_SEGS iterations in the call, hit counts that wrap the reference's uint32 arithmetic, or more than 2^32 hits in the call
// (the filter indexes hits with 32 bits).
uint32_t td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, int K, const uint32_t* bpos, int rm, uint32_t* words_out) {
    hipStream_t st = sl->stream;
    const uint32_t start = bpos[0], end = bpos[K];
    const uint32_t tmask = seed_tmask();
    const uint32_t words = 1u + (uint32_t)__builtin_popcount(tmask);
    *words_out = words;
    if (end <= start) return 0;
    const uint32_t n = end - start;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    sl->td_toff.ensure(n, "probe scratch");
    sl->td_tcnt.ensure(n, "probe scratch");
    sl->td_rec.ensure((size_t)n + 1, "probe records");
    sl->td_chunk.ensure(TD_CHUNK_CAP, "probe chunk starts");
    sl->td_partial.ensure(probe_partial_bytes(n), "probe partials");
    // the chunks of a call tile [start, end) with wga_chunk-sized pieces (the last may be short or empty): the probe finds the
    // boundaries arithmetically.  Every caller builds its bounds that way; anything else is a programming error
    TdBounds tb;
    tb.nb = K + 1;
    tb.start = start;
    tb.end = end;
    tb.chunk = K > 1 ? g_wga_chunk : std::max(n, 1u);
    for (int c = 0; c <= K; c++)
        if (bpos[c] != (uint32_t)std::min<uint64_t>((uint64_t)start + (uint64_t)c * tb.chunk, end) || K + 1 > TD_MAX_BOUNDS || tb.chunk == 0) {
            fprintf(stderr, "Error: table-direct call with chunk bounds off the wga_chunk grid (chunk %d of %d)\n", c, K);
            exit(15);
        }
    {
        ProfScope p(sl, "seed_probe");
        launch_probe_lookup(qcodes, start, n, sh, dc->nbr_start, dc->nkeys, sl->td_toff.p, sl->td_tcnt.p, sl->td_partial.p, st);
    }
    // head-bit map for the class filter: sized for 128 hits per position; a denser call regrows it (it stays) and repeats the compaction
    const bool want_bits = dc->nbr_ctx != nullptr;
    if (want_bits) sl->td_bits.ensure(std::max<size_t>((size_t)n * 4 + 64, 1u << 16), "probe head bits");
    // the device-side state the later stages of the call expect zeroed is cleared by the probe's own clearing kernel
    sl->l2_counts.ensure((size_t)L2_NSUB * L2_CNT_STRIDE, "second-level counters");
    sl->chain_bucket_cnt.ensure(chain_num_buckets(), "chain buckets");
    ZeroList zl;
    zl.p[0] = reinterpret_cast<uint32_t*>(sl->d_cnt);  zl.n[0] = (uint32_t)(sizeof(Counters) / sizeof(uint32_t));
    zl.p[1] = sl->l2_counts.p;                         zl.n[1] = (uint32_t)(L2_NSUB * L2_CNT_STRIDE);
    zl.p[2] = sl->chain_bucket_cnt.p;                  zl.n[2] = chain_num_buckets();
    zl.p[3] = sl->d_seg_info;                          zl.n[3] = dedup_seg_info_words();
    for (bool first_pass = true;; first_pass = false) {
        {
            ProfScope p(sl, "probe_compact");
            launch_probe_compact(start, n, sl->td_toff.p, sl->td_tcnt.p, sl->td_partial.p, sl->d_td_bounds, sl->td_rec.p, sl->td_chunk.p, TD_CHUNK_CAP,
                                 want_bits ? sl->td_bits.p : nullptr, (uint32_t)std::min<size_t>(sl->td_bits.cap, 0xFFFFFFFFu), zl, tb, first_pass, st);
        }
        {
            ProfScope p(sl, "iteration_plan");
            launch_probe_plan(qcodes, sh, tmask, dc->bucket_start, sl->d_td_bounds, K, sl->td_rec.p, (uint64_t)(uint32_t)g_max_hits, rm ? 0 : 1, sl->d_td_plan,
                              sl->d_seg_end, st);
        }
        check_launch("probe");
        check_memcpy(hipMemcpyAsync(sl->h_td_plan, sl->d_td_plan, sizeof(TdPlan) * K, hipMemcpyDeviceToHost, st), "probe plan");
        check_sync(st, "probe plan");
        const uint64_t call_hits = sl->h_td_plan[K - 1].hit_base + sl->h_td_plan[K - 1].num_hits;
        const uint64_t need_words = ((call_hits + 63) >> 6) * 2 + 16;  // (the filter reads up to six 64-bit words past the last buffer)
        if (!want_bits || need_words <= sl->td_bits.cap || call_hits > 0xFFFFFFFFull) break;
        sl->td_bits.ensure((size_t)need_words + need_words / 4, "probe head bits(grow)");
        t_front_flags |= SA_PATH_HEAD_BITS_REGROWN;
    }
    uint64_t nvalid = 0, n_iter = 0;
    for (int c = 0; c < K; c++) {
        const TdPlan& tp = sl->h_td_plan[c];
        nvalid += tp.num_valid;
        // a chunk at or above MAX_HITS stays table-direct as long as the reference's greedy groups (:725-741) number <= TD_MAX_ITER and its
        // counts cannot wrap the reference's uint32 arithmetic (probe_plan_kernel); otherwise the general path plans it
        if (tp.n_iter == TD_PLAN_OVERFLOW) return 0xFFFFFFFFu;
        if (!rm && tp.num_hits > 0xFFFFFFFFull) return 0xFFFFFFFFu;
        n_iter += tp.n_iter;
    }
    if (n_iter > (uint64_t)MAX_SEGS) return 0xFFFFFFFFu;  // (more iterations than one extension batch resolves: the pass is halved, api_calls.hip)
    if (sl->h_td_plan[K - 1].hit_base + sl->h_td_plan[K - 1].num_hits >= 0xFFFFFFFFull) return 0xFFFFFFFFu;  // (hit indices are 32-bit, 2^32 - 1 is a sentinel)
    if (nvalid * words >= 0xFFFFFFFFull) return 0xFFFFFFFFu;
    return (uint32_t)(nvalid * words);
}

// ---------------------------------------------------------------------------------------------------------------------
// KEY-ORDERED calls (join.h): the front of a call whose hits are enumerated per seed key
// ---------------------------------------------------------------------------------------------------------------------
// A call goes key-ordered when that pays: the context table is resident, the seed keys are short enough for the two-level partition
// build (<= 24 bits), a query position collects enough hits that the per-position work (sort, field words: ~110 bytes of traffic per
// position) disappears behind them, and the call holds about one position per seed key or more -- below that a key's run is fetched
// for one position anyway and the streamed filter (extend.hip 1d) is as good (tools/micro/join_proto.hip: 139 / 174 / 215 / 257 G hits/s
// at 0.75 / 1.5 / 3 / 6 positions per key against 140-148 streamed).
bool join_wanted(DevCtx* dc, int K, uint32_t n_positions) {
    if (!g_key_order || !dc->nbr_ctx || dc->nbr_state != 1 || g_audit_cap > (1u << 27)) return false;
    if (2 * g_shape.weight > 24 || !table_partition_build_supported(g_shape.weight) || dc->nkeys != (1u << (2 * g_shape.weight))) return false;
    if (g_key_order == 2) return true;
    const double per_pos = (double)dc->nbr_total / (double)std::max<uint32_t>(dc->nkeys, 1);
    const uint64_t min_pos = g_key_order_min_pos > 0 ? (uint64_t)g_key_order_min_pos : (uint64_t)dc->nkeys;
    return K > 1 && per_pos >= 16.0 && (uint64_t)n_positions >= min_pos;
}

// Sort the call's positions by key, plan its chunks, lay out the entries and the query field words -- everything the filter (extend.hip
// 1e) and saf_core need; one D2H, one sync.  Returns the number of seed words the reference would have been handed (0: nothing to do),
// or UINT32_MAX when the call must be split (a chunk with num_hits >= MAX_HITS needs the general path's iteration plan).
uint32_t join_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, uint32_t qlen, int K, const uint32_t* bpos, const PackedBuf* q2_own, const PackedBuf* q2_other,
                    uint32_t* words_out) {
    hipStream_t st = sl->stream;
    const uint32_t start = bpos[0], end = bpos[K];
    const uint32_t tmask = seed_tmask();
    const uint32_t words = 1u + (uint32_t)__builtin_popcount(tmask);
    *words_out = words;
    if (end <= start) return 0;
    const uint32_t n = end - start;
    const uint32_t chunk = K > 1 ? g_wga_chunk : std::max(n, 1u);
    for (int c = 0; c <= K; c++)
        if (bpos[c] != (uint32_t)std::min<uint64_t>((uint64_t)start + (uint64_t)c * chunk, end) || K > SA_MAX_CHUNKS) {
            fprintf(stderr, "Error: key-ordered call with chunk bounds off the wga_chunk grid (chunk %d of %d)\n", c, K);
            exit(15);
        }
    sl->jq_chunk = chunk;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    const uint32_t nkeys = dc->nkeys;
    // ---- 1. the positions of the call sorted by key: the table build's radix partition on the query strand (table.hip) ----
    const size_t pw = table_partition_part_start_words();
    sl->jq_keys.ensure(n, "join keys");
    sl->jq_pairs.ensure((size_t)4 * n, "join partition pairs");
    sl->jq_misc.ensure(3 * pw + 1024 + 16, "join partition scratch");
    sl->jq_start.ensure((size_t)nkeys + 1, "join key starts");
    sl->jq_pos.ensure(n, "join positions");
    const size_t ent_cap = (size_t)std::min<uint32_t>(nkeys, n) + n / JOIN_CMAX + 64;
    sl->jq_scan.ensure(std::max(scan_temp_bytes(std::max<size_t>(pw, (size_t)1 << 18)), scan_temp_bytes(ent_cap)), "join scan temp");
    sl->jq_stats.ensure((size_t)2 * SA_MAX_CHUNKS, "join chunk statistics");  // hits (u64) | valid, last (u32 each)
    sl->jq_ent.ensure(ent_cap, "join entries");
    sl->jq_ent_nt.ensure(ent_cap, "join entry run lengths");
    sl->jq_vstart.ensure(ent_cap + 1, "join entry starts");
    sl->jq_qx.ensure((size_t)n * JOIN_QX_DW + 64, "join query field words");
    uint32_t* coarse = sl->jq_misc.p;
    uint32_t* part_start = coarse + pw;
    uint32_t* cursor = part_start + pw;
    uint8_t* part_unsorted = reinterpret_cast<uint8_t*>(cursor + pw);
    uint32_t* key_a = sl->jq_pairs.p;
    {
        ProfScope p(sl, "join_sort");
        check_memcpy(hipMemsetAsync(coarse, 0, pw * sizeof(uint32_t), st), "join coarse histogram");
        launch_table_keys(qcodes, n, start, 1u, sh, sl->jq_keys.p, coarse, st);
        launch_exclusive_scan_u32(coarse, part_start, pw - 1, sl->jq_scan.p, st);
        // (the number of valid positions is only known on the device: the second pass runs over all n slots, the unused ones invalid)
        check_memcpy(hipMemsetAsync(key_a, 0xFF, (size_t)n * sizeof(uint32_t), st), "join partition pairs");
        launch_table_partition_build(sl->jq_keys.p, n, start, 1u, sh.weight, part_start, n, cursor, key_a, key_a + n, key_a + 2 * (size_t)n, key_a + 3 * (size_t)n,
                                     part_unsorted, sl->jq_start.p, sl->jq_pos.p, nullptr, sl->jq_scan.p, nullptr, st);
    }
    // ---- 2. chunk statistics + the reference's iteration split per chunk ----
    unsigned long long* d_hits = sl->jq_stats.p;
    uint32_t* d_valid = reinterpret_cast<uint32_t*>(sl->jq_stats.p + SA_MAX_CHUNKS);
    uint32_t* d_last = d_valid + SA_MAX_CHUNKS;
    sl->l2_counts.ensure((size_t)L2_NSUB * L2_CNT_STRIDE, "second-level counters");
    sl->chain_bucket_cnt.ensure(chain_num_buckets(), "chain buckets");
    {
        ProfScope p(sl, "join_plan");
        check_memcpy(hipMemsetAsync(sl->jq_stats.p, 0, (size_t)2 * SA_MAX_CHUNKS * sizeof(unsigned long long), st), "join chunk statistics");
        check_memcpy(hipMemsetAsync(sl->d_jhead, 0, sizeof(JoinHead), st), "join head");
        launch_join_stats(sl->jq_start.p, sl->jq_pos.p, dc->nbr_start, nkeys, start, chunk, K, d_hits, d_valid, d_last, st);
        launch_join_plan(qcodes, sh, tmask, dc->bucket_start, d_hits, d_valid, d_last, K, sl->d_jplan, sl->d_seg_end, sl->d_jhead, st);
        // the device-side state the later stages of the call expect zeroed
        ZeroList zl;
        zl.p[0] = reinterpret_cast<uint32_t*>(sl->d_cnt);  zl.n[0] = (uint32_t)(sizeof(Counters) / sizeof(uint32_t));
        zl.p[1] = sl->l2_counts.p;                         zl.n[1] = (uint32_t)(L2_NSUB * L2_CNT_STRIDE);
        zl.p[2] = sl->chain_bucket_cnt.p;                  zl.n[2] = chain_num_buckets();
        zl.p[3] = sl->d_seg_info;                          zl.n[3] = dedup_seg_info_words();
        launch_call_clear(zl, st);
    }
    check_launch("join sort/plan");
    check_memcpy(hipMemcpyAsync(sl->h_jplan, sl->d_jplan, sizeof(JoinChunk) * K, hipMemcpyDeviceToHost, st), "join plan");
    // ---- 3. entries by class, their starts in the virtual record space, the query field words (queued behind the plan: the host's
    //         decision below only ever drops the call, it never changes these) ----
    {
        ProfScope p(sl, "join_entries");
        check_memcpy(hipMemsetAsync(sl->jq_ent_nt.p, 0, ent_cap * sizeof(uint32_t), st), "join entry run lengths");
        launch_join_entries(sl->jq_start.p, dc->nbr_start, nkeys, sl->d_jhead, sl->jq_ent.p, sl->jq_ent_nt.p, (uint32_t)std::min<size_t>(ent_cap, 0xFFFFFFFFu), st);
        launch_exclusive_scan_u64(sl->jq_ent_nt.p, reinterpret_cast<uint64_t*>(sl->jq_vstart.p), ent_cap, sl->jq_scan.p, st);
        launch_join_finish(sl->d_jhead, sl->jq_vstart.p, st);
    }
    {
        ProfScope p(sl, "join_qx");
        launch_join_qx(sl->jq_start.p, nkeys, sl->jq_pos.p, q2_own->base, q2_other->base, qlen, g_seed_size, dc->nbr_left_skip, sl->jq_qx.p, st);
    }
    check_launch("join entries");
    check_sync(st, "join plan");
    uint64_t nvalid = 0, hit_base = 0;
    for (int c = 0; c < K; c++) {
        const JoinChunk& jc = sl->h_jplan[c];
        TdPlan& tp = sl->h_td_plan[c];
        tp.hit_base = hit_base;
        tp.num_hits = jc.hits;
        tp.n_iter = jc.hits ? 2u : 0u;  // (hit offsets only size the lists of a key-ordered call: its segments are resolved per hit, extend.hip seg_of)
        tp.upto[0] = hit_base;
        tp.upto[1] = hit_base + jc.hits;
        tp.num_valid = jc.valid;
        tp.m_lo = tp.m_hi = 0;
        hit_base += jc.hits;
        nvalid += jc.valid;
        if (jc.hits >= (uint64_t)(uint32_t)g_max_hits || jc.hits > 0xFFFFFFFFull) return 0xFFFFFFFFu;
    }
    if (nvalid * words >= 0xFFFFFFFFull) return 0xFFFFFFFFu;
    // the call was sized by the hits a chunk of RANDOM sequence collects (sa_get_chunks_per_call); repeat-rich sequence collects a
    // multiple (lumpy stand-in: 3.3 x), and the call's lists are sized by its hits: well above the bound it is halved like any
    // other pass that cannot hold its chunks
    if (K > 1 && hit_base > (uint64_t)g_key_order_hits + (uint64_t)g_key_order_hits / 2) return 0xFFFFFFFFu;
    t_front_flags |= SA_PATH_KEY_ORDERED;
    return (uint32_t)(nvalid * words);
}

// DROP-IN FAST PATH.  g_SeedAndFilter hands the engine a host seed vector (src/seeder.cpp:57-78).  When that vector is exactly
// what the device seeder would emit for the positions it spans -- checked on the device, one lane per position group, plus the
// probe's own count of valid positions -- the call is the same as sa_seed_and_filter_range(first, last + 1) and takes the
// table-direct path (one probe per position, record-stream filter).  Anything else keeps the reference-shaped path on the
// uploaded words: hand-made vectors, and the reference's minus-strand arena when the query holds other IUPAC letters (H14).
// Returns the seed-word count of the table-direct call, or UINT32_MAX.
uint32_t dropin_td_front(DevCtx* dc, Slot* sl, const uint8_t* qcodes, uint32_t qlen, const uint64_t* host_seeds, size_t n,
                                const PackedBuf* q4, const PackedBuf* q2_own, const PackedBuf* q2_other, int rm, uint32_t* first_out,
                                uint32_t* end_out, uint32_t* words_out) {
    if (n == 0 || !td_eligible(dc, q4, q2_own, q2_other)) return 0xFFFFFFFFu;
    const uint32_t tmask = seed_tmask();
    const uint32_t per = 1u + (uint32_t)__builtin_popcount(tmask);
    if (n % per != 0 || n > 0xFFFFFFFFull) return 0xFFFFFFFFu;
    const uint32_t first = (uint32_t)host_seeds[0], last = (uint32_t)host_seeds[n - 1];
    if (last < first || (uint64_t)last + g_seed_size > qlen) return 0xFFFFFFFFu;
    hipStream_t st = sl->stream;
    SeedShape sh = g_shape;
    sh.span = (int)g_seed_size;
    *sl->h_verify = 0;
    check_memcpy(hipMemsetAsync(sl->d_verify, 0xFF, sizeof(uint32_t), st), "seed verify flag");
    {
        ProfScope p(sl, "seed_verify");
        launch_seed_verify(sl->seeds.p, (uint32_t)(n / per), per, qcodes, qlen, sh, tmask, sl->d_verify, st);
    }
    check_memcpy(hipMemcpyAsync(sl->h_verify, sl->d_verify, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "seed verify flag");
    const uint32_t bp[2] = {first, last + 1u};
    const uint32_t ns = td_front(dc, sl, qcodes, 1, bp, rm, words_out);  // (synchronises the stream: the flag has arrived)
    if (ns == 0xFFFFFFFFu || *sl->h_verify == 0u || (uint64_t)ns != (uint64_t)n) return 0xFFFFFFFFu;
    *first_out = first;
    *end_out = last + 1u;
    return ns;
}

void set_query2(CoreArgs& ca, DevCtx* dc, uint32_t buffer, int rev) {  // plain calls: strand copies of query buffer `buffer`
    ca.q2_own = rev ? &dc->query2_rc[buffer] : &dc->query2[buffer];
    ca.q2_other = rev ? &dc->query2[buffer] : &dc->query2_rc[buffer];
    ca.q_present = dc->query_present[buffer];
}
void set_query2_rm(CoreArgs& ca, DevCtx* dc, int rev) {  // repeat masker: the query is the target
    ca.q2_own = rev ? &dc->refq2_rc : &dc->refq2;
    ca.q2_other = rev ? &dc->refq2 : &dc->refq2_rc;
    ca.q_present = dc->ref_present;
}

}  // namespace sa
