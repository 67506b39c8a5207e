// core.hip -- SeedAndFilter core (src/seed_filter.cu:682-828; rm :724-876): seeds already on the device (or a table-direct front
// already run): iteration plan, filter levels, exact extension, entropy, ordering + de-duplication, return vectors.
#include "engine_internal.h"

namespace sa {

size_t saf_core(DevCtx* dc, Slot* sl, uint32_t num_seeds, const CoreArgs& ca, sa_segment_pair** out) {
    hipStream_t st = sl->stream;
    memset(&t_stats, 0, sizeof(t_stats));
    t_stats.num_seeds = num_seeds;
    t_stats.device = dc->index;

    uint64_t num_hits = 0;
    uint32_t n_final = 0;
    uint32_t survivors = 0;
    uint64_t n_cand_total = 0, n_ent_total = 0, n_fwd_total = 0;
    uint32_t path_flags = t_front_flags;  // SA_PATH_* of this call (the front of the call may have set some already)
    t_front_flags = 0;
    // chunks of the seed vector (one for an ordinary call)
    const int K = ca.nchunks > 1 ? ca.nchunks : 1;
    uint32_t sbound[SA_MAX_CHUNKS + 1] = {0, num_seeds};
    if (ca.nchunks > 1) memcpy(sbound, ca.seed_bound, sizeof(uint32_t) * (K + 1));
    uint64_t chunk_hits[SA_MAX_CHUNKS] = {0};
    uint32_t chunk_first_seg[SA_MAX_CHUNKS + 1] = {0};
    bool have_seg = false;  // sl->h_seg holds the segment of every final record

    if (num_seeds > 0) {
        // flat list of reference iterations ("segments") over all chunks: global seed / hit offsets of their ends
        struct SegEnd { int64_t seed_hi; uint64_t hit_hi; };
        std::vector<SegEnd> segs;
        if (ca.raw_hits) {
            check_memcpy(hipMemsetAsync(sl->d_cnt, 0, sizeof(Counters), st), "counters");
            memset(sl->h_cnt, 0, sizeof(Counters));
            segs.push_back({1, ca.raw_hits});
            chunk_hits[0] = num_hits = ca.raw_hits;
            chunk_first_seg[1] = 1;
        } else if (ca.td) {
            // ---- table-direct: td_front has probed the positions and planned every chunk (probe.hip); counters were cleared there ----
            memset(sl->h_cnt, 0, sizeof(Counters));
            for (int c = 0; c < K; c++) {
                const TdPlan& tp = sl->h_td_plan[c];
                chunk_first_seg[c] = (uint32_t)segs.size();
                chunk_hits[c] = tp.num_hits;
                sbound[c + 1] = sbound[c] + tp.num_valid * ca.td_words;
                // the chunk's reference iterations (:718-745) as probe_plan_kernel planned them: two below MAX_HITS, the greedy groups above
                for (uint32_t i = 0; i < tp.n_iter && tp.n_iter != TD_PLAN_OVERFLOW; i++) segs.push_back({0, tp.upto[i]});
                if (tp.n_iter != TD_PLAN_OVERFLOW) t_stats.num_iter += tp.n_iter;
                num_hits = tp.hit_base + tp.num_hits;
            }
            chunk_first_seg[K] = (uint32_t)segs.size();
        } else {
        // ---- bucket lookup + prefix (find_num_hits :157-182 ; inclusive_scan :714) ----
        sl->start.ensure(num_seeds, "seed start");
        sl->count.ensure(num_seeds, "seed count");
        sl->prefix.ensure((size_t)num_seeds + 1, "hit prefix");
        sl->scan_temp.ensure(scan_temp_bytes(num_seeds), "scan temp");
        {
            ProfScope p(sl, "seed_lookup");
            launch_seed_lookup(sl->seeds.p, num_seeds, dc->bucket_start, dc->nkeys, sl->start.p, sl->count.p, st);
        }
        {
            ProfScope p(sl, "hit_prefix_scan");
            launch_exclusive_scan_u64(sl->count.p, sl->prefix.p, num_seeds, sl->scan_temp.p, st);
        }
        // ---- iteration plan (:718-745) of every chunk, one D2H ----
        {
            ProfScope p(sl, "iteration_plan");
            for (int c = 0; c < K; c++)
                launch_plan(sl->prefix.p + sbound[c], sbound[c + 1] - sbound[c], (uint64_t)(uint32_t)g_max_hits, ca.rm ? 0 : 1,
                            sl->d_plan + c, st);
        }
        check_launch("lookup/scan/plan");
        check_memcpy(hipMemcpyAsync(sl->h_plan, sl->d_plan, sizeof(IterPlan) * K, hipMemcpyDeviceToHost, st), "plan");
        check_memcpy(hipMemsetAsync(sl->d_cnt, 0, sizeof(Counters), st), "counters");
        check_sync(st, "plan");
        memset(sl->h_cnt, 0, sizeof(Counters));
        {
            uint64_t hit_base = 0;
            for (int c = 0; c < K; c++) {
                const IterPlan& plan = sl->h_plan[c];
                if (plan.overflow) {
                    fprintf(stderr, "Error: SeedAndFilter needs %u iterations (> %u); MAX_HITS=%ld is too small for %lu hits\n",
                            plan.overflow, PLAN_MAX_ITER, (long)g_max_hits, (unsigned long)plan.num_hits);
                    exit(15);
                }
                chunk_first_seg[c] = (uint32_t)segs.size();
                chunk_hits[c] = plan.num_hits;
                if (plan.num_hits > 0)
                    for (uint32_t i = 0; i < plan.num_iter; i++)
                        segs.push_back({(int64_t)sbound[c] + plan.limit_pos[i] + 1, hit_base + plan.upto[i]});
                hit_base += plan.num_hits;
                t_stats.num_iter += plan.num_iter;
            }
            chunk_first_seg[K] = (uint32_t)segs.size();
            num_hits = hit_base;
        }
        }

        auto ensure_host_out = [&](size_t n) {
            if (sl->h_out_cap >= n) return;
            if (sl->h_out) hipHostFree(sl->h_out);
            sl->h_out_cap = std::max<size_t>(n, 1u << 16);
            if (hipHostMalloc((void**)&sl->h_out, sl->h_out_cap * sizeof(sa_segment_pair)) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for hsp_output failed\n");
                exit(12);
            }
        };
        auto ensure_host_seg = [&](size_t n) {
            if (sl->h_seg_cap >= n) return;
            if (sl->h_seg) hipHostFree(sl->h_seg);
            sl->h_seg_cap = std::max<size_t>(n, 1u << 16);
            if (hipHostMalloc((void**)&sl->h_seg, sl->h_seg_cap * sizeof(uint32_t)) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for the segment ids failed\n");
                exit(12);
            }
        };
        bool spec_tried = false; // ... was launched (a segment too large for LDS is not worth a second attempt)
        bool spec_done = false;  // the speculative LDS chain of a single-batch call delivered the final records
        if (num_hits > 0 && !segs.empty()) {
            // ---- batches of consecutive iterations: expand (find_hits) + extend (find_hsps) ----
            const uint64_t HIT_BATCH = 1ull << 27;  // 128 Mi hits (1 GiB of 8-byte hits) per batch unless one iteration is larger
            sl->recA.ensure((size_t)std::max<uint64_t>(1u << 20, std::min<uint64_t>(num_hits, 1ull << 26)), "survivors");
            uint32_t it = 0;
            int64_t seed_lo = 0;
            uint64_t hit_lo = 0;
            while (it < segs.size()) {
                ExtendArgs ea;
                memset(&ea, 0, sizeof(ea));
                int nseg = 0;
                int64_t b_seed_lo = seed_lo, b_seed_hi = seed_lo;
                uint64_t b_hit_lo = hit_lo, b_hit_hi = hit_lo;
                uint32_t it0 = it;
                while (it < segs.size() && nseg < (ca.td ? MAX_SEGS : MAX_SEGS_ABS)) {
                    uint64_t upto = segs[it].hit_hi;
                    if (!ca.td && nseg > 0 && upto - b_hit_lo > HIT_BATCH) break;  // (no hit list in a table-direct call)
                    sl->h_seg_end[nseg++] = upto;
                    b_seed_hi = std::max(b_seed_hi, segs[it].seed_hi);
                    b_hit_hi = upto;
                    it++;
                }
                seed_lo = b_seed_hi;
                hit_lo = b_hit_hi;
                const uint64_t bh = b_hit_hi - b_hit_lo;
                if (bh == 0 || (!ca.td && b_seed_hi <= b_seed_lo)) continue;  // iterations without hits produce nothing (H5)
                // (the segment ends of the batch: a device array the candidate-stage kernels search; the previous batch has been
                //  synchronised, so the pinned staging copy is free)
                // (a table-direct call is one batch whose ends probe_plan_kernel has already written from the same chunk plans)
                if (!ca.td) check_memcpy(hipMemcpyAsync(sl->d_seg_end, sl->h_seg_end, (size_t)nseg * sizeof(uint64_t), hipMemcpyHostToDevice, st), "segment ends");
                ea.seg_end = sl->d_seg_end;
                if (ca.td) {
                    ea.td = 1;
                    ea.td_rec = sl->td_rec.p;
                    ea.td_chunk = sl->td_chunk.p;
                    ea.td_m = sl->h_td_plan[K - 1].m_hi;
                    ea.td_pos = dc->nbr_pos;
                    ea.seed_size = g_seed_size;
                    // (class filter: td_eligible has made sure the 2-bit copies of both strands are there and addressable)
                    if (dc->nbr_ctx && !q2_usable(ca.q2_own, ca.q2_other)) {
                        fprintf(stderr, "Error: table-direct call against the context table without usable 2-bit query copies\n");
                        exit(15);
                    }
                    if (dc->nbr_ctx) {
                        ea.td_ctx = dc->nbr_ctx;
                        ea.td_bits = reinterpret_cast<const uint64_t*>(sl->td_bits.p);
                        ea.q2_own = ca.q2_own->base;
                        ea.q2_other = ca.q2_other->base;
                        ea.q2_stride = ca.q2_own->stride;
                        class_scores(dc->ref_present, ca.q_present, ea.cls);
                    }
                    if (ea.td_ctx) {
                        // (a sub-list can take a whole chunk; SEGALIGN_AMD_L2_CAP: tests start small to reach the regrow-and-rerun path)
                        sl->l2_list.ensure(g_l2_cap_test ? (size_t)g_l2_cap_test
                                                         : (size_t)std::max<uint64_t>((uint64_t)L2_NSUB * TD_CHUNK_HITS, bh / 8), "second-level list");
                        sl->l2_counts.ensure((size_t)L2_NSUB * L2_CNT_STRIDE, "second-level counters");
                    }
                    ea.l2_count = sl->l2_counts.p;
                    ea.l2_total = &sl->d_cnt->n_l2;
                    ea.l2_max = &sl->d_cnt->n_l2_max;
                    ea.l2_blocks = (uint32_t)g_l2_blocks;
                    ea.ctx_waves = (uint32_t)g_ctx_waves;
                    ea.ctx_threads = (uint32_t)g_ctx_threads;
                    // (the one-copy form unless the sixteen copies were asked for AND built: the option is read at InitializeProcessor, the
                    //  copies are made at SendQueryWriteRequest)
                    ea.cls_one_copy = (ca.q2_own->copies == (uint32_t)Q2_COPIES && ca.q2_other->copies == (uint32_t)Q2_COPIES && opt_value("cls_one_copy") == 2) ? 2u : 1u;
                    if (ca.join) {  // key-ordered call (join.h): segments are resolved per hit from the chunk table join_plan_kernel left in d_seg_end
                        ea.join = 1;
                        ea.join_q_lo = ca.q_lo;
                        ea.join_chunk = sl->jq_chunk;
                    }
                } else if (!ca.raw_hits) {
                    sl->hits.ensure((size_t)bh, "hits");
                    ProfScope p(sl, "expand_hits");
                    launch_expand_hits(sl->seeds.p, sl->start.p, sl->count.p, sl->prefix.p, (uint32_t)b_seed_lo,
                                       (uint32_t)b_seed_hi, b_hit_lo, dc->pos_table, g_seed_size, sl->hits.p, st);
                }
                ea.hits = ca.td ? nullptr : sl->hits.p;
                ea.ref8 = dc->ref8.codes;
                ea.fin_batch = g_fin_batch;
                ea.bufs_per_wave = g_bufs_per_wave;
                ea.query = ca.query;
                ea.ref_len = dc->ref.len;
                ea.query_len = ca.query_len;
                ea.sub_mat = dc->d_sub_mat;
                ea.xdrop = g_xdrop;
                ea.hspthresh = g_hspthresh;
                ea.noentropy = g_noentropy;
                ea.left_skip = dc->nbr_left_skip;
                ea.l2_right_state = (uint32_t)opt_value("l2_right_state");
                ea.log4_double = g_log4_double;
                ea.entropy_ulps = g_entropy_ulps;
                ea.num_hits = bh;
                ea.hit_base = b_hit_lo;
                ea.num_segs = nseg;
                ea.seg_base = it0;
                ea.out_count = &sl->d_cnt->survivors;
                ea.examined = g_count_examined ? &sl->d_cnt->examined : nullptr;
                ea.rm = ca.rm;
                ea.rm_rev = ca.rm_rev;
                ea.rm_win_start = ca.rm_win_start;
                ea.rm_win_end = ca.rm_win_end;
                ea.long_cap = (uint32_t)g_long_cap;
                ea.cand_count = &sl->d_cnt->n_long;
                ea.fast_filter = g_fast_filter;
                if (g_packed_filter && ca.query4 && ca.query4->base && dc->ref2.base && !g_count_examined) {
                    ea.fast_filter = 3;  // packed upper-bound filter (extend.hip 1b)
                    ea.ref2 = dc->ref2.base;
                    ea.ref2_stride = dc->ref2.stride;
                    ea.query4 = ca.query4->base;
                    ea.query4_stride = ca.query4->stride;
                }
                ea.ent_count = &sl->d_cnt->n_ent;
                ea.long_blocks = (uint32_t)g_long_blocks;
                ea.max_waves = (uint32_t)(ea.fast_filter == 3 ? g_packed_waves : g_max_waves);
                ea.ent_blocks = 256;  // (grid-stride over a count that lives on the device: a few thousand records usually, millions on repeats)
                sl->cand_list.ensure((size_t)std::max<uint64_t>(1u << 16, bh / 16), "candidate list");
                // chain shortcut: valid for the plain X-drop recurrence (xdrop >= 0), kept to query blocks below 2^29 bases on the
                // general path (the envelope it is tested in; the 64-bit sort key of rounds 1-4 needed it, chain_key32 does not), and
                // off while E is being counted.  The repeat masker takes it too: its window only decides WHICH
                // hits are extended (all candidates lie inside it), and its chain starts with an exact-duplicate unique
                // (rm :819-823), so the duplicates the shortcut never produces would be removed there anyway
                const bool chain_rel = ca.td && ca.q_hi > ca.q_lo;  // table-direct call: anchors relative to the call's first position
                const bool chain = g_chain && g_xdrop >= 0 && (chain_rel || (nseg <= MAX_SEGS_ABS && ca.query_len < (1u << 29))) && !g_count_examined;
                ea.chain_q_bits = chain_rel ? 32u : 29u;  // (32: table-direct call, no iteration in the sort key's hash -- kernels.h)
                ea.chain_cap = chain ? CHAIN_CAP : 0u;
                ea.chain_buckets = (uint32_t)g_chain_buckets;  // (0: the device sizes them by the candidates it finds)
                ea.chain_bucket_target = (uint32_t)g_chain_bucket_target;
                ea.chain_sort_blocks = (uint32_t)g_chain_sort_blocks;
                ea.chain_group_max = (uint32_t)g_chain_group_max;
                ea.chain_no_link = (uint32_t)opt_value("chain_no_link");
                ea.chain_sort_threads = (uint32_t)g_chain_sort_threads;
                if (chain) {
                    sl->chain_tmp.ensure(CHAIN_CAP, "chain candidates");
                    sl->chain_sorted.ensure(CHAIN_CAP, "chain candidates");
                    sl->chain_is_head.ensure(CHAIN_CAP, "chain flags");
                    sl->chain_heads.ensure(CHAIN_CAP, "chain heads");
                    sl->chain_bucket_cnt.ensure(chain_num_buckets(), "chain buckets");
                    sl->chain_bucket_start.ensure(chain_num_buckets() + 1, "chain buckets");
                    ea.chain_tmp = sl->chain_tmp.p;
                    ea.chain_sorted = sl->chain_sorted.p;
                    ea.chain_bucket_cnt = sl->chain_bucket_cnt.p;
                    ea.chain_bucket_start = sl->chain_bucket_start.p;
                    ea.chain_is_head = sl->chain_is_head.p;
                    ea.chain_heads = sl->chain_heads.p;
                    ea.chain_head_count = &sl->d_cnt->n_heads;
                    ea.chain_big = &sl->d_cnt->n_chain_big;
                }
                sl->ent_list.ensure((size_t)std::max<uint64_t>(1u << 16, bh / 32), "entropy list");
                Counters before = *sl->h_cnt;  // counters as of the previous batch (zero for the first)
                before.n_long = 0;
                before.n_ent = 0;
                before.n_heads = 0;
                before.n_l2 = 0;
                before.n_l2_max = 0;
                // (a table-direct call is ONE batch, and td_front's clearing kernel has just zeroed the counters, the sub-list counters,
                //  the chain buckets and the segment info: the memsets below only run for later batches and for reruns)
                bool cleared = ca.td && it0 == 0;
                if (!cleared) check_memcpy(hipMemsetAsync(&sl->d_cnt->n_long, 0, 6 * sizeof(uint32_t), st), "counters");
                if (g_audit_cap && ca.td) {
                    sl->audit.ensure(g_audit_cap, "audit list");
                    ea.audit_list = sl->audit.p;
                    ea.audit_count = &sl->d_cnt->n_audit;
                    ea.audit_cap = g_audit_cap;
                }
                before.n_audit = 0;
                before.n_chain_big = 0;
                for (;;) {  // rerun the batch with larger lists if one overflowed (device writes are guarded)
                    ea.out = sl->recA.p;
                    ea.out_cap = (uint32_t)std::min<size_t>(sl->recA.cap, 0xFFFFFFFFu);
                    ea.cand_list = sl->cand_list.p;
                    ea.cand_cap_recs = (uint32_t)std::min<size_t>(sl->cand_list.cap, 0xFFFFFFFFu);
                    ea.ent_list = sl->ent_list.p;
                    ea.ent_cap_recs = (uint32_t)std::min<size_t>(sl->ent_list.cap, 0xFFFFFFFFu);
                    if (ea.td && ea.td_ctx) {
                        // context / class filter over the table's own records, then the packed filter on what it could not decide
                        ea.l2_list = sl->l2_list.p;
                        ea.l2_cap = (uint32_t)std::min<size_t>(sl->l2_list.cap / L2_NSUB, 0xFFFFFFu);  // per sub-list
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->l2_counts.p, 0, (size_t)L2_NSUB * L2_CNT_STRIDE * sizeof(uint32_t), st), "second-level counters");
                        if (ca.join) {
                            JoinArgs ja;
                            ja.head = sl->d_jhead;
                            ja.head_rw = sl->d_jhead;
                            ja.ent = sl->jq_ent.p;
                            ja.vstart = sl->jq_vstart.p;
                            ja.qx = sl->jq_qx.p;
                            check_memcpy(hipMemsetAsync(&sl->d_jhead->work_next, 0, sizeof(unsigned long long), st), "join work cursor");  // (reruns start over)
                            ProfScope p(sl, "extend_filter");
                            launch_join_filter(ea, ja, st);
                        } else {
                            ProfScope p(sl, "extend_filter");
                            if (sl->stream_lo) {  // (option filter_prio: the class filter alone on the slot's low-priority stream)
                                hipEventRecord(sl->ev_lo_a, st);
                                hipStreamWaitEvent(sl->stream_lo, sl->ev_lo_a, 0);
                                launch_extend_filter_cls(ea, sl->stream_lo);
                                hipEventRecord(sl->ev_lo_b, sl->stream_lo);
                                hipStreamWaitEvent(st, sl->ev_lo_b, 0);
                            } else {
                                launch_extend_filter_cls(ea, st);
                            }
                        }
                        ExtendArgs e2 = ea;
                        e2.td = 0;
                        e2.src_cand = 1;
                        { ProfScope p(sl, "extend_filter2"); launch_extend_filter(e2, st); }
                    } else {
                        ProfScope p(sl, "extend_filter");
                        launch_extend_filter(ea, st);
                    }
                    if (ea.chain_cap) {
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->chain_bucket_cnt.p, 0, chain_num_buckets() * sizeof(uint32_t), st), "chain buckets");
                        { ProfScope p(sl, "chain_group"); launch_chain_group(ea, st); }
                    }
                    if (ea.chain_cap) { ProfScope p(sl, "extend_exact_chain"); launch_extend_exact_chain(ea, st); }
                    else              { ProfScope p(sl, "extend_exact");       launch_extend_exact(ea, st); }
                    { ProfScope p(sl, "extend_entropy"); launch_extend_entropy(ea, st); }
                    check_launch("expand/extend");
                    // A call that is ONE batch (every table-direct call) does not wait for the survivor count: the per-segment LDS
                    // chain (:776-782) is launched on the device-side count and its first SPEC_RECS records travel with the
                    // counters -- one host sync for extension + chain + output instead of two
                    const bool spec = ca.td && !ca.rm && !ca.raw_hits && it0 == 0 && it == segs.size() &&
                                      segs.size() <= dedup_small_max_segs() && !g_no_small_dedup && g_spec_dedup;
                    const uint32_t seg_words = dedup_seg_info_words();
                    if (spec) {
                        spec_tried = true;
                        sl->out16.ensure(dedup_seg_max_total(), "out16");
                        ensure_host_out(dedup_seg_max_total());
                        ensure_host_seg(std::max<size_t>(dedup_seg_max_total(), seg_words));
                        if (!cleared) check_memcpy(hipMemsetAsync(sl->d_seg_info, 0, seg_words * sizeof(uint32_t), st), "segment info");
                        { ProfScope p(sl, "dedup_seg"); launch_dedup_seg(ea.out, 0, &sl->d_cnt->survivors, (uint32_t)segs.size(), sl->out16.p, sl->d_seg_info, (uint32_t)g_dedup_threads, g_dedup_seg_max, st); }
                        check_launch("dedup seg");
                        check_memcpy(hipMemcpyAsync(sl->h_seg_info, sl->d_seg_info, seg_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "segment info");
                        check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)SPEC_RECS * sizeof(sa_segment_pair), hipMemcpyDeviceToHost, st),
                                     "hsp_output");  // :788
                    }
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "extend");
                    spec_done = spec && sl->h_seg_info[seg_words - 1] == 0;
                    if (ea.chain_cap && sl->h_cnt->n_long > ea.chain_cap && sl->h_cnt->n_long <= ea.cand_cap_recs) {
                        spec_done = false;  // (the chain ran on an unfinished survivor list)
                        path_flags |= SA_PATH_CHAIN_SLICED;
                        // more candidates than the chain buffers hold: the chain kernels left the batch alone (device-side test on
                        // the same counter).  Run the chain stages over the candidate list SLICE BY SLICE: a chain that crosses a
                        // slice border restarts there (one more extension), nothing else changes -- repeat-rich sequence puts tens of
                        // millions of candidates into one call, and extending each on its own took 160 ms per call
                        const uint32_t n_long = sl->h_cnt->n_long;
                        for (uint32_t first = 0; first < n_long; first += ea.chain_cap) {
                            ExtendArgs es = ea;
                            es.cand_sliced = 1;
                            es.cand_first = first;
                            check_memcpy(hipMemsetAsync(sl->chain_bucket_cnt.p, 0, chain_num_buckets() * sizeof(uint32_t), st), "chain buckets");
                            check_memcpy(hipMemsetAsync(&sl->d_cnt->n_heads, 0, sizeof(uint32_t), st), "chain heads");
                            { ProfScope p(sl, "chain_group"); launch_chain_group(es, st); }
                            { ProfScope p(sl, "extend_exact_chain"); launch_extend_exact_chain(es, st); }
                        }
                        { ProfScope p(sl, "extend_entropy"); launch_extend_entropy(ea, st); }
                        check_launch("extend (sliced chain)");
                        check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                        check_sync(st, "extend (sliced chain)");
                    }
                    const Counters& c = *sl->h_cnt;
                    const bool l2_ok = !(ea.td && ea.td_ctx) || c.n_l2_max <= ea.l2_cap;
                    if (c.survivors <= ea.out_cap && c.n_long <= ea.cand_cap_recs && c.n_ent <= ea.ent_cap_recs && l2_ok) break;
                    path_flags |= SA_PATH_LIST_REGROWN;
                    if (!l2_ok)  // (the later stages saw a truncated list)
                        sl->l2_list.ensure((size_t)c.n_l2_max * L2_NSUB + ((size_t)c.n_l2_max * L2_NSUB) / 4, "second-level list(grow)");
                    // an overflowing long list also truncates what the later kernels saw: size everything from the
                    // counts of this attempt (upper bounds for the rerun: survivors <= hits, entropy candidates <= hits)
                    if (c.n_long > ea.cand_cap_recs) sl->cand_list.ensure((size_t)c.n_long, "candidate list(grow)");
                    if (c.n_ent > ea.ent_cap_recs || c.n_long > ea.cand_cap_recs)
                        sl->ent_list.ensure((size_t)std::min<uint64_t>(bh, (uint64_t)c.n_ent + c.n_long), "entropy list(grow)");
                    if (c.survivors > ea.out_cap || c.n_long > ea.cand_cap_recs)
                        sl->recA.ensure((size_t)std::min<uint64_t>((uint64_t)before.survivors + bh, (uint64_t)c.survivors + c.n_long + c.n_ent),
                                        "survivors(grow)", true, st);
                    check_memcpy(hipMemcpy(sl->d_cnt, &before, sizeof(Counters), hipMemcpyHostToDevice), "counter reset");
                    cleared = false;  // (the rerun of the batch clears its lists itself)
                }
                survivors = sl->h_cnt->survivors;
                if (sl->h_cnt->n_chain_big) path_flags |= SA_PATH_CHAIN_BUCKET_OVERFLOW;
                n_cand_total += sl->h_cnt->n_long;
                n_fwd_total += sl->h_cnt->n_l2;
                n_ent_total += sl->h_cnt->n_ent;
            }
            if (g_audit_cap && ca.td) {  // (tests) the rejected hits of the last batch
                const uint32_t na = std::min(sl->h_cnt->n_audit, g_audit_cap);
                t_audit.resize(na);
                if (na) {
                    check_memcpy(hipMemcpyAsync(t_audit.data(), sl->audit.p, (size_t)na * sizeof(uint2), hipMemcpyDeviceToHost, st), "audit");
                    check_sync(st, "audit");
                }
            }
            t_stats.num_examined = sl->h_cnt->examined;
            t_stats.num_examined_filter = sl->h_cnt->examined_filter;
            t_stats.num_candidates = n_cand_total;
            t_stats.num_forwarded = n_fwd_total;
            t_stats.num_entropy = n_ent_total;

            // ---- order + de-duplicate (:776-782 ; rm :819-831) ----
            if (survivors > 0 && spec_done && survivors <= dedup_seg_max_total()) {
                if (survivors > SPEC_RECS) {  // the records beyond the speculative prefix
                    check_memcpy(hipMemcpyAsync(sl->h_out + SPEC_RECS, sl->out16.p + SPEC_RECS, (size_t)(survivors - SPEC_RECS) * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");
                    check_sync(st, "hsp_output");
                }
                const uint32_t S = dedup_small_max_segs();
                size_t pos = 0;
                for (uint32_t g = 0; g < (uint32_t)segs.size(); g++) {  // close the gaps the unique step left
                    const uint32_t m2 = sl->h_seg_info[g], off = sl->h_seg_info[S + g];
                    if (m2 && pos != off) memmove(sl->h_out + pos, sl->h_out + off, (size_t)m2 * sizeof(sa_segment_pair));
                    for (uint32_t i = 0; i < m2; i++) sl->h_seg[pos + i] = g;
                    pos += m2;
                }
                n_final = (uint32_t)pos;
                have_seg = true;
            } else if (survivors > 0 && ca.raw_hits) {  // the extension stage's own output, unordered
                sl->out16.ensure(survivors, "out16");
                ensure_host_out(survivors);
                launch_strip(sl->recA.p, survivors, sl->out16.p, nullptr, st);
                check_launch("strip");
                check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)survivors * sizeof(sa_segment_pair), hipMemcpyDeviceToHost, st),
                             "hsp_output");
                check_sync(st, "hsp_output");
                n_final = survivors;
            } else if (survivors > 0) {
                sl->recB.ensure(std::max<size_t>(survivors, sl->recA.cap), "survivors B");
                size_t tb = sort_temp_bytes(survivors);
                sl->sort_temp.ensure(tb, "sort temp");
                sl->unique_temp.ensure(unique_temp_bytes(survivors), "unique temp");
                HspRec* fin = nullptr;
                bool done = false;
                if (!ca.rm && survivors <= dedup_seg_max_total() && segs.size() <= dedup_small_max_segs() && !g_no_small_dedup && !spec_tried) {
                    // the whole chain in LDS, one workgroup per segment; one D2H of the (gapped) records + the segment counts
                    const uint32_t words = dedup_seg_info_words();
                    sl->out16.ensure(survivors, "out16");
                    ensure_host_out(survivors);
                    ensure_host_seg(std::max<size_t>(survivors, words));
                    check_memcpy(hipMemsetAsync(sl->d_seg_info, 0, words * sizeof(uint32_t), st), "segment info");
                    { ProfScope p(sl, "dedup_seg"); launch_dedup_seg(sl->recA.p, survivors, nullptr, (uint32_t)segs.size(), sl->out16.p, sl->d_seg_info, (uint32_t)g_dedup_threads, g_dedup_seg_max, st); }
                    check_launch("dedup seg");
                    check_memcpy(hipMemcpyAsync(sl->h_seg_info, sl->d_seg_info, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "segment info");
                    check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)survivors * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");  // :788
                    check_sync(st, "hsp_output");
                    if (sl->h_seg_info[words - 1] == 0) {  // (else a segment was too large for LDS: library sorts below)
                        const uint32_t S = dedup_small_max_segs();
                        size_t pos = 0;
                        for (uint32_t g = 0; g < (uint32_t)segs.size(); g++) {  // close the gaps the unique step left
                            const uint32_t m2 = sl->h_seg_info[g], off = sl->h_seg_info[S + g];
                            if (m2 && pos != off) memmove(sl->h_out + pos, sl->h_out + off, (size_t)m2 * sizeof(sa_segment_pair));
                            for (uint32_t i = 0; i < m2; i++) sl->h_seg[pos + i] = g;
                            pos += m2;
                        }
                        n_final = (uint32_t)pos;
                        have_seg = true;
                        done = true;
                    }
                }
                if (!done && !ca.rm && segs.size() <= dedup_small_max_segs() && !g_no_small_dedup) path_flags |= SA_PATH_DEDUP_FALLBACK;
                if (done) {
                    // nothing left to do on the device
                } else if (!ca.rm) {
                    { ProfScope p(sl, "sort_diag");  launch_sort(sl->recA.p, sl->recB.p, survivors, ORDER_DIAG, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");     launch_unique(sl->recB.p, sl->recA.p, survivors, 0, &sl->d_cnt->uniq, sl->unique_temp.p, st); }
                    check_launch("sort/unique");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique");
                    n_final = sl->h_cnt->uniq;
                    { ProfScope p(sl, "sort_lastz"); launch_sort(sl->recA.p, sl->recB.p, n_final, ORDER_LASTZ, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    fin = sl->recB.p;
                } else {
                    { ProfScope p(sl, "sort_rm_first"); launch_sort(sl->recA.p, sl->recB.p, survivors, ORDER_RM_FIRST, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");        launch_unique(sl->recB.p, sl->recA.p, survivors, 1, &sl->d_cnt->uniq, sl->unique_temp.p, st); }
                    check_launch("sort/unique");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique");
                    uint32_t n1 = sl->h_cnt->uniq;
                    { ProfScope p(sl, "sort_rm_diag");  launch_sort(sl->recA.p, sl->recB.p, n1, ORDER_RM_DIAG, sl->sort_temp.p, sl->sort_temp.cap, st); }
                    { ProfScope p(sl, "unique");        launch_unique(sl->recB.p, sl->recA.p, n1, 0, &sl->d_cnt->uniq2, sl->unique_temp.p, st); }
                    check_launch("sort/unique 2");
                    check_memcpy(hipMemcpyAsync(sl->h_cnt, sl->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st), "counters");
                    check_sync(st, "unique 2");
                    n_final = sl->h_cnt->uniq2;
                    if (ca.cov_diff) {  // coverage is order-independent: the final sort (rm :831) and the D2H are not needed
                        ProfScope p(sl, "coverage_add");
                        launch_coverage_add_hsprec(sl->recA.p, n_final, ca.cov_diff, ca.cov_diff_len, sl->d_cov_range, st);
                        check_launch("coverage add");
                    } else {
                        { ProfScope p(sl, "sort_rm_final"); launch_sort(sl->recA.p, sl->recB.p, n_final, ORDER_RM_FINAL, sl->sort_temp.p, sl->sort_temp.cap, st); }
                        fin = sl->recB.p;
                    }
                }
                if (n_final > 0 && fin) {
                    sl->out16.ensure(n_final, "out16");
                    ensure_host_out(n_final);
                    if (K > 1) { sl->out_seg.ensure(n_final, "out seg"); ensure_host_seg(n_final); }
                    { ProfScope p(sl, "strip"); launch_strip(fin, n_final, sl->out16.p, K > 1 ? sl->out_seg.p : nullptr, st); }
                    check_launch("final sort/strip");
                    if (K > 1) {
                        check_memcpy(hipMemcpyAsync(sl->h_seg, sl->out_seg.p, (size_t)n_final * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "hsp segs");
                        have_seg = true;
                    }
                    check_memcpy(hipMemcpyAsync(sl->h_out, sl->out16.p, (size_t)n_final * sizeof(sa_segment_pair),
                                                hipMemcpyDeviceToHost, st), "hsp_output");  // :788
                }
                check_sync(st, "hsp_output");
            }
        }
    }
    prof_flush(sl);

    t_stats.path_flags = path_flags;
    t_stats.lookup_path = ca.td ? ((dc->nbr_ctx && ca.q2_own && ca.q2_own->base) ? 2 : 1) : 0;
    t_stats.num_hits = num_hits;
    t_stats.num_survivors = survivors;
    t_stats.num_anchors = n_final;
    if (K > 1 && ca.outs) {
        // ---- one return vector per chunk: records are ordered by segment, chunk c owns segments
        //      [chunk_first_seg[c], chunk_first_seg[c+1]) ; a chunk without seeds returns nothing (seeder.cpp:76) ----
        size_t pos = 0;
        for (int c = 0; c < K; c++) {
            size_t n_c = 0;
            if (have_seg)
                while (pos + n_c < n_final && sl->h_seg[pos + n_c] < chunk_first_seg[c + 1]) n_c++;
            if (sbound[c + 1] == sbound[c]) {
                ca.outs[c] = nullptr;
                ca.counts[c] = 0;
            } else {
                sa_segment_pair* r = (sa_segment_pair*)malloc((n_c + 1) * sizeof(sa_segment_pair));
                memset(&r[0], 0, sizeof(sa_segment_pair));
                r[0].len = (uint32_t)n_c;
                r[0].score = (int32_t)(uint32_t)chunk_hits[c];
                if (n_c) memcpy(r + 1, sl->h_out + pos, n_c * sizeof(sa_segment_pair));
                ca.outs[c] = r;
                ca.counts[c] = n_c + 1;
            }
            pos += n_c;
        }
        return (size_t)n_final + K;
    }
    if (out == nullptr) return (size_t)n_final + 1;  // coverage mode: nothing is returned to the host

    // ---- return vector: header + HSPs (:804-827 ; rm :857-861) ----
    sa_segment_pair* res = (sa_segment_pair*)malloc(((size_t)n_final + 1) * sizeof(sa_segment_pair));
    memset(&res[0], 0, sizeof(sa_segment_pair));
    if (!ca.rm) {
        res[0].len = n_final;
        res[0].score = (int32_t)(uint32_t)num_hits;
    } else {
        uint64_t ta = n_final;
        res[0].ref_start = (uint32_t)(num_hits & 0xFFFFFFFFull);
        res[0].query_start = (uint32_t)(num_hits >> 32);
        res[0].len = (uint32_t)(ta & 0xFFFFFFFFull);
        res[0].score = (int32_t)(ta >> 32);
    }
    if (n_final) memcpy(res + 1, sl->h_out, (size_t)n_final * sizeof(sa_segment_pair));
    *out = res;
    return (size_t)n_final + 1;
}

}  // namespace sa
