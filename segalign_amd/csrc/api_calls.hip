// api_calls.hip -- C-ABI, the hot calls: g_SeedAndFilter (src/seed_filter.cu:682-828) and its additive forms.
#include "engine_internal.h"

using namespace sa;

extern "C" {

// ---- hot calls ------------------------------------------------------------------------------------------------------
size_t sa_seed_and_filter(const uint64_t* seeds, size_t num_seeds, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_proc("SeedAndFilter", buffer);
    if ((int64_t)num_seeds > g_max_seeds) {  // :688-692
        printf("MAX_SEEDS exceeded\n");
        fflush(stdout);
        fprintf(stderr, "Assertion `num_seeds <= MAX_SEEDS' failed.\n");
        abort();
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    upload_seeds(sl, seeds, num_seeds);
    CoreArgs ca = {rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes, g_query_len[buffer], 0, 0, 0, 0, 0, 0, nullptr, 0,
                   rev ? &dc->query4_rc[buffer] : &dc->query4[buffer]};  // :762-767
    set_query2(ca, dc, buffer, rev);
    uint32_t lo = 0, hi = 0, words = 0;
    if (dropin_td_front(dc, sl, ca.query, ca.query_len, seeds, num_seeds, ca.query4, ca.q2_own, ca.q2_other, 0, &lo, &hi, &words) != 0xFFFFFFFFu) {
        ca.td = 1;
        ca.td_words = words;
        ca.q_lo = lo;
        ca.q_hi = hi;
    }
    size_t n = saf_core(dc, sl, (uint32_t)num_seeds, ca, out);
    release_slot(sl);
    return n;
}

size_t sa_seed_and_filter_range(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_proc("SeedAndFilterRange", buffer);
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    uint32_t qlen = g_query_len[buffer];
    // a seed window must lie inside the block: positions j with j + span <= len
    uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    if (end > lim) end = lim;
    const PackedBuf* q4 = rev ? &dc->query4_rc[buffer] : &dc->query4[buffer];
    uint32_t ns = 0xFFFFFFFFu, words = 0;
    if (td_eligible(dc, q4, rev ? &dc->query2_rc[buffer] : &dc->query2[buffer], rev ? &dc->query2[buffer] : &dc->query2_rc[buffer])) {
        const uint32_t bp[2] = {start, std::max(start, end)};
        ns = td_front(dc, sl, q, 1, bp, 0, &words);
    }
    const bool td = ns != 0xFFFFFFFFu;
    if (!td) { ns = device_seeds(sl, q, start, end); t_front_flags |= SA_PATH_GENERAL_FALLBACK; }
    size_t n = 0;
    *out = nullptr;
    if (ns > 0) {  // seeder.cpp:76: the engine is only called for a non-empty seed vector
        CoreArgs ca = {q, qlen, 0, 0, 0, 0, start, end, nullptr, 0, q4};
        set_query2(ca, dc, buffer, rev);
        ca.td = td ? 1 : 0;
        ca.td_words = words;
        n = saf_core(dc, sl, ns, ca, out);
    } else {
        prof_flush(sl);
        memset(&t_stats, 0, sizeof(t_stats));
        t_front_flags = 0;
    }
    release_slot(sl);
    return n;
}

// Up to SA_MAX_CHUNKS consecutive wga_chunk-sized chunks of one strand in ONE pass over the kernels: the chunks share the
// seeding, lookup, expansion, extension, grouping and ordering launches and the host syncs, while every chunk keeps its own
// iteration plan, dedup scope and return vector -- bit for bit what one sa_seed_and_filter_range call per chunk returns.
int sa_max_chunks_per_call(void) { return SA_MAX_CHUNKS; }
// What the interval entries hand to one call, and what a host that builds its own call lists should use: option chunks_per_call
// (default 40), raised for the RESIDENT target when its seed hits are sparse -- a call is sized by hits, not by chunks: option
// call_hits (default 128 M: beyond that a call no longer gets cheaper per hit, and a pass has too few calls to fill the slots) / (table entries per key x wga_chunk positions), at most sa_max_chunks_per_call().  With
// --notransition one 250 kbp chunk holds ~1.5 M hits instead of ~13 M: forty-chunk calls would spend two thirds of their time in
// per-call fixed costs.
int sa_get_chunks_per_call(void) {
    int k = g_chunks_per_call;
    if (g_call_hits > 0 && g_ndev > 0 && g_proc_init) {
        DevCtx* dc = g_dev[0];
        if (dc->nbr_state == 1 && dc->nkeys > 0 && g_wga_chunk > 0) {
            const double per_pos = (double)dc->nbr_total / (double)dc->nkeys;           // hits of a query position with a random k-mer
            const double per_chunk = std::max(1.0, per_pos * (double)g_wga_chunk);
            const double want = (double)g_call_hits / per_chunk;
            if (want > (double)k) k = (int)std::min<double>(want, (double)SA_MAX_CHUNKS);
            // ... and lowered when they are dense (a 500 Mbp target block: ~70 M hits per chunk): a call stays below ~option
            // call_hits_max = 1 G hits by this estimate (its lists are sized by its hits; 2^32 is the hard limit of a pass)
            const double most = (double)g_call_hits_max / per_chunk;
            if (g_call_hits_max > 0 && most < (double)k) k = (int)std::max(1.0, most);
            // key-ordered calls (join.h) want MANY positions per call -- a key's run is fetched once for all the call's positions that
            // carry the key: option key_order_chunks (200 chunks = 50 Mbp = three positions per 12-mer), as long as the call's lists
            // stay in bounds (option key_order_hits) and the call still holds about a position per key
            if (g_key_order && dc->nbr_ctx) {
                const int kj = (int)std::min<double>((double)g_key_order_chunks, std::max(1.0, (double)g_key_order_hits / per_chunk));
                if (kj > k && join_wanted(dc, kj, (uint32_t)std::min<uint64_t>((uint64_t)kj * g_wga_chunk, 0xFFFFFFFFull))) k = kj;
            }
        }
    }
    return k;
}

static void stats_add(sa_call_stats& a, const sa_call_stats& b) {
    a.num_seeds += b.num_seeds; a.num_hits += b.num_hits; a.num_survivors += b.num_survivors; a.num_anchors += b.num_anchors;
    a.num_examined += b.num_examined; a.num_examined_filter += b.num_examined_filter; a.num_candidates += b.num_candidates;
    a.num_forwarded += b.num_forwarded; a.num_entropy += b.num_entropy; a.num_iter += b.num_iter;
    a.device = b.device;
    a.lookup_path = b.lookup_path;
    a.path_flags |= b.path_flags;
}

// The chunks of [start, end) in one pass -- or, when the pass cannot hold them, in two halves (recursively): a table-direct call
// indexes its hits with 32 bits and needs every chunk below MAX_HITS, the general path plans at most SA_MAX_CHUNKS_GENERAL chunks
// per pass.  Splitting changes nothing in the results (every chunk keeps its own plan and dedup scope either way); the calling
// thread's statistics are the sums over the passes.
static size_t chunks_pass(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** outs, size_t* counts) {
    const uint32_t chunk = g_wga_chunk;
    const int K = end > start ? (int)(((uint64_t)end - start + chunk - 1) / chunk) : 0;
    if (K == 0) return 0;
    if (K == 1) {
        counts[0] = sa_seed_and_filter_range(start, end, rev, buffer, &outs[0]);
        return counts[0];
    }
    auto halves = [&]() {
        const int k0 = K / 2;
        const uint32_t mid = start + (uint32_t)k0 * chunk;
        sa_call_stats sum;
        size_t total = chunks_pass(start, mid, rev, buffer, outs, counts);
        sum = t_stats;
        total += chunks_pass(mid, end, rev, buffer, outs + k0, counts + k0);
        stats_add(sum, t_stats);
        t_stats = sum;
        return total;
    };
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    const uint32_t qlen = g_query_len[buffer];
    const uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    const uint32_t send = std::min(end, lim);  // a seed window must lie inside the block
    uint32_t bpos[SA_MAX_CHUNKS + 1], bseed[SA_MAX_CHUNKS_GENERAL + 1];
    for (int c = 0; c <= K; c++) bpos[c] = (uint32_t)std::min<uint64_t>((uint64_t)start + (uint64_t)c * chunk, send);
    const PackedBuf* q4 = rev ? &dc->query4_rc[buffer] : &dc->query4[buffer];
    uint32_t ns = 0xFFFFFFFFu, words = 0;
    const bool eligible = td_eligible(dc, q4, rev ? &dc->query2_rc[buffer] : &dc->query2[buffer], rev ? &dc->query2[buffer] : &dc->query2_rc[buffer]);
    const PackedBuf* q2o = rev ? &dc->query2_rc[buffer] : &dc->query2[buffer];
    const PackedBuf* q2x = rev ? &dc->query2[buffer] : &dc->query2_rc[buffer];
    const bool join = eligible && send > start && q2_usable(q2o, q2x) && join_wanted(dc, K, send - start);
    if (join) ns = join_front(dc, sl, q, qlen, K, bpos, q2o, q2x, &words);
    else if (eligible) ns = td_front(dc, sl, q, K, bpos, 0, &words);
    const bool td = ns != 0xFFFFFFFFu;
    if (!td && (eligible || K > SA_MAX_CHUNKS_GENERAL)) {
        // one of the chunks needs the general path (num_hits >= MAX_HITS), the pass holds 2^32 hits or more, or it is too long for
        // the general path: halve it -- the chunks that can stay table-direct do
        prof_flush(sl);
        release_slot(sl);
        return halves();
    }
    if (!td) { ns = device_seeds(sl, q, start, send, K + 1, bpos, bseed); t_front_flags |= SA_PATH_GENERAL_FALLBACK; }
    size_t total = 0;
    if (ns > 0) {
        CoreArgs ca = {q, qlen, 0, 0, 0, 0, start, send, nullptr, 0, q4};
        set_query2(ca, dc, buffer, rev);
        ca.nchunks = K;
        ca.td = td ? 1 : 0;
        ca.join = (td && join) ? 1 : 0;
        ca.td_words = words;
        for (int c = 0; c <= K; c++) ca.seed_bound[c] = td ? 0u : bseed[c];  // (a table-direct call derives them from its chunk plans)
        ca.outs = outs;
        ca.counts = counts;
        saf_core(dc, sl, ns, ca, nullptr);
        for (int c = 0; c < K; c++) total += counts[c];
    } else {
        prof_flush(sl);
        memset(&t_stats, 0, sizeof(t_stats));
        t_front_flags = 0;
    }
    release_slot(sl);
    return total;
}

size_t sa_seed_and_filter_chunks(uint32_t start, uint32_t end, int rev, uint32_t buffer, sa_segment_pair** outs, size_t* counts) {
    require_proc("SeedAndFilterChunks", buffer);
    const uint32_t chunk = g_wga_chunk;
    const int K = end > start ? (int)(((uint64_t)end - start + chunk - 1) / chunk) : 0;
    if (K > SA_MAX_CHUNKS) {
        fprintf(stderr, "Error: SeedAndFilterChunks takes at most %d chunks per call\n", SA_MAX_CHUNKS);
        exit(1);
    }
    for (int c = 0; c < K; c++) { outs[c] = nullptr; counts[c] = 0; }
    if (K == 0) {
        memset(&t_stats, 0, sizeof(t_stats));
        return 0;
    }
    return chunks_pass(start, end, rev, buffer, outs, counts);
}

void sa_free_segments(sa_segment_pair* p) { free(p); }

// Introspection (tests): the extension stage alone -- find_hsps + compaction of the passing hits (src/seed_filter.cu:232-680)
// -- for caller-supplied anchors {ref_loc, query_loc} on the resident target / query strand.  Returns 1 + the number of
// passing hits; out[0] is a header {len = count}, the records follow in no particular order and are NOT de-duplicated,
// except that exact duplicates may already be merged (the chain shortcut extends one member of a run of anchors that
// provably produce the identical record).
size_t sa_extend_hits(const uint32_t* ref_query_pairs, size_t num_hits, int rev, uint32_t buffer, sa_segment_pair** out) {
    require_init("ExtendHits");
    if (buffer >= SA_BUFFER_DEPTH) {
        fprintf(stderr, "Error: query buffer %u out of range\n", buffer);
        exit(1);
    }
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    size_t n = 0;
    *out = nullptr;
    if (num_hits > 0) {
        sl->hits.ensure(num_hits, "hits");
        check_memcpy(hipMemcpyAsync(sl->hits.p, ref_query_pairs, num_hits * sizeof(Hit), hipMemcpyHostToDevice, sl->stream), "hits");
        CoreArgs ca = {rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes, g_query_len[buffer], 0, 0, 0, 0, 0, 0, nullptr, 0,
                       rev ? &dc->query4_rc[buffer] : &dc->query4[buffer]};
        set_query2(ca, dc, buffer, rev);
        ca.raw_hits = num_hits;
        n = saf_core(dc, sl, 1, ca, out);
    }
    release_slot(sl);
    return n;
}

// seeder_body::operator() of src/seeder.cpp:12-127 for one query interval: plus-strand chunks [start, end) in steps of
// wga_chunk, then the minus-strand chunks of the same interval in reverse-complement coordinates (:33-34,89-91), every
// chunk through sa_seed_and_filter_range; `threads` chunk calls are kept in flight (the reference keeps one per TBB
// worker).  HSPs are concatenated per strand in chunk order, headers removed (:80-85,115-120).
size_t sa_seed_interval(uint32_t start, uint32_t end, uint32_t q_len, int strands, uint32_t buffer, int threads,
                        sa_segment_pair** out_fw, size_t* n_fw, sa_segment_pair** out_rc, size_t* n_rc, sa_call_stats* totals) {
    require_proc("SeedInterval", buffer);
    struct Job { uint32_t a, b; int rev; int k; sa_segment_pair* res[SA_MAX_CHUNKS]; size_t n[SA_MAX_CHUNKS]; };
    std::vector<Job> jobs;
    const int per_job = sa_get_chunks_per_call();  // chunks of one strand that share one pass over the kernels
    for (int rev = 0; rev < 2; rev++) {
        if (!(strands & (rev ? SA_STRAND_MINUS : SA_STRAND_PLUS))) continue;
        const uint32_t a = rev ? q_len - end : start, b = rev ? q_len - start : end;
        // equal groups: 40 chunks go as 14 + 14 + 12, not 16 + 16 + 8 (the calls of an interval finish together)
        const uint64_t nchunks = b > a ? ((uint64_t)b - a + g_wga_chunk - 1) / g_wga_chunk : 0;
        const uint64_t ncalls = (nchunks + per_job - 1) / per_job;
        const uint64_t group = ncalls ? (nchunks + ncalls - 1) / ncalls : 1;
        for (uint64_t i = a; i < b; i += (uint64_t)g_wga_chunk * group) {
            Job jb;
            memset(&jb, 0, sizeof(jb));
            jb.a = (uint32_t)i;
            jb.b = (uint32_t)std::min<uint64_t>(i + (uint64_t)g_wga_chunk * group, b);
            jb.rev = rev;
            jb.k = (int)(((uint64_t)jb.b - jb.a + g_wga_chunk - 1) / g_wga_chunk);
            jobs.push_back(jb);
        }
    }
    sa_call_stats tot;
    memset(&tot, 0, sizeof(tot));
    std::mutex mu;
    run_parallel(jobs.size(), threads, [&](size_t j) {  // (pool threads of the engine: nothing is created per interval)
        Job& jb = jobs[j];
        sa_seed_and_filter_chunks(jb.a, jb.b, jb.rev, buffer, jb.res, jb.n);
        std::lock_guard<std::mutex> lk(mu);
        stats_add(tot, t_stats);
    });
    size_t cnt[2] = {0, 0};
    for (const Job& jb : jobs)
        for (int c = 0; c < jb.k; c++) if (jb.n[c] > 1) cnt[jb.rev] += jb.n[c] - 1;
    sa_segment_pair* dst[2];
    for (int r = 0; r < 2; r++) dst[r] = (sa_segment_pair*)malloc(std::max<size_t>(cnt[r], 1) * sizeof(sa_segment_pair));
    size_t off[2] = {0, 0};
    for (Job& jb : jobs)
        for (int c = 0; c < jb.k; c++) {
            if (jb.n[c] > 1) {
                memcpy(dst[jb.rev] + off[jb.rev], jb.res[c] + 1, (jb.n[c] - 1) * sizeof(sa_segment_pair));
                off[jb.rev] += jb.n[c] - 1;
            }
            free(jb.res[c]);
        }
    *out_fw = dst[0]; *n_fw = cnt[0];
    *out_rc = dst[1]; *n_rc = cnt[1];
    if (totals) *totals = tot;
    return cnt[0] + cnt[1];
}

// A list of independent calls -- each up to sa_max_chunks_per_call() consecutive chunks of one strand -- run with `threads` of
// them in flight on the engine's worker pool.  This is the unit a multi-GPU host deals out: any subset of the calls of a pass
// may run on any device.  results[i]: the HSPs of call i, its chunks concatenated in order, headers removed.
size_t sa_seed_calls(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, sa_call_result* results,
                     sa_call_stats* totals) {
    require_proc("SeedCalls", buffer);
    sa_call_stats tot;
    memset(&tot, 0, sizeof(tot));
    std::mutex mu;
    run_parallel(num_calls, threads, [&](size_t i) {
        sa_segment_pair* res[SA_MAX_CHUNKS];
        size_t cnt[SA_MAX_CHUNKS];
        const sa_call_desc& c = calls[i];
        const int K = c.end > c.start ? (int)(((uint64_t)c.end - c.start + g_wga_chunk - 1) / g_wga_chunk) : 0;
        sa_seed_and_filter_chunks(c.start, c.end, c.rev, buffer, res, cnt);
        size_t n = 0;
        for (int k = 0; k < K; k++)
            if (cnt[k] > 0) n += cnt[k] - 1;
        sa_segment_pair* out = (sa_segment_pair*)malloc(std::max<size_t>(n, 1) * sizeof(sa_segment_pair));
        size_t off = 0;
        for (int k = 0; k < K; k++) {
            if (cnt[k] > 1) {
                memcpy(out + off, res[k] + 1, (cnt[k] - 1) * sizeof(sa_segment_pair));
                off += cnt[k] - 1;
            }
            free(res[k]);
        }
        results[i].device = K > 0 ? t_stats.device : 0;
        results[i].reserved = 0;
        results[i].hsps = out;
        results[i].num_hsps = n;
        results[i].num_hits = K > 0 ? t_stats.num_hits : 0;  // (an empty call leaves the thread's statistics of its previous call alone)
        std::lock_guard<std::mutex> lk(mu);
        stats_add(tot, t_stats);
    });
    size_t total = 0;
    for (size_t i = 0; i < num_calls; i++) total += results[i].num_hsps;
    if (totals) *totals = tot;
    return total;
}

// The seed hits of every call of a list, lookup only (see the header): what a multi-GPU host weighs the calls of a pass with.
void sa_count_chunk_hits(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, uint64_t* hits, uint64_t* chunk_hits) {
    require_proc("CountCallHits", buffer);
    run_parallel(num_calls, threads, [&](size_t i) {
        const sa_call_desc& c = calls[i];
        hits[i] = 0;
        uint64_t* per_chunk = chunk_hits ? chunk_hits + i * (size_t)SA_MAX_CHUNKS : nullptr;
        const uint32_t chunk = g_wga_chunk;
        const int K = c.end > c.start ? (int)(((uint64_t)c.end - c.start + chunk - 1) / chunk) : 0;
        if (K == 0) return;
        if (K > SA_MAX_CHUNKS) {
            fprintf(stderr, "Error: CountCallHits takes at most %d chunks per call\n", SA_MAX_CHUNKS);
            exit(1);
        }
        Slot* sl = acquire_slot();
        DevCtx* dc = sl->ctx;
        const int rev = c.rev ? 1 : 0;
        const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
        const uint32_t qlen = g_query_len[buffer];
        const uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
        const uint32_t send = std::min(c.end, lim);
        uint32_t bpos[SA_MAX_CHUNKS + 1];
        for (int k = 0; k <= K; k++) bpos[k] = (uint32_t)std::min<uint64_t>((uint64_t)c.start + (uint64_t)k * chunk, send);
        const PackedBuf* q4 = rev ? &dc->query4_rc[buffer] : &dc->query4[buffer];
        uint32_t ns = 0xFFFFFFFFu, words = 0;
        if (td_eligible(dc, q4, rev ? &dc->query2_rc[buffer] : &dc->query2[buffer], rev ? &dc->query2[buffer] : &dc->query2_rc[buffer]))
            ns = td_front(dc, sl, q, K, bpos, 0, &words);
        if (ns != 0xFFFFFFFFu) {
            if (ns > 0) hits[i] = sl->h_td_plan[K - 1].hit_base + sl->h_td_plan[K - 1].num_hits;
            if (per_chunk) for (int k = 0; k < K; k++) per_chunk[k] = ns > 0 ? sl->h_td_plan[k].num_hits : 0;
            prof_flush(sl);
            release_slot(sl);
            return;
        }
        // no table-direct lookup for this call (general path, MAX_HITS split): the call's own statistics
        prof_flush(sl);
        release_slot(sl);
        sa_segment_pair* res[SA_MAX_CHUNKS];
        size_t cnt[SA_MAX_CHUNKS];
        sa_seed_and_filter_chunks(c.start, c.end, rev, buffer, res, cnt);
        for (int k = 0; k < K; k++) {
            if (per_chunk) per_chunk[k] = cnt[k] ? (uint64_t)(uint32_t)res[k][0].score : 0;  // (header: the chunk's hit count, :806-809)
            free(res[k]);
        }
        hits[i] = t_stats.num_hits;
    });
}
void sa_count_call_hits(const sa_call_desc* calls, size_t num_calls, uint32_t buffer, int threads, uint64_t* hits) {
    sa_count_chunk_hits(calls, num_calls, buffer, threads, hits, nullptr);
}
uint32_t sa_get_wga_chunk(void) { return g_wga_chunk; }

size_t sa_device_make_seeds(uint32_t start, uint32_t end, int rev, uint32_t buffer, uint64_t* dst, size_t cap) {
    require_proc("DeviceMakeSeeds", buffer);
    Slot* sl = acquire_slot();
    DevCtx* dc = sl->ctx;
    const uint8_t* q = rev ? dc->query_rc[buffer].codes : dc->query[buffer].codes;
    uint32_t qlen = g_query_len[buffer];
    uint32_t lim = qlen >= g_seed_size ? qlen - g_seed_size + 1 : 0;
    if (end > lim) end = lim;
    uint32_t ns = device_seeds(sl, q, start, end);
    size_t ncopy = std::min<size_t>(ns, cap);
    if (ncopy) {
        check_memcpy(hipMemcpyAsync(dst, sl->seeds.p, ncopy * sizeof(uint64_t), hipMemcpyDeviceToHost, sl->stream), "seeds d2h");
        check_sync(sl->stream, "seeds d2h");
    }
    prof_flush(sl);
    release_slot(sl);
    return ns;
}

}  // extern "C"
