// segalign_rm_host.cpp -- host harness of the repeat-masker flavour: FASTA in, tmp<i>.block<b>.intervals files out.
//
// SURVEY.md 8(f) row 4.  The reference binary (repeat_masker_src/main.cpp + seeder.cpp + segment_printer.cpp) needs TBB,
// boost and kseq; this is the small owned driver that walks the same plan on the engine's C-ABI:
//
//   arena               repeat_masker_src/main.cpp:283-309   all records joined by '&' (no trailing one)
//   plan                repeat_masker_src/main.cpp:316-436   blocks with neighbour overlap; per lastz_interval a seed range
//                                                            and a target window of `neighbor_proportion` of the intervals
//   engine call order   main.cpp:251-252,483-497,551          InitializeInterface, InitializeProcessor, per block: ClearRef/
//                                                            ClearQuery, SendRefWriteRequest, SendQueryWriteRequest(),
//                                                            GenerateSeedPosTable
//   seeder body         repeat_masker_src/seeder.cpp:28-195  ONE engine call per interval (sa_rm_mask_interval): chunks,
//                                                            strands, coverage counters and run extraction stay on the GPU
//   interval printer    repeat_masker_src/segment_printer.cpp:8-65  tmp<i>.block<b>.intervals, "chr\tstart\tend" with
//                                                            end = start + len + 1 (:56), optional --markend line
//
// --host-loop keeps the reference's structure instead (host seeding per chunk, sa_rm_seed_and_filter per chunk and
// strand, HSPs back to the host, sa_rm_coverage_intervals for the counting); both modes write identical files.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "segalign_amd.h"
#include "host_common.hpp"

struct Config {  // repeat_masker_src/graph.h:37-76 with the defaults of repeat_masker_src/main.cpp:46-83
    std::string seq_file, outdir = ".";
    std::string strand = "both", seed_shape = "12of19", ambiguous = "", scoring_file = "";
    float prop_neigh = 0.2f;
    uint32_t step = 1, M = 1;
    bool transition = true, noentropy = false, markend = false, debug = false, host_loop = false;
    int xdrop = 910, hspthresh = 3000;
    uint32_t wga_chunk = 250000, lastz_interval = 10000000, seq_block_size = 1000000000;
    int num_gpu = -1, num_threads = 0;
    std::string shape;
    uint32_t seed_size = 19;
    int kmer_size = 12;
};
static Config cfg;

struct Task {  // struct seed_interval + seq_block, repeat_masker_src/graph.h:80-93
    int block_index;
    size_t block_start;
    uint32_t block_len, start, end, ref_start, ref_end, num_invoked, num_intervals;
};

static std::string seq;      // seq_DRAM
static std::string seq_rc;   // seq_rc_DRAM (only --host-loop reads it)
static std::vector<std::string> chr_name;
static std::vector<size_t> chr_start;
static std::vector<uint32_t> chr_len;
static int shape_pos[32], shape_weight, transition_pos[32];

// repeat_masker_src/main.cpp:316-436
static std::vector<Task> make_plan(size_t seq_len) {
    if (cfg.seq_block_size == 1000000000u) cfg.seq_block_size -= cfg.seq_block_size % cfg.lastz_interval;  // :255-258
    uint32_t total_query_intervals = (uint32_t)ceil((float)seq_len / cfg.lastz_interval);
    uint32_t num_neigh = (uint32_t)ceil((float)cfg.prop_neigh * total_query_intervals);
    uint32_t left_intervals = (uint32_t)ceil((float)(num_neigh - 1) / 2);
    uint32_t right_intervals = num_neigh - 1 - left_intervals;
    uint32_t left_overlap = left_intervals * cfg.lastz_interval;
    uint32_t right_overlap = right_intervals * cfg.lastz_interval;
    uint32_t max_interval_seq_len = left_overlap + cfg.lastz_interval + right_overlap;
    if (cfg.debug)
        fprintf(stderr, "len: %zu lastz_interval: %u\ntotal_intervals: %u neigh_intervals: %u\nleft_intervals: %u 1 right_intervals: %u\n",
                seq_len, cfg.lastz_interval, total_query_intervals, num_neigh, left_intervals, right_intervals);
    std::vector<Task> plan;
    int block_index = 0;
    for (size_t l = 0; l < seq_len; l += cfg.seq_block_size) {
        size_t bstart = l < left_overlap ? l : l - left_overlap;
        uint32_t blen;
        if (l + cfg.seq_block_size + right_overlap > seq_len) blen = (uint32_t)(seq_len - bstart);
        else blen = (uint32_t)(l - bstart + cfg.seq_block_size) + right_overlap;
        uint32_t start_pos = (uint32_t)(l - bstart), end_pos;
        if (blen < cfg.seq_block_size) end_pos = start_pos + blen - (uint32_t)(l - bstart) - cfg.seed_size;
        else end_pos = start_pos + cfg.seq_block_size - cfg.seed_size;
        size_t first = plan.size();
        while (start_pos < end_pos) {
            Task t;
            t.block_index = block_index;
            t.block_start = bstart;
            t.block_len = blen;
            t.start = start_pos;
            t.end = std::min(end_pos, start_pos + cfg.lastz_interval);
            const bool left_limit = t.start < left_overlap;
            const bool right_limit = (t.end + right_overlap) > blen;
            if (left_limit) {
                t.ref_start = 0;
                t.ref_end = right_limit ? blen : (max_interval_seq_len > blen ? blen : max_interval_seq_len);
            } else if (right_limit) {
                t.ref_end = blen;
                t.ref_start = blen < max_interval_seq_len ? 0 : blen - max_interval_seq_len;
            } else {
                t.ref_start = t.start - left_overlap;
                t.ref_end = t.end + right_overlap;
            }
            plan.push_back(t);
            start_pos += cfg.lastz_interval;
        }
        for (size_t i = first; i < plan.size(); i++) {
            plan[i].num_invoked = (uint32_t)(i - first + 1);  // :528-535
            plan[i].num_intervals = (uint32_t)(plan.size() - first);
        }
        block_index++;
    }
    return plan;
}

static char rc_char(char c, bool& keep) {  // common/ntcoding.cpp:63-105: characters outside ACGTacgtNn& are dropped
    keep = true;
    switch (c) {
        case 'a': return 't'; case 'A': return 'T'; case 'c': return 'g'; case 'C': return 'G';
        case 'g': return 'c'; case 'G': return 'C'; case 't': return 'a'; case 'T': return 'A';
        case 'n': case 'N': case '&': return c;
        default: keep = false; return c;
    }
}

static uint32_t host_kmer(const char* s, size_t pos) {  // common/ntcoding.cpp:43-61: any non-ACGT in the span is invalid
    uint32_t code[32];
    for (uint32_t i = 0; i < cfg.seed_size; i++) {
        switch (s[pos + i]) {
            case 'A': code[i] = 0; break; case 'C': code[i] = 1; break;
            case 'G': code[i] = 2; break; case 'T': code[i] = 3; break;
            default: return 1u << 31;
        }
    }
    uint32_t k = 0;
    for (int i = 0; i < shape_weight; i++) k = (k << 2) + code[shape_pos[i]];
    return k;
}

static void host_seeds(const char* buf, size_t base, uint32_t s0, uint32_t s1, std::vector<uint64_t>& v) {  // seeder.cpp:84-101
    v.clear();
    for (uint32_t j = s0; j < s1; j++) {
        uint64_t k = host_kmer(buf, base + j);
        if (k == (1u << 31)) continue;
        v.push_back((k << 32) + j);
        if (cfg.transition)
            for (int t = 0; t < shape_weight; t++)
                if (transition_pos[t]) v.push_back(((k ^ ((uint64_t)2 << (2 * t))) << 32) + j);
    }
}

static std::atomic<uint64_t> g_num_seeds(0), g_num_hits(0), g_num_hsps(0);

// seeder_body::operator(), repeat_masker_src/seeder.cpp:28-195
static std::vector<sa_interval> mask_interval(const Task& t) {
    const int strands = cfg.strand == "plus" ? SA_STRAND_PLUS : cfg.strand == "minus" ? SA_STRAND_MINUS : SA_STRAND_BOTH;
    sa_interval* iv = nullptr;
    size_t n;
    if (!cfg.host_loop) {
        uint64_t tot[3];
        n = sa_rm_mask_interval(t.start, t.end, t.ref_start, t.ref_end, strands, cfg.M, &iv, tot);
        g_num_seeds += tot[0]; g_num_hits += tot[1]; g_num_hsps += tot[2];
    } else {
        const uint32_t end_pos_rc = t.block_len - 1 - t.start;
        const size_t rc_block_start = seq.size() - 1 - t.block_start - (t.block_len - 1);  // :48
        const uint32_t lim = t.block_len - cfg.seed_size + 1;
        std::vector<sa_segment_pair> all;
        std::vector<uint64_t> seeds;
        for (uint64_t i = t.start; i < t.end; i += cfg.wga_chunk) {
            const uint32_t start = (uint32_t)i, end = (uint32_t)std::min<uint64_t>(i + cfg.wga_chunk, t.end);
            for (int rev = 0; rev < 2; rev++) {
                if (!(strands & (rev ? SA_STRAND_MINUS : SA_STRAND_PLUS))) continue;
                uint32_t s0 = start, s1 = end;
                if (rev) { s0 = t.block_len - 1 - end; s1 = (uint32_t)std::min<uint64_t>((uint64_t)s0 + cfg.wga_chunk, end_pos_rc); }  // :118-119
                s1 = std::min(s1, lim);
                host_seeds(rev ? seq_rc.data() : seq.data(), rev ? rc_block_start : t.block_start, s0, s1, seeds);
                if (seeds.empty()) continue;
                sa_segment_pair* out = nullptr;
                size_t m = sa_rm_seed_and_filter(seeds.data(), seeds.size(), rev, t.ref_start, t.ref_end, &out);
                g_num_seeds += seeds.size();
                g_num_hits += ((uint64_t)out[0].query_start << 32) + out[0].ref_start;  // :107
                g_num_hsps += m - 1;
                all.insert(all.end(), out + 1, out + m);
                sa_free_segments(out);
            }
        }
        n = sa_rm_coverage_intervals(all.data(), all.size(), t.block_len, cfg.M, &iv);
    }
    std::vector<sa_interval> v(iv, iv + n);
    sa_free_intervals(iv);
    return v;
}

// interval_printer_body::operator(), repeat_masker_src/segment_printer.cpp:8-65
static void print_intervals(const Task& t, const std::vector<sa_interval>& ivs) {
    if (ivs.empty()) return;
    auto chr_of = [&](size_t pos) { return (size_t)(std::upper_bound(chr_start.begin(), chr_start.end(), pos) - chr_start.begin() - 1); };
    size_t c = chr_of(t.block_start);
    size_t c_start = chr_start[c], c_end = c_start + chr_len[c];
    std::string fn = cfg.outdir + "/tmp" + std::to_string(t.num_invoked) + ".block" + std::to_string(t.block_index) + ".intervals";
    FILE* f = fopen(fn.c_str(), "w");
    if (!f) die(8, "cant write file: %s", fn.c_str());
    for (const sa_interval& e : ivs) {
        size_t q = t.block_start + e.query_start;
        if (q < c_start || q >= c_end) {
            c = chr_of(q);
            c_start = chr_start[c];
            c_end = c_start + chr_len[c];
        }
        fprintf(f, "%s\t%lu\t%lu\n", chr_name[c].c_str(), (unsigned long)(q - c_start), (unsigned long)(q + e.len + 1 - c_start));  // :56
    }
    if (cfg.markend) fprintf(f, "# segalign_repeat_masker end-of-file\n");
    fclose(f);
}

static void usage() {
    fprintf(stderr,
            "Usage: segalign_rm_host seq.fa [options]\n"
            "  --strand=plus|minus|both --neighbor_proportion=F --seed=12of19|14of22|<0/1 pattern> --step=N --notransition\n"
            "  --xdrop=N --hspthresh=N --noentropy --M=N --markend --ambiguous=x|n|iupac[,reward,penalty] --scoring=FILE\n"
            "  --wga_chunk_size=N --lastz_interval_size=N --seq_block_size=N --num_gpu=N --num_threads=N --outdir=DIR\n"
            "  --host-loop (seeds, HSPs and the chunk loop on the host like repeat_masker_src/seeder.cpp) --debug\n"
            "  --plan-only=SEQ_LEN (print the block / interval plan for a sequence of that length and leave)\n");
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // the host's own choice, before the first HIP call: one hardware queue per engine slot
    std::vector<std::string> pos;
    unsigned long long plan_only = 0;
    for (int i = 1; i < argc; i++) {
        std::string v;
        const char* a = argv[i];
        if (a[0] != '-') pos.push_back(a);
        else if (!strcmp(a, "--help")) { usage(); return 0; }
        else if (opt(a, "--strand", v)) cfg.strand = v;
        else if (opt(a, "--neighbor_proportion", v)) cfg.prop_neigh = (float)atof(v.c_str());
        else if (opt(a, "--seed", v)) cfg.seed_shape = v;
        else if (opt(a, "--step", v)) cfg.step = (uint32_t)atoi(v.c_str());
        else if (!strcmp(a, "--notransition")) cfg.transition = false;
        else if (opt(a, "--xdrop", v)) cfg.xdrop = atoi(v.c_str());
        else if (opt(a, "--hspthresh", v)) cfg.hspthresh = atoi(v.c_str());
        else if (!strcmp(a, "--noentropy")) cfg.noentropy = true;
        else if (opt(a, "--M", v)) cfg.M = (uint32_t)atol(v.c_str());
        else if (!strcmp(a, "--markend")) cfg.markend = true;
        else if (opt(a, "--ambiguous", v)) cfg.ambiguous = v;
        else if (opt(a, "--scoring", v)) cfg.scoring_file = v;
        else if (opt(a, "--wga_chunk_size", v)) cfg.wga_chunk = (uint32_t)atol(v.c_str());
        else if (opt(a, "--lastz_interval_size", v)) cfg.lastz_interval = (uint32_t)atol(v.c_str());
        else if (opt(a, "--seq_block_size", v)) cfg.seq_block_size = (uint32_t)atol(v.c_str());
        else if (opt(a, "--num_gpu", v)) cfg.num_gpu = atoi(v.c_str());
        else if (opt(a, "--num_threads", v)) cfg.num_threads = atoi(v.c_str());
        else if (opt(a, "--outdir", v)) cfg.outdir = v;
        else if (opt(a, "--plan-only", v)) plan_only = strtoull(v.c_str(), nullptr, 10);  // print the plan for a sequence of this length and leave (no engine, no GPU)
        else if (!strcmp(a, "--host-loop")) cfg.host_loop = true;
        else if (!strcmp(a, "--debug")) cfg.debug = true;
        else { fprintf(stderr, "unknown option %s\n", a); usage(); return 1; }
    }
    if (plan_only) {  // the plan alone (tests/test_rm_plan_golden.py holds it against the reference's own text): block index, start, length,
        if (cfg.seed_shape == "14of22") cfg.seed_size = 22;  // seed range, target window per interval task
        else if (cfg.seed_shape != "12of19") cfg.seed_size = (uint32_t)cfg.seed_shape.size();
        for (const Task& t : make_plan((size_t)plan_only))
            printf("%d %zu %u %u %u %u %u\n", t.block_index, t.block_start, t.block_len, t.start, t.end, t.ref_start, t.ref_end);
        return 0;
    }
    if (pos.size() < 1) {
        fprintf(stderr, "You must specify a sequence file \n");
        usage();
        return 1;
    }
    cfg.seq_file = pos[0];
    if (cfg.seed_shape == "12of19") cfg.shape = "TTT0T00TT00T0T0TTTT";  // repeat_masker_src/main.cpp:136-153
    else if (cfg.seed_shape == "14of22") cfg.shape = "TTT0T0TT00TT00T0T0TTTT";
    else { cfg.shape = cfg.seed_shape; for (auto& c : cfg.shape) c = (c == '1') ? 'T' : '0'; }
    cfg.seed_size = (uint32_t)cfg.shape.size();
    shape_weight = 0;
    for (size_t i = 0; i < cfg.shape.size(); i++)
        if (cfg.shape[i] == '1' || cfg.shape[i] == 'T') { transition_pos[shape_weight] = cfg.shape[i] == 'T'; shape_pos[shape_weight++] = (int)i; }
    cfg.kmer_size = shape_weight;
    if (cfg.num_threads <= 0) cfg.num_threads = std::max(2u, std::thread::hardware_concurrency());
    cfg.num_threads = std::min(cfg.num_threads, 64);

    int sub_mat[64];
    build_sub_mat(sub_mat, cfg.ambiguous, cfg.scoring_file, cfg.xdrop);
    fprintf(stderr, "Using %d threads\n", cfg.num_threads);
    cfg.num_gpu = sa_initialize_interface(cfg.num_gpu);                                                                    // :251
    sa_generate_shape_pos(cfg.shape.c_str());                                                                             // :155
    {   // one engine slot per seeder thread, up to four calls in flight per device (more only queue behind the filter kernels)
        char slots[16];
        snprintf(slots, sizeof(slots), "%d", std::max(2, std::min(4, cfg.num_threads)));
        setenv("SEGALIGN_AMD_SLOTS", slots, 0);
    }
    sa_initialize_processor(cfg.transition, cfg.wga_chunk, cfg.seed_size, sub_mat, cfg.xdrop, cfg.hspthresh, cfg.noentropy);  // :252

    auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "\nReading target file ...\n");
    read_fasta(cfg.seq_file, [&](const std::string& name, const std::string& s) {  // :283-305
        chr_name.push_back(name);
        chr_start.push_back(seq.size());
        chr_len.push_back((uint32_t)s.size());
        seq += s;
        seq += '&';
    });
    if (seq.empty()) die(9, "no sequence in %s", cfg.seq_file.c_str());
    seq.pop_back();  // :307
    if (cfg.host_loop) {  // RevComp of the whole arena, :311
        seq_rc.reserve(seq.size() + 64);
        for (size_t i = seq.size(); i > 0; i--) {
            bool keep;
            char c = rc_char(seq[i - 1], keep);
            if (keep) seq_rc.push_back(c);
        }
        seq_rc.append(64, '\0');  // window reads of the last positions stay inside the buffer
    }
    const size_t seq_len = seq.size();
    std::vector<Task> plan = make_plan(seq_len);
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "\nStart alignment ...\n");

    double table_ms = 0;
    size_t ti = 0;
    int blocks_sent = 0;
    while (ti < plan.size()) {
        const Task& b = plan[ti];
        fprintf(stderr, "\nSending block %d ...\n", b.block_index);
        if (blocks_sent > 0) { sa_clear_ref(); sa_rm_clear_query(); }                       // :483-486
        sa_send_ref_write_request(seq.data(), b.block_start, b.block_len);                  // :488
        sa_rm_send_query_write_request();                                                   // :489
        auto ta = std::chrono::steady_clock::now();
        sa_generate_seed_pos_table(seq.data(), b.block_start, b.block_len, cfg.step, (int)cfg.seed_size, cfg.kmer_size);  // :495
        table_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
        blocks_sent++;
        size_t te = ti;
        while (te < plan.size() && plan[te].block_index == b.block_index) te++;
        std::atomic<size_t> next(ti);
        auto worker = [&]() {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= te) return;
                const Task& t = plan[i];
                fprintf(stderr, "Chromosome block %d interval %u/%u (%zu:%zu) with ref (%u:%u)\n", t.block_index, t.num_invoked,
                        t.num_intervals, t.block_start + t.start, t.block_start + t.end, t.ref_start, t.ref_end);  // seeder.cpp:71
                print_intervals(t, mask_interval(t));
            }
        };
        std::vector<std::thread> pool;
        int nt = (int)std::min<size_t>((size_t)cfg.num_threads, te - ti);
        for (int k = 0; k < nt; k++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        ti = te;
    }
    auto t2 = std::chrono::steady_clock::now();
    sa_shutdown_processor();                                                                // :551
    if (cfg.debug) {  // :553-560
        fprintf(stderr, "Time elapsed (loading sequence): %.3f sec\n", std::chrono::duration<double>(t1 - t0).count());
        fprintf(stderr, "Time elapsed (seed position table create on GPU): %.1f msec\n", table_ms);
        fprintf(stderr, "Time elapsed (complete pipeline): %.3f sec \n\n", std::chrono::duration<double>(t2 - t1).count());
        fprintf(stderr, "#seeds: %lu \n#seed hits: %lu \n#HSPs: %lu \n", (unsigned long)g_num_seeds.load(), (unsigned long)g_num_hits.load(),
                (unsigned long)g_num_hsps.load());
    }
    return 0;
}
