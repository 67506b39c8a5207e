// segalign_host.cpp -- host harness for the MI355X engine: FASTA in, LASTZ ".segments" files + lastz command lines out.
//
// SURVEY.md 8(f) rows 2 and 3.  The reference host (src/main.cpp + src/seeder.cpp + src/segment_printer.cpp) needs
// TBB, boost and kentUtils and "stays intact" upstream; this program is the small owned driver that walks the same
// engine boundary in the same order so that the whole path can be run and measured end to end without them:
//
//   sequence arenas     src/main.cpp:300-549   records joined by '&', blocks closed once they exceed seq_block_size,
//                                              reverse-complement arena per query block, *_block<k>.name files
//   work plan           src/main.cpp:383-393   10 Mbp intervals per query block ; src/seeder.cpp:48-51 250 kbp chunks
//   engine call order   src/main.cpp:297-298,613-621,649-685,743 (InitializeInterface, InitializeProcessor, per target
//                       block: ClearRef/SendRef/GenerateSeedPosTable, query blocks through BUFFER_DEPTH=2 device buffers)
//   seeder body         src/seeder.cpp:12-127  per interval: plus strand chunks, then minus strand chunks in rc coordinates
//   segment printer     src/segment_printer.cpp:11-173  tmp<i>.block<q>.r<rstart>.{plus,minus}.segments, 1-based,
//                       minus strand emitted in reverse vector order, one lastz command line per file on stdout
//
// Differences by design: std::thread workers instead of a TBB flow graph; the next query block is uploaded by a
// background thread while the current one is processed (the reference's reader lambda does the same through its
// buffer state machine, src/main.cpp:649-685); seeds are generated on the device by default (--host-seeding restores
// the reference's host loop; both give identical files).  Only the engine's C-ABI is used (include/segalign_amd.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "segalign_amd.h"
#include "host_common.hpp"

struct Config {  // src/graph.h:32-76 with the defaults of src/main.cpp:61-124
    std::string target, query, data_folder = "./", outdir = ".";
    std::string strand = "both", seed_shape = "12of19", ambiguous = "", scoring_file = "", output_format = "maf-";
    uint32_t step = 1;
    bool transition = true, noentropy = false, gapped = true, notrivial = false, debug = false, host_seeding = false;
    int xdrop = 910, hspthresh = 3000, ydrop = 9430, gappedthresh = -1;
    uint32_t wga_chunk = 250000, lastz_interval = 10000000, seq_block_size = 500000000;
    int num_gpu = -1, num_threads = 0;
    // derived
    std::string shape;
    uint32_t seed_size = 19;
    int kmer_size = 12;
};
static Config cfg;

struct Arena {  // common/DRAM.h: one contiguous buffer per sequence set
    std::string buf;
    std::vector<std::string> chr_name;
    std::vector<size_t> chr_start;
    std::vector<uint32_t> chr_len;
    std::vector<size_t> block_start;
    std::vector<uint32_t> block_len;
};
struct Interval { uint32_t start, end; };

static Arena R, Q;
static std::string Qrc;  // query_rc_DRAM
static std::vector<std::string> rc_chr_name;
static std::vector<size_t> rc_chr_start;
static std::vector<uint32_t> rc_chr_len;
static std::vector<std::vector<Interval>> q_intervals;  // per query block
static int shape_pos[32], shape_weight, transition_pos[32];

static char rc_char(char c) {  // common/ntcoding.cpp:63-105
    switch (c) {
        case 'a': return 't'; case 'A': return 'T'; case 'c': return 'g'; case 'C': return 'G';
        case 'g': return 'c'; case 'G': return 'C'; case 't': return 'a'; case 'T': return 'A';
        default: return c;  // n N & stay; anything else is reported there as "Bad Nt char" -- kept as is here
    }
}

// ---- arenas + plan: src/main.cpp:320-549 -----------------------------------------------------------------------------
static void load_set(const std::string& path, Arena& A, bool is_query, const char* tag) {
    uint32_t block_no = 0, seq_block_len = 0;
    size_t seq_block_start = 0;
    std::vector<uint32_t> block_chrs;
    A.block_start.push_back(0);
    FILE* names = fopen((cfg.outdir + "/" + tag + "_block" + std::to_string(block_no) + ".name").c_str(), "w");
    auto close_block = [&](uint32_t len) {
        A.block_len.push_back(len);
        if (is_query) {
            for (int i = (int)block_chrs.size() - 1; i >= 0; i--) {  // :369-374
                uint32_t c = block_chrs[i];
                rc_chr_name.push_back(A.chr_name[c]);
                rc_chr_start.push_back(2 * seq_block_start + len - A.chr_start[c] - A.chr_len[c]);
                rc_chr_len.push_back(A.chr_len[c]);
            }
            Qrc.resize(seq_block_start + len, 'N');  // RevComp of the block, :381 / :421
            for (uint32_t i = 0; i < len; i++) Qrc[seq_block_start + i] = rc_char(A.buf[seq_block_start + len - 1 - i]);
            std::vector<Interval> iv;  // :383-393
            uint32_t end_pos = len - cfg.seed_size;
            for (uint32_t cur = 0; len > cfg.seed_size && cur < end_pos; cur += cfg.lastz_interval)
                iv.push_back({cur, std::min(end_pos, cur + cfg.lastz_interval)});
            q_intervals.push_back(iv);
        }
    };
    read_fasta(path, [&](const std::string& name, const std::string& seq) {
        fprintf(names, "%s\n", name.c_str());
        uint32_t c = (uint32_t)A.chr_name.size();
        A.chr_name.push_back(name);
        A.chr_start.push_back(A.buf.size());
        A.chr_len.push_back((uint32_t)seq.size());
        block_chrs.push_back(c);
        A.buf += seq;
        seq_block_len += (uint32_t)seq.size();
        if (seq_block_len > cfg.seq_block_size) {  // :359 / :515
            close_block(seq_block_len);
            seq_block_start = A.buf.size();
            A.block_start.push_back(seq_block_start);
            seq_block_len = 0;
            block_chrs.clear();
            block_no++;
            fclose(names);
            names = fopen((cfg.outdir + "/" + tag + "_block" + std::to_string(block_no) + ".name").c_str(), "w");
        } else {
            A.buf += '&';  // :405-409
            seq_block_len += 1;
        }
    });
    if (seq_block_len > 0) close_block(seq_block_len - 1);  // drop the trailing '&', :411-413
    else A.block_start.pop_back();
    fclose(names);
}

// ---- host seeding: common/ntcoding.cpp:43-61 + src/seeder.cpp:57-74 (only with --host-seeding) ----------------------
static uint32_t host_kmer(const char* s, size_t pos) {
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

struct Hsps { std::vector<sa_segment_pair> fw, rc; };
static std::atomic<uint64_t> g_num_seed_hits(0), g_num_hsps(0);

// seeder_body::operator(), src/seeder.cpp:12-127
static void seed_interval(size_t q_block_start, uint32_t q_len /* block_len - seed_size */, Interval iv, uint32_t buffer, Hsps& out) {
    for (int rev = 0; rev < 2; rev++) {
        if (rev == 0 && !(cfg.strand == "plus" || cfg.strand == "both")) continue;
        if (rev == 1 && !(cfg.strand == "minus" || cfg.strand == "both")) continue;
        uint32_t a = rev ? q_len - iv.end : iv.start, b = rev ? q_len - iv.start : iv.end;  // :33-34
        std::vector<sa_segment_pair>& dst = rev ? out.rc : out.fw;
        if (!cfg.host_seeding) {
            // device seeding: up to sa_max_chunks_per_call() consecutive chunks share one pass over the kernels; every
            // chunk still gets its own return vector, identical to one call per chunk
            const int kmax = sa_get_chunks_per_call();  // (chunks_per_call, or more when the resident target's seed hits are sparse)
            std::vector<sa_segment_pair*> res((size_t)kmax, nullptr);
            std::vector<size_t> n((size_t)kmax, 0);
            for (uint64_t i = a; i < b; i += (uint64_t)cfg.wga_chunk * kmax) {
                const uint32_t e = (uint32_t)std::min<uint64_t>(i + (uint64_t)cfg.wga_chunk * kmax, b);
                const int nc = (int)((e - i + cfg.wga_chunk - 1) / cfg.wga_chunk);  // chunks of THIS call: only their slots are written
                sa_seed_and_filter_chunks((uint32_t)i, e, rev, buffer, res.data(), n.data());
                for (int c = 0; c < nc; c++) {
                    if (!n[c]) continue;
                    g_num_seed_hits += (uint32_t)res[c][0].score;
                    if (n[c] > 1) {
                        dst.insert(dst.end(), res[c] + 1, res[c] + n[c]);
                        g_num_hsps += n[c] - 1;
                    }
                    sa_free_segments(res[c]);
                }
            }
            continue;
        }
        for (uint32_t i = a; i < b; i += cfg.wga_chunk) {  // the reference's loop: seed words built on the host, :57-74
            uint32_t e = std::min(i + cfg.wga_chunk, b);
            sa_segment_pair* res = nullptr;
            size_t n = 0;
            const std::string& buf = rev ? Qrc : Q.buf;
            std::vector<uint64_t> seeds;
            for (uint32_t j = i; j < e; j++) {
                uint64_t k = host_kmer(buf.data(), q_block_start + j);
                if (k != (1u << 31)) {
                    seeds.push_back((k << 32) + j);
                    if (cfg.transition)
                        for (int t = 0; t < shape_weight; t++)
                            if (transition_pos[t]) seeds.push_back(((k ^ ((uint64_t)2 << (2 * t))) << 32) + j);
                }
            }
            if (!seeds.empty()) n = sa_seed_and_filter(seeds.data(), seeds.size(), rev, buffer, &res);
            if (n) {
                g_num_seed_hits += (uint32_t)res[0].score;
                if (n > 1) {
                    dst.insert(dst.end(), res + 1, res + n);
                    g_num_hsps += n - 1;
                }
                sa_free_segments(res);
            }
        }
    }
}

static size_t chr_of(const std::vector<size_t>& starts, size_t pos) {
    return (size_t)(std::upper_bound(starts.begin(), starts.end(), pos) - starts.begin()) - 1;
}
static std::mutex io_lock;

// segment_printer_body::operator(), src/segment_printer.cpp:11-173
static void print_segments(int r_block_index, int q_block_index, size_t r_block_start, size_t q_block_start, uint32_t index,
                           const Hsps& h) {
    for (int rev = 0; rev < 2; rev++) {
        const std::vector<sa_segment_pair>& v = rev ? h.rc : h.fw;
        if (v.empty()) continue;
        const std::vector<std::string>& qn = rev ? rc_chr_name : Q.chr_name;
        const std::vector<size_t>& qs = rev ? rc_chr_start : Q.chr_start;
        std::string base = "tmp" + std::to_string(index) + ".block" + std::to_string(q_block_index) + ".r" +
                           std::to_string(r_block_start) + (rev ? ".minus" : ".plus");
        std::string seg_name = base + ".segments";
        FILE* f = fopen((cfg.outdir + "/" + seg_name).c_str(), "w");
        if (!f) die(7, "cant open file: %s", seg_name.c_str());
        auto emit = [&](const sa_segment_pair& e) {
            size_t seg_r = e.ref_start + r_block_start, seg_q = e.query_start + q_block_start;
            size_t ri = chr_of(R.chr_start, seg_r), qi = chr_of(qs, seg_q);
            fprintf(f, "%s\t%zu\t%zu\t%s\t%zu\t%zu\t%c\t%d\n", R.chr_name[ri].c_str(), seg_r + 1 - R.chr_start[ri],
                    seg_r + e.len + 1 - R.chr_start[ri], qn[qi].c_str(), seg_q + 1 - qs[qi], seg_q + e.len + 1 - qs[qi],
                    rev ? '-' : '+', e.score);
        };
        if (!rev) for (size_t i = 0; i < v.size(); i++) emit(v[i]);
        else for (size_t i = v.size(); i-- > 0;) emit(v[i]);  // :130: reverse vector order on the minus strand
        fclose(f);
        if (cfg.gapped) {  // :96-113 / :151-168
            std::string cmd = "lastz " + cfg.data_folder + "ref.2bit[nameparse=darkspace][multiple][subset=ref_block" +
                              std::to_string(r_block_index) + ".name] " + cfg.data_folder +
                              "query.2bit[nameparse=darkspace][subset=query_block" + std::to_string(q_block_index) +
                              ".name] --format=" + cfg.output_format + " --ydrop=" + std::to_string(cfg.ydrop) +
                              " --gappedthresh=" + std::to_string(cfg.gappedthresh) + " --strand=" + (rev ? "minus" : "plus");
            if (cfg.ambiguous != "") cmd += " --ambiguous=" + cfg.ambiguous;
            if (cfg.notrivial) cmd += " --notrivial";
            if (cfg.scoring_file != "") cmd += " --scoring=" + cfg.scoring_file;
            cmd += " --segments=" + seg_name + " --output=" + base + "." + cfg.output_format + " 2> " + base + ".err";
            std::lock_guard<std::mutex> lk(io_lock);
            printf("%s\n", cmd.c_str());
        }
    }
}

static void usage() {
    fprintf(stderr,
            "Usage: segalign_host target.fa query.fa [data_folder/] [options]\n"
            "  --strand=plus|minus|both  --seed=12of19|14of22|<pattern of 0/1>  --step=N  --notransition\n"
            "  --xdrop=N --hspthresh=N --noentropy --nogapped --ydrop=N --gappedthresh=N --notrivial --format=F\n"
            "  --ambiguous=x|n|iupac[,reward,penalty] --scoring=FILE\n"
            "  --wga_chunk=N --lastz_interval=N --seq_block_size=N --num_gpu=N --num_threads=N --outdir=DIR\n"
            "  --host-seeding (build seed vectors on the host like src/seeder.cpp) --debug\n");
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // the host's own choice, before the first HIP call: one hardware queue per engine slot
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string v;
        const char* a = argv[i];
        if (a[0] != '-') pos.push_back(a);
        else if (!strcmp(a, "--help")) { usage(); return 0; }
        else if (opt(a, "--strand", v)) cfg.strand = v;
        else if (opt(a, "--seed", v)) cfg.seed_shape = v;
        else if (opt(a, "--step", v)) cfg.step = (uint32_t)atoi(v.c_str());
        else if (!strcmp(a, "--notransition")) cfg.transition = false;
        else if (opt(a, "--xdrop", v)) cfg.xdrop = atoi(v.c_str());
        else if (opt(a, "--hspthresh", v)) cfg.hspthresh = atoi(v.c_str());
        else if (!strcmp(a, "--noentropy")) cfg.noentropy = true;
        else if (!strcmp(a, "--nogapped")) cfg.gapped = false;
        else if (opt(a, "--ydrop", v)) cfg.ydrop = atoi(v.c_str());
        else if (opt(a, "--gappedthresh", v)) cfg.gappedthresh = atoi(v.c_str());
        else if (!strcmp(a, "--notrivial")) cfg.notrivial = true;
        else if (opt(a, "--format", v)) cfg.output_format = v;
        else if (opt(a, "--ambiguous", v)) cfg.ambiguous = v;
        else if (opt(a, "--scoring", v)) cfg.scoring_file = v;
        else if (opt(a, "--wga_chunk", v)) cfg.wga_chunk = (uint32_t)atol(v.c_str());
        else if (opt(a, "--lastz_interval", v)) cfg.lastz_interval = (uint32_t)atol(v.c_str());
        else if (opt(a, "--seq_block_size", v)) cfg.seq_block_size = (uint32_t)atol(v.c_str());
        else if (opt(a, "--num_gpu", v)) cfg.num_gpu = atoi(v.c_str());
        else if (opt(a, "--num_threads", v)) cfg.num_threads = atoi(v.c_str());
        else if (opt(a, "--outdir", v)) cfg.outdir = v;
        else if (!strcmp(a, "--host-seeding")) cfg.host_seeding = true;
        else if (!strcmp(a, "--debug")) cfg.debug = true;
        else { fprintf(stderr, "unknown option %s\n", a); usage(); return 1; }
    }
    if (pos.size() < 2) {
        fprintf(stderr, "You must specify a target file and a query file\n");
        usage();
        return 1;
    }
    cfg.target = pos[0];
    cfg.query = pos[1];
    if (pos.size() > 2) cfg.data_folder = pos[2];
    if (cfg.gappedthresh < 0) cfg.gappedthresh = cfg.hspthresh;  // src/main.cpp:182-183
    // seed shape, src/main.cpp:160-180
    if (cfg.seed_shape == "12of19") cfg.shape = "TTT0T00TT00T0T0TTTT";
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
    cfg.num_gpu = sa_initialize_interface(cfg.num_gpu);                                                        // main.cpp:297
    sa_generate_shape_pos(cfg.shape.c_str());                                                                 // main.cpp:180
    {   // one engine slot per seeder thread, up to four calls in flight per device (more only queue behind the filter kernels)
        char slots[16];
        snprintf(slots, sizeof(slots), "%d", std::max(2, std::min(4, cfg.num_threads)));
        setenv("SEGALIGN_AMD_SLOTS", slots, 0);
    }
    sa_initialize_processor(cfg.transition, cfg.wga_chunk, cfg.seed_size, sub_mat, cfg.xdrop, cfg.hspthresh, cfg.noentropy);  // :298

    auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "\nReading query file ...\n");
    load_set(cfg.query, Q, true, "query");
    fprintf(stderr, "\nReading target file ...\n");
    load_set(cfg.target, R, false, "ref");
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "\nStart alignment ...\n");

    double table_ms = 0;
    uint64_t query_bases_done = 0;
    for (size_t rb = 0; rb < R.block_len.size(); rb++) {
        fprintf(stderr, "\nSending reference block %zu ...\n", rb);
        if (rb > 0) sa_clear_ref();                                                                            // :613
        sa_send_ref_write_request(R.buf.data(), R.block_start[rb], R.block_len[rb]);                           // :615
        auto ta = std::chrono::steady_clock::now();
        sa_generate_seed_pos_table(R.buf.data(), R.block_start[rb], R.block_len[rb], cfg.step, (int)cfg.seed_size, cfg.kmer_size);  // :621
        table_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();

        const size_t nqb = Q.block_len.size();
        std::thread uploader;
        auto upload = [&](size_t qb) {  // :659-661 / :680-681
            uint32_t buffer = (uint32_t)(qb % SA_BUFFER_DEPTH);
            if (rb > 0 || qb >= SA_BUFFER_DEPTH) sa_clear_query(buffer);
            sa_send_query_write_request(Q.buf.data(), Q.block_start[qb], Q.block_len[qb], buffer);
        };
        if (nqb > 0) upload(0);
        for (size_t qb = 0; qb < nqb; qb++) {
            if (uploader.joinable()) uploader.join();
            if (qb + 1 < nqb) uploader = std::thread(upload, qb + 1);  // next block into the other device buffer while this one runs
            const uint32_t buffer = (uint32_t)(qb % SA_BUFFER_DEPTH);
            const std::vector<Interval>& ivs = q_intervals[qb];
            const uint32_t q_len = Q.block_len[qb] - cfg.seed_size;                                             // :708
            std::atomic<size_t> next(0);
            auto worker = [&]() {
                for (;;) {
                    size_t i = next.fetch_add(1);
                    if (i >= ivs.size()) return;
                    fprintf(stderr, "Query block %zu, interval %zu/%zu (%u:%u) with buffer %u\n", qb, i + 1, ivs.size(),
                            ivs[i].start, ivs[i].end, buffer);                                                   // seeder.cpp:45
                    Hsps h;
                    seed_interval(Q.block_start[qb], q_len, ivs[i], buffer, h);
                    print_segments((int)rb, (int)qb, R.block_start[rb], Q.block_start[qb], (uint32_t)i + 1, h);
                }
            };
            std::vector<std::thread> pool;
            int nt = (int)std::min<size_t>((size_t)cfg.num_threads, std::max<size_t>(ivs.size(), 1));
            for (int t = 0; t < nt; t++) pool.emplace_back(worker);
            for (auto& t : pool) t.join();
            query_bases_done += Q.block_len[qb];
        }
        if (uploader.joinable()) uploader.join();
    }
    auto t2 = std::chrono::steady_clock::now();
    sa_shutdown_processor();                                                                                   // :743
    if (cfg.debug) {  // src/main.cpp:617-629,745-752
        double load_s = std::chrono::duration<double>(t1 - t0).count(), run_s = std::chrono::duration<double>(t2 - t1).count();
        fprintf(stderr, "Time elapsed (loading sequences): %.3f sec\n", load_s);
        fprintf(stderr, "Time elapsed (seed position table create on GPU): %.1f msec\n", table_ms);
        fprintf(stderr, "Time elapsed (complete pipeline): %.3f sec  (%.4f Gbp of query x %zu target block(s) per sec)\n", run_s,
                query_bases_done / run_s / 1e9, R.block_len.size());
        fprintf(stderr, "#seed hits: %lu \n#HSPs: %lu \n", (unsigned long)g_num_seed_hits.load(), (unsigned long)g_num_hsps.load());
    }
    return 0;
}
