// compat_driver.cpp -- a miniature of the reference HOST (src/main.cpp:297-298,613-621,659-661 + src/seeder.cpp:47-121)
// that talks to the engine ONLY through the reference's own symbols (g_InitializeInterface ... g_SeedAndFilter,
// GenerateSeedPosTable), which include/segalign_amd_compat.hpp defines on top of the C-ABI.
// It is test code: tests/test_gpu_compat.py compiles it with g++, runs it on the GPU box and compares every HSP with
// the oracle.  Host threads call g_SeedAndFilter concurrently like the TBB seeder bodies do.
//
// Built twice (tests/test_compat_header.py): plain = the src/ binary's symbols; with -DCOMPAT_DRIVER_RM = the repeat
// masker's symbols of the same names (repeat_masker_src/seed_filter.h:4-14): g_SendQueryWriteRequest(),
// g_SeedAndFilter(seeds, rev, ref_start, ref_end), g_ClearQuery(), driven like repeat_masker_src/seeder.cpp:73-146 (the query
// IS the target; the minus-strand chunk is derived from the plus-strand chunk end, :118-119); <query.txt> is then ignored
// and the two extra arguments are the target window.
//
// usage: compat_driver <target.txt> <query.txt> <shape> <chunk> <transition 0|1> <threads> [rm: <ref_start> <ref_end>]
//   target/query: one block each, records already joined by '&' (src/main.cpp:343-409)
// output (stdout): for every (strand, chunk) in order:  "C <rev> <start> <end> <n_hsps> <num_hits>" then n lines
//   "<ref_start> <query_start> <len> <score>"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#ifdef COMPAT_DRIVER_RM
#define SEGALIGN_AMD_COMPAT_DEFINE_RM
#else
#define SEGALIGN_AMD_COMPAT_DEFINE
#endif
#include "segalign_amd_compat.hpp"

// ---- the part of common/ntcoding.cpp the host keeps (restated for this test host, not copied) -----------------------
int shape_pos[32];
int shape_size;
int transition_pos[32];

static int HostGenerateShapePos(const std::string& shape) {  // same contract as ntcoding.cpp:21-37
    shape_size = 0;
    for (size_t i = 0; i < shape.size(); i++)
        if (shape[i] == '1' || shape[i] == 'T') {
            transition_pos[shape_size] = shape[i] == 'T';
            shape_pos[shape_size++] = (int)i;
        }
    return shape_size;
}
static uint32_t HostKmer(const char* s, size_t pos, uint32_t span) {  // same contract as ntcoding.cpp:43-61
    uint32_t code[64];
    for (uint32_t i = 0; i < span; i++) {
        switch (s[pos + i]) {
            case 'A': code[i] = 0; break;
            case 'C': code[i] = 1; break;
            case 'G': code[i] = 2; break;
            case 'T': code[i] = 3; break;
            default: return 1u << 31;
        }
    }
    uint32_t k = 0;
    for (int i = 0; i < shape_size; i++) k = (k << 2) + code[shape_pos[i]];
    return k;
}
static std::string HostRevComp(const std::string& s) {  // same contract as ntcoding.cpp:63-105
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[s.size() - 1 - i], o = c;
        switch (c) {
            case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break;
            case 'a': o = 't'; break; case 'c': o = 'g'; break; case 'g': o = 'c'; break; case 't': o = 'a'; break;
            default: o = c;
        }
        r[i] = o;
    }
    return r;
}
static std::string slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string s = ss.str();
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
    return s;
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s target query shape chunk transition threads\n", argv[0]);
        return 1;
    }
    std::string target = slurp(argv[1]), query = slurp(argv[2]), shape = argv[3];
    const uint32_t chunk = (uint32_t)atoi(argv[4]);
    const bool transition = atoi(argv[5]) != 0;
    const int nthreads = std::max(1, atoi(argv[6]));
    const int xdrop = 910, hspthresh = 3000;
    const uint32_t span = (uint32_t)shape.size();

    // HOXD70 + L/N/X/E exactly as the reference host builds it (src/main.cpp:187-268, default --ambiguous)
    int sub_mat[64];
    {
        const int core[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
        for (int i = 0; i < 64; i++) sub_mat[i] = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) sub_mat[i * 8 + j] = core[i][j];
        for (int i = 0; i < 5; i++) sub_mat[i * 8 + 4] = sub_mat[4 * 8 + i] = -1000;
        for (int i = 0; i < 6; i++) sub_mat[i * 8 + 5] = sub_mat[5 * 8 + i] = -1000;
        for (int i = 0; i < 4; i++) sub_mat[i * 8 + 6] = sub_mat[6 * 8 + i] = -100;
        for (int i = 4; i < 6; i++) sub_mat[i * 8 + 6] = sub_mat[6 * 8 + i] = -1000;
        sub_mat[6 * 8 + 6] = -100;
        for (int i = 0; i < 8; i++) sub_mat[i * 8 + 7] = sub_mat[7 * 8 + i] = -10 * xdrop;
    }

    const int kmer_size = HostGenerateShapePos(shape);
    int ngpu = g_InitializeInterface(1);                                                              // main.cpp:297
    (void)ngpu;
    g_InitializeProcessor(transition, chunk, span, sub_mat, xdrop, hspthresh, false);                 // main.cpp:298
    g_SendRefWriteRequest(&target[0], 0, (uint32_t)target.size());                                    // main.cpp:615
    GenerateSeedPosTable(&target[0], 0, (uint32_t)target.size(), 1, (int)span, kmer_size);           // main.cpp:621
    struct Job { bool rev; uint32_t s, e; std::vector<segmentPair> out; int path; };
    std::vector<Job> jobs;
#ifdef COMPAT_DRIVER_RM
    if (argc < 9) {
        fprintf(stderr, "rm mode needs <ref_start> <ref_end>\n");
        return 1;
    }
    const uint32_t rm_ref_start = (uint32_t)strtoul(argv[7], nullptr, 10), rm_ref_end = (uint32_t)strtoul(argv[8], nullptr, 10);
    query = target;                                                                                   // rm main.cpp: one arena
    g_SendQueryWriteRequest();                                                                        // rm main.cpp:420
    std::string query_rc = HostRevComp(query);
    {
        const uint32_t L = (uint32_t)target.size(), start_pos = 0, end_pos = L - span;                // one block, one interval
        const uint32_t end_pos_rc = L - 1 - start_pos;                                                // rm seeder.cpp:46-47
        for (uint32_t i = start_pos; i < end_pos; i += chunk) {                                       // rm seeder.cpp:73-77
            const uint32_t e = std::min(i + chunk, end_pos);
            jobs.push_back({false, i, e, {}, -1});
            const uint32_t s_rc = L - 1 - e;                                                          // rm seeder.cpp:118-119
            jobs.push_back({true, s_rc, std::min(std::min(s_rc + chunk, end_pos_rc), L - span + 1), {}, -1});
        }
    }
#else
    segalign_amd_compat::query_arena() = &query[0];                                                   // query_DRAM->buffer
    g_SendQueryWriteRequest(0, (uint32_t)query.size(), 0);                                            // main.cpp:661
    std::string query_rc = HostRevComp(query);                                                        // main.cpp:372
    const uint32_t end_pos = (uint32_t)query.size() - span;                                           // main.cpp:383
    for (int rev = 0; rev < 2; rev++)
        for (uint32_t i = 0; i < end_pos; i += chunk) jobs.push_back({rev != 0, i, std::min(i + chunk, end_pos), {}, -1});
#endif

    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (;;) {
            size_t j = next.fetch_add(1);
            if (j >= jobs.size()) return;
            Job& job = jobs[j];
            const std::string& buf = job.rev ? query_rc : query;
            std::vector<uint64_t> seeds;                                                              // seeder.cpp:53-74
            for (uint32_t p = job.s; p < job.e; p++) {
                uint64_t k = HostKmer(buf.data(), p, span);
                if (k != (1u << 31)) {
                    seeds.push_back((k << 32) + p);
                    if (transition)
                        for (int t = 0; t < kmer_size; t++)
                            if (transition_pos[t] == 1) seeds.push_back(((k ^ ((uint64_t)2 << (2 * t))) << 32) + p);
                }
            }
#ifdef COMPAT_DRIVER_RM
            if (!seeds.empty()) job.out = g_SeedAndFilter(seeds, job.rev, rm_ref_start, rm_ref_end);  // rm seeder.cpp:103-105,140-142
#else
            if (!seeds.empty()) job.out = g_SeedAndFilter(seeds, job.rev, 0);                         // seeder.cpp:76-78
#endif
            if (!seeds.empty()) {  // (test introspection, not a reference symbol: which lookup path did the engine give this call?)
                sa_call_stats st;
                sa_get_last_call_stats(&st);
                job.path = st.lookup_path;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();

    for (auto& job : jobs) {
        size_t n = job.out.empty() ? 0 : job.out.size() - 1;
#ifdef COMPAT_DRIVER_RM  // 64-bit counts packed into the header (repeat_masker_src/seed_filter.cu:857-861)
        long hits = job.out.empty() ? 0 : (long)(((uint64_t)job.out[0].query_start << 32) | job.out[0].ref_start);
#else
        long hits = job.out.empty() ? 0 : job.out[0].score;
#endif
        printf("C %d %u %u %zu %ld %d\n", job.rev ? 1 : 0, job.s, job.e, n, hits, job.path);
        for (size_t i = 1; i < job.out.size(); i++)
            printf("%u %u %u %d\n", job.out[i].ref_start, job.out[i].query_start, job.out[i].len, job.out[i].score);
    }
#ifdef COMPAT_DRIVER_RM
    g_ClearQuery();
#else
    g_ClearQuery(0);
#endif
    g_ClearRef();
    g_ShutdownProcessor();                                                                            // main.cpp:743
    return 0;
}
