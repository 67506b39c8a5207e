// host_common.hpp -- pieces shared by the two host harnesses (segalign_host.cpp, segalign_rm_host.cpp):
// FASTA reading, the substitution matrix of src/main.cpp:187-268 (identical in repeat_masker_src/main.cpp:163-247)
// and option parsing.  Header-only, plain C++11 + zlib.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

static void die(int code, const char* fmt, const char* a = "") {
    fprintf(stderr, fmt, a);
    fprintf(stderr, "\n");
    exit(code);
}

// ---- FASTA (the reference uses klib's kseq over zlib; record name = first word of the header) ---------------------
template <class F>
static void read_fasta(const std::string& path, F&& on_record) {
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) die(7, "cant open file: %s", path.c_str());  // src/main.cpp:313-316
    std::string name, seq, line;
    bool have = false;
    char buf[1 << 16];
    auto flush = [&]() { if (have) on_record(name, seq); };
    while (gzgets(f, buf, sizeof(buf))) {
        size_t n = strlen(buf);
        bool eol = n && buf[n - 1] == '\n';
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
        if (line.empty() && buf[0] == '>') {
            flush();
            have = true;
            seq.clear();
            const char* p = buf + 1;
            size_t k = 0;
            while (p[k] && p[k] != ' ' && p[k] != '\t') k++;
            name.assign(p, k);
            // (header lines longer than the buffer are truncated to their first word, which is all that is used)
            while (!eol && gzgets(f, buf, sizeof(buf))) { size_t m = strlen(buf); eol = m && buf[m - 1] == '\n'; }
        } else {
            seq.append(buf, n);
        }
        line.clear();
    }
    flush();
    gzclose(f);
}

static void build_sub_mat(int* m, const std::string& ambiguous, const std::string& scoring_file, int xdrop) {  // src/main.cpp:187-268
    int reward = -100, penalty = -100;
    const int fill = -100, bad = -1000;
    std::string field = "x";
    if (!ambiguous.empty()) {
        std::vector<std::string> parts;
        size_t p = 0;
        while (true) { size_t q = ambiguous.find(',', p); parts.push_back(ambiguous.substr(p, q - p)); if (q == std::string::npos) break; p = q + 1; }
        field = parts[0];
        if (parts.size() == 3) { reward = atoi(parts[1].c_str()); penalty = -atoi(parts[2].c_str()); }
        else if (ambiguous == "n" || ambiguous == "iupac") { reward = 0; penalty = 0; }
    }
    for (int i = 0; i < 64; i++) m[i] = 0;
    if (!scoring_file.empty()) return;  // :205: a scoring file leaves the matrix all-zero (hazard H10)
    const int core[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i * 8 + j] = core[i][j];
    for (int i = 0; i < 4; i++) m[i * 8 + 4] = m[4 * 8 + i] = bad;
    m[4 * 8 + 4] = bad;
    bool n_amb = field == "n" || field == "iupac";
    for (int i = 0; i < 5; i++) m[i * 8 + 5] = m[5 * 8 + i] = n_amb ? penalty : bad;
    m[5 * 8 + 5] = n_amb ? reward : bad;
    if (field == "iupac") {
        for (int i = 0; i < 6; i++) m[i * 8 + 6] = m[6 * 8 + i] = penalty;
        m[6 * 8 + 6] = reward;
    } else {
        for (int i = 0; i < 4; i++) m[i * 8 + 6] = m[6 * 8 + i] = fill;
        for (int i = 4; i < 6; i++) m[i * 8 + 6] = m[6 * 8 + i] = bad;
        m[6 * 8 + 6] = fill;
    }
    for (int i = 0; i < 7; i++) m[i * 8 + 7] = m[7 * 8 + i] = -10 * xdrop;
    m[7 * 8 + 7] = -10 * xdrop;
}

static bool opt(const char* arg, const char* name, std::string& val) {
    size_t n = strlen(name);
    if (strncmp(arg, name, n) == 0 && arg[n] == '=') { val = arg + n + 1; return true; }
    return false;
}

