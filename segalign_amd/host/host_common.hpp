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

// ---- FASTA: the records klib's kseq_read returns (common/kseq.h:177-218, instantiated at src/main.cpp:21, looped over at :336 / :494 until it
//      fails), restated as the state machine it is -- not a line filter: the first '>' or '@' ANYWHERE opens the first header, the name ends at the
//      first isspace(), the sequence runs until a LINE that starts with '>', '@' or '+', empty lines are skipped, one trailing CR is taken off the
//      accumulated sequence after every line (when it is longer than one character, :141), '+' opens a quality block that swallows lines until it
//      is as long as the sequence, a block of another length ends the whole read.  Pinned to the real header: tests/golden/kseq_golden.json
//      (the header itself compiled as it lies) and a fuzz against that binary (tests/test_fasta_kseq.py). ----
struct GzChars {
    gzFile f;
    unsigned char buf[1 << 16];
    int begin = 0, end = 0;
    bool eof = false;
    bool fill() {
        if (begin < end) return true;
        if (eof) return false;
        end = gzread(f, buf, sizeof(buf));
        begin = 0;
        if (end <= 0) { eof = true; end = 0; return false; }
        return true;
    }
    int getc() { return fill() ? buf[begin++] : -1; }
    // appends up to (not including) the next delimiter, which is consumed; false when the stream had nothing left at all (ks_getuntil2's -1)
    template <class P>
    bool getuntil(std::string& s, P is_delim, int* dret) {
        bool gotany = false;
        if (dret) *dret = 0;
        while (fill()) {
            int i = begin;
            while (i < end && !is_delim(buf[i])) i++;
            gotany = true;
            s.append((const char*)buf + begin, (size_t)(i - begin));
            begin = i + 1;
            if (i < end) { if (dret) *dret = buf[i]; break; }
        }
        return gotany;
    }
};

template <class F>
static void read_fasta(const std::string& path, F&& on_record) {
    GzChars* ks = new GzChars;
    ks->f = gzopen(path.c_str(), "r");
    if (!ks->f) die(7, "cant open file: %s", path.c_str());  // src/main.cpp:313-316
    auto is_space = [](unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
    auto is_nl = [](unsigned char ch) { return ch == '\n'; };
    std::string name, seq, rest;
    int last = 0, c;
    for (;;) {
        if (last == 0) {  // jump to the next header character (:182-186)
            while ((c = ks->getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) break;
            last = c;
        }
        name.clear();
        seq.clear();
        if (!ks->getuntil(name, is_space, &c)) break;                          // :188
        if (c != '\n') { rest.clear(); ks->getuntil(rest, is_nl, nullptr); }   // the comment (:189)
        while ((c = ks->getc()) >= 0 && c != '>' && c != '+' && c != '@') {    // :194-198
            if (c == '\n') continue;
            seq.push_back((char)c);
            if (ks->getuntil(seq, is_nl, nullptr) && seq.size() > 1 && seq.back() == '\r') seq.pop_back();
        }
        if (c == '>' || c == '@') last = c;
        if (c != '+') { on_record(name, seq); continue; }
        while ((c = ks->getc()) >= 0 && c != '\n') {}                          // FASTQ: the rest of the '+' line (:211-212)
        if (c < 0) break;
        rest.clear();
        while (ks->getuntil(rest, is_nl, nullptr)) {                           // :213
            if (rest.size() > 1 && rest.back() == '\r') rest.pop_back();
            if (rest.size() >= seq.size()) break;
        }
        last = 0;
        if (rest.size() != seq.size()) break;                                  // :216: kseq_read < 0 ends the loop of main.cpp:336
        on_record(name, seq);
    }
    gzclose(ks->f);
    delete ks;
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

