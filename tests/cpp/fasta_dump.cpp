// fasta_dump.cpp -- the C++ hosts' FASTA reader (segalign_amd/host/host_common.hpp::read_fasta) behind the output format of
// oracle/_ref/kseq_dump: one line per record, name <TAB> sequence length <TAB> sequence.  Test program (tests/test_fasta_kseq.py).
#include "host_common.hpp"

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    read_fasta(argv[1], [](const std::string& name, const std::string& seq) {
        printf("%s\t%lu\t", name.c_str(), (unsigned long)seq.size());
        fwrite(seq.data(), 1, seq.size(), stdout);
        fputc('\n', stdout);
    });
    return 0;
}
