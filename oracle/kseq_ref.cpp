// kseq_ref.cpp -- a door onto the REAL FASTA reader of the reference: klib's kseq.h as vendored under /root/reference/common,
// instantiated exactly as src/main.cpp:21 instantiates it (KSEQ_INIT2(, gzFile, gzread)) and read the way :318,:336-341 read it.
// Built only by oracle/Makefile target `_ref`, with the header included from where it lies (never copied); needs nothing the image
// lacks (zlib).  Output: oracle/_ref/kseq_dump (git-ignored, travels to the GPU box).  This file contains no reference code.
//   kseq_dump FILE  ->  one line per record: name <TAB> sequence length <TAB> sequence
#include <zlib.h>
#include <stdio.h>
#include "kseq.h"
KSEQ_INIT2(, gzFile, gzread)

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    gzFile f = gzopen(argv[1], "r");
    if (!f) return 3;
    kseq_t* rd = kseq_init(f);
    while (kseq_read(rd) >= 0) {
        printf("%s\t%lu\t", rd->name.s ? rd->name.s : "", (unsigned long)rd->seq.l);
        fwrite(rd->seq.s, 1, rd->seq.l, stdout);
        fputc('\n', stdout);
    }
    kseq_destroy(rd);
    gzclose(f);
    return 0;
}
