#!/usr/bin/env python3
"""Generator of tests/golden/loader_golden.json: how the reference lays FASTA records out in its DRAM arenas, by a second route.

What runs: the reference's own text of src/main.cpp -- the query loader (from `gzFile f_rd = gzopen(cfg.query_filename ...` to
`total_query_intervals = interval_list.size();`, :312-462) and the target loader (from `f_rd = gzopen(cfg.reference_filename ...` to its
`gzclose(f_rd);`, :479-541) -- verbatim inside a function of the harness, compiled with g++ against the reference's own graph.h, store.h,
common/kseq.h and linked with the real common/ntcoding.cpp (RevComp) and zlib.  Stand-ins: TBB's header (as in make_seeder_golden.py), a
DRAM constructor with a plain buffer, and ONE edited constant -- the loaders close a block when it exceeds DEFAULT_SEQ_BLOCK_SIZE (the
#define, 500 Mbp, not the --seq_block_size option): the harness re-defines it to a few kilobases behind graph.h so that small files
make several blocks.  What the vectors add (a second route): records '&'-joined inside a block, no separator behind
a block's last record, a block closed AFTER the record that crosses the size, the minus-strand chromosome table (names reversed per
block, start = 2 * block_start + block_len - chr_start - chr_len), the per-block interval lists over [0, block_len - seed_size), the
query_block<i>.name / ref_block<i>.name files -- what tests/host_model.py::Arena and segalign_host.cpp restate (8f-2, 8f-3).

usage: python tests/golden/make_loader_golden.py   (needs /root/reference, g++, zlib)
"""
import base64
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_printer_golden import FAKE_TBB  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "loader_golden.json")

HARNESS = r'''
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include "graph.h"
#undef DEFAULT_SEQ_BLOCK_SIZE
#define DEFAULT_SEQ_BLOCK_SIZE SA_BLOCK      /* the ONE edit: see the generator's doc string */
#include "kseq.h"
#include "ntcoding.h"
#include "store.h"
KSEQ_INIT2(, gzFile, gzread)                  // src/main.cpp:21
// ---- what src/main.cpp declares at file scope (:30-60), this repository's code ----
Configuration cfg;
DRAM *ref_DRAM, *query_DRAM, *query_rc_DRAM;
DRAM::DRAM() : size(1u << 22), seqSize(0), bufferPosition(0) { buffer = (char*)calloc(size, 1); }
DRAM::~DRAM() {}
std::vector<std::string> q_chr_name, rc_q_chr_name, r_chr_name;
std::vector<uint32_t> q_chr_file_name, rc_q_chr_file_name, r_chr_file_name, q_chr_len, rc_q_chr_len, r_chr_len;
std::vector<size_t> q_chr_start, rc_q_chr_start, r_chr_start;
std::vector<uint32_t> q_buffer;
std::vector<size_t> query_block_start, ref_block_start;
std::vector<uint32_t> query_block_len, ref_block_len;
static void dump_table(const char* tag, const std::vector<std::string>& n, const std::vector<size_t>& s, const std::vector<uint32_t>& l) {
  for (size_t i = 0; i < n.size(); i++) printf("%s %s %zu %u\n", tag, n[i].c_str(), s[i], l[i]); }
int main(int argc, char** argv) {
  cfg.query_filename = argv[1]; cfg.reference_filename = argv[2]; cfg.seed.size = atoi(argv[3]); cfg.lastz_interval_size = (uint32_t)atoi(argv[4]); cfg.debug = false;
  ref_DRAM = new DRAM; query_DRAM = new DRAM; query_rc_DRAM = new DRAM;
  FILE* block_name_file;
#include "ref_loader_q.inc"
#include "ref_loader_t.inc"
  dump_table("QCHR", q_chr_name, q_chr_start, q_chr_len);
  dump_table("RCCHR", rc_q_chr_name, rc_q_chr_start, rc_q_chr_len);
  dump_table("TCHR", r_chr_name, r_chr_start, r_chr_len);
  for (size_t b = 0; b < query_block_len.size(); b++) printf("QBLOCK %zu %u %u\n", query_block_start[b], query_block_len[b], block_num_intervals[b]);
  for (size_t b = 0; b < ref_block_len.size(); b++) printf("TBLOCK %zu %u\n", ref_block_start[b], ref_block_len[b]);
  for (auto& iv : interval_list) printf("IV %u %u\n", iv.start, iv.end);
  printf("QSIZE %zu %zu TSIZE %zu\n", query_DRAM->seqSize, query_rc_DRAM->seqSize, ref_DRAM->bufferPosition);
  FILE* o = fopen("arenas.bin", "wb");
  fwrite(query_DRAM->buffer, 1, query_DRAM->seqSize, o); fwrite(query_rc_DRAM->buffer, 1, query_rc_DRAM->seqSize, o); fwrite(ref_DRAM->buffer, 1, ref_DRAM->bufferPosition, o);
  fclose(o);
  return 0;
}
'''


def write_fasta(path, records, width=60):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(">%s some description\n" % name)
            s = bytes(seq).decode("ascii")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def b64(raw):
    return base64.b64encode(zlib.compress(raw, 9)).decode()


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from segalign_amd import synth
    lines = open(os.path.join(REF, "src", "main.cpp")).read().split("\n")
    q0 = next(i for i, l in enumerate(lines) if 'gzFile f_rd = gzopen(cfg.query_filename.c_str(), "r");' in l)
    q1 = next(i for i, l in enumerate(lines) if l.strip() == "total_query_intervals = interval_list.size();")
    t0 = next(i for i, l in enumerate(lines) if 'f_rd = gzopen(cfg.reference_filename.c_str(), "r");' in l)
    t1 = next(i for i in range(t0, len(lines)) if lines[i].strip() == "gzclose(f_rd);")
    assert (q0, q1, t0, t1) == (311, 461, 478, 540), (q0, q1, t0, t1)          # :312-462, :479-541
    cases = []
    rng = np.random.default_rng(31)
    for ci, (block, interval, seed, qlens, tlens) in enumerate(((3000, 1000, 19, (900, 1200, 1500, 400, 2900, 3100, 50, 700), (2000, 1100, 1000, 3500, 200)),
                                                              (100000, 1500, 19, (1800, 2200), (2600,)),
                                                              (2500, 700, 22, (2501, 10, 2490, 1300, 1300), (600, 600, 600, 600, 600, 600)))):
        tmp = tempfile.mkdtemp(prefix="sa_loader_golden_")
        os.makedirs(os.path.join(tmp, "tbb"))
        open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
        open(os.path.join(tmp, "ref_loader_q.inc"), "w").write("\n".join(lines[q0:q1 + 1]) + "\n")
        open(os.path.join(tmp, "ref_loader_t.inc"), "w").write("\n".join(lines[t0:t1 + 1]) + "\n")
        open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
        exe = os.path.join(tmp, "harness")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-DSA_BLOCK=%d" % block, "-I", tmp, "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "common"),
                               os.path.join(tmp, "harness.cpp"), os.path.join(REF, "common", "ntcoding.cpp"), "-lz", "-o", exe])

        def recs(prefix, lens):
            out = []
            for i, n in enumerate(lens):
                s = synth.random_dna(n, 1000 * ci + 17 * i + len(prefix)).copy()
                if n > 300:
                    s[50:120] = np.frombuffer(bytes(s[50:120]).lower(), dtype=np.uint8)
                    s[200:210] = ord("N")
                out.append(("%s%d" % (prefix, i + 1), s))
            return out
        qrecs, trecs = recs("scaffold_", qlens), recs("chr", tlens)
        write_fasta(os.path.join(tmp, "q.fa"), qrecs)
        write_fasta(os.path.join(tmp, "t.fa"), trecs, width=71)
        out = subprocess.check_output([exe, "q.fa", "t.fa", str(seed), str(interval)], cwd=tmp, stderr=subprocess.DEVNULL).decode()
        raw = open(os.path.join(tmp, "arenas.bin"), "rb").read()
        rows = [l.split() for l in out.split("\n") if l]
        sizes = next(r for r in rows if r[0] == "QSIZE")
        qs, rcs, ts = int(sizes[1]), int(sizes[2]), int(sizes[4])
        names = {f: open(os.path.join(tmp, f)).read() for f in sorted(os.listdir(tmp)) if f.endswith(".name")}
        case = dict(seq_block_size=block, lastz_interval_size=interval, seed_size=seed,
                    query=[[n, bytes(s).decode("ascii")] for n, s in qrecs], target=[[n, bytes(s).decode("ascii")] for n, s in trecs],
                    q_chr=[[r[1], int(r[2]), int(r[3])] for r in rows if r[0] == "QCHR"], rc_q_chr=[[r[1], int(r[2]), int(r[3])] for r in rows if r[0] == "RCCHR"],
                    r_chr=[[r[1], int(r[2]), int(r[3])] for r in rows if r[0] == "TCHR"],
                    q_blocks=[[int(r[1]), int(r[2]), int(r[3])] for r in rows if r[0] == "QBLOCK"], r_blocks=[[int(r[1]), int(r[2])] for r in rows if r[0] == "TBLOCK"],
                    intervals=[[int(r[1]), int(r[2])] for r in rows if r[0] == "IV"], name_files=names,
                    q_arena=b64(raw[:qs]), q_rc_arena=b64(raw[qs:qs + rcs]), r_arena=b64(raw[qs + rcs:qs + rcs + ts]))
        print("case %d: %d query records -> %d blocks, %d intervals; %d target records -> %d blocks" % (ci, len(qrecs), len(case["q_blocks"]), len(case["intervals"]),
              len(trecs), len(case["r_blocks"])), flush=True)
        cases.append(case)
    json.dump(dict(note="the DRAM arenas and tables src/main.cpp's loaders build (:312-462 query, :479-541 target; reference text inside a harness function, "
                        "DEFAULT_SEQ_BLOCK_SIZE re-defined to seq_block_size: tests/golden/make_loader_golden.py).  chromosome tables = name, start, length; q_blocks = "
                        "start, length, number of intervals; intervals = start, end in block order; arenas zlib + base64.", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
