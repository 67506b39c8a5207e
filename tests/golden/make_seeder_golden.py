#!/usr/bin/env python3
"""Generator of tests/golden/seeder_golden.json: the reference's host seeding loop by a second route.

What runs: src/seeder.cpp -- the WHOLE file as it lies, unedited -- compiled with g++ against the reference's own src/graph.h,
src/seed_filter.h, src/store.h, common/DRAM.h, common/parameters.h and linked with the REAL common/ntcoding.cpp.  The one thing the image
lacks is TBB: a four-line `tbb/flow_graph.h` in the temporary directory maps tbb::flow::tuple / get onto std::tuple / std::get (which is
what they are in oneTBB) and declares an empty multifunction_node; the harness (this repository's code) defines the globals main.cpp owns
-- cfg, query_DRAM, query_rc_DRAM with plain buffers, g_SeedAndFilter -- fills the minus-strand buffer with the real RevComp exactly as
src/main.cpp:377 does, and records every g_SeedAndFilter call seeder_body::operator() makes for the intervals it is handed.
A build with a stand-in header does not pin the oracle; what the vectors add: the chunk loop, the minus strand's
interval arithmetic (q_block_len - q_inter_end ...), the seed word order (k-mer, then one word per transition position in ascending t),
invalid windows skipped, "no seed words -> no call", and q_block_start + j as the sequence offset come out of the reference's own object
code, not out of reading it (a-7 host logic, 8f-1).

usage: python tests/golden/make_seeder_golden.py   (needs /root/reference and g++)
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_rm_golden import pack_rows  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "seeder_golden.json")

FAKE_TBB = r'''#pragma once
// stand-in for the one TBB header src/graph.h includes: the tuple names (std::tuple in oneTBB) and an empty node type
#include <tuple>
#include <cstddef>
namespace tbb { namespace flow { using std::tuple; using std::get; template <class In, class Out> struct multifunction_node { typedef int output_ports_type; }; } }
'''

HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "graph.h"
#include "ntcoding.h"
#include "seed_filter.h"
#include "store.h"
// ---- what src/main.cpp owns (this repository's code) ----
Configuration cfg;
DRAM *ref_DRAM, *query_DRAM, *query_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }   // (common/DRAM.cpp needs TBB's allocator and 6 GB)
DRAM::~DRAM() {}
InitializeProcessor_ptr g_InitializeProcessor; SendQueryWriteRequest_ptr g_SendQueryWriteRequest; SeedAndFilter_ptr g_SeedAndFilter;
ClearQuery_ptr g_ClearQuery; ShutdownProcessor_ptr g_ShutdownProcessor;
static FILE* g_out;
static std::vector<segmentPair> capture(std::vector<uint64_t> seeds, bool rev, uint32_t buffer) {   // what g_SeedAndFilter is handed
  uint32_t h[3] = {rev ? 1u : 0u, buffer, (uint32_t)seeds.size()};
  fwrite(h, 4, 3, g_out); fwrite(seeds.data(), 8, seeds.size(), g_out);
  std::vector<segmentPair> r(1); r[0].ref_start = r[0].query_start = r[0].len = 0; r[0].score = 0; return r;
}
// in: u32 block_len, q_block_start, chunk, transition, strand (1 plus, 2 minus, 3 both), shape_len, n_intervals ; shape ; sequence (the DRAM arena
//     from 0: q_block_start bases of another block in front, then the block) ; n_intervals x {start, end}      out: per call u32 rev, buffer, n + n seed words
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[7];
  if (fread(hdr, 4, 7, f) != 7) return 2;
  std::string shape(hdr[5], ' ');
  const uint32_t arena = hdr[1] + hdr[0];
  std::vector<char> fw(arena + 64, 'N'), rc(arena + 64, 'N');
  if (fread(&shape[0], 1, hdr[5], f) != hdr[5] || fread(fw.data(), 1, arena, f) != arena) return 2;
  std::vector<uint32_t> iv(2 * hdr[6]);
  if (fread(iv.data(), 4, iv.size(), f) != iv.size()) return 2;
  fclose(f);
  cfg.seed.shape = shape; cfg.seed.size = (int)shape.size(); cfg.seed.kmer_size = GenerateShapePos(shape);   // src/main.cpp:283-286
  cfg.seed.transition = hdr[3] != 0; cfg.wga_chunk_size = hdr[2];
  cfg.strand = hdr[4] == 1 ? "plus" : hdr[4] == 2 ? "minus" : "both";
  query_DRAM = new DRAM; query_rc_DRAM = new DRAM;
  query_DRAM->buffer = fw.data(); query_rc_DRAM->buffer = rc.data();
  RevComp(query_rc_DRAM->buffer, query_DRAM->buffer, hdr[1], hdr[1], hdr[0]);   // src/main.cpp:377: the WHOLE block (bufferPosition == seq_block_start here)
  g_SeedAndFilter = capture;
  g_out = fopen(argv[2], "wb");
  seeder_body body;
  for (uint32_t k = 0; k < hdr[6]; k++) {
    seq_block b; b.r_index = 0; b.q_index = 0; b.r_start = 0; b.q_start = hdr[1]; b.r_len = 0; b.q_len = hdr[0] - (uint32_t)cfg.seed.size;   // src/main.cpp:708
    seed_interval s; s.start = iv[2 * k]; s.end = iv[2 * k + 1]; s.num_invoked = k + 1; s.num_intervals = hdr[6]; s.buffer = k & 1;
    uint32_t mark[3] = {0xFFFFFFFFu, k, 0}; fwrite(mark, 4, 3, g_out);   // interval marker
    body(seeder_input(seeder_payload(b, s), (size_t)0));
  }
  fclose(g_out);
  return 0;
}
'''


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from segalign_amd import synth
    import re
    shape22 = re.search(r'"([T0]{22})"', open(os.path.join(ROOT, "tests", "test_gpu_edge_cases.py")).read()).group(1)
    tmp = tempfile.mkdtemp(prefix="sa_seeder_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "common"),
                           os.path.join(tmp, "harness.cpp"), os.path.join(REF, "src", "seeder.cpp"), os.path.join(REF, "common", "ntcoding.cpp"), "-o", exe])
    cases = []
    S19 = "TTT0T00TT00T0T0TTTT"
    for (shape, transition, strand, chunk, n, qstart, ivlen) in ((S19, 1, 3, 700, 5000, 0, 2000), (S19, 0, 3, 1000, 4000, 0, 1500), (S19, 1, 1, 500, 2500, 300, 2500),
                                                                 (S19, 1, 2, 900, 3000, 120, 1000), (shape22, 1, 3, 800, 3000, 0, 3000), (S19, 1, 3, 250, 600, 0, 600)):
        q = synth.random_dna(n, 700 + len(cases)).copy()
        if n > 1000:
            q[200:330] = np.frombuffer(bytes(q[200:330]).lower(), dtype=np.uint8)
            q[900:1700] = ord("N")                       # a chunk without a single seed word: no call
            q[n // 2] = ord("&")
            if len(cases) % 2 == 0:
                q[n // 2 + 300] = ord("R")           # (hazard H14: the real RevComp does not advance over it -- the minus strand shifts)
            q[n - 40] = ord("n")
        arena = np.concatenate([synth.random_dna(qstart, 99), q]) if qstart else q
        q_len = n - len(shape)                            # the q_len main.cpp hands the seeder (:708: block length - seed size)
        ivs = [(s, min(s + ivlen, q_len)) for s in range(0, q_len, ivlen)]
        inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<7I", n, qstart, chunk, transition, strand, len(shape), len(ivs)))
            f.write(shape.encode())
            f.write(arena.tobytes())
            for (a, b) in ivs:
                f.write(struct.pack("<2I", a, b))
        subprocess.check_call([exe, inp, outp], stderr=subprocess.DEVNULL)
        raw = open(outp, "rb").read()
        off, calls, k = 0, [], -1
        while off < len(raw):
            a, b, c = struct.unpack_from("<3I", raw, off)
            off += 12
            if a == 0xFFFFFFFF:
                k = b
                continue
            seeds = np.frombuffer(raw, dtype="<u8", count=c, offset=off)
            off += 8 * c
            calls.append(dict(interval=k, rev=int(a), buffer=int(b), n=int(c), seeds=pack_rows(seeds.tobytes())))
        print("shape %d transition %d strand %d chunk %d, %d bp from %d, %d intervals: %d calls, %d seed words" %
              (len(shape), transition, strand, chunk, n, qstart, len(ivs), len(calls), sum(c["n"] for c in calls)), flush=True)
        cases.append(dict(iupac=int(len(cases) % 2 == 0 and n > 1000), shape=shape, transition=transition, strand=strand, chunk=chunk, q_block_start=qstart, q_len=q_len,
                          block_len=n, arena=arena.tobytes().decode("ascii"), intervals=ivs, calls=calls))
    json.dump(dict(note="every g_SeedAndFilter call of seeder_body::operator() (src/seeder.cpp compiled as it lies + the real ntcoding.cpp; TBB's tuple header stood "
                        "in for, tests/golden/make_seeder_golden.py): per call the interval it belongs to, strand, buffer, the seed words (u64: key << 32 | position; "
                        "zlib + base64).  arena = the query DRAM from 0: q_block_start bases of another block, then the block (block_len bases; the minus-strand "
                        "buffer is RevComp of the whole block, src/main.cpp:377; the seeder is handed q_len = block_len - seed size, :708).", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
