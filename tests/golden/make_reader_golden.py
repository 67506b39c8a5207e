#!/usr/bin/env python3
"""Generator of tests/golden/reader_golden.json: the reference's block / buffer protocol -- the source node of src/main.cpp -- by a second route.

What runs: the reference's own text of src/main.cpp:575-598 (the reader's state) and :603-735 (the body of the source node's lambda: which target
block is uploaded and indexed when, which query block goes into which of the BUFFER_DEPTH = 2 device buffers, when g_ClearRef / g_ClearQuery are
called, which (block, interval) payload is handed to the seeder next), verbatim inside a lambda of the harness, together with src/seeder.cpp AS IT LIES
(its counters num_seeded_regions / total_xdrop steer the reader) and the real common/ntcoding.cpp.  The harness (this repository's code) owns what the
rest of main.cpp owns -- cfg, the DRAM arenas, the block and interval lists (taken from tests/host_model.py::Arena, which tests/golden/
loader_golden.json holds against the loaders' own text) -- stands in for TBB's header, logs every g_* call, and drives the graph SERIALLY: one payload
out of the reader, through the seeder, back to the reader (cfg.num_threads tickets bound how far the reader may run ahead in the real graph; one ticket
is one of its schedules).  A build with stand-ins does not pin anything.  What the vectors add: the call protocol a drop-in
engine is driven with (8f-3, and a-3's Send* / Clear* entries under buffer reuse), and the payload fields the printer names its files by
(r_index counts from 1, q_index from 0, q_len = block length - seed size, num_invoked from 1).

usage: python tests/golden/make_reader_golden.py   (needs /root/reference and g++)
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_seeder_golden import FAKE_TBB  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "reader_golden.json")
S19 = "TTT0T00TT00T0T0TTTT"

HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/time.h>
#include "graph.h"
#include "ntcoding.h"
#include "seed_filter.h"
#include "seed_filter_interface.h"
#include "store.h"
// ---- what src/main.cpp owns outside the extracted text (this repository's code; names as in main.cpp:23-24,47-54,319-326,487) ----
struct timeval start_time, end_time, start_time_complete, end_time_complete;
long useconds, seconds, mseconds;
Configuration cfg;
DRAM *ref_DRAM, *query_DRAM, *query_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }
DRAM::~DRAM() {}
std::vector<uint32_t> q_buffer;
std::vector<size_t>   query_block_start;
std::vector<uint32_t> query_block_len;
std::vector<size_t>   ref_block_start;
std::vector<uint32_t> ref_block_len;
static FILE* g_out;
static void ev(uint32_t tag, uint64_t a = 0, uint64_t b = 0, uint64_t c = 0) { uint64_t r[4] = {tag, a, b, c}; fwrite(r, 8, 4, g_out); }
// the engine's entries: logged (tags 1..6), nothing else
static int  L_InitializeInterface(int n) { return n; }
static void L_SendRefWriteRequest(char*, size_t addr, uint32_t len) { ev(1, addr, len); }
static void L_ClearRef() { ev(2); }
static void L_SendQueryWriteRequest(size_t addr, uint32_t len, uint32_t buffer) { ev(3, addr, len, buffer); }
static void L_ClearQuery(uint32_t buffer) { ev(4, buffer); }
static std::vector<segmentPair> L_SeedAndFilter(std::vector<uint64_t> seeds, bool rev, uint32_t buffer) {
  ev(6, rev, buffer, seeds.size());
  std::vector<segmentPair> r(1); r[0].ref_start = r[0].query_start = r[0].len = 0; r[0].score = 0; return r; }
InitializeInterface_ptr g_InitializeInterface = L_InitializeInterface; SendRefWriteRequest_ptr g_SendRefWriteRequest = L_SendRefWriteRequest;
ClearRef_ptr g_ClearRef = L_ClearRef; ShutdownProcessor_ptr g_ShutdownProcessor;
InitializeProcessor_ptr g_InitializeProcessor; SendQueryWriteRequest_ptr g_SendQueryWriteRequest = L_SendQueryWriteRequest;
SeedAndFilter_ptr g_SeedAndFilter = L_SeedAndFilter; ClearQuery_ptr g_ClearQuery = L_ClearQuery;
void GenerateSeedPosTable(char*, size_t start_addr, uint32_t ref_length, uint32_t step, int shape_size, int kmer_size) {
  ev(5, start_addr, ref_length, ((uint64_t)step << 32) | ((uint64_t)shape_size << 8) | (uint64_t)kmer_size); }
// in : u32 chunk, transition, step, shape_len, n_rblocks, n_qblocks, n_intervals, arena_len ; shape ; query arena ; rc arena ;
//      n_rblocks x {u64 start, u32 len} ; n_qblocks x {u64 start, u32 len, u32 n_intervals} ; n_intervals x {u32 start, end}
// out: events of 4 x u64: 1 SendRef, 2 ClearRef, 3 SendQuery, 4 ClearQuery, 5 GenerateSeedPosTable, 6 SeedAndFilter,
//      7 payload {r_index, q_index, r_start | q_start, ...} in two records (7 and 8)
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[8];
  if (fread(hdr, 4, 8, f) != 8) return 2;
  std::string shape(hdr[3], ' ');
  std::vector<char> fw(hdr[7] + 64, 'N'), rc(hdr[7] + 64, 'N');
  if (fread(&shape[0], 1, hdr[3], f) != hdr[3] || fread(fw.data(), 1, hdr[7], f) != hdr[7] || fread(rc.data(), 1, hdr[7], f) != hdr[7]) return 2;
  std::vector<seed_interval> interval_list; std::vector<uint32_t> block_num_intervals;
  uint32_t total_q_blocks = hdr[5], total_r_blocks = hdr[4], total_query_intervals = hdr[6];
  for (uint32_t i = 0; i < hdr[4]; i++) { uint64_t s; uint32_t l; if (fread(&s, 8, 1, f) != 1 || fread(&l, 4, 1, f) != 1) return 2; ref_block_start.push_back(s); ref_block_len.push_back(l); }
  for (uint32_t i = 0; i < hdr[5]; i++) { uint64_t s; uint32_t l, n; if (fread(&s, 8, 1, f) != 1 || fread(&l, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    query_block_start.push_back(s); query_block_len.push_back(l); block_num_intervals.push_back(n); q_buffer.push_back(0); }   // (:345: one q_buffer slot per block)
  for (uint32_t i = 0; i < hdr[6]; i++) { uint32_t se[2]; if (fread(se, 4, 2, f) != 2) return 2; seed_interval s; s.start = se[0]; s.end = se[1]; s.num_invoked = 0; s.num_intervals = 0; s.buffer = 0; interval_list.push_back(s); }
  fclose(f);
  cfg.seed.shape = shape; cfg.seed.size = (int)shape.size(); cfg.seed.kmer_size = GenerateShapePos(shape);
  cfg.seed.transition = hdr[1] != 0; cfg.wga_chunk_size = hdr[0]; cfg.step = hdr[2]; cfg.strand = "both"; cfg.debug = false;
  ref_DRAM = new DRAM; query_DRAM = new DRAM; query_rc_DRAM = new DRAM;
  ref_DRAM->buffer = fw.data(); query_DRAM->buffer = fw.data(); query_rc_DRAM->buffer = rc.data();
  g_out = fopen(argv[2], "wb");
#include "ref_reader_state.inc"
  auto reader = [&](seeder_payload &op) -> bool {
#include "ref_reader_body.inc"
  };
  seeder_body seeder;
  seeder_payload op;
  while (reader(op)) {
    const seq_block& b = get<0>(op); const seed_interval& s = get<1>(op);
    ev(7, ((uint64_t)(uint32_t)b.r_index << 32) | (uint32_t)b.q_index, b.r_start, b.q_start);
    ev(8, ((uint64_t)b.r_len << 32) | b.q_len, ((uint64_t)s.start << 32) | s.end, ((uint64_t)s.num_invoked << 40) | ((uint64_t)s.num_intervals << 16) | s.buffer);
    seeder(seeder_input(op, (size_t)0));
  }
  ev(9);
  fclose(g_out);
  return 0;
}
'''


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from host_model import Arena
    from segalign_amd import synth
    lines = open(os.path.join(REF, "src", "main.cpp")).read().split("\n")
    assert lines[574].strip() == "bool send_r_block = true;" and lines[597].strip() == "uint32_t block_intervals_num;", (lines[574], lines[597])
    assert lines[601].strip() == "[&](seeder_payload &op) -> bool {" and lines[736].strip() == "}, true);", (lines[601], lines[736])
    tmp = tempfile.mkdtemp(prefix="sa_reader_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "ref_reader_state.inc"), "w").write("\n".join(lines[574:598]) + "\n")     # :575-598
    open(os.path.join(tmp, "ref_reader_body.inc"), "w").write("\n".join(lines[602:735]) + "\n")      # :603-735
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "common"),
                           os.path.join(tmp, "harness.cpp"), os.path.join(REF, "src", "seeder.cpp"), os.path.join(REF, "common", "ntcoding.cpp"), "-o", exe])
    cases = []
    rng = np.random.default_rng(41)
    #        target records (lengths)        query records                     block size  interval chunk transition step
    plan = (((1500, 1200),                   (900, 1100, 700),                 10 ** 9,    800,     500,  1,         1),    # one block each, several intervals
            ((1500, 1200, 1400),             (900, 1100, 700, 1300, 800, 600), 2000,       900,     600,  1,         1),    # 2 target blocks x 4 query blocks: buffers re-used
            ((2500,),                        (800, 900, 1000, 700, 650),       1500,       700,     400,  0,         2),    # 1 x 3, two intervals per block
            ((1200, 1300, 1250, 1100, 900),  (1000, 1200),                     2100,       5000,    800,  1,         1))    # 3 target blocks x 1 query block
    for ci, (t_lens, q_lens, bsize, interval, chunk, trans, step) in enumerate(plan):
        trecs = [("chr%d" % (i + 1), synth.random_dna(n, 4000 + 10 * ci + i).tobytes()) for i, n in enumerate(t_lens)]
        qrecs = []
        tcat = np.frombuffer(b"".join(s for _, s in trecs), dtype=np.uint8)
        comp = np.full(256, ord("N"), np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        for i, n in enumerate(q_lens):   # query records = 5 %-diverged slices of the target, every second one from the other strand: the replay on the GPU ends in HSPs
            src = int(rng.integers(0, tcat.size - n))
            piece = tcat[src:src + n].copy()
            m = rng.random(n) < 0.05
            piece[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            qrecs.append(("ctg%d" % (i + 1), (comp[piece[::-1]] if i % 2 else piece).tobytes()))
        R = Arena(trecs, bsize, 19, interval, False)
        Q = Arena(qrecs, bsize, 19, interval, True)
        ivs = [iv for qb in range(len(Q.block_len)) for iv in Q.intervals[qb]]
        arena = bytes(Q.buf)
        rc = bytes(Q.rc) + b"N" * (len(arena) - len(Q.rc))
        inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<8I", chunk, trans, step, 19, len(R.block_len), len(Q.block_len), len(ivs), len(arena)))
            f.write(S19.encode())
            f.write(arena)
            f.write(rc[:len(arena)])
            for s, l in zip(R.block_start, R.block_len):
                f.write(struct.pack("<QI", s, l))
            for qb, (s, l) in enumerate(zip(Q.block_start, Q.block_len)):
                f.write(struct.pack("<QII", s, l, len(Q.intervals[qb])))
            for a, b in ivs:
                f.write(struct.pack("<2I", a, b))
        subprocess.check_call([exe, inp, outp], stderr=subprocess.DEVNULL)
        raw = np.frombuffer(open(outp, "rb").read(), dtype="<u8").reshape(-1, 4)
        assert int(raw[-1][0]) == 9
        events, i = [], 0
        names = {1: "SendRef", 2: "ClearRef", 3: "SendQuery", 4: "ClearQuery", 5: "Table", 6: "SeedAndFilter"}
        while i < len(raw) - 1:
            tag, a, b, c = (int(x) for x in raw[i])
            if tag == 7:
                _, a2, b2, c2 = (int(x) for x in raw[i + 1])
                events.append(["Payload", a >> 32, a & 0xFFFFFFFF, b, c, a2 >> 32, a2 & 0xFFFFFFFF, b2 >> 32, b2 & 0xFFFFFFFF, c2 >> 40, (c2 >> 16) & 0xFFFFFF, c2 & 0xFFFF])
                i += 2
                continue
            if tag == 5:
                events.append(["Table", a, b, c >> 32, (c >> 8) & 0xFFFFFF, c & 0xFF])
            elif tag == 6:
                if events and events[-1][0] == "SeedAndFilter" and events[-1][1:3] == [a, b]:
                    events[-1][3] += 1                  # consecutive calls of one strand and buffer: counted
                else:
                    events.append(["SeedAndFilter", a, b, 1])
            else:
                events.append([names[tag], a, b, c][: {1: 3, 2: 1, 3: 4, 4: 2}[tag]])
            i += 1
        print("case %d: %d target blocks x %d query blocks, %d intervals: %d events, %d payloads" %
              (ci, len(R.block_len), len(Q.block_len), len(ivs), len(events), sum(e[0] == "Payload" for e in events)), flush=True)
        cases.append(dict(target_records=[[n, s.decode()] for n, s in trecs], query_records=[[n, s.decode()] for n, s in qrecs], seq_block_size=bsize, interval=interval,
                          chunk=chunk, transition=trans, step=step, shape=S19, events=events))
    json.dump(dict(note="the calls the reference's source node (src/main.cpp:575-598, :603-735, verbatim) makes, driven serially through src/seeder.cpp compiled as it lies "
                        "(tests/golden/make_reader_golden.py): SendRef [addr, len], ClearRef, Table [addr, len, step, seed size, kmer size], SendQuery [addr, len, buffer], "
                        "ClearQuery [buffer], Payload [r_index, q_index, r_start, q_start, r_len, q_len, interval start, end, num_invoked, num_intervals, buffer], "
                        "SeedAndFilter [rev, buffer, consecutive calls].  Blocks and intervals are tests/host_model.py::Arena's for the records, block size and interval given.",
                   cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
