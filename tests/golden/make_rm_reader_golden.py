#!/usr/bin/env python3
"""Generator of tests/golden/rm_reader_golden.json: the repeat masker's block protocol -- the source node of repeat_masker_src/main.cpp -- by a second
route, the way make_reader_golden.py does it for the src/ binary.

What runs: the reference's own text of repeat_masker_src/main.cpp:469-479 (the reader's state) and :485-552 (the body of the source node's lambda)
verbatim inside a lambda of the harness, driven serially through repeat_masker_src/seeder.cpp AS IT LIES (its counter num_seeded_regions tells the
reader when a block is finished) + the real common/ntcoding.cpp; TBB's headers stood in for (make_rm_host_golden.py), the harness owns the block and
interval lists (segalign_amd/shard.py::rm_plan, which tests/golden/rm_plan_golden.json holds against the plan's own text) and logs every g_* call.
A build with stand-ins does not pin anything.  What the vectors add: per block g_ClearRef + g_ClearQuery (from the second block
on), g_SendRefWriteRequest, g_SendQueryWriteRequest, GenerateSeedPosTable -- in that order -- before its first interval task, a block's tasks in plan
order with index = the block's number from 0, num_invoked from 1.

usage: python tests/golden/make_rm_reader_golden.py   (needs /root/reference and g++)
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
from make_printer_golden import FAKE_TBB  # noqa: E402
from make_rm_host_golden import FAKE_ALLOC  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "rm_reader_golden.json")
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
// ---- what repeat_masker_src/main.cpp owns outside the extracted text (this repository's code; names as in main.cpp) ----
struct timeval start_time, end_time, start_time_complete, end_time_complete;
long useconds, seconds, mseconds;
Configuration cfg;
DRAM *seq_DRAM, *seq_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }
DRAM::~DRAM() {}
std::vector<std::string> chr_name; std::vector<size_t> chr_start; std::vector<uint32_t> chr_len;
std::vector<size_t>   block_start;
std::vector<uint32_t> block_len;
static FILE* g_out;
static void ev(uint32_t tag, uint64_t a = 0, uint64_t b = 0, uint64_t c = 0) { uint64_t r[4] = {tag, a, b, c}; fwrite(r, 8, 4, g_out); }
static int  L_InitializeInterface(int n) { return n; }
static void L_SendRefWriteRequest(char*, size_t addr, uint32_t len) { ev(1, addr, len); }
static void L_ClearRef() { ev(2); }
static void L_SendQueryWriteRequest() { ev(3); }
static void L_ClearQuery() { ev(4); }
static std::vector<segmentPair> L_SeedAndFilter(std::vector<uint64_t> seeds, bool rev, uint32_t ref_start, uint32_t ref_end) {
  ev(6, rev, ref_start, ref_end);
  std::vector<segmentPair> r(1); r[0].ref_start = r[0].query_start = r[0].len = 0; r[0].score = 0; return r; }
InitializeInterface_ptr g_InitializeInterface = L_InitializeInterface; SendRefWriteRequest_ptr g_SendRefWriteRequest = L_SendRefWriteRequest;
ClearRef_ptr g_ClearRef = L_ClearRef; ShutdownProcessor_ptr g_ShutdownProcessor;
InitializeProcessor_ptr g_InitializeProcessor; SendQueryWriteRequest_ptr g_SendQueryWriteRequest = L_SendQueryWriteRequest;
SeedAndFilter_ptr g_SeedAndFilter = L_SeedAndFilter; ClearQuery_ptr g_ClearQuery = L_ClearQuery;
void GenerateSeedPosTable(char*, size_t start_addr, uint32_t ref_length, uint32_t step, int shape_size, int kmer_size) {
  ev(5, start_addr, ref_length, ((uint64_t)step << 32) | ((uint64_t)shape_size << 8) | (uint64_t)kmer_size); }
// in : u32 seq_len, chunk, n_blocks, n_intervals ; sequence ; n_blocks x {u64 start, u32 len, u32 n_intervals} ; n_intervals x {start, end, ref_start, ref_end}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[4];
  if (fread(hdr, 4, 4, f) != 4) return 2;
  std::vector<char> fw(hdr[0] + 64, 'N'), rc(hdr[0] + 64, 'N');
  if (fread(fw.data(), 1, hdr[0], f) != hdr[0]) return 2;
  std::vector<seed_interval> interval_list; std::vector<uint32_t> block_num_intervals;
  uint32_t total_r_blocks = hdr[2];
  for (uint32_t i = 0; i < hdr[2]; i++) { uint64_t s; uint32_t l, n; if (fread(&s, 8, 1, f) != 1 || fread(&l, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    block_start.push_back(s); block_len.push_back(l); block_num_intervals.push_back(n); }
  for (uint32_t i = 0; i < hdr[3]; i++) { uint32_t v[4]; if (fread(v, 4, 4, f) != 4) return 2; seed_interval s; s.start = v[0]; s.end = v[1]; s.ref_start = v[2]; s.ref_end = v[3];
    s.num_invoked = 0; s.num_intervals = 0; interval_list.push_back(s); }
  fclose(f);
  std::string shape = "TTT0T00TT00T0T0TTTT";
  cfg.seed.shape = shape; cfg.seed.size = 19; cfg.seed.kmer_size = GenerateShapePos(shape); cfg.seed.transition = true;
  cfg.wga_chunk_size = hdr[1]; cfg.step = 1; cfg.strand = "both"; cfg.debug = false; cfg.M = 1; cfg.seq_len = hdr[0];
  seq_DRAM = new DRAM; seq_rc_DRAM = new DRAM; seq_DRAM->buffer = fw.data(); seq_rc_DRAM->buffer = rc.data();
  RevComp(seq_rc_DRAM->buffer, seq_DRAM->buffer, 0, 0, cfg.seq_len);
  g_out = fopen(argv[2], "wb");
#include "ref_reader_state.inc"
  auto reader = [&](seeder_payload &op) -> bool {
#include "ref_reader_body.inc"
  };
  seeder_body seeder;
  seeder_payload op;
  while (reader(op)) {
    const seq_block& b = get<0>(op); const seed_interval& s = get<1>(op);
    ev(7, ((uint64_t)(uint32_t)b.index << 32) | b.len, b.start, ((uint64_t)s.start << 32) | s.end);
    ev(8, ((uint64_t)s.ref_start << 32) | s.ref_end, s.num_invoked, s.num_intervals);
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
    from segalign_amd import shard, synth
    lines = open(os.path.join(REF, "repeat_masker_src", "main.cpp")).read().split("\n")
    assert lines[468].strip() == "size_t send_block_start;" and lines[478].strip() == "seeder_body::total_xdrop = 0;", (lines[468], lines[478])
    assert lines[482].strip() == "[&](seeder_payload &op) -> bool {" and lines[553].strip() == "}, true);", (lines[482], lines[553])
    tmp = tempfile.mkdtemp(prefix="sa_rm_reader_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "tbb", "scalable_allocator.h"), "w").write(FAKE_ALLOC)
    open(os.path.join(tmp, "ref_reader_state.inc"), "w").write("\n".join(lines[468:479]) + "\n")     # :469-479
    open(os.path.join(tmp, "ref_reader_body.inc"), "w").write("\n".join(lines[483:552]) + "\n")      # :484-552
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    rm = os.path.join(REF, "repeat_masker_src")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", rm, "-I", os.path.join(REF, "common"), os.path.join(tmp, "harness.cpp"),
                           os.path.join(rm, "seeder.cpp"), os.path.join(REF, "common", "ntcoding.cpp"), "-o", exe])
    cases = []
    for ci, (seq_len, block_size, interval, prop, chunk) in enumerate(((5000, 10 ** 9, 1500, 0.5, 600), (9000, 3000, 1000, 0.4, 700), (7000, 2000, 1000, 0.0, 1000))):
        seq = synth.random_dna(seq_len, 6000 + ci)
        tasks = shard.rm_plan(seq_len, block_size, interval, prop, 19)
        blocks = []
        for t in tasks:
            if not blocks or blocks[-1][0] != t["block_index"]:
                blocks.append([t["block_index"], t["block_start"], t["block_len"], 0])
            blocks[-1][3] += 1
        inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<4I", seq_len, chunk, len(blocks), len(tasks)))
            f.write(seq.tobytes())
            for _, s, l, n in blocks:
                f.write(struct.pack("<QII", s, l, n))
            for t in tasks:
                f.write(struct.pack("<4I", t["start"], t["end"], t["ref_start"], t["ref_end"]))
        subprocess.check_call([exe, inp, outp], stderr=subprocess.DEVNULL)
        raw = np.frombuffer(open(outp, "rb").read(), dtype="<u8").reshape(-1, 4)
        assert int(raw[-1][0]) == 9
        names = {1: "SendRef", 2: "ClearRef", 3: "SendQuery", 4: "ClearQuery"}
        events, i = [], 0
        while i < len(raw) - 1:
            tag, a, b, c = (int(x) for x in raw[i])
            if tag == 7:
                _, a2, b2, c2 = (int(x) for x in raw[i + 1])
                events.append(["Payload", a >> 32, b, a & 0xFFFFFFFF, c >> 32, c & 0xFFFFFFFF, a2 >> 32, a2 & 0xFFFFFFFF, b2, c2])
                i += 2
                continue
            if tag == 5:
                events.append(["Table", a, b, c >> 32, (c >> 8) & 0xFFFFFF, c & 0xFF])
            elif tag == 6:
                if events and events[-1][0] == "SeedAndFilter" and events[-1][1:3] == [b, c]:
                    events[-1][3] += 1
                else:
                    events.append(["SeedAndFilter", b, c, 1])
            else:
                events.append([names[tag], a, b][: {1: 3, 2: 1, 3: 1, 4: 1}[tag]])
            i += 1
        print("case %d: %d blocks, %d tasks: %d events" % (ci, len(blocks), len(tasks), len(events)), flush=True)
        cases.append(dict(seq_len=seq_len, seq_block_size=block_size, interval=interval, neighbor_proportion=prop, chunk=chunk, events=events))
    json.dump(dict(note="the calls the repeat masker's source node (repeat_masker_src/main.cpp:469-479, :484-552, verbatim) makes, driven serially through "
                        "repeat_masker_src/seeder.cpp compiled as it lies (tests/golden/make_rm_reader_golden.py): SendRef [addr, len], ClearRef, SendQuery, ClearQuery, "
                        "Table [addr, len, step, seed size, kmer size], Payload [index, block start, block len, start, end, ref_start, ref_end, num_invoked, num_intervals], "
                        "SeedAndFilter [ref_start, ref_end, consecutive calls].  Blocks and tasks are segalign_amd/shard.py::rm_plan's for the parameters given.",
                   cases=cases), open(OUT, "w"))
    print("wrote %s" % OUT)


if __name__ == "__main__":
    main()
