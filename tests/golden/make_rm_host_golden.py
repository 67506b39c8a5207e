#!/usr/bin/env python3
"""Generator of tests/golden/rm_host_golden.json: the repeat masker's host side by a second route.

What runs: repeat_masker_src/seeder.cpp and repeat_masker_src/segment_printer.cpp -- both files as they lie, unedited -- compiled with
g++ against the fork's own graph.h / store.h / seed_filter.h and linked with the real common/ntcoding.cpp.  TBB is stood in for by two
tiny headers (tbb/flow_graph.h: tuple = std::tuple and a port that swallows the token; tbb/scalable_allocator.h: calloc / free); the
harness (this repository's code) owns what repeat_masker_src/main.cpp owns -- cfg, seq_DRAM, seq_rc_DRAM filled with the real RevComp as
:311 does, the chromosome table -- and a g_SeedAndFilter that RECORDS what it is handed and RETURNS designed HSPs (a small LCG; piles of
300 HSPs on the same bases among them, to reach the uint8 counter's wrap).  So the reference's own object code runs the chunk loop with
the minus piece derived from the plus piece's end (:118-119), the coverage counting (:153-160, uint8), the run extraction (:166-186, a run
still open at the block end is dropped) and writes the .intervals file (segment_printer.cpp:8-65).  A build with stand-in headers does
not pin anything: a second route for a-10's host half and 8f-4.

usage: python tests/golden/make_rm_host_golden.py   (needs /root/reference and g++)
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
from make_printer_golden import FAKE_TBB  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "rm_host_golden.json")

FAKE_ALLOC = r'''#pragma once
// stand-in for tbb/scalable_allocator.h
#include <cstdlib>
static inline void* scalable_calloc(size_t n, size_t s) { return calloc(n, s); }
static inline void scalable_free(void* p) { free(p); }
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
// ---- what repeat_masker_src/main.cpp owns (this repository's code) ----
Configuration cfg;
DRAM *seq_DRAM, *seq_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }
DRAM::~DRAM() {}
std::vector<std::string> chr_name; std::vector<size_t> chr_start; std::vector<uint32_t> chr_len;
InitializeProcessor_ptr g_InitializeProcessor; SendQueryWriteRequest_ptr g_SendQueryWriteRequest; SeedAndFilter_ptr g_SeedAndFilter;
ClearQuery_ptr g_ClearQuery; ShutdownProcessor_ptr g_ShutdownProcessor;
static FILE* g_out; static uint32_t g_block_len, g_lcg = 12345, g_calls = 0, g_pile = 0;
static uint32_t lcg() { g_lcg = g_lcg * 1664525u + 1013904223u; return g_lcg >> 8; }
// what g_SeedAndFilter is handed, and the designed HSPs it answers with (element 0 = the 64-bit header, rm seed_filter.cu:857-861)
static std::vector<segmentPair> capture(std::vector<uint64_t> seeds, bool rev, uint32_t ref_start, uint32_t ref_end) {
  g_calls++;
  std::vector<segmentPair> r(1);
  r[0].ref_start = 1000u + g_calls; r[0].query_start = 0; r[0].len = 0; r[0].score = 0;
  uint32_t n = lcg() % 5;                                   // a few scattered HSPs
  if (g_pile && g_calls % 3 == 0) n = 300;                   // ... or a pile: 300 HSPs over the same bases -> the uint8 counter wraps
  const uint32_t base = lcg() % (g_block_len > 400 ? g_block_len - 400 : 1);
  for (uint32_t i = 0; i < n; i++) {
    segmentPair h; h.ref_start = lcg() % g_block_len; h.score = 3000 + (int)(lcg() % 9000);
    if (n == 300) { h.query_start = base + (i % 7); h.len = 120 + (i % 5); }
    else { h.query_start = lcg() % (g_block_len - 1); h.len = 1 + lcg() % 250; if (h.query_start + h.len > g_block_len) h.len = g_block_len - h.query_start; }
    r.push_back(h);
  }
  uint32_t hdr[5] = {rev ? 1u : 0u, ref_start, ref_end, (uint32_t)seeds.size(), (uint32_t)r.size() - 1};
  fwrite(hdr, 4, 5, g_out); fwrite(seeds.data(), 8, seeds.size(), g_out); fwrite(r.data() + 1, 16, r.size() - 1, g_out);
  return r;
}
// in: u32 seq_len, block_start, block_len, chunk, transition, strand, M, markend, pile, shape_len, n_chr, n_intervals ; shape ; sequence ;
//     n_chr x {name_len u32, name, start u32, len u32} ; n_intervals x {start, end, ref_start, ref_end}
// out (binary): per interval a marker {0xFFFFFFFF, k, n_runs} + runs {query_start, len} AFTER its calls; calls as written by capture
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[12];
  if (fread(hdr, 4, 12, f) != 12) return 2;
  std::string shape(hdr[9], ' ');
  std::vector<char> fw(hdr[0] + 64, 'N'), rc(hdr[0] + 64, 'N');
  if (fread(&shape[0], 1, hdr[9], f) != hdr[9] || fread(fw.data(), 1, hdr[0], f) != hdr[0]) return 2;
  for (uint32_t i = 0; i < hdr[10]; i++) { uint32_t nl, st, ln; if (fread(&nl, 4, 1, f) != 1) return 2; std::string nm(nl, ' ');
    if (fread(&nm[0], 1, nl, f) != nl || fread(&st, 4, 1, f) != 1 || fread(&ln, 4, 1, f) != 1) return 2; chr_name.push_back(nm); chr_start.push_back(st); chr_len.push_back(ln); }
  std::vector<uint32_t> iv(4 * hdr[11]);
  if (fread(iv.data(), 4, iv.size(), f) != iv.size()) return 2;
  fclose(f);
  cfg.seed.shape = shape; cfg.seed.size = (int)shape.size(); cfg.seed.kmer_size = GenerateShapePos(shape);
  cfg.seed.transition = hdr[4] != 0; cfg.wga_chunk_size = hdr[3]; cfg.M = hdr[6]; cfg.markend = hdr[7] != 0;
  cfg.strand = hdr[5] == 1 ? "plus" : hdr[5] == 2 ? "minus" : "both";
  cfg.seq_len = hdr[0];                                                      // repeat_masker_src/main.cpp:309
  seq_DRAM = new DRAM; seq_rc_DRAM = new DRAM; seq_DRAM->buffer = fw.data(); seq_rc_DRAM->buffer = rc.data();
  RevComp(seq_rc_DRAM->buffer, seq_DRAM->buffer, 0, 0, cfg.seq_len);         // :311
  g_SeedAndFilter = capture; g_block_len = hdr[2]; g_pile = hdr[8];
  g_out = fopen(argv[2], "wb");
  seeder_body seeder; interval_printer_body printer; printer_node::output_ports_type ports;
  for (uint32_t k = 0; k < hdr[11]; k++) {
    seq_block b; b.index = 0; b.start = hdr[1]; b.len = hdr[2];
    seed_interval s; s.start = iv[4 * k]; s.end = iv[4 * k + 1]; s.ref_start = iv[4 * k + 2]; s.ref_end = iv[4 * k + 3]; s.num_invoked = k + 1; s.num_intervals = hdr[11];
    printer_input out = seeder(seeder_input(seeder_payload(b, s), (size_t)k));
    const interval_output& runs = get<2>(get<0>(out));
    uint32_t mark[3] = {0xFFFFFFFFu, k, (uint32_t)runs.size()};
    fwrite(mark, 4, 3, g_out);
    for (auto& r : runs) { uint32_t p[2] = {r.query_start, r.len}; fwrite(p, 4, 2, g_out); }
    printer(out, ports);                                                     // writes tmp<k+1>.block0.intervals into the working directory
  }
  fclose(g_out);
  return 0;
}
'''


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from segalign_amd import synth
    tmp = tempfile.mkdtemp(prefix="sa_rm_host_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "tbb", "scalable_allocator.h"), "w").write(FAKE_ALLOC)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    rm = os.path.join(REF, "repeat_masker_src")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", rm, "-I", os.path.join(REF, "common"), os.path.join(tmp, "harness.cpp"),
                           os.path.join(rm, "seeder.cpp"), os.path.join(rm, "segment_printer.cpp"), os.path.join(REF, "common", "ntcoding.cpp"), "-o", exe])
    S19 = "TTT0T00TT00T0T0TTTT"
    cases = []
    for ci, (transition, strand, chunk, M, markend, pile, rec_lens, block) in enumerate((
            (1, 3, 700, 1, 0, 0, (1500, 900, 1700), None), (0, 3, 500, 2, 1, 1, (2000, 1200), None), (1, 1, 800, 1, 0, 1, (1800, 1500), (600, 2400)),
            (1, 2, 450, 1, 1, 0, (2500,), None))):
        recs = [synth.random_dna(n, 800 + 10 * ci + i).copy() for i, n in enumerate(rec_lens)]
        recs[0][100:220] = np.frombuffer(bytes(recs[0][100:220]).lower(), dtype=np.uint8)
        recs[-1][300:700] = ord("N")
        seq = np.concatenate([np.concatenate([r, np.frombuffer(b"&", dtype=np.uint8)]) for r in recs])[:-1].copy()
        names = ["chr%d" % (i + 1) for i in range(len(recs))]
        starts = [int(sum(len(r) + 1 for r in recs[:i])) for i in range(len(recs))]
        bs, bl = block if block else (0, seq.size)
        ivs = []
        for s in range(0, bl - 19, 1000):   # interval tasks with windows like the plan's (repeat_masker_src/main.cpp:367-420): a window around the interval
            e = min(s + 1000, bl - 19)
            ivs.append((s, e, max(0, s - 1000), min(bl, e + 1000)))
        wd = os.path.join(tmp, "case%d" % ci)
        os.makedirs(wd)
        inp, outp = os.path.join(wd, "in.bin"), os.path.join(wd, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<12I", seq.size, bs, bl, chunk, transition, strand, M, markend, pile, len(S19), len(names), len(ivs)))
            f.write(S19.encode())
            f.write(seq.tobytes())
            for nm, st, r in zip(names, starts, recs):
                f.write(struct.pack("<I", len(nm)) + nm.encode() + struct.pack("<2I", st, len(r)))
            for t in ivs:
                f.write(struct.pack("<4I", *t))
        subprocess.check_call([exe, inp, outp], cwd=wd, stderr=subprocess.DEVNULL)
        raw = open(outp, "rb").read()
        off, tasks, cur = 0, [], []
        while off < len(raw):
            a = struct.unpack_from("<I", raw, off)[0]
            if a == 0xFFFFFFFF:
                _, k, nr = struct.unpack_from("<3I", raw, off)
                off += 12
                runs = np.frombuffer(raw, dtype="<u4", count=2 * nr, offset=off).reshape(-1, 2).tolist()
                off += 8 * nr
                fn = "tmp%d.block0.intervals" % (k + 1)
                text = open(os.path.join(wd, fn)).read() if os.path.exists(os.path.join(wd, fn)) else None
                tasks.append(dict(interval=list(ivs[k]), calls=cur, runs=runs, file=text))
                cur = []
                continue
            rev, rs, re_, ns, nh = struct.unpack_from("<5I", raw, off)
            off += 20
            seeds = np.frombuffer(raw, dtype="<u8", count=ns, offset=off)
            off += 8 * ns
            hs = np.frombuffer(raw, dtype="<i4", count=4 * nh, offset=off).reshape(-1, 4)
            off += 16 * nh
            cur.append(dict(rev=int(rev), ref_start=int(rs), ref_end=int(re_), n=int(ns), first=int(seeds[0] & 0xFFFFFFFF) if ns else None,
                            seeds=pack_rows(seeds.tobytes()), hsps=pack_rows(np.ascontiguousarray(hs, dtype="<i4").tobytes())))
        print("case %d: %d bp, block (%d, %d), %d interval tasks, %d calls, %d runs, %d files" % (ci, seq.size, bs, bl, len(tasks), sum(len(t["calls"]) for t in tasks),
              sum(len(t["runs"]) for t in tasks), sum(t["file"] is not None for t in tasks)), flush=True)
        cases.append(dict(shape=S19, transition=transition, strand=strand, chunk=chunk, M=M, markend=markend, seq=seq.tobytes().decode("ascii"),
                          block_start=bs, block_len=bl, chr=[names, starts, [len(r) for r in recs]], tasks=tasks))
    json.dump(dict(note="the repeat masker's host side (repeat_masker_src/seeder.cpp + segment_printer.cpp compiled as they lie + the real ntcoding.cpp; TBB stood in for, "
                        "tests/golden/make_rm_host_golden.py): per interval task {start, end, ref_start, ref_end} the g_SeedAndFilter calls in order (strand, window, seed words "
                        "zlib + base64, and the designed HSPs -- ref_start, query_start, len, score as int32 x 4, zlib + base64 -- the harness answered with), the runs the seeder returns and the text of "
                        "the .intervals file (null: none written).", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
