#!/usr/bin/env python3
"""Generator of tests/golden/printer_golden.json: the reference's segment writer by a second route.

What runs: src/segment_printer.cpp -- the WHOLE file as it lies, unedited -- compiled with g++ against the reference's own src/graph.h and
src/store.h.  TBB is stood in for by a tiny `tbb/flow_graph.h` (tuple = std::tuple, a multifunction_node whose output port swallows the
token); the harness (this repository's code) defines the globals src/main.cpp owns -- cfg and the chromosome tables r_chr_* / q_chr_* /
rc_q_chr_* -- from the case file, hands segment_printer_body::operator() one printer_input per interval and collects the .segments files
it writes into the working directory and the lastz command lines it prints.  A build with a stand-in header does not pin anything
(DESIGN.md section 5); what the vectors add: file names (tmp<num_invoked>.block<q>.r<r_block_start>.{plus,minus}.segments), 1-based
inclusive coordinates relative to the chromosome, the minus strand written in REVERSE order against the rc chromosome table, the
`r_index - 1` of the ref block in the command line, which options the command carries -- from the reference's object code (8f-2).

usage: python tests/golden/make_printer_golden.py   (needs /root/reference and g++)
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference"
OUT = os.path.join(HERE, "printer_golden.json")

FAKE_TBB = r'''#pragma once
// stand-in for the one TBB header src/graph.h includes: the tuple names (std::tuple in oneTBB) and a node type whose port takes the token
#include <tuple>
#include <cstddef>
namespace tbb { namespace flow {
using std::tuple; using std::get;
struct sa_port { bool try_put(size_t) { return true; } };
template <class In, class Out> struct multifunction_node { typedef std::tuple<sa_port> output_ports_type; };
} }
'''

HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "graph.h"
#include "store.h"
// ---- what src/main.cpp owns (this repository's code) ----
Configuration cfg;
DRAM *ref_DRAM, *query_DRAM, *query_rc_DRAM;
std::vector<std::string> q_chr_name, rc_q_chr_name, r_chr_name;
std::vector<uint32_t> q_chr_file_name, rc_q_chr_file_name, r_chr_file_name, q_chr_len, rc_q_chr_len, r_chr_len;
std::vector<size_t> q_chr_start, rc_q_chr_start, r_chr_start;
// case file (text): see the generator
int main(int argc, char** argv) {
  std::ifstream in(argv[1]);
  std::string tok;
  auto table = [&](std::vector<std::string>& name, std::vector<size_t>& start, std::vector<uint32_t>& len, std::vector<uint32_t>& fn) {
    size_t n; in >> n;
    for (size_t i = 0; i < n; i++) { std::string s; size_t a; uint32_t l; in >> s >> a >> l; name.push_back(s); start.push_back(a); len.push_back(l); fn.push_back((uint32_t)i); } };
  in >> cfg.gapped >> cfg.data_folder >> cfg.output_format >> cfg.ydrop >> cfg.gappedthresh >> cfg.ambiguous >> cfg.notrivial >> cfg.scoring_file;
  if (cfg.ambiguous == "-") cfg.ambiguous = "";
  if (cfg.scoring_file == "-") cfg.scoring_file = "";
  table(r_chr_name, r_chr_start, r_chr_len, r_chr_file_name);
  table(q_chr_name, q_chr_start, q_chr_len, q_chr_file_name);
  table(rc_q_chr_name, rc_q_chr_start, rc_q_chr_len, rc_q_chr_file_name);
  size_t n_int; in >> n_int;
  segment_printer_body body;
  printer_node::output_ports_type ports;
  for (size_t k = 0; k < n_int; k++) {
    seq_block b; seed_interval s; size_t nf, nr;
    in >> b.r_index >> b.q_index >> b.r_start >> b.q_start >> b.r_len >> b.q_len >> s.start >> s.end >> s.num_invoked >> s.num_intervals >> s.buffer >> nf >> nr;
    hsp_output fw(nf), rc(nr);
    for (auto& e : fw) in >> e.ref_start >> e.query_start >> e.len >> e.score;
    for (auto& e : rc) in >> e.ref_start >> e.query_start >> e.len >> e.score;
    body(printer_input(printer_payload(seeder_payload(b, s), fw, rc), (size_t)k), ports);
  }
  return 0;
}
'''


class T:   # chromosome tables as tests/host_model.py's Arena exposes them
    pass


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from host_model import Arena
    tmp = tempfile.mkdtemp(prefix="sa_printer_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "common"),
                           os.path.join(tmp, "harness.cpp"), os.path.join(REF, "src", "segment_printer.cpp"), "-o", exe])
    rng = np.random.default_rng(23)
    cases = []
    for ci, (gapped, folder, fmt, ydrop, gth, amb, notriv, scoring, block_size) in enumerate((
            (1, "./", "maf-", 9430, 3000, "-", 0, "-", 500000000), (1, "/data/run1/", "axt", 5000, 2200, "iupac", 1, "scores.txt", 2500),
            (0, "./", "maf-", 9430, 3000, "-", 0, "-", 1800))):
        def recs(prefix, n, lo, hi):
            return [("%s%d" % (prefix, i + 1), b"A" * int(rng.integers(lo, hi))) for i in range(n)]
        R = Arena(recs("chr", 5, 400, 1500), block_size, 19, 1000, False)
        Q = Arena(recs("contig_", 7, 300, 1200), block_size, 19, 1000, True)
        intervals = []
        for rb, (rs, rl) in enumerate(zip(R.block_start, R.block_len)):
            for qb, (qs, ql) in enumerate(zip(Q.block_start, Q.block_len)):
                q_len = ql - 19
                for i, (a, b) in enumerate(Q.intervals[qb]):
                    def hsps(rev, n):
                        out = []
                        chr_s = Q.rc_start if rev else Q.chr_start
                        chr_l = Q.rc_len if rev else Q.chr_len
                        lo, hi = (q_len - b, q_len - a) if rev else (a, b)
                        for _ in range(n):
                            # a query start inside the interval (in that strand's coordinates) that is a real base of some record
                            for _try in range(50):
                                q0 = int(rng.integers(lo, max(hi, lo + 1)))
                                cidx = max(k for k in range(len(chr_s)) if chr_s[k] <= qs + q0)
                                if qs + q0 < chr_s[cidx] + chr_l[cidx]:
                                    break
                            ridx = int(rng.integers(0, len(R.chr_start)))
                            if not (rs <= R.chr_start[ridx] < rs + rl):
                                ridx = max(k for k in range(len(R.chr_start)) if R.chr_start[k] <= rs)
                            r0 = R.chr_start[ridx] - rs + int(rng.integers(0, R.chr_len[ridx]))
                            out.append((r0, q0, int(rng.integers(19, 300)), int(rng.integers(2200, 40000))))
                        return sorted(out, key=lambda h: (h[1], h[0]))
                    nf, nr = int(rng.integers(0, 6)), int(rng.integers(0, 6))
                    intervals.append(dict(r_index=rb + 1, q_index=qb, r_start=rs, q_start=qs, r_len=rl, q_len=q_len, start=a, end=b, num_invoked=i + 1,
                                          num_intervals=len(Q.intervals[qb]), buffer=qb & 1, fw=hsps(False, nf), rc=hsps(True, nr)))
        case_txt = ["%d %s %s %d %d %s %d %s" % (gapped, folder, fmt, ydrop, gth, amb, notriv, scoring)]
        for names, starts, lens in ((R.chr_name, R.chr_start, R.chr_len), (Q.chr_name, Q.chr_start, Q.chr_len), (Q.rc_name, Q.rc_start, Q.rc_len)):
            case_txt.append(str(len(names)))
            case_txt += ["%s %d %d" % (n, s, l) for n, s, l in zip(names, starts, lens)]
        case_txt.append(str(len(intervals)))
        for it in intervals:
            case_txt.append("%d %d %d %d %d %d %d %d %d %d %d %d %d" % (it["r_index"], it["q_index"], it["r_start"], it["q_start"], it["r_len"], it["q_len"], it["start"],
                                                                     it["end"], it["num_invoked"], it["num_intervals"], it["buffer"], len(it["fw"]), len(it["rc"])))
            case_txt += ["%d %d %d %d" % h for h in it["fw"] + it["rc"]]
        wd = os.path.join(tmp, "case%d" % ci)
        os.makedirs(wd)
        open(os.path.join(wd, "case.txt"), "w").write("\n".join(case_txt) + "\n")
        out = subprocess.check_output([exe, "case.txt"], cwd=wd).decode()
        files = {f: open(os.path.join(wd, f)).read() for f in sorted(os.listdir(wd)) if f.endswith(".segments")}
        cmds = [l for l in out.split("\n") if l]
        print("case %d: %d target / %d query blocks, %d intervals -> %d files, %d command lines" % (ci, len(R.block_start), len(Q.block_start), len(intervals), len(files), len(cmds)), flush=True)
        cases.append(dict(gapped=gapped, data_folder=folder, output_format=fmt, ydrop=ydrop, gappedthresh=gth, ambiguous="" if amb == "-" else amb, notrivial=notriv,
                          scoring_file="" if scoring == "-" else scoring,
                          r_chr=[R.chr_name, R.chr_start, R.chr_len], q_chr=[Q.chr_name, Q.chr_start, Q.chr_len], rc_q_chr=[Q.rc_name, Q.rc_start, Q.rc_len],
                          intervals=intervals, files=files, cmds=cmds))
    json.dump(dict(note="what segment_printer_body::operator() (src/segment_printer.cpp compiled as it lies; TBB's header stood in for, tests/golden/"
                        "make_printer_golden.py) writes for the listed intervals: the .segments files by name and the lastz command lines in order.  Chromosome "
                        "tables = name / start in the DRAM arena / length; HSPs = ref_start, query_start, len, score relative to their blocks.", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
