#!/usr/bin/env python3
"""Generator of tests/golden/rm_path_golden.json: the repeat masker binary's whole path executed from the reference's own files, end to end.

What runs -- every file as it lies under /root/reference, compiled with g++ into ONE program (the method of make_path_golden.py):
    repeat_masker_src/seed_filter.cu     InitializeProcessor, SendQueryWriteRequest (rev_comp_string), SeedAndFilter (u64 scans, the plan over MAX_HITS,
                                         sort / unique / diagonal sort / diagonal unique / final sort) and every kernel of the fork
    common/seed_filter_interface.cu      InitializeInterface, SendRefWriteRequest (compress_string)
    common/seed_pos_table.cu             GenerateSeedPosTable
    common/ntcoding.cpp                  k-mers, RevComp                                                         (unedited, no stand-in)
    repeat_masker_src/seeder.cpp         the chunk loop, both strands, uint8 coverage, run extraction                  (unedited)
    repeat_masker_src/segment_printer.cpp    the .intervals text                                                       (unedited)
in the order repeat_masker_src/main.cpp calls them (:256-257, :311, :498-499, :505).  Recorded per interval task: every g_SeedAndFilter return
(64-bit header + HSPs in the reference's order), the runs the seeder returns, the text of the .intervals file.

Stand-ins (this repository's code, temporary directory only): as in make_path_golden.py (CUDA runtime names, `<<< >>>` -> the fiber SIMT
emulation, thrust over std:: with adjacent-input unique_copy, serial TBB, DRAM's constructor, the H1 edit) + tbb/scalable_allocator.h = calloc /
free.  A build with stand-ins does not pin the oracle.  What the vectors add over rm_golden.json (kernels, orchestration
restated) and rm_host_golden.json (host files, designed HSPs): the fork's own orchestration text and the two halves joined -- real HSPs of a
self-alignment (the diagonal, tandem and inverted repeats) through the coverage counters into the file.

usage: python tests/golden/make_rm_path_golden.py   (needs /root/reference and g++)
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
from make_find_hsps_golden import hoxd70  # noqa: E402
from make_path_golden import FAKE_TBB_SORT, FAKE_THRUST, PRELUDE, S19, edited_copy, mutate  # noqa: E402
from make_printer_golden import FAKE_TBB  # noqa: E402
from make_rm_golden import pack_rows  # noqa: E402
from make_rm_host_golden import FAKE_ALLOC  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "rm_path_golden.json")

HARNESS = r'''
#include <string>
#include "graph.h"
#include "ntcoding.h"
#include "seed_filter.h"
#include "seed_filter_interface.h"
#include "store.h"
// ---- what repeat_masker_src/main.cpp owns (this repository's code) ----
size_t sa_fake_global_mem;
Configuration cfg;
DRAM *seq_DRAM, *seq_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }
DRAM::~DRAM() {}
std::vector<std::string> chr_name; std::vector<size_t> chr_start; std::vector<uint32_t> chr_len;
static FILE* g_out;
static SeedAndFilter_ptr g_real;
static std::vector<segmentPair> record(std::vector<uint64_t> seeds, bool rev, uint32_t ref_start, uint32_t ref_end) {
  std::vector<segmentPair> r = g_real(seeds, rev, ref_start, ref_end);
  uint32_t h[9] = {rev ? 1u : 0u, ref_start, ref_end, (uint32_t)seeds.size(), (uint32_t)r.size() - 1u, r[0].ref_start, r[0].query_start, r[0].len, (uint32_t)r[0].score};
  fwrite(h, 4, 9, g_out); fwrite(r.data() + 1, 16, r.size() - 1, g_out);
  return r;
}
// in : u64 fake_mem ; u32 seq_len, block_start, block_len, chunk, transition, strand, M, markend, step, shape_len, noentropy, n_chr, n_intervals ; i32 xdrop, hspthresh ;
//      64 x i32 matrix ; shape ; sequence ; n_chr x {name_len, name, start, len} ; n_intervals x {start, end, ref_start, ref_end}
// out: calls as written by record; per interval AFTER its calls a marker {0xFFFFFFFF, k, n_runs, 0 x 6} + runs {query_start, len}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint64_t mem; uint32_t hdr[13]; int par[2]; int mat[64];
  if (fread(&mem, 8, 1, f) != 1 || fread(hdr, 4, 13, f) != 13 || fread(par, 4, 2, f) != 2 || fread(mat, 4, 64, f) != 64) return 2;
  std::string shape(hdr[9], ' ');
  std::vector<char> fw(hdr[0] + 64, 'N'), rc(hdr[0] + 64, 'N');
  if (fread(&shape[0], 1, hdr[9], f) != hdr[9] || fread(fw.data(), 1, hdr[0], f) != hdr[0]) return 2;
  for (uint32_t i = 0; i < hdr[11]; i++) { uint32_t nl, st, ln; if (fread(&nl, 4, 1, f) != 1) return 2; std::string nm(nl, ' ');
    if (fread(&nm[0], 1, nl, f) != nl || fread(&st, 4, 1, f) != 1 || fread(&ln, 4, 1, f) != 1) return 2; chr_name.push_back(nm); chr_start.push_back(st); chr_len.push_back(ln); }
  std::vector<uint32_t> iv(4 * hdr[12]);
  if (fread(iv.data(), 4, iv.size(), f) != iv.size()) return 2;
  fclose(f);
  sa_fake_global_mem = (size_t)mem;
  cfg.seed.shape = shape; cfg.seed.size = (int)shape.size(); cfg.seed.kmer_size = GenerateShapePos(shape);
  cfg.seed.transition = hdr[4] != 0; cfg.wga_chunk_size = hdr[3]; cfg.M = hdr[6]; cfg.markend = hdr[7] != 0; cfg.step = hdr[8];
  cfg.strand = hdr[5] == 1 ? "plus" : hdr[5] == 2 ? "minus" : "both";
  cfg.xdrop = par[0]; cfg.hspthresh = par[1]; cfg.noentropy = hdr[10] != 0;
  for (int i = 0; i < 64; i++) cfg.sub_mat[i] = mat[i];
  cfg.num_gpu = g_InitializeInterface(1);                                                                                   // main.cpp:256
  g_InitializeProcessor(cfg.seed.transition, cfg.wga_chunk_size, cfg.seed.size, cfg.sub_mat, cfg.xdrop, cfg.hspthresh, cfg.noentropy);    // :257
  cfg.seq_len = hdr[0];                                                                                                      // :309
  seq_DRAM = new DRAM; seq_rc_DRAM = new DRAM; seq_DRAM->buffer = fw.data(); seq_rc_DRAM->buffer = rc.data();
  RevComp(seq_rc_DRAM->buffer, seq_DRAM->buffer, 0, 0, cfg.seq_len);                                                         // :311
  g_SendRefWriteRequest(seq_DRAM->buffer, hdr[1], hdr[2]);                                                                   // :498
  g_SendQueryWriteRequest();                                                                                                 // :499
  GenerateSeedPosTable(seq_DRAM->buffer, hdr[1], hdr[2], cfg.step, cfg.seed.size, cfg.seed.kmer_size);                       // :505
  g_real = g_SeedAndFilter; g_SeedAndFilter = record;
  g_out = fopen(argv[2], "wb");
  seeder_body seeder; interval_printer_body printer; printer_node::output_ports_type ports;
  for (uint32_t k = 0; k < hdr[12]; k++) {
    seq_block b; b.index = 0; b.start = hdr[1]; b.len = hdr[2];
    seed_interval s; s.start = iv[4 * k]; s.end = iv[4 * k + 1]; s.ref_start = iv[4 * k + 2]; s.ref_end = iv[4 * k + 3]; s.num_invoked = k + 1; s.num_intervals = hdr[12];
    printer_input out = seeder(seeder_input(seeder_payload(b, s), (size_t)k));
    const interval_output& runs = get<2>(get<0>(out));
    uint32_t mark[9] = {0xFFFFFFFFu, k, (uint32_t)runs.size(), 0, 0, 0, 0, 0, 0};
    fwrite(mark, 4, 9, g_out);
    for (auto& r : runs) { uint32_t p[2] = {r.query_start, r.len}; fwrite(p, 4, 2, g_out); }
    printer(out, ports);                                                       // writes tmp<k+1>.block0.intervals into the working directory
  }
  fclose(g_out);
  return 0;
}
'''


def design(seed, rec_lens):
    """'&'-joined records with what a repeat masker looks for: a tandem repeat, diverged copies of pieces elsewhere in the sequence, half of
    them reverse-complemented (so that minus-strand calls have hits too), a soft-masked and an N stretch."""
    from segalign_amd import synth
    rng = np.random.default_rng(seed)
    recs = [synth.random_dna(n, 9500 + 10 * seed + i).copy() for i, n in enumerate(rec_lens)]
    seq = np.concatenate([np.concatenate([r, np.frombuffer(b"&", dtype=np.uint8)]) for r in recs])[:-1].copy()
    L = seq.size
    comp = np.full(256, ord("N"), np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    at = 150
    while at + 260 < L:
        n = int(rng.integers(70, 160))
        src = int(rng.integers(0, L - n))
        piece = mutate(rng, seq[src:src + n], 0.05)
        piece[piece == ord("&")] = ord("A")
        if rng.random() < 0.5:
            piece = comp[piece[::-1]]
        keep = seq[at:at + n] == ord("&")
        seq[at:at + n] = np.where(keep, seq[at:at + n], piece)
        at += n + int(rng.integers(60, 200))
    unit = synth.random_dna(23, 9600 + seed)
    for c in range(9):
        p = L // 4 + 23 * c
        seq[p:p + 23] = np.where(seq[p:p + 23] == ord("&"), seq[p:p + 23], mutate(rng, unit, 0.03))
    seq[60:130] = np.frombuffer(bytes(seq[60:130]).lower(), dtype=np.uint8)
    seq[L - 420:L - 380] = np.where(seq[L - 420:L - 380] == ord("&"), ord("&"), ord("N"))
    names = ["chr%d" % (i + 1) for i in range(len(recs))]
    starts = [int(sum(len(r) + 1 for r in recs[:i])) for i in range(len(recs))]
    return seq, names, starts, [len(r) for r in recs]


def check_design(seq, bs, bl, chunk, trans, strand, step, ivs, mem):
    """every call must be one the reference's plan can run: hits beyond its first seed word (H5), no seed word with MAX_HITS hits, no iteration
    (the last one included) above MAX_HITS (H16) -- see make_path_golden.check_design"""
    from oracle import oracle as O
    from host_model import rm_chunk_calls
    O.build(with_ref=False)
    k = O.generate_shape_pos(S19)
    raw = seq.tobytes()
    L = len(raw)
    index, _ = O.generate_seed_pos_table(raw, bs, bl, step, 19, k)
    counts = np.diff(np.concatenate([[0], index.astype(np.int64)]))
    max_hits = int(np.float32(4194304) * np.float32(mem / 1073741824.0))
    rc = O.rev_comp_ascii(raw, 0, L)
    rc_block_start = L - 1 - bs - (bl - 1)
    for (s, e, ws, we) in ivs:
        for (rev, s0, s1) in rm_chunk_calls(s, e, bl, chunk, strand):
            seeds = O.make_seeds(rc, rc_block_start, s0, s1, 19, k, bool(trans)) if rev else O.make_seeds(raw, bs, s0, s1, 19, k, bool(trans))
            if seeds.size == 0:
                continue
            assert seeds.size <= (13 if trans else 1) * chunk
            per = counts[(seeds >> np.uint64(32)).astype(np.int64)]
            assert per.sum() > per[0], ("a call whose hits all sit on its first seed word", s, e, rev, s0, s1, int(per.sum()))
            assert per.max() < max_hits, ("a seed word with MAX_HITS hits", int(per.max()))
            scan = np.cumsum(per)
            num_hits = int(scan[-1])
            if num_hits >= max_hits:
                num_iter, limit, start = num_hits // max_hits + 2, max_hits, 0
                for i in range(num_iter - 1):
                    pos = int(np.searchsorted(scan, limit, side="left")) - 1
                    assert pos >= 0 and int(scan[pos]) - start <= max_hits and (i == 0 or int(scan[pos]) > start), ("plan", i, pos)
                    start = int(scan[pos])
                    limit = min(start + max_hits, num_hits)
                assert num_hits - start <= max_hits, ("the last iteration overruns the buffers", num_hits - start, max_hits)


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    tmp = tempfile.mkdtemp(prefix="sa_rm_path_golden_")
    for d in ("thrust/iterator", "tbb"):
        os.makedirs(os.path.join(tmp, d))
    open(os.path.join(tmp, "prelude.h"), "w").write(PRELUDE)
    for h in ("binary_search.h", "device_vector.h", "execution_policy.h", "iterator/constant_iterator.h", "scan.h", "unique.h"):
        open(os.path.join(tmp, "thrust", h), "w").write(FAKE_THRUST)
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "tbb", "parallel_sort.h"), "w").write(FAKE_TBB_SORT)
    open(os.path.join(tmp, "tbb", "scalable_allocator.h"), "w").write(FAKE_ALLOC)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    srcs, launches = [os.path.join(tmp, "harness.cpp")], 0
    for rel, h1 in (("repeat_masker_src/seed_filter.cu", True), ("common/seed_filter_interface.cu", False), ("common/seed_pos_table.cu", False)):
        p, n = edited_copy(tmp, rel, h1)
        srcs.append(p)
        launches += n
    assert launches == 7, launches   # find_num_hits, find_hits, find_hsps x 2, compress_output, rev_comp_string, compress_string
    rm = os.path.join(REF, "repeat_masker_src")
    srcs += [os.path.join(rm, "seeder.cpp"), os.path.join(rm, "segment_printer.cpp"), os.path.join(REF, "common", "ntcoding.cpp")]
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-w", "-include", os.path.join(tmp, "prelude.h"), "-I", tmp, "-I", rm, "-I", os.path.join(REF, "common")]
                          + srcs + ["-o", exe])
    cases = []
    GB = 1 << 30
    #        trans strand chunk M markend step rec_lens            block          mem       xdrop hspthresh noentropy ivlen window
    plan = ((1,    3,     700,  1, 0,      1,   (1500, 900, 1700),  None,          GB,       910,  3000,     0,        1400,  1000),
            (1,    3,     600,  2, 1,      1,   (2000, 1500),       None,          1 << 17,  910,  3000,     0,        1200,  900),    # MAX_HITS 512
            (0,    3,     500,  1, 0,      1,   (1800, 1600),       (500, 2600),   1 << 16,  910,  3000,     1,        1000,  3000),   # a block inside the sequence, --noentropy
            (1,    2,     800,  1, 1,      1,   (2600,),            None,          1 << 17,  500,  2200,     0,        1600,  700))    # minus strand only, other thresholds
    for ci, (trans, strand, chunk, M, markend, step, rec_lens, block, mem, xdrop, hspthresh, noentropy, ivlen, window) in enumerate(plan):
        for attempt in range(40):
            seq, names, starts, lens = design(100 * ci + attempt, rec_lens)
            bs, bl = block if block else (0, seq.size)
            ivs = []
            for s in range(0, bl - 19, ivlen):   # interval tasks with a target window around the interval (repeat_masker_src/main.cpp:367-420 makes such windows)
                e = min(s + ivlen, bl - 19)
                ivs.append((s, e, max(0, s - window), min(bl, e + window)))
            try:
                check_design(seq, bs, bl, chunk, trans, strand, step, ivs, mem)
                break
            except AssertionError as e:
                print("case %d design %d: %s" % (ci, attempt, e.args[0][0] if e.args else e), flush=True)
        else:
            raise SystemExit("no design for case %d" % ci)
        mat = hoxd70(xdrop)
        wd = os.path.join(tmp, "case%d" % ci)
        os.makedirs(wd)
        inp, outp = os.path.join(wd, "in.bin"), os.path.join(wd, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<Q13I2i", mem, seq.size, bs, bl, chunk, trans, strand, M, markend, step, len(S19), noentropy, len(names), len(ivs), xdrop, hspthresh))
            f.write(mat.astype("<i4").tobytes())
            f.write(S19.encode())
            f.write(seq.tobytes())
            for nm, st, ln in zip(names, starts, lens):
                f.write(struct.pack("<I", len(nm)) + nm.encode() + struct.pack("<2I", st, ln))
            for t in ivs:
                f.write(struct.pack("<4I", *t))
        subprocess.check_call([exe, inp, outp], cwd=wd, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        raw = open(outp, "rb").read()
        off, tasks, cur = 0, [], []
        while off < len(raw):
            h = struct.unpack_from("<9I", raw, off)
            off += 36
            if h[0] == 0xFFFFFFFF:
                k, nr = h[1], h[2]
                runs = np.frombuffer(raw, dtype="<u4", count=2 * nr, offset=off).reshape(-1, 2).tolist()
                off += 8 * nr
                fn = os.path.join(wd, "tmp%d.block0.intervals" % (k + 1))
                tasks.append(dict(interval=list(ivs[k]), calls=cur, runs=runs, file=open(fn).read() if os.path.exists(fn) else None))
                cur = []
                continue
            num_hits, anchors = h[5] | (h[6] << 32), h[7] | (h[8] << 32)
            assert anchors == h[4], h
            cur.append(dict(rev=h[0], ref_start=h[1], ref_end=h[2], n_seeds=h[3], n_hsps=h[4], num_hits=num_hits, hsps=pack_rows(raw[off:off + 16 * h[4]])))
            off += 16 * h[4]
        max_hits = int(np.float32(4194304) * np.float32(mem / 1073741824.0))
        calls = [c for t in tasks for c in t["calls"]]
        print("case %d: %d bp, block (%d, %d), %d tasks, %d calls, %d HSPs, %d seed hits, MAX_HITS %d (calls above it: %d), %d runs, %d files" %
              (ci, seq.size, bs, bl, len(tasks), len(calls), sum(c["n_hsps"] for c in calls), sum(c["num_hits"] for c in calls), max_hits,
               sum(c["num_hits"] >= max_hits for c in calls), sum(len(t["runs"]) for t in tasks), sum(t["file"] is not None for t in tasks)), flush=True)
        cases.append(dict(shape=S19, transition=trans, strand=strand, chunk=chunk, M=M, markend=markend, step=step, total_global_mem=mem, max_hits=max_hits,
                          xdrop=xdrop, hspthresh=hspthresh, noentropy=noentropy, sub_mat=mat.tolist(), seq=seq.tobytes().decode("ascii"),
                          block_start=bs, block_len=bl, chr=[names, starts, lens], tasks=tasks))
    json.dump(dict(note="the repeat masker binary's own files run end to end (tests/golden/make_rm_path_golden.py: repeat_masker_src/seed_filter.cu, seeder.cpp, "
                        "segment_printer.cpp, common/seed_filter_interface.cu, seed_pos_table.cu, ntcoding.cpp; CUDA runtime / thrust / TBB stood in for, kernels under SIMT "
                        "emulation): per interval task {start, end, ref_start, ref_end} every g_SeedAndFilter return in call order (strand, window, seed words handed over, "
                        "seed hits and HSP count of the 64-bit header, HSPs as u32 x 3 + i32 rows, zlib + base64), the runs the seeder returns, the .intervals text (null: none).",
                   cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
