#!/usr/bin/env python3
"""Generator of tests/golden/rm_golden.json: the REPEAT-MASKER fork's device code executed once, by a second route than reading it.

What runs: the reference's own text of repeat_masker_src/seed_filter.cu:45-722 -- the comparator functors (hspDiagEqual, hspDiagComp,
hspEqual, hspFinalComp, hspComp), rev_comp_string, find_num_hits, find_hits (the target-window flag, :239-244), find_hsps (the skip
of flagged hits, :305-333) and compress_output (the reverse-strand coordinate flip, :705-708) -- on the CPU under the SIMT emulation
of make_find_hsps_golden.py (one std::thread per CUDA thread).  The extracted text goes to a temporary directory only; what is
committed is data: inputs (a small self-alignment problem) and the lists every stage leaves behind.

Around the kernels the harness restates the host's orchestration for a call with num_hits < MAX_HITS (repeat_masker_src/
seed_filter.cu:754-831: inclusive scans, the two-iteration plan, stable_sort / unique_copy chain).  Those restated parts are this
repository's reading of thrust -- std::stable_sort for thrust::stable_sort, head flags on adjacent input pairs for
thrust::unique_copy (hazard H3; tests/cpp/thrust_unique.cpp checks that reading against rocThrust on the GPU box) -- so the `final`
rows are weaker evidence than the `hits` / `ext` / `reduced` rows, which come out of reference text alone.

Status under the project's rules: a build that needs a stand-in for the CUDA runtime does NOT pin the oracle; DESIGN.md section 5
keeps "parity unpinned" for a-10.  The same H1 edit as in make_find_hsps_golden.py is applied to find_hsps.

usage: python tests/golden/make_rm_golden.py   (needs /root/reference, g++ with C++20 <barrier>, oracle/liboracle.so)
"""
import base64
import json
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_find_hsps_golden import SHIM, hoxd70  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "rm_golden.json")
SHAPE = "TTT0T00TT00T0T0TTTT"

SHIM_EXTRA = r'''
// one block per listed id (find_hits is launched with one block per seed word; blocks of empty buckets do nothing and are left out)
template <class F> static void launch_blocks(const std::vector<unsigned>& ids, unsigned grid, unsigned block, F f) {
  gridDim = {grid,1,1}; blockDim = {block,1,1}; g_warps.clear();
  for (unsigned i = 0; i < block/32; i++) g_warps.emplace_back(new WarpCtx());
  g_block_bar.reset(new std::barrier<>(block));
  for (unsigned b : ids) { std::vector<std::thread> ts;
    for (unsigned t = 0; t < block; t++) ts.emplace_back([=]{ threadIdx = {t,0,0}; blockIdx = {b,0,0}; f(); });
    for (auto& t : ts) t.join(); } }
'''

HARNESS = r'''
#include "simt_shim.h"
#include "parameters.h"
#include <algorithm>
#include <numeric>
#include <cstdlib>
#include <cstring>
struct segmentPair { uint32_t ref_start; uint32_t query_start; uint32_t len; int score; };
#include "ref_rm.inc"
// in : u32 ref_len, num_seeds, rev, win_start, win_end, noentropy, seed_size, num_pos ; i32 xdrop, hspthresh ; 64 x i32 matrix ;
//      ref codes ; seeds (u64) ; index table (4^12 x u32, inclusive ends) ; pos table (num_pos x u32)
// out: sections, each "u32 tag, u32 rows" + rows: 1 = hits (find_hits), 2 = ext (find_hsps) {seg, done}, 3 = reduced (compress_output),
//      4 = final (sort / unique chain), 5 = rc codes (one byte per row)
static void put(FILE* o, uint32_t tag, const void* p, size_t rows, size_t row_bytes) {
  uint32_t h[2] = {tag, (uint32_t)rows}; fwrite(h, 4, 2, o); if (rows) fwrite(p, row_bytes, rows, o); }
template <class Eq> static size_t unique_adjacent(const std::vector<segmentPair>& in, size_t n, std::vector<segmentPair>& out, Eq eq) {
  size_t m = 0;   // thrust::unique_copy on the device back ends: head flags on ADJACENT INPUT pairs (hazard H3) -- restated
  for (size_t i = 0; i < n; i++) if (i == 0 || !eq(in[i - 1], in[i])) out[m++] = in[i];
  return m; }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[8]; int par[2]; int mat[64];
  if (fread(hdr, 4, 8, f) != 8 || fread(par, 4, 2, f) != 2 || fread(mat, 4, 64, f) != 64) return 2;
  const uint32_t ref_len = hdr[0], num_seeds = hdr[1], win_start = hdr[3], win_end = hdr[4], seed_size = hdr[6], num_pos = hdr[7];
  const bool rev = hdr[2] != 0, noentropy = hdr[5] != 0;
  std::vector<char> ref(ref_len), rc(ref_len);
  std::vector<uint64_t> seeds(num_seeds), hit_num(num_seeds);
  std::vector<uint32_t> index(1u << 24), pos(num_pos);
  if (fread(ref.data(), 1, ref_len, f) != ref_len || fread(seeds.data(), 8, num_seeds, f) != num_seeds ||
      fread(index.data(), 4, index.size(), f) != index.size() || fread(pos.data(), 4, num_pos, f) != num_pos) return 2;
  fclose(f);
  launch(2, 64, [&]{ rev_comp_string(ref_len, ref.data(), rc.data()); });                                     // :959
  launch(2, 64, [&]{ find_num_hits((int)num_seeds, index.data(), seeds.data(), hit_num.data()); });           // :754
  std::partial_sum(hit_num.begin(), hit_num.end(), hit_num.begin());                                           // thrust::inclusive_scan :756
  const uint64_t num_hits = hit_num[num_seeds - 1];
  // the plan of :760-787 for num_hits < MAX_HITS: everything before the last hit-bearing seed word, then that word and the rest
  int64_t limit_pos[2]; int num_iter = 2;
  limit_pos[0] = (int64_t)(std::lower_bound(hit_num.begin(), hit_num.end(), num_hits) - hit_num.begin()) - 1;
  limit_pos[1] = (int64_t)num_seeds - 1;
  if (limit_pos[0] < 0) return 3;                       // (hazard H5: the reference reads prefix[-1]; the designed cases avoid it)
  if (limit_pos[1] == limit_pos[0]) num_iter = 1;
  std::vector<segmentPair> all_hits, all_ext, all_red, all_fin; std::vector<uint32_t> all_done;
  uint32_t start_seed_index = 0; uint64_t start_hit_val = 0;
  for (int it = 0; it < num_iter; it++) {
    const uint32_t iter_num_seeds = (uint32_t)(limit_pos[it] + 1 - start_seed_index);
    const uint64_t upto = hit_num[limit_pos[it]], iter_num_hits = upto - start_hit_val;
    if (iter_num_seeds > 0 && iter_num_hits > 0) {
      std::vector<segmentPair> hsp(iter_num_hits, segmentPair{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, -7}), red(iter_num_hits), tmp(iter_num_hits);
      std::vector<uint32_t> done(iter_num_hits, 7u), ids;
      for (uint32_t b = 0; b < iter_num_seeds; b++) {
        const uint32_t s = b + start_seed_index;
        if (hit_num[s] != (s ? hit_num[s - 1] : 0)) ids.push_back(b);
      }
      launch_blocks(ids, iter_num_seeds, BLOCK_SIZE, [&]{ find_hits(index.data(), pos.data(), seeds.data(), seed_size, hit_num.data(), iter_num_hits,
                                                                    hsp.data(), start_seed_index, start_hit_val, win_start, win_end); });   // :803
      all_hits.insert(all_hits.end(), hsp.begin(), hsp.end());
      launch(16, BLOCK_SIZE, [&]{ find_hsps(ref.data(), rev ? rc.data() : ref.data(), ref_len, ref_len, mat, noentropy, par[0], par[1], (int)iter_num_hits,
                                            hsp.data(), done.data()); });                                                                  // :806-809
      all_ext.insert(all_ext.end(), hsp.begin(), hsp.end());
      all_done.insert(all_done.end(), done.begin(), done.end());
      std::partial_sum(done.begin(), done.end(), done.begin());                                                                            // :812
      size_t na = done[iter_num_hits - 1];
      if (na > 0) {
        launch(2, 64, [&]{ compress_output(done.data(), hsp.data(), red.data(), (int)iter_num_hits, rev, ref_len); });                     // :817
        all_red.insert(all_red.end(), red.begin(), red.begin() + na);
        std::stable_sort(red.begin(), red.begin() + na, hspComp());                                                                        // :819
        na = unique_adjacent(red, na, tmp, hspEqual());                                                                                    // :821
        std::stable_sort(tmp.begin(), tmp.begin() + na, hspDiagComp());                                                                    // :825
        na = unique_adjacent(tmp, na, red, hspDiagEqual());                                                                                // :827
        std::stable_sort(red.begin(), red.begin() + na, hspFinalComp());                                                                   // :831
        all_fin.insert(all_fin.end(), red.begin(), red.begin() + na);
      }
    }
    start_seed_index = (uint32_t)(limit_pos[it] + 1);
    start_hit_val = upto;
  }
  FILE* o = fopen(argv[2], "wb");
  put(o, 1, all_hits.data(), all_hits.size(), 16);
  std::vector<uint32_t> ext5(all_ext.size() * 5);
  for (size_t i = 0; i < all_ext.size(); i++) { memcpy(&ext5[5 * i], &all_ext[i], 16); ext5[5 * i + 4] = all_done[i]; }
  put(o, 2, ext5.data(), all_ext.size(), 20);
  put(o, 3, all_red.data(), all_red.size(), 16);
  put(o, 4, all_fin.data(), all_fin.size(), 16);
  put(o, 5, rc.data(), ref_len, 1);
  fclose(o);
  return 0;
}
'''


def design(seed, records=9, rec_len=270):
    """A small self-alignment problem.  Short records ('&'-joined like the reference's arena, src/main.cpp:320-549) keep every extension
    short -- in a self-alignment each hit on the main diagonal extends to its record's ends, and the emulation pays per base --; a
    100 bp repeat unit sits in most records, forward or reverse-complemented, at 3-9 % divergence (off-diagonal HSPs on both strands);
    plus a microsatellite, a lower-case run (not seeded), an N and another IUPAC letter."""
    from segalign_amd import synth
    rng = np.random.default_rng(seed)
    unit = synth.random_dna(100, 200 + seed)
    recs = []
    for i in range(records):
        r = synth.random_dna(rec_len + int(rng.integers(-20, 21)), 100 * seed + i).copy()
        if i % 4 != 3:
            cp = synth.mutate(unit, 300 + 10 * seed + i, 0.03 + 0.0075 * i)
            cp = cp if i % 3 else synth.reverse_complement(cp)
            at = int(rng.integers(20, r.size - cp.size - 20))
            r[at:at + cp.size] = cp
        recs.append(r)
    recs[3][40:100] = np.frombuffer(b"CAG" * 20, dtype=np.uint8)
    recs[5][150:200] = np.frombuffer(bytes(recs[5][150:200]).lower(), dtype=np.uint8)
    recs[1][int(rng.integers(30, 200))] = ord("N")
    recs[6][int(rng.integers(30, 200))] = ord("R")
    return np.concatenate([np.concatenate([r, np.frombuffer(b"&", dtype=np.uint8)]) for r in recs])[:-1].copy()


def unpack(buf):
    out, off = {}, 0
    while off < len(buf):
        tag, rows = struct.unpack_from("<2I", buf, off)
        off += 8
        width = {1: 16, 2: 20, 3: 16, 4: 16, 5: 1}[tag]
        out[tag] = buf[off:off + rows * width]
        off += rows * width
    return out


def pack_rows(raw):
    return base64.b64encode(zlib.compress(raw, 9)).decode()


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from oracle import oracle as O
    O.build(with_ref=False)
    src = os.path.join(REF, "repeat_masker_src", "seed_filter.cu")
    lines = open(src).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.startswith("struct hspDiagEqual"))
    last = next(i for i, l in enumerate(lines) if l.startswith("std::vector<segmentPair> SeedAndFilter"))
    assert (first, last) == (44, 723), (first, last)   # :45 .. :723 (1-based), as cited in the doc string
    tmp = tempfile.mkdtemp(prefix="sa_rm_golden_")
    open(os.path.join(tmp, "simt_shim.h"), "w").write(SHIM + SHIM_EXTRA)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    inc = os.path.join(tmp, "ref_rm.inc")
    open(inc, "w").write("\n".join(lines[first:last]) + "\n")
    subprocess.check_call(["sed", "-i", r"s/short count\[4\];/short count[8] = {0};/; s/short count_del\[4\];/short count_del[8] = {0};/; "
                                        r"s/^    char r_chr;/    char r_chr = 0;/; s/^    char q_chr;/    char q_chr = 1;/", inc])
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-pthread", "-w", "-I", tmp, "-I", os.path.join(REF, "common"),
                           os.path.join(tmp, "harness.cpp"), "-o", exe])
    k = O.generate_shape_pos(SHAPE)
    cases = []
    for seed in (1, 2):
        t = design(seed)
        L = t.size
        codes = O.encode(t.tobytes())
        index, pos = O.generate_seed_pos_table(t.tobytes(), 0, L, 1, 19, k)
        rc_ascii = O.rev_comp_ascii(t.tobytes(), 0, L)
        for rev in (0, 1):
            for (ws, we, hspthresh, noentropy) in ((0, L, 3000, 0), (L // 3, 2 * L // 3, 3000, 0), (L // 4, L // 2, 2200, 1)):
                start, end = (0, L // 2) if ws == 0 else (L // 5, 4 * L // 5)   # (seed range of the call; the flagged hits lie outside the window)
                buf = rc_ascii if rev else t.tobytes()
                seeds = O.make_seeds(buf, 0, start, end, 19, k, True)
                mat = hoxd70(910)
                inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
                with open(inp, "wb") as f:
                    f.write(struct.pack("<8I2i", L, seeds.size, rev, ws, we, noentropy, 19, pos.size, 910, hspthresh))
                    f.write(mat.astype("<i4").tobytes())
                    f.write(np.ascontiguousarray(codes, np.uint8).tobytes())
                    f.write(np.ascontiguousarray(seeds, "<u8").tobytes())
                    f.write(np.ascontiguousarray(index, "<u4").tobytes())
                    f.write(np.ascontiguousarray(pos, "<u4").tobytes())
                subprocess.check_call([exe, inp, outp])
                sec = unpack(open(outp, "rb").read())
                n_hits, n_red, n_fin = len(sec[1]) // 16, len(sec[3]) // 16, len(sec[4]) // 16
                hits = np.frombuffer(sec[1], dtype="<i4").reshape(-1, 4)
                print("seed %d rev %d window [%d, %d] hspthresh %d noentropy %d: %d seeds, %d hits (%d outside the window), %d anchors, %d final"
                      % (seed, rev, ws, we, hspthresh, noentropy, seeds.size, n_hits, int(np.count_nonzero(hits[:, 3] < 0)), n_red, n_fin), flush=True)
                cases.append(dict(seed=seed, target=t.tobytes().decode("ascii"), rev=rev, win_start=ws, win_end=we, start=start, end=end,
                                  xdrop=910, hspthresh=hspthresh, noentropy=noentropy, num_seeds=int(seeds.size), sub_mat=mat.tolist(),
                                  hits=pack_rows(sec[1]), ext=pack_rows(sec[2]), reduced=pack_rows(sec[3]), final=pack_rows(sec[4]),
                                  rc_codes=pack_rows(sec[5])))
    json.dump(dict(note="lists left behind by the repeat-masker fork's device code (repeat_masker_src/seed_filter.cu:45-722) under the SIMT emulation of "
                        "tests/golden/make_rm_golden.py.  hits / reduced / final rows = ref_start, query_start, len, score (little-endian int32 x 4); "
                        "ext rows = the same + done; rc_codes = one code per base; every list zlib + base64.  Seeds = the host loop's words "
                        "(seeder.cpp:57-74, oracle make_seeds, pinned to ntcoding.cpp) for [start, end) of the target (rev: of its reverse "
                        "complement), transitions on, shape 12of19.", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
