#!/usr/bin/env python3
"""Generator of tests/golden/table_golden.json: the reference's seed position table by a second route.

What runs: the reference's own text of GenerateSeedPosTable (common/seed_pos_table.cu:49-109), compiled with g++ together with the REAL
common/ntcoding.cpp (GenerateShapePos, GetKmerIndexAtPos -- the part of the reference that is pinned anyway), on small sequences.  The
function needs TBB and the CUDA runtime, which this image lacks; the harness stands in for them in the smallest possible way --
tbb::parallel_for runs its body once over the whole range (serial: positions arrive in ascending order, one of the orders the atomic
`__sync_fetch_and_add` admits), InclusivePrefixScan is std::partial_sum (the reference calls thrust::inclusive_scan on the host), and
SendSeedPosTable keeps the two arrays instead of uploading them.  A build with stand-ins does not pin the oracle;
what the vectors add is that the table's conventions -- `index_table + 1` handed on, i.e. INCLUSIVE bucket ends; offset = (shape_size + 1)
% step, start_offset = step - offset, so position 0 is never indexed for step 1 (hazard H6); windows with anything but upper-case ACGT
skipped -- come out of reference text, not out of reading it.  The extracted text goes to a temporary directory only.

usage: python tests/golden/make_table_golden.py   (needs /root/reference and g++)
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
OUT = os.path.join(HERE, "table_golden.json")

HARNESS = r'''
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cassert>
#include <numeric>
#include <string>
#include <vector>
#include "ntcoding.h"
// ---- stand-ins (this repository's code): serial TBB, host scan, no upload ----
namespace tbb {
template <class T> struct blocked_range { T b, e; blocked_range(T b_, T e_, size_t) : b(b_), e(e_) {} T begin() const { return b; } T end() const { return e; } };
template <class R, class F> void parallel_for(const R& r, F f) { f(r); }
}
static void InclusivePrefixScan(uint32_t* data, uint32_t len) { std::partial_sum(data, data + len, data); }   // seed_pos_table.cu:8-31: thrust::inclusive_scan(thrust::host, ...)
static std::vector<uint32_t> g_index, g_pos;
static void SendSeedPosTable(uint32_t* index_table, uint32_t index_table_size, uint32_t* pos_table, uint32_t num_index) {   // :33-47 uploads exactly these
  g_index.assign(index_table, index_table + index_table_size); g_pos.assign(pos_table, pos_table + num_index); }
#include "ref_table.inc"
// in: u32 len, step, shape_len ; shape string ; sequence (ASCII)   out: u32 kmer_size, n_index, n_pos ; index ; pos
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 2;
  std::string shape(hdr[2], ' ');
  std::vector<char> seq(hdr[0] + 64, 'N');
  if (fread(&shape[0], 1, hdr[2], f) != hdr[2] || fread(seq.data(), 1, hdr[0], f) != hdr[0]) return 2;
  fclose(f);
  const int kmer = GenerateShapePos(shape);                                    // src/main.cpp:286
  GenerateSeedPosTable(seq.data(), 0, hdr[0], hdr[1], (int)hdr[2], kmer);      // src/main.cpp:621
  FILE* o = fopen(argv[2], "wb");
  uint32_t oh[3] = {(uint32_t)kmer, (uint32_t)g_index.size(), (uint32_t)g_pos.size()};
  fwrite(oh, 4, 3, o); fwrite(g_index.data(), 4, g_index.size(), o); fwrite(g_pos.data(), 4, g_pos.size(), o);
  fclose(o);
  return 0;
}
'''

SHAPES = {"12of19": "TTT0T00TT00T0T0TTTT", "14of22": None, "11of18": "TT0T0T00TT00T0TTTT"[:18]}


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from segalign_amd import synth
    import re
    src = open(os.path.join(ROOT, "tests", "test_gpu_edge_cases.py")).read()
    shape22 = re.search(r'"([T0]{22})"', src).group(1)     # src/main.cpp:164-167, as the tests hold it
    lines = open(os.path.join(REF, "common", "seed_pos_table.cu")).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.startswith("void GenerateSeedPosTable"))
    last = max(i for i, l in enumerate(lines) if l.startswith("}"))
    assert (first, last) == (48, 108), (first, last)   # :49 .. :109
    tmp = tempfile.mkdtemp(prefix="sa_table_golden_")
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    open(os.path.join(tmp, "ref_table.inc"), "w").write("\n".join(lines[first:last + 1]) + "\n")
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", os.path.join(REF, "common"), os.path.join(tmp, "harness.cpp"),
                           os.path.join(REF, "common", "ntcoding.cpp"), "-o", exe])
    rng = np.random.default_rng(17)
    cases = []
    for (name, shape, step, n) in (("12of19", "TTT0T00TT00T0T0TTTT", 1, 6000), ("12of19", "TTT0T00TT00T0T0TTTT", 2, 6000), ("12of19", "TTT0T00TT00T0T0TTTT", 5, 4000),
                                   ("12of19", "TTT0T00TT00T0T0TTTT", 19, 4000), ("12of19", "TTT0T00TT00T0T0TTTT", 20, 4000), ("14of22", shape22, 1, 5000),
                                   ("14of22", shape22, 3, 5000), ("12of19", "TTT0T00TT00T0T0TTTT", 1, 40)):
        t = synth.random_dna(n, 500 + len(cases)).copy()
        if n > 1000:
            t[300:420] = np.frombuffer(bytes(t[300:420]).lower(), dtype=np.uint8)       # soft-masked: not indexed
            t[1000:1010] = ord("N")
            t[2000] = ord("&")
            t[2500] = ord("R")
            t[3000:3300] = ord("A")                                                    # one heavy bucket
            t[3500:3560] = np.frombuffer(b"AC" * 30, dtype=np.uint8)
            t[n - 25:n - 22] = ord("n")
        inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<3I", t.size, step, len(shape)))
            f.write(shape.encode())
            f.write(t.tobytes())
        subprocess.check_call([exe, inp, outp])
        raw = open(outp, "rb").read()
        kmer, n_index, n_pos = struct.unpack_from("<3I", raw, 0)
        index = np.frombuffer(raw, dtype="<u4", count=n_index, offset=12)
        pos = np.frombuffer(raw, dtype="<u4", count=n_pos, offset=12 + 4 * n_index)
        assert n_index == 4 ** kmer and int(index[-1]) == n_pos
        nz = np.nonzero(np.diff(np.concatenate([[0], index.astype(np.int64)])))[0].astype("<u4")   # keys with a non-empty bucket
        cases.append(dict(shape_name=name, shape=shape, step=step, kmer_size=int(kmer), target=t.tobytes().decode("ascii"), num_index=int(n_pos),
                          keys=pack_rows(nz.tobytes()), ends=pack_rows(index[nz].astype("<u4").tobytes()), pos=pack_rows(pos.astype("<u4").tobytes())))
        print("%s step %d, %d bp: %d positions in %d buckets, largest %d" % (name, step, n, n_pos, nz.size,
              int(np.diff(np.concatenate([[0], index[nz].astype(np.int64)])).max()) if nz.size else 0), flush=True)
    json.dump(dict(note="GenerateSeedPosTable (common/seed_pos_table.cu:49-109, reference text + the real common/ntcoding.cpp; TBB and the upload stood in for, "
                        "tests/golden/make_table_golden.py): keys = the seed keys with a non-empty bucket, ends = the table's entry for each of them (INCLUSIVE "
                        "bucket end = d_index_table[key]; every other entry repeats its predecessor), pos = the position table (serial arrival order = ascending "
                        "inside a bucket); little-endian u32, zlib + base64.", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
