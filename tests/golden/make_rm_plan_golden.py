#!/usr/bin/env python3
"""Generator of tests/golden/rm_plan_golden.json: the repeat masker's block / interval plan by a second route.

What runs: the reference's own text of repeat_masker_src/main.cpp:259-262 (the block size rounded down to whole intervals) and :323-433
(neighbour intervals, overlaps, blocks, per interval the seed range and the target window -- float / ceil arithmetic included), verbatim
inside a function of the harness, compiled with g++ against the fork's own graph.h (TBB's header stood in for as in
make_seeder_golden.py).  The harness (this repository's code) declares the two vectors main.cpp declares at file scope and cfg, sets the
five configuration values and prints the lists.  A second route for the plan restated in oracle/segalign_oracle.c (orc_rm_plan) and
segalign_amd/shard.py (rm_plan), not a pin.

usage: python tests/golden/make_rm_plan_golden.py   (needs /root/reference and g++)
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_printer_golden import FAKE_TBB  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "rm_plan_golden.json")

HARNESS = r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "graph.h"
Configuration cfg;
std::vector<size_t>   block_start;      // repeat_masker_src/main.cpp:35-36
std::vector<uint32_t> block_len;
int main(int argc, char** argv) {
  cfg.seq_len = strtoull(argv[1], 0, 10); cfg.seq_block_size = (uint32_t)strtoul(argv[2], 0, 10); cfg.lastz_interval_size = (uint32_t)strtoul(argv[3], 0, 10);
  cfg.prop_neigh_interval = (float)atof(argv[4]); cfg.seed.size = atoi(argv[5]); cfg.debug = false;
#include "ref_plan_a.inc"
#include "ref_plan_b.inc"
  size_t k = 0;
  for (size_t b = 0; b < block_num_intervals.size(); b++)
    for (uint32_t i = 0; i < block_num_intervals[b]; i++, k++)
      printf("%zu %zu %u %u %u %u %u\n", b, block_start[b], block_len[b], interval_list[k].start, interval_list[k].end, interval_list[k].ref_start, interval_list[k].ref_end);
  return 0;
}
'''


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    lines = open(os.path.join(REF, "repeat_masker_src", "main.cpp")).read().split("\n")
    a0 = next(i for i, l in enumerate(lines) if "if(cfg.seq_block_size == DEFAULT_SEQ_BLOCK_SIZE){" in l)
    assert a0 == 258 and lines[a0 + 3].strip() == "}", a0                      # :259-262
    b0 = next(i for i, l in enumerate(lines) if l.strip().startswith("uint32_t total_query_intervals = ceil("))
    b1 = next(i for i, l in enumerate(lines) if l.strip() == "total_query_intervals = interval_list.size();")
    assert (b0, b1) == (322, 433), (b0, b1)                                    # :323-433
    tmp = tempfile.mkdtemp(prefix="sa_rm_plan_golden_")
    os.makedirs(os.path.join(tmp, "tbb"))
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB)
    open(os.path.join(tmp, "ref_plan_a.inc"), "w").write("\n".join(lines[a0:a0 + 4]) + "\n")
    open(os.path.join(tmp, "ref_plan_b.inc"), "w").write("\n".join(lines[b0:b1]) + "\n")
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", tmp, "-I", os.path.join(REF, "repeat_masker_src"), "-I", os.path.join(REF, "common"),
                           os.path.join(tmp, "harness.cpp"), "-o", exe])
    cases = []
    for (seq_len, block, interval, prop, seed) in ((100_000_000, 1_000_000_000, 10_000_000, 0.2, 19), (100_286_401, 1_000_000_000, 10_000_000, 0.2, 19),
                                                   (3_100_000_000, 1_000_000_000, 10_000_000, 0.2, 19), (2_500_000_123, 1_000_000_000, 10_000_000, 0.05, 19),
                                                   (95_000_000, 30_000_000, 10_000_000, 0.3, 19), (95_000_000, 30_000_000, 10_000_000, 1.0, 22),
                                                   (1_234_567, 400_000, 100_000, 0.25, 19), (1_234_567, 1_000_000_000, 100_000, 0.0, 19), (50_000, 1_000_000_000, 10_000_000, 0.2, 19),
                                                   (777_777, 250_000, 70_000, 0.5, 19), (4_000_000_000, 1_000_000_000, 10_000_000, 0.01, 19)):
        out = subprocess.check_output([exe, str(seq_len), str(block), str(interval), repr(prop), str(seed)]).decode()
        rows = [[int(x) for x in l.split()] for l in out.split("\n") if l]
        print("seq_len %d block %d interval %d prop %g: %d blocks, %d interval tasks" % (seq_len, block, interval, prop, len({r[0] for r in rows}), len(rows)), flush=True)
        cases.append(dict(seq_len=seq_len, seq_block_size=block, lastz_interval_size=interval, prop_neigh_interval=prop, seed_size=seed, tasks=rows))
    json.dump(dict(note="repeat_masker_src/main.cpp:259-262 + :323-433 (reference text inside a harness function, tests/golden/make_rm_plan_golden.py): per interval task "
                        "block index, block start, block length, seed range start / end, target window ref_start / ref_end", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
