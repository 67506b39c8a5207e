"""CPU: the repeat masker's host side as the repository restates it, against the reference's own object code (repeat_masker_src/seeder.cpp
and segment_printer.cpp compiled as they lie + the real ntcoding.cpp, TBB stood in for, g_SeedAndFilter answering with designed HSPs:
tests/golden/make_rm_host_golden.py):
  * which seed ranges are handed to g_SeedAndFilter, in which order, with which window -- tests/host_model.py::rm_chunk_calls (the minus
    piece derived from the plus piece's end, seeder.cpp:118-119) + the oracle's seed words;
  * the coverage counting and run extraction on the returned HSPs -- oracle.rm_coverage_intervals (uint8 counters: a pile of 300 HSPs
    wraps to 44 and counts as covered, 256 would not; a run still open at the block end is dropped, seeder.cpp:153-188);
  * the text of tmp<i>.block<b>.intervals -- tests/host_model.py::rm_interval_lines (segment_printer.cpp:8-65, --markend included).
A second route for a-10's host half and 8f-4."""
import json
import os

import numpy as np
import pytest

from host_model import rm_chunk_calls, rm_interval_lines
from rm_golden import SEG, _rows

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_host_golden.json")
CASES = json.load(open(PATH))["cases"]


def hsps_of(call):
    return _rows(call["hsps"], SEG)


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_calls_runs_and_files_equal_the_reference_hosts(oracle, idx):
    O, c = oracle, CASES[idx]
    k = O.generate_shape_pos(c["shape"])
    seq = c["seq"].encode("ascii")
    L, bs, bl, span = len(seq), c["block_start"], c["block_len"], len(c["shape"])
    rc = O.rev_comp_ascii(seq, 0, L)                                     # repeat_masker_src/main.cpp:311: the whole sequence
    rc_block_start = L - 1 - bs - (bl - 1)                               # seeder.cpp:49
    names, starts, _ = c["chr"]
    for ti, t in enumerate(c["tasks"]):
        s, e, ws, we = t["interval"]
        want = []
        for (rev, s0, s1) in rm_chunk_calls(s, e, bl, c["chunk"], c["strand"]):
            seeds = O.make_seeds(rc, rc_block_start, s0, s1, span, k, bool(c["transition"])) if rev else O.make_seeds(seq, bs, s0, s1, span, k, bool(c["transition"]))
            if seeds.size:                                               # seeder.cpp:101 / :139
                want.append((int(rev), seeds))
        assert len(want) == len(t["calls"]), (ti, len(want), len(t["calls"]))
        allh = []
        for (rev, seeds), g in zip(want, t["calls"]):
            assert (g["rev"], g["ref_start"], g["ref_end"], g["n"]) == (rev, ws, we, seeds.size)
            assert np.array_equal(_rows(g["seeds"], np.dtype("<u8")), seeds)
            allh.append(hsps_of(g))
        allh = np.concatenate(allh) if allh else np.zeros(0, dtype=SEG)
        runs = O.rm_coverage_intervals(allh, bl, c["M"])
        assert [[int(r["query_start"]), int(r["len"])] for r in runs] == t["runs"], ti
        text = "".join(rm_interval_lines(names, starts, bs, t["runs"], bool(c["markend"]))) if t["runs"] else None
        assert text == t["file"], ti


def test_the_golden_set_reaches_the_uint8_wrap_and_an_open_run():
    wrapped = 0
    for c in CASES:
        for t in c["tasks"]:
            hs = [hsps_of(g) for g in t["calls"]]
            if not hs:
                continue
            allh = np.concatenate(hs)
            depth = np.zeros(c["block_len"] + 512, dtype=np.int64)
            for h in allh:
                depth[int(h["query_start"]):int(h["query_start"]) + int(h["len"])] += 1
            wrapped += int(np.count_nonzero(depth >= 256))
    assert wrapped > 0
    assert any(c["block_start"] > 0 for c in CASES) and {c["strand"] for c in CASES} == {1, 2, 3} and any(c["markend"] for c in CASES)
