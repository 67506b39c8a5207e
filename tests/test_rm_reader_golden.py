"""CPU: the repeat masker's block protocol (the source node of repeat_masker_src/main.cpp:469-479, :484-552 verbatim, driven serially through
repeat_masker_src/seeder.cpp compiled as it lies: tests/golden/make_rm_reader_golden.py) against the repository's restatement: the interval tasks in
segalign_amd/shard.py::rm_plan's order with the fields the printer names files by (index = the block's number from 0, num_invoked from 1), and per
block g_ClearRef + g_ClearQuery (from the second block on), g_SendRefWriteRequest, g_SendQueryWriteRequest, GenerateSeedPosTable in that order in
front of its first task -- the order segalign_rm_host.cpp drives the engine in.  A second route for 8f-4's host loop."""
import json
import os

import pytest

from segalign_amd import shard

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_reader_golden.json")
CASES = json.load(open(PATH))["cases"]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_tasks_and_block_protocol(idx):
    c = CASES[idx]
    tasks = shard.rm_plan(c["seq_len"], c["seq_block_size"], c["interval"], c["neighbor_proportion"], 19)
    per_block = {}
    for t in tasks:
        per_block.setdefault(t["block_index"], []).append(t)
    want = []
    for b in sorted(per_block):
        ts = per_block[b]
        if b > 0:
            want += [["ClearRef"], ["ClearQuery"]]
        want += [["SendRef", ts[0]["block_start"], ts[0]["block_len"]], ["SendQuery"], ["Table", ts[0]["block_start"], ts[0]["block_len"], 1, 19, 12]]
        for i, t in enumerate(ts):
            want.append(["Payload", b, t["block_start"], t["block_len"], t["start"], t["end"], t["ref_start"], t["ref_end"], i + 1, len(ts)])
    got = [e for e in c["events"] if e[0] != "SeedAndFilter"]
    assert got == want
    ev = c["events"]
    for i, e in enumerate(ev):          # every call of a task carries the task's window (seeder.cpp:102,:140)
        if e[0] == "SeedAndFilter":
            p = max(k for k in range(i) if ev[k][0] == "Payload")
            assert e[1:3] == ev[p][6:8]


def test_the_golden_set_switches_blocks():
    assert any(sum(e[0] == "ClearRef" for e in c["events"]) >= 2 for c in CASES)
