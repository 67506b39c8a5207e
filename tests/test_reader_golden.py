"""CPU: the block / buffer protocol of the reference's source node (src/main.cpp:575-598, :603-735 verbatim, driven serially through src/seeder.cpp
compiled as it lies: tests/golden/make_reader_golden.py) against what the repository restates of it:
  * the payloads -- which (target block, query block, interval) the seeder is handed, in which order, with which fields (r_index from 1, q_index from 0,
    q_len = block length - seed size, num_invoked from 1): tests/host_model.py's loops (Arena + expected_outputs' nesting), i.e. what names the files;
  * the protocol a drop-in engine is driven with: a target block is sent and indexed before its first payload, g_ClearRef in front of every later one;
    the first BUFFER_DEPTH query blocks go up right behind the table (after g_ClearQuery of their buffers from the second target block on); a later
    block takes the buffer whose block is finished, behind g_ClearQuery; every g_SeedAndFilter call names the buffer that holds its block;
  * one call per chunk and strand, plus strand first (src/seeder.cpp:47-121).
A second route for 8f-3."""
import json
import os

import pytest

from host_model import Arena

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reader_golden.json")
CASES = json.load(open(PATH))["cases"]


def arenas(c):
    span = len(c["shape"])
    R = Arena([(n, s.encode()) for n, s in c["target_records"]], c["seq_block_size"], span, c["interval"], False)
    Q = Arena([(n, s.encode()) for n, s in c["query_records"]], c["seq_block_size"], span, c["interval"], True)
    return R, Q


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_payloads_are_the_host_models_loops(idx):
    c = CASES[idx]
    R, Q = arenas(c)
    span = len(c["shape"])
    want = []
    for rb, (rs, rl) in enumerate(zip(R.block_start, R.block_len)):
        for qb, (qs, ql) in enumerate(zip(Q.block_start, Q.block_len)):
            for i, (a, b) in enumerate(Q.intervals[qb]):
                want.append([rb + 1, qb, rs, qs, rl, ql - span, a, b, i + 1, len(Q.intervals[qb])])
    got = [e[1:11] for e in c["events"] if e[0] == "Payload"]
    assert got == want


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_the_protocol_an_engine_is_driven_with(idx):
    c = CASES[idx]
    span, chunk = len(c["shape"]), c["chunk"]
    ref, table, buffers, cleared = None, None, {}, set()
    ref_blocks = 0
    ev = c["events"]
    for i, e in enumerate(ev):
        if e[0] == "SendRef":
            assert ref is None and (ref_blocks == 0 or ev[i - 1] == ["ClearRef"])
            ref, table = (e[1], e[2]), None
            ref_blocks += 1
            assert ev[i + 1][:3] == ["Table", e[1], e[2]] and ev[i + 1][3:] == [c["step"], span, 12]
        elif e[0] == "ClearRef":
            assert ref is not None
            ref = None
        elif e[0] == "Table":
            table = (e[1], e[2])
        elif e[0] == "ClearQuery":
            assert e[1] in buffers                      # only buffers that hold a block are cleared
            del buffers[e[1]]
        elif e[0] == "SendQuery":
            assert e[3] not in buffers and e[3] in (0, 1) and table == ref   # into an empty buffer, the target indexed
            buffers[e[3]] = (e[1], e[2])
        elif e[0] == "Payload":
            r_start, q_start, r_len, q_len, a, b, buf = e[3], e[4], e[5], e[6], e[7], e[8], e[11]
            assert ref == (r_start, r_len) == table and buffers[buf] == (q_start, q_len + span) and e[1] == ref_blocks
            n_chunks = -(-(b - a) // chunk)
            assert ev[i + 1] == ["SeedAndFilter", 0, buf, n_chunks] and ev[i + 2] == ["SeedAndFilter", 1, buf, n_chunks]


def test_the_golden_set_reuses_buffers_and_switches_target_blocks():
    assert any(sum(e[0] == "SendRef" for e in c["events"]) >= 2 and sum(e[0] == "SendQuery" for e in c["events"]) >= 5 for c in CASES)
    assert any(sum(e[0] == "ClearRef" for e in c["events"]) == 2 for c in CASES)
