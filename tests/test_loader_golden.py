"""CPU: tests/host_model.py::Arena -- the model of how src/main.cpp lays FASTA records out in its DRAM arenas, against which the owned C++
host's files are compared byte for byte (tests/test_gpu_host.py) -- against the reference's own loader text executed in a harness
(src/main.cpp:312-462 query, :479-541 target; tests/golden/make_loader_golden.py): '&'-joined records, a block closed after the record
that crosses the block size, no separator behind a block's last record, the minus-strand chromosome table, the per-block interval lists,
the reverse-complement arena, the .name files.  A second route (8f-2, 8f-3)."""
import base64
import json
import os
import zlib

import pytest

from host_model import Arena

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_golden.json")
CASES = json.load(open(PATH))["cases"]


def raw(s):
    return zlib.decompress(base64.b64decode(s))


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_arena_model_equals_the_reference_loaders(idx):
    c = CASES[idx]
    q = Arena([(n, s.encode("ascii")) for n, s in c["query"]], c["seq_block_size"], c["seed_size"], c["lastz_interval_size"], True)
    t = Arena([(n, s.encode("ascii")) for n, s in c["target"]], c["seq_block_size"], c["seed_size"], c["lastz_interval_size"], False)
    assert [[n, s, l] for n, s, l in zip(q.chr_name, q.chr_start, q.chr_len)] == c["q_chr"]
    assert [[n, s, l] for n, s, l in zip(q.rc_name, q.rc_start, q.rc_len)] == c["rc_q_chr"]
    assert [[n, s, l] for n, s, l in zip(t.chr_name, t.chr_start, t.chr_len)] == c["r_chr"]
    assert [[s, l, len(iv)] for s, l, iv in zip(q.block_start, q.block_len, q.intervals)] == c["q_blocks"]
    assert [[s, l] for s, l in zip(t.block_start, t.block_len)] == c["r_blocks"]
    assert [list(iv) for blk in q.intervals for iv in blk] == c["intervals"]
    qa, qrc, ta = raw(c["q_arena"]), raw(c["q_rc_arena"]), raw(c["r_arena"])
    assert bytes(q.buf)[:len(qa)] == qa and bytes(t.buf)[:len(ta)] == ta
    for s, l in zip(q.block_start, q.block_len):               # the minus-strand arena, block by block (bytes between blocks are never read)
        assert bytes(q.rc[s:s + l]) == qrc[s:s + l]
    for k, names in enumerate(q.block_names):
        assert c["name_files"]["query_block%d.name" % k] == "".join(n + "\n" for n in names)
    for k, names in enumerate(t.block_names):
        assert c["name_files"]["ref_block%d.name" % k] == "".join(n + "\n" for n in names)
    want = {"query_block%d.name" % k for k in range(len(q.block_names))} | {"ref_block%d.name" % k for k in range(len(t.block_names))}
    for a, tag in ((q, "query"), (t, "ref")):   # a last record that closes its block leaves the next block's .name file behind, empty
        if a.stray_name_file is not None:
            want.add("%s_block%d.name" % (tag, a.stray_name_file))
            assert c["name_files"]["%s_block%d.name" % (tag, a.stray_name_file)] == ""
    assert set(c["name_files"]) == want


def test_the_golden_set_splits_blocks():
    assert any("" in c["name_files"].values() for c in CASES)   # the stray empty .name file
    assert any(len(c["q_blocks"]) > 2 for c in CASES) and any(len(c["r_blocks"]) > 1 for c in CASES) and any(len(c["q_blocks"]) == 1 for c in CASES)
