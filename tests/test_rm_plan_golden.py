"""CPU: the repeat masker's block / interval plan as restated in oracle/segalign_oracle.c (orc_rm_plan) and segalign_amd/shard.py (rm_plan,
what bench.py and the C++ repeat-masker host walk) against the reference's own text of repeat_masker_src/main.cpp:259-262 + :323-433
executed inside a harness function (tests/golden/make_rm_plan_golden.py): float / ceil arithmetic of the neighbour intervals, unsigned
wrap with neighbor_proportion 0, overlaps, several blocks, genomes above 2^32 / 2 bases, a sequence shorter than one interval."""
import json
import os

import pytest

from segalign_amd import shard

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_plan_golden.json")
CASES = json.load(open(PATH))["cases"]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_plan_restatements_equal_the_reference_text(oracle, idx):
    c = CASES[idx]
    want = [tuple(r) for r in c["tasks"]]
    o = oracle.rm_plan(c["seq_len"], c["seq_block_size"], c["lastz_interval_size"], c["prop_neigh_interval"], c["seed_size"])
    got_o = [(int(t["block_index"]), int(t["block_start"]), int(t["block_len"]), int(t["start"]), int(t["end"]), int(t["ref_start"]), int(t["ref_end"])) for t in o]
    assert got_o == want
    s = shard.rm_plan(c["seq_len"], c["seq_block_size"], c["lastz_interval_size"], c["prop_neigh_interval"], c["seed_size"])
    got_s = [(t["block_index"], t["block_start"], t["block_len"], t["start"], t["end"], t["ref_start"], t["ref_end"]) for t in s]
    assert got_s == want


def test_the_cpp_hosts_plan_equals_the_reference_text():
    """segalign_rm_host --plan-only walks make_plan() without touching the engine: the owned C++ host's restatement (uint32 arithmetic like
    the reference's) against the same vectors"""
    import subprocess
    from segalign_amd.build import build_host, RM_HOST_BIN
    build_host()
    for c in CASES:
        seed = {19: "12of19", 22: "14of22"}[c["seed_size"]]
        out = subprocess.check_output([RM_HOST_BIN, "--plan-only=%d" % c["seq_len"], "--seq_block_size=%d" % c["seq_block_size"],
                                       "--lastz_interval_size=%d" % c["lastz_interval_size"], "--neighbor_proportion=%r" % c["prop_neigh_interval"],
                                       "--seed=%s" % seed], stderr=subprocess.DEVNULL).decode()
        got = [tuple(int(x) for x in l.split()) for l in out.split("\n") if l]
        assert got == [tuple(r) for r in c["tasks"]], (c["seq_len"], c["prop_neigh_interval"])
