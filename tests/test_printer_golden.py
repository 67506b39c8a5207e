"""CPU: the segment writer as tests/host_model.py restates it (segment_files -- what the byte-exact tests of the owned C++ host,
tests/test_gpu_host.py, compare segalign_host's files with) against what the reference's own segment_printer_body::operator() writes
(src/segment_printer.cpp compiled as it lies, TBB's header stood in for: tests/golden/make_printer_golden.py): file names, 1-based
chromosome-relative coordinates, the minus strand in reverse order against the rc chromosome table, the lastz command lines with and
without --ambiguous / --notrivial / --scoring, no command without --gapped, several target and query blocks.  A second route (8f-2)."""
import json
import os

import pytest

from host_model import segment_files

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "printer_golden.json")
CASES = json.load(open(PATH))["cases"]


class Tables:
    pass


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_host_model_writes_what_the_reference_printer_writes(idx):
    c = CASES[idx]
    R, Q = Tables(), Tables()
    R.chr_name, R.chr_start, R.chr_len = c["r_chr"]
    Q.chr_name, Q.chr_start, Q.chr_len = c["q_chr"]
    Q.rc_name, Q.rc_start, Q.rc_len = c["rc_q_chr"]
    files, cmds = {}, []
    for it in c["intervals"]:
        f, m = segment_files(R, Q, it["r_index"] - 1, it["r_start"], it["q_index"], it["q_start"], it["num_invoked"], [tuple(h) for h in it["fw"]],
                             [tuple(h) for h in it["rc"]], gapped=bool(c["gapped"]), data_folder=c["data_folder"], output_format=c["output_format"],
                             ydrop=c["ydrop"], gappedthresh=c["gappedthresh"], ambiguous=c["ambiguous"], notrivial=bool(c["notrivial"]),
                             scoring_file=c["scoring_file"])
        assert not (set(f) & set(files))
        files.update(f)
        cmds.extend(m)
    assert sorted(files) == sorted(c["files"])
    for name in files:
        assert files[name] == c["files"][name], name
    assert cmds == c["cmds"]


def test_the_golden_set_has_both_strands_and_several_blocks():
    assert any(n.endswith(".minus.segments") for c in CASES for n in c["files"]) and any(n.endswith(".plus.segments") for c in CASES for n in c["files"])
    assert any(len({it["r_start"] for it in c["intervals"]}) > 1 and len({it["q_start"] for it in c["intervals"]}) > 1 for c in CASES)
    assert any(not c["gapped"] and not c["cmds"] for c in CASES) and any(c["ambiguous"] and c["notrivial"] and c["scoring_file"] for c in CASES)
