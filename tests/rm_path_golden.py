"""Reader of tests/golden/rm_path_golden.json (generator: tests/golden/make_rm_path_golden.py): what the repeat masker binary's own files --
repeat_masker_src/seed_filter.cu, seeder.cpp, segment_printer.cpp + the common files -- return and write when they run end to end on small
self-alignment problems (CUDA runtime / thrust / TBB stood in for, kernels under SIMT emulation)."""
import json
import os

import numpy as np

from rm_golden import SEG, _rows

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_path_golden.json")


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["sub_mat"] = np.array(c["sub_mat"], dtype=np.int32)
        c["tasks"] = [dict(t, calls=[dict(k, hsps=_rows(k["hsps"], SEG)) for k in t["calls"]]) for t in c["tasks"]]
        yield c


def case_id(c):
    return "%s-strand%d-chunk%d-M%d-block%d+%d-maxhits%d" % ("tr" if c["transition"] else "notr", c["strand"], c["chunk"], c["M"], c["block_start"], c["block_len"], c["max_hits"])


def header(segs):
    """(seed hits, HSPs) of the fork's 64-bit header element (repeat_masker_src/seed_filter.cu:857-861)"""
    return (int(segs[0]["ref_start"]) | (int(segs[0]["query_start"]) << 32), int(segs[0]["len"]) | ((int(segs[0]["score"]) & 0xFFFFFFFF) << 32))
