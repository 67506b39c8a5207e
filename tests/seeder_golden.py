"""Reader of tests/golden/seeder_golden.json (generator: tests/golden/make_seeder_golden.py): every g_SeedAndFilter call the reference's own
seeder_body::operator() (src/seeder.cpp compiled as it lies + the real ntcoding.cpp) makes for small query blocks."""
import json
import os

import numpy as np

from rm_golden import _rows

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeder_golden.json")


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["arena"] = np.frombuffer(c["arena"].encode("ascii"), dtype=np.uint8)
        c["calls"] = [dict(k, seeds=_rows(k["seeds"], np.dtype("<u8"))) for k in c["calls"]]
        yield c


def case_id(c):
    return "span%d-%s-strand%d-chunk%d-from%d" % (len(c["shape"]), "tr" if c["transition"] else "notr", c["strand"], c["chunk"], c["q_block_start"])
