"""Reader of tests/golden/path_golden.json (generator: tests/golden/make_path_golden.py): what every g_SeedAndFilter call returns when the
reference's own files -- src/seed_filter.cu, common/seed_filter_interface.cu, common/seed_pos_table.cu, common/ntcoding.cpp, src/seeder.cpp --
run end to end on small block pairs (CUDA runtime / thrust / TBB stood in for, kernels under SIMT emulation)."""
import json
import os

import numpy as np

from rm_golden import SEG, _rows

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_golden.json")


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["target_arena"] = np.frombuffer(c["target_arena"].encode("ascii"), dtype=np.uint8)
        c["query_arena"] = np.frombuffer(c["query_arena"].encode("ascii"), dtype=np.uint8)
        c["sub_mat"] = np.array(c["sub_mat"], dtype=np.int32)
        c["calls"] = [dict(k, hsps=_rows(k["hsps"], SEG)) for k in c["calls"]]
        yield c


def case_id(c):
    return "span%d-%s-strand%d-chunk%d-step%d-maxhits%d" % (len(c["shape"]), "tr" if c["transition"] else "notr", c["strand"], c["chunk"], c["step"], c["max_hits"])


def chunk_calls(c, shard):
    """the (interval, rev, start, end) of every chunk the seeder walks, in its order (src/seeder.cpp:47-121) -- chunks without a seed word make no call"""
    q_len = c["q_len"] - len(c["shape"])
    for kk, (s, e) in enumerate(c["intervals"]):
        for rev in (False, True):
            if c["strand"] & (2 if rev else 1):
                for (a, b) in shard.chunks_of((s, e), c["chunk"], q_len, rev):
                    yield kk, rev, a, b
