"""Reader of tests/golden/table_golden.json (generator: tests/golden/make_table_golden.py): seed position tables built by the reference's
own GenerateSeedPosTable text + the real ntcoding.cpp."""
import json
import os

import numpy as np

from rm_golden import _rows

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "table_golden.json")
U32 = np.dtype("<u4")


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["target"] = np.frombuffer(c["target"].encode("ascii"), dtype=np.uint8)
        for k in ("keys", "ends", "pos"):
            c[k] = _rows(c[k], U32)
        yield c


def case_id(c):
    return "%s-step%d-%dbp" % (c["shape_name"], c["step"], c["target"].size)


def check_table(c, index, pos):
    """index = the table as handed on (inclusive bucket ends per key), pos = positions, ascending inside a bucket"""
    index = np.asarray(index)
    assert index.size == 4 ** c["kmer_size"] and int(index[-1]) == c["num_index"] == np.asarray(pos).size
    assert np.array_equal(index[c["keys"]], c["ends"])
    starts = np.concatenate([[0], index[:-1].astype(np.int64)])
    nz = np.nonzero(index.astype(np.int64) - starts)[0]
    assert np.array_equal(nz.astype(np.uint32), c["keys"])          # no other key has a bucket
    assert np.array_equal(np.asarray(pos), c["pos"])
