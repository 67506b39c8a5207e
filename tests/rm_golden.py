"""Reader of tests/golden/rm_golden.json (generator: tests/golden/make_rm_golden.py): the lists the repeat-masker fork's device code
-- the reference's own text under SIMT emulation -- leaves behind on small self-alignment problems."""
import base64
import json
import os
import zlib

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_golden.json")
SEG = np.dtype([("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4")])
EXT = np.dtype([("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4"), ("done", "<u4")])
SHAPE = "TTT0T00TT00T0T0TTTT"


def _rows(s, dt):
    return np.frombuffer(zlib.decompress(base64.b64decode(s)), dtype=dt)


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["target"] = np.frombuffer(c["target"].encode("ascii"), dtype=np.uint8)
        c["sub_mat"] = np.array(c["sub_mat"], dtype=np.int32)
        c["hits"], c["reduced"], c["final"] = _rows(c["hits"], SEG), _rows(c["reduced"], SEG), _rows(c["final"], SEG)
        c["ext"] = _rows(c["ext"], EXT)
        c["rc_codes"] = _rows(c["rc_codes"], np.uint8)
        yield c


def case_id(c):
    return "seed%d-rev%d-win%d_%d-thr%d" % (c["seed"], c["rev"], c["win_start"], c["win_end"], c["hspthresh"])
