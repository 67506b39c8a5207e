"""Reader of tests/golden/src_golden.json (generator: tests/golden/make_src_golden.py): the lists the src/ binary's device code -- the
reference's own text under SIMT emulation -- leaves behind on small block pairs."""
import json
import os

import numpy as np

from rm_golden import EXT, SEG, SHAPE, _rows  # noqa: F401

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "src_golden.json")


def cases():
    for c in json.load(open(PATH))["cases"]:
        c = dict(c)
        c["target"] = np.frombuffer(c["target"].encode("ascii"), dtype=np.uint8)
        c["query"] = np.frombuffer(c["query"].encode("ascii"), dtype=np.uint8)
        c["sub_mat"] = np.array(c["sub_mat"], dtype=np.int32)
        for k in ("hits", "reduced", "final"):
            c[k] = _rows(c[k], SEG)
        c["ext"] = _rows(c["ext"], EXT)
        for k in ("t_codes", "q_codes", "q_rc_codes"):
            c[k] = _rows(c[k], np.uint8)
        yield c


def case_id(c):
    return "seed%d-rev%d-thr%d-%s" % (c["seed"], c["rev"], c["hspthresh"], "tr" if c["transition"] else "notr")
