"""GPU: a C++ host that uses ONLY the reference's own symbols (g_* function pointers + GenerateSeedPosTable from
include/segalign_amd_compat.hpp), with several host threads like the TBB seeder bodies, gives HSP lists identical
to the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import Case, SHAPE_12OF19
from segalign_amd import synth
from test_compat_header import build_driver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads", [1, 4])
def test_reference_symbols_end_to_end(oracle, tmp_path, threads):
    t, q = synth.make_pair(240000, 31, 32, sub_rate=0.10, mask_frac=0.1, records=2, indel_every=500)
    chunk = 60000
    c = Case(t, q, chunk=chunk).oracle_setup(oracle)
    tp, qp = tmp_path / "t.txt", tmp_path / "q.txt"
    tp.write_bytes(t.tobytes())
    qp.write_bytes(q.tobytes())
    exe = build_driver()
    out = subprocess.check_output([exe, str(tp), str(qp), SHAPE_12OF19, str(chunk), "1", str(threads)],
                                  stderr=subprocess.DEVNULL).decode().split("\n")
    i = 0
    total = 0
    while i < len(out) and out[i]:
        tag, rev, s, e, n, hits = out[i].split()
        assert tag == "C"
        rev, s, e, n, hits = int(rev), int(s), int(e), int(n), int(hits)
        got = np.array([tuple(int(x) for x in out[i + 1 + j].split()) for j in range(n)],
                       dtype=[("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4")])
        seeds = c.host_seeds(s, e, bool(rev))
        want, st = c.oracle_saf(seeds, bool(rev))
        assert hits == st["num_hits"]
        assert got.size == want.size - 1 and np.all(got == want[1:]), (rev, s, e)
        total += n
        i += 1 + n
    assert total > 0
