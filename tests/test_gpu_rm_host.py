"""GPU: the repeat-masker host harness (segalign_amd/host/segalign_rm_host.cpp) against the test-side
restatement of repeat_masker_src/{main,seeder,segment_printer}.cpp driven by the oracle (tests/host_model.py).
The .intervals files must match byte for byte in both modes: everything on the device (sa_rm_mask_interval) and the
reference's structure (--host-loop: host seeding, HSPs back to the host, sa_rm_coverage_intervals)."""
import os
import subprocess

import numpy as np
import pytest

from host_model import rm_expected_outputs, write_fasta
from segalign_amd import synth
from segalign_amd.build import build_host, RM_HOST_BIN

pytestmark = pytest.mark.gpu


def repeat_rich_records():
    unit = synth.random_dna(500, 7)
    recs = []
    rng = np.random.default_rng(11)
    for r, n in enumerate((90000, 40000, 70000)):
        s = synth.random_dna(n, 20 + r)
        for i in range(n // 2500):
            p = int(rng.integers(0, n - 600))
            cp = synth.mutate(unit, 100 * r + i, 0.05)
            s[p:p + cp.size] = cp if i % 3 else synth.reverse_complement(cp)
        recs.append(("chr%d" % (r + 1), synth.soft_mask(s, 40 + r, 0.04, 100, 400)))
    return recs


@pytest.mark.parametrize("mode,extra", [("device", []), ("host-loop", ["--host-loop"]),
                                         ("device-M2", ["--M=2", "--markend", "--strand=plus"])])
def test_rm_host_writes_the_reference_files(oracle, tmp_path, mode, extra):
    recs = repeat_rich_records()
    fa = tmp_path / "seq.fa"
    write_fasta(fa, recs)
    params = dict(chunk=20000, interval=30000, seq_block_size=120000, prop=0.3)
    kw = dict(params)
    if "--M=2" in extra:
        kw.update(M=2, markend=True, strand="plus")
    files = rm_expected_outputs(oracle, [(n, s.tobytes()) for n, s in recs], **kw)
    outdir = tmp_path / "out"
    outdir.mkdir()
    build_host()
    cmd = [RM_HOST_BIN, str(fa), "--wga_chunk_size=%d" % params["chunk"], "--lastz_interval_size=%d" % params["interval"],
           "--seq_block_size=%d" % params["seq_block_size"], "--neighbor_proportion=%g" % params["prop"],
           "--outdir=%s" % outdir, "--num_threads=3", "--num_gpu=1"] + extra
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    got = {f: open(os.path.join(outdir, f)).read() for f in os.listdir(outdir)}
    assert sorted(got) == sorted(files), sorted(set(got) ^ set(files))
    for f in files:
        assert got[f] == files[f], f
    assert len(files) >= 4 and len({f.split(".")[1] for f in files}) >= 2  # several intervals, more than one block
