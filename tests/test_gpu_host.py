"""GPU: the C++ host harness (segalign_amd/host/segalign_host.cpp: FASTA -> arenas -> blocks/intervals -> engine ->
.segments files + lastz command lines) against the test-side restatement of the reference host
driven by the oracle (tests/host_model.py).  Files must match byte for byte; command lines as a set (worker threads
print them in completion order, like the reference's TBB printer)."""
import os
import subprocess

import pytest

from host_model import expected_outputs, write_fasta
from segalign_amd import synth
from segalign_amd.build import build_host

pytestmark = pytest.mark.gpu


def make_records(seed, lens, sub=0.0, invert=False):
    recs = []
    for i, n in enumerate(lens):
        s = synth.random_dna(n, seed + i)
        recs.append(("chr%s%d" % ("T" if not sub else "Q", i + 1), s))
    return recs


@pytest.mark.parametrize("host_seeding,threads,chunk,block", [(False, 4, 20000, 60000), (True, 2, 20000, 60000),
                                                              (False, 3, 2500, 10**6)])  # 18 chunks per interval: calls of 16 + 2
def test_host_harness_writes_the_reference_files(oracle, tmp_path, host_seeding, threads, chunk, block):
    t_recs = make_records(100, [40000, 15000, 60000, 9000, 30000])
    q_recs = []
    for i, (name, s) in enumerate(t_recs[::-1]):
        m = synth.mutate(s, 200 + i, 0.09, indel_every=700)
        m = synth.soft_mask(m, 300 + i, 0.08, 100, 600)
        if i % 2:
            m = synth.invert_blocks(m, 400 + i, block=5000, frac=0.5)
        q_recs.append(("q%d" % (i + 1), m))
    t_recs = [(n, synth.soft_mask(s, 500 + i, 0.05, 100, 400)) for i, (n, s) in enumerate(t_recs)]
    tf, qf = tmp_path / "target.fa", tmp_path / "query.fa"
    write_fasta(tf, t_recs)
    write_fasta(qf, q_recs, width=70)
    params = dict(chunk=chunk, interval=45000, seq_block_size=block)  # several target blocks, query blocks, intervals
    files, cmds = expected_outputs(oracle, [(n, s.tobytes()) for n, s in t_recs], [(n, s.tobytes()) for n, s in q_recs], **params)
    outdir = tmp_path / "out"
    outdir.mkdir()
    exe = build_host()
    cmd = [exe, str(tf), str(qf), "./", "--wga_chunk=%d" % params["chunk"], "--lastz_interval=%d" % params["interval"],
           "--seq_block_size=%d" % params["seq_block_size"], "--outdir=%s" % outdir, "--num_threads=%d" % threads, "--num_gpu=1"]
    if host_seeding:
        cmd.append("--host-seeding")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    got = {f: open(os.path.join(outdir, f)).read() for f in os.listdir(outdir)}
    assert sorted(got) == sorted(files), (sorted(set(got) ^ set(files)))
    for f in files:
        assert got[f] == files[f], f
    assert sorted(res.stdout.decode().strip().split("\n")) == sorted(cmds)
    assert sum(1 for f in files if f.endswith(".segments")) >= (6 if block < 10**6 else 2) and any(".minus." in f for f in files)
