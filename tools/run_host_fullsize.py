#!/usr/bin/env python3
"""End-to-end runs of the owned C++ host (segalign_amd/host/segalign_host.cpp: FASTA -> arenas -> blocks / intervals -> engine ->
.segments files + lastz command lines, the reference's src/main.cpp:601-737 + src/segment_printer.cpp:70-168) at BASELINE configs[1]
size, on the GPU box:

  python tools/run_host_fullsize.py [out.txt] [target_mbp]

  A  100 Mbp target x 100 Mbp query (the stand-in bench.py times), written as FASTA: process wall time, the host's own clock
     (FASTA open -> sequences loaded -> last .segments file closed), Gbp/s.  ONE query block: nothing hides its upload or the table build.
  B  the same target x FOUR such query blocks (the query records four times under other names, --seq_block_size just below one copy):
     the steady state bench.py's `value` describes -- block k + 1 uploaded into the other device buffer while block k's calls run.
  C  one interval of run A (both strands: tmp<k>.block0.r0.{plus,minus}.segments) byte for byte against tests/host_model.py driven by
     the oracle on its own table (80 chunk calls on the CPU).
"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segalign_amd import synth  # noqa: E402
from segalign_amd.build import build_host  # noqa: E402


def records_of(joined):
    return bytes(joined).split(b"&")


def write_fasta(path, names, recs):
    with open(path, "wb") as f:
        for n, r in zip(names, recs):
            f.write(b">" + n + b"\n")
            for j in range(0, len(r), 1 << 20):
                f.write(r[j:j + (1 << 20)] + b"\n")


def run_host(exe, tfa, qfa, out, threads, extra=()):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, SEGALIGN_AMD_SLOTS=str(threads))
    t0 = time.time()
    res = subprocess.run([exe, tfa, qfa, "./", "--outdir=" + out, "--debug", "--num_threads=%d" % threads, "--num_gpu=1"] + list(extra),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t0
    err = res.stderr.decode()
    keep = [l for l in err.split("\n") if l.startswith("Time elapsed") or l.startswith("#")]
    load = float(re.search(r"loading sequences\): ([0-9.]+) sec", err).group(1))
    pipe = float(re.search(r"complete pipeline\): ([0-9.]+) sec", err).group(1))
    nseg = sum(1 for f in os.listdir(out) if f.endswith(".segments"))
    nlines = sum(sum(1 for _ in open(os.path.join(out, f))) for f in os.listdir(out) if f.endswith(".segments"))
    ncmd = len([l for l in res.stdout.decode().strip().split("\n") if l])
    return dict(rc=res.returncode, wall=wall, load=load, pipeline=pipe, lines=keep, segment_files=nseg, hsp_lines=nlines, lastz_cmds=ncmd)


def main():
    out_txt = sys.argv[1] if len(sys.argv) > 1 else None
    mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    threads = 6
    log = []

    def say(s=""):
        print(s)
        sys.stdout.flush()
        log.append(s)

    t, q = synth.make_pair(int(mbp * 1e6), 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, invert_frac=0.3, invert_block=100_000)
    t_recs, q_recs = records_of(t), records_of(q)
    t_names = [b"chrT%d" % (i + 1) for i in range(len(t_recs))]
    q_names = [b"chrQ%d" % (i + 1) for i in range(len(q_recs))]
    d = tempfile.mkdtemp(prefix="sa_host_")
    tfa, qfa, q4fa = os.path.join(d, "t.fa"), os.path.join(d, "q.fa"), os.path.join(d, "q4.fa")
    write_fasta(tfa, t_names, t_recs)
    write_fasta(qfa, q_names, q_recs)
    write_fasta(q4fa, [b"c%d_" % k + n for k in range(4) for n in q_names], q_recs * 4)
    exe = build_host()
    qbases = sum(len(r) for r in q_recs)
    say("owned C++ host end to end (tools/run_host_fullsize.py), %d host threads = engine slots, 1 GPU; stand-in of bench.py: %.0f Mbp target x %.0f Mbp query, 7 records each" % (threads, mbp, mbp))
    say("=" * 150)

    a = run_host(exe, tfa, qfa, os.path.join(d, "outA"), threads)
    say("A. one query block (configs[1] as the reference would run it: one target block, one query block, 10 intervals, both strands)")
    for l in a["lines"]:
        say("     " + l)
    say("   exit %d; process wall %.2f s (engine start + FASTA + pipeline + teardown); FASTA open -> last .segments closed: %.3f s = load %.3f + pipeline %.3f"
        % (a["rc"], a["wall"], a["load"] + a["pipeline"], a["load"], a["pipeline"]))
    say("   pipeline alone (target upload + table build + query upload + calls + segment files): %.4f Gbp of query per second; with the FASTA load: %.4f"
        % (qbases / a["pipeline"] / 1e9, qbases / (a["load"] + a["pipeline"]) / 1e9))
    say("   %d segment files, %d HSP lines, %d lastz command lines" % (a["segment_files"], a["hsp_lines"], a["lastz_cmds"]))

    b = run_host(exe, tfa, q4fa, os.path.join(d, "outB"), threads, ["--seq_block_size=%d" % (qbases - 1000)])
    say("B. four query blocks of the same size against the resident target (block k + 1 uploaded into the other device buffer while block k runs)")
    for l in b["lines"]:
        say("     " + l)
    say("   exit %d; pipeline %.3f s for %d query bases = %.4f Gbp/s; per added block (B - A) / 3: %.1f ms = %.4f Gbp/s  <- the steady state bench.py's `value` describes"
        % (b["rc"], b["pipeline"], 4 * qbases, 4 * qbases / b["pipeline"] / 1e9, 1e3 * (b["pipeline"] - a["pipeline"]) / 3.0,
           qbases / max((b["pipeline"] - a["pipeline"]) / 3.0, 1e-9) / 1e9))
    say("   %d segment files, %d HSP lines (4 x A's: %s)" % (b["segment_files"], b["hsp_lines"], b["hsp_lines"] == 4 * a["hsp_lines"]))

    # ---- C: one interval of run A against the host model + oracle ----
    from oracle import oracle as O
    import host_model as HM
    O.build(with_ref=False)
    interval_index = 6   # num_invoked of the printer = 1-based interval number
    seed_size, chunk, interval = 19, 250000, 10_000_000
    t0 = time.time()
    R = HM.Arena(list(zip([n.decode() for n in t_names], t_recs)), 500_000_000, seed_size, interval, False)
    Q = HM.Arena(list(zip([n.decode() for n in q_names], q_recs)), 500_000_000, seed_size, interval, True)
    kmer = O.generate_shape_pos("TTT0T00TT00T0T0TTTT")
    sub_mat = O.build_sub_mat(910)
    tblock, qblock, qrc = bytes(R.buf[:R.block_len[0]]), bytes(Q.buf[:Q.block_len[0]]), bytes(Q.rc[:Q.block_len[0]])
    ref_codes = O.encode(tblock)
    index, pos = O.generate_seed_pos_table(tblock, 0, len(tblock), 1, seed_size, kmer)
    q_codes, qrc_codes = O.encode_rev_comp(qblock)
    q_len = len(qblock) - seed_size
    ia, ib = Q.intervals[0][interval_index - 1]
    hs = {False: [], True: []}
    for rev in (False, True):
        s, e = (q_len - ib, q_len - ia) if rev else (ia, ib)
        for c in range(s, e, chunk):
            seeds = O.make_seeds(qrc if rev else qblock, 0, c, min(c + chunk, e), seed_size, kmer, True)
            if seeds.size == 0:
                continue
            segs, _ = O.seed_and_filter(ref_codes, qrc_codes if rev else q_codes, index, pos, seeds, sub_mat)
            hs[rev].extend(segs[1:].tolist())
    files, cmds = HM.segment_files(R, Q, 0, 0, 0, 0, interval_index, hs[False], hs[True], gappedthresh=3000)
    ok = True
    for f, text in sorted(files.items()):
        p = os.path.join(d, "outA", f)
        got = open(p).read() if os.path.exists(p) else None
        same = got == text
        ok = ok and same
        say("C. %s: %d lines by the host model + oracle (own table), host file %s" % (f, text.count("\n"), "IDENTICAL byte for byte" if same else "DIFFERS"))
    say("   (%.0f s of oracle: its own 100 Mbp table + 80 chunk calls)  ->  %s" % (time.time() - t0, "ok" if ok and files else "MISMATCH"))
    if out_txt:
        os.makedirs(os.path.dirname(os.path.abspath(out_txt)), exist_ok=True)
        with open(out_txt, "w") as f:
            f.write("\n".join(log) + "\n")
    sys.exit(0 if (ok and a["rc"] == 0 and b["rc"] == 0) else 1)


if __name__ == "__main__":
    main()
