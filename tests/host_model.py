"""Test-side restatement of the reference HOST around the engine: arena layout, block/interval planning, the seeder
body and the segment printer -- src/main.cpp:320-549, src/seeder.cpp:12-127, src/segment_printer.cpp:11-173 --
driven by the oracle's SeedAndFilter.  Returns {file name: text} and the lastz command lines, i.e. exactly what the
reference binary would leave on disk / print for the same FASTA input.  Used to check segalign_amd/host/segalign_host."""
import bisect

import numpy as np


def write_fasta(path, records, width=60):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(">%s some description\n" % name)
            s = bytes(seq).decode()
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


class Arena:
    def __init__(self, records, seq_block_size, seed_size, interval, is_query):
        self.buf = bytearray()
        self.chr_name, self.chr_start, self.chr_len = [], [], []
        self.block_start, self.block_len, self.block_names = [0], [], [[]]
        self.rc = bytearray()
        self.rc_name, self.rc_start, self.rc_len = [], [], []
        self.intervals = []
        block_chrs, seq_block_len, seq_block_start = [], 0, 0

        def close(length):
            nonlocal block_chrs
            self.block_len.append(length)
            if is_query:
                for c in reversed(block_chrs):  # main.cpp:369-374
                    self.rc_name.append(self.chr_name[c])
                    self.rc_start.append(2 * seq_block_start + length - self.chr_start[c] - self.chr_len[c])
                    self.rc_len.append(self.chr_len[c])
                comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
                blk = bytes(self.buf[seq_block_start:seq_block_start + length])
                self.rc[seq_block_start:seq_block_start + length] = blk[::-1].translate(comp)
                end_pos = length - seed_size
                self.intervals.append([(s, min(end_pos, s + interval)) for s in range(0, max(end_pos, 0), interval)]
                                      if length > seed_size else [])

        for name, seq in records:
            self.block_names[-1].append(name)
            c = len(self.chr_name)
            self.chr_name.append(name)
            self.chr_start.append(len(self.buf))
            self.chr_len.append(len(seq))
            block_chrs.append(c)
            self.buf += bytes(seq)
            seq_block_len += len(seq)
            if seq_block_len > seq_block_size:  # main.cpp:359
                close(seq_block_len)
                seq_block_start = len(self.buf)
                self.block_start.append(seq_block_start)
                self.block_names.append([])
                seq_block_len, block_chrs = 0, []
            else:
                self.buf += b"&"
                seq_block_len += 1
        self.stray_name_file = None
        if seq_block_len > 0:
            close(seq_block_len - 1)
        else:
            self.block_start.pop()
            self.block_names.pop()
            if self.chr_name:  # the last record closed a block: the loader has already opened the NEXT block's .name file and leaves it
                self.stray_name_file = len(self.block_names)  # empty (main.cpp:404 / :528; tests/golden/loader_golden.json)


def segment_files(R, Q, rb, rs, qb, qs, index, fw_hsps, rc_hsps, gapped=True, data_folder="./", output_format="maf-", ydrop=9430,
                  gappedthresh=3000, ambiguous="", notrivial=False, scoring_file=""):
    """segment_printer_body::operator() (src/segment_printer.cpp:11-175) for one interval: the .segments files of the two strands and the
    lastz command lines.  R / Q: objects with chr_name / chr_start (and rc_name / rc_start for the query's minus strand); rb / qb block
    indices (rb = the printer's r_index - 1), rs / qs block starts in the arenas, index = num_invoked.  HSPs as (ref_start, query_start,
    len, score).  Held against the real printer by tests/test_printer_golden.py."""
    files, cmds = {}, []
    for rev in (False, True):  # segment_printer.cpp:70-168
        hs = rc_hsps if rev else fw_hsps
        if not hs:
            continue
        base = "tmp%d.block%d.r%d.%s" % (index, qb, rs, "minus" if rev else "plus")
        names = Q.rc_name if rev else Q.chr_name
        starts = Q.rc_start if rev else Q.chr_start
        lines = []
        for (r0, q0, ln, sc) in (reversed(hs) if rev else hs):
            seg_r, seg_q = r0 + rs, q0 + qs
            ri = bisect.bisect_right(R.chr_start, seg_r) - 1
            qi = bisect.bisect_right(starts, seg_q) - 1
            lines.append("%s\t%d\t%d\t%s\t%d\t%d\t%s\t%d\n" % (
                R.chr_name[ri], seg_r + 1 - R.chr_start[ri], seg_r + ln + 1 - R.chr_start[ri], names[qi],
                seg_q + 1 - starts[qi], seg_q + ln + 1 - starts[qi], "-" if rev else "+", sc))
        files[base + ".segments"] = "".join(lines)
        if gapped:
            cmd = ("lastz %sref.2bit[nameparse=darkspace][multiple][subset=ref_block%d.name] "
                   "%squery.2bit[nameparse=darkspace][subset=query_block%d.name] --format=%s --ydrop=%d "
                   "--gappedthresh=%d --strand=%s" % (data_folder, rb, data_folder, qb, output_format, ydrop, gappedthresh,
                                                      "minus" if rev else "plus"))
            if ambiguous != "":
                cmd += " --ambiguous=" + ambiguous
            if notrivial:
                cmd += " --notrivial"
            if scoring_file != "":
                cmd += " --scoring=" + scoring_file
            cmds.append(cmd + " --segments=%s.segments --output=%s.%s 2> %s.err" % (base, base, output_format, base))
    return files, cmds


def expected_outputs(O, target_records, query_records, shape="TTT0T00TT00T0T0TTTT", transition=True, step=1, xdrop=910,
                     hspthresh=3000, noentropy=False, chunk=250000, interval=10000000, seq_block_size=500000000,
                     gapped=True, data_folder="./", output_format="maf-", ydrop=9430, strand="both"):
    seed_size = len(shape)
    R = Arena(target_records, seq_block_size, seed_size, interval, False)
    Q = Arena(query_records, seq_block_size, seed_size, interval, True)
    kmer = O.generate_shape_pos(shape)
    sub_mat = O.build_sub_mat(xdrop)
    files, cmds = {}, []
    for k, names in enumerate(R.block_names):
        files["ref_block%d.name" % k] = "".join(n + "\n" for n in names)
    for k, names in enumerate(Q.block_names):
        files["query_block%d.name" % k] = "".join(n + "\n" for n in names)
    if R.stray_name_file is not None:
        files["ref_block%d.name" % R.stray_name_file] = ""
    if Q.stray_name_file is not None:
        files["query_block%d.name" % Q.stray_name_file] = ""
    for rb, (rs, rl) in enumerate(zip(R.block_start, R.block_len)):
        tblock = bytes(R.buf[rs:rs + rl])
        ref_codes = O.encode(tblock)
        index, pos = O.generate_seed_pos_table(tblock, 0, rl, step, seed_size, kmer)
        for qb, (qs, ql) in enumerate(zip(Q.block_start, Q.block_len)):
            qblock = bytes(Q.buf[qs:qs + ql])
            qrc_block = bytes(Q.rc[qs:qs + ql])
            q_codes, qrc_codes = O.encode_rev_comp(qblock)
            q_len = ql - seed_size
            for i, (a, b) in enumerate(Q.intervals[qb]):
                hs = {False: [], True: []}
                for rev in (False, True):
                    if (not rev and strand not in ("plus", "both")) or (rev and strand not in ("minus", "both")):
                        continue
                    s, e = (q_len - b, q_len - a) if rev else (a, b)  # seeder.cpp:33-34
                    for c in range(s, e, chunk):
                        seeds = O.make_seeds(qrc_block if rev else qblock, 0, c, min(c + chunk, e), seed_size, kmer, transition)
                        if seeds.size == 0:
                            continue  # seeder.cpp:76
                        segs, _ = O.seed_and_filter(ref_codes, qrc_codes if rev else q_codes, index, pos, seeds, sub_mat,
                                                    seed_size=seed_size, xdrop=xdrop, hspthresh=hspthresh, noentropy=noentropy)
                        hs[rev].extend(segs[1:].tolist())
                f2, c2 = segment_files(R, Q, rb, rs, qb, qs, i + 1, hs[False], hs[True], gapped=gapped, data_folder=data_folder,
                                       output_format=output_format, ydrop=ydrop, gappedthresh=hspthresh)
                files.update(f2)
                cmds.extend(c2)
    return files, cmds


# ---- repeat masker host (repeat_masker_src/main.cpp:283-436, seeder.cpp:28-195, segment_printer.cpp:8-65) -------------
def rm_chunk_calls(start_pos, end_pos, block_len, chunk, strands):
    """The seed ranges [s0, s1) the repeat masker's seeder hands g_SeedAndFilter for one interval, in call order (repeat_masker_src/
    seeder.cpp:73-146): per wga_chunk piece of [start_pos, end_pos) the plus-strand piece, then a minus-strand piece derived from the
    plus piece's END (:118-119).  Held against the real seeder by tests/test_rm_host_golden.py."""
    end_pos_rc = block_len - 1 - start_pos
    for i in range(start_pos, end_pos, chunk):
        start, end = i, min(i + chunk, end_pos)
        for rev in (False, True):
            if not (strands & (2 if rev else 1)):
                continue
            if rev:
                s0 = block_len - 1 - end
                yield True, s0, min(s0 + chunk, end_pos_rc)
            else:
                yield False, start, end


def rm_interval_lines(chr_name, chr_start, block_start, intervals, markend=False):
    """interval_printer_body::operator() (repeat_masker_src/segment_printer.cpp:8-65): the lines of tmp<i>.block<b>.intervals for the
    (query_start, len) pairs of one interval task"""
    lines = []
    for (qs, ln) in intervals:
        q = block_start + int(qs)
        c = bisect.bisect_right(chr_start, q) - 1
        lines.append("%s\t%d\t%d\n" % (chr_name[c], q - chr_start[c], q + int(ln) + 1 - chr_start[c]))
    if markend:
        lines.append("# segalign_repeat_masker end-of-file\n")
    return lines


def rm_mask_interval(O, blk, ctx, start_pos, end_pos, ref_start, ref_end, strands, M, chunk, seed_size, kmer_size, transition,
                     saf_kwargs):
    """seeder_body::operator() of the repeat masker for one interval of block `blk` (ASCII bytes); ctx caches the
    oracle-side encoded block, its reverse complement and its seed table."""
    L = len(blk)
    if "ref" not in ctx:
        ctx["ref"] = O.encode(blk)
        ctx["rc_ascii"] = O.rev_comp_ascii(blk, 0, L)
        ctx["rc"] = O.rev_comp_codes(ctx["ref"])
        ctx["index"], ctx["pos"] = O.generate_seed_pos_table(blk, 0, L, saf_kwargs.get("step", 1), seed_size, kmer_size)
    kw = {k: v for k, v in saf_kwargs.items() if k != "step"}
    hsps = []
    tot = dict(num_seeds=0, num_hits=0, num_hsps=0)
    for (rev, s0, s1) in rm_chunk_calls(start_pos, end_pos, L, chunk, strands):
        if True:
            s1 = min(s1, L - seed_size + 1)
            seeds = O.make_seeds(ctx["rc_ascii"] if rev else blk, 0, s0, s1, seed_size, kmer_size, transition)
            if seeds.size == 0:
                continue
            segs, st = O.seed_and_filter(ctx["ref"], ctx["rc"] if rev else ctx["ref"], ctx["index"], ctx["pos"], seeds,
                                         seed_size=seed_size, rm=(rev, ref_start, ref_end), **kw)
            tot["num_seeds"] += int(seeds.size)
            tot["num_hits"] += int(st["num_hits"])
            tot["num_hsps"] += int(segs.size - 1)
            hsps.append(segs[1:])
    allh = np.concatenate(hsps) if hsps else np.zeros(0, dtype=O.SEG_DTYPE)
    return O.rm_coverage_intervals(allh, L, M), tot


def rm_expected_outputs(O, records, shape="TTT0T00TT00T0T0TTTT", transition=True, step=1, xdrop=910, hspthresh=3000,
                        noentropy=False, chunk=250000, interval=10000000, seq_block_size=1000000000, prop=0.2, M=1,
                        strand="both", markend=False):
    """{file name: text} the reference repeat masker would write for this FASTA content."""
    kmer_size = O.generate_shape_pos(shape)
    seed_size = len(shape)
    sub_mat = O.build_sub_mat(xdrop)
    buf = bytearray()
    chr_name, chr_start, chr_len = [], [], []
    for name, seq in records:  # main.cpp:283-305
        chr_name.append(name)
        chr_start.append(len(buf))
        chr_len.append(len(seq))
        buf += bytes(seq) + b"&"
    buf = bytes(buf[:-1])
    strands = {"plus": 1, "minus": 2, "both": 3}[strand]
    tasks = O.rm_plan(len(buf), seq_block_size, interval, prop, seed_size)
    files = {}
    ctxs = {}
    counters = {}
    for t in tasks:
        b = int(t["block_index"])
        counters[b] = counters.get(b, 0) + 1
        bs, bl = int(t["block_start"]), int(t["block_len"])
        blk = buf[bs:bs + bl]
        ctx = ctxs.setdefault(b, {})
        ivs, _ = rm_mask_interval(O, blk, ctx, int(t["start"]), int(t["end"]), int(t["ref_start"]), int(t["ref_end"]), strands, M,
                                  chunk, seed_size, kmer_size, transition,
                                  dict(sub_mat=sub_mat, xdrop=xdrop, hspthresh=hspthresh, noentropy=noentropy, step=step))
        if ivs.size == 0:
            continue
        lines = rm_interval_lines(chr_name, chr_start, bs, [(int(iv["query_start"]), int(iv["len"])) for iv in ivs], markend)
        files["tmp%d.block%d.intervals" % (counters[b], b)] = "".join(lines)
    return files
