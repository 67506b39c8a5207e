"""Deterministic synthetic genomes for the BASELINE.md configurations (numpy only, no I/O).

The generators mimic what matters to the seed-filter-extend path: i.i.d. ACGT background, diverged copies
(substitutions + sparse indels) so real HSPs exist, soft-masked (lower-case) runs, N runs and multi-record
blocks joined by '&' the way src/main.cpp:343-409 lays sequences out in its DRAM arena.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(n, seed):
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def mutate(seq, seed, sub_rate=0.08, indel_every=0, max_indel=10):
    """Diverged copy: i.i.d. substitutions at sub_rate; one 1..max_indel bp insertion or deletion about every
    `indel_every` bases (0 = none)."""
    rng = np.random.default_rng(seed)
    out = seq.copy()
    m = rng.random(out.size) < sub_rate
    # substitute by a DIFFERENT base: add 1..3 in ACGT index space
    idx = np.searchsorted(_ACGT, np.where(out >= 97, out - 32, out))
    idx = np.clip(idx, 0, 3)
    new = _ACGT[(idx + rng.integers(1, 4, size=out.size)) % 4]
    out = np.where(m, new, out)
    if indel_every:
        pieces, p = [], 0
        while p < out.size:
            step = int(rng.integers(indel_every // 2, indel_every * 3 // 2 + 1))
            q = min(out.size, p + step)
            pieces.append(out[p:q])
            k = int(rng.integers(1, max_indel + 1))
            if rng.random() < 0.5:
                pieces.append(_ACGT[rng.integers(0, 4, size=k)])  # insertion
                p = q
            else:
                p = q + k  # deletion
        out = np.concatenate(pieces)
    return out


def soft_mask(seq, seed, frac=0.2, run_lo=200, run_hi=2000):
    """Lower-case runs covering ~frac of the sequence."""
    rng = np.random.default_rng(seed)
    out = seq.copy()
    if frac <= 0:
        return out
    mean_run = (run_lo + run_hi) / 2
    nruns = int(out.size * frac / mean_run)
    starts = rng.integers(0, max(out.size - run_hi, 1), size=nruns)
    lens = rng.integers(run_lo, run_hi + 1, size=nruns)
    for s, l in zip(starts, lens):
        out[s:s + l] |= 0x20
    return out


def join_records(records):
    """Block layout of src/main.cpp:343-409: records separated by a single '&', none after the last."""
    parts = []
    for i, r in enumerate(records):
        if i:
            parts.append(np.frombuffer(b"&", dtype=np.uint8))
        parts.append(r)
    return np.concatenate(parts)


_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtNn&", b"TGCAtgcaNn&"):
    _COMP[_a] = _b


def reverse_complement(seq):
    return _COMP[seq[::-1]]


def invert_blocks(seq, seed, block=20000, frac=0.3):
    """Reverse-complement ~frac of the `block`-sized pieces in place (inversions -> minus-strand HSPs)."""
    rng = np.random.default_rng(seed)
    out = seq.copy()
    for s in range(0, out.size - block + 1, block):
        if rng.random() < frac:
            out[s:s + block] = reverse_complement(out[s:s + block])
    return out


def make_pair(target_len, seed_t, seed_q, sub_rate=0.08, mask_frac=0.0, records=1, indel_every=0,
              n_runs=0, invert_frac=0.3, invert_block=20000):
    """(target_ascii, query_ascii) uint8 arrays."""
    per = target_len // records
    t_recs = [random_dna(per, seed_t + 1000 * i) for i in range(records)]
    q_recs = [mutate(r, seed_q + 1000 * i, sub_rate, indel_every) for i, r in enumerate(t_recs)]
    if invert_frac > 0:
        q_recs = [invert_blocks(r, seed_q + 31 + i, invert_block, invert_frac) for i, r in enumerate(q_recs)]
    if mask_frac > 0:
        t_recs = [soft_mask(r, seed_t + 77 + i, mask_frac) for i, r in enumerate(t_recs)]
        q_recs = [soft_mask(r, seed_q + 77 + i, mask_frac) for i, r in enumerate(q_recs)]
    if n_runs:
        rng = np.random.default_rng(seed_t + 5)
        for r in t_recs + q_recs:
            for _ in range(n_runs):
                s = int(rng.integers(0, max(r.size - 500, 1)))
                r[s:s + int(rng.integers(1, 300))] = ord("N")
    return join_records(t_recs), join_records(q_recs)


# ------------------------------------------------------------------------------------------------------------------
# Realistic-composition stand-in ("lumpy"): what real assemblies have and i.i.d. DNA has not -- a skewed k-mer spectrum.
# Real ce11 / cb4 cannot be fetched here (reference README.md:69-78 downloads them from UCSC); this generator builds a pair
# with the features that decide how seed hits are distributed over buckets and chunks:
#   * an AT-rich 2nd-order Markov background (C. elegans: 35 % GC, CpG-poor, poly-A/T-prone),
#   * microsatellites (1-6 bp units, 50-500 bp) -- most of them soft-masked, as TRF / RepeatMasker leave them,
#   * dispersed repeat families (transposon-like: 0.3-6 kb consensus, hundreds to thousands of fragments at 5-25 %
#     divergence), ~70 % of the copies soft-masked, the rest is what a repeat masker misses,
#   * a few 50-200 kb segmental duplications at 1-3 % divergence (unmasked),
#   * N gaps, several records joined by '&'.
# The query is a diverged, rearranged copy (conserved islands at 4-10 %, neutral sequence at 25-35 %, 30 % of 100 kb blocks
# inverted) with lineage-specific repeat insertions and microsatellites of its own from the SAME families.
# ------------------------------------------------------------------------------------------------------------------
def markov_dna(n, seed, gc=0.355, chains=8192):
    """AT-rich 2nd-order Markov background: `chains` independent chains stepped in lockstep (numpy), concatenated."""
    rng = np.random.default_rng(seed)
    base = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    P = np.tile(base, (16, 1))
    for a in range(4):
        for b in range(4):
            s = 4 * a + b
            P[s, b] *= 1.35 if b in (0, 3) else 1.15          # homopolymer runs (A/T more than C/G)
            if a == b:
                P[s, b] *= 1.25                                # ... growing with the run
            if b == 1:
                P[s, 2] *= 0.45                                # CpG depletion
            if b == 0:
                P[s, 3] *= 1.10                                # AT / TA steps
            if b == 3:
                P[s, 0] *= 0.85
            P[s] /= P[s].sum()
    cum = np.cumsum(P, axis=1)[:, :3]
    steps = (n + chains - 1) // chains
    out = np.empty((steps, chains), dtype=np.uint8)
    prev2 = rng.integers(0, 4, chains)
    prev1 = rng.integers(0, 4, chains)
    for i in range(steps):
        u = rng.random(chains)
        c = cum[4 * prev2 + prev1]
        nxt = (u > c[:, 0]).astype(np.int64) + (u > c[:, 1]) + (u > c[:, 2])
        out[i] = nxt
        prev2, prev1 = prev1, nxt
    return _ACGT[out.T.reshape(-1)[:n]]


def _lumpy_pos(rng, n, hot):
    """Insertion site of a repeat copy: 15 % into one of the record's few hot regions (200 kb clusters of nested insertions), 60 % on
    the chromosome 'arms' (the outer quarters, repeat-rich in C. elegans), 25 % in the centre half."""
    u = rng.random()
    if u < 0.15 and len(hot):
        return int(min(n - 1, hot[int(rng.integers(0, len(hot)))] + rng.integers(0, 200_000)))
    if u < 0.75:
        x = int(rng.integers(0, max(n // 4, 1)))
        return x if rng.random() < 0.5 else n - 1 - x
    return int(n // 4 + rng.integers(0, max(n // 2, 1)))


def _overlay(seq, pos, piece, lower):
    n = min(piece.size, seq.size - pos)
    if n > 0:
        seq[pos:pos + n] = (piece[:n] | 0x20) if lower else piece[:n]


def repeat_families(seed, families=20):
    """Consensus sequences of the dispersed repeat families (0.3-6 kb), shared by target and query."""
    rng = np.random.default_rng(seed)
    return [markov_dna(int(rng.integers(300, 6001)), seed + 17 * (k + 1), gc=0.30 + 0.2 * rng.random(), chains=8) for k in range(families)]


def add_repeats(seq, seed, fams, repeat_frac=0.12, masked=0.7):
    """Overwrite ~repeat_frac of `seq` with fragments of the family consensi, each 5-25 % diverged, `masked` of them lower case."""
    rng = np.random.default_rng(seed)
    budget = int(seq.size * repeat_frac)
    hot = rng.integers(0, max(seq.size - 200_000, 1), max(1, seq.size // 5_000_000))
    per_family = rng.dirichlet(np.full(len(fams), 0.8)) * budget
    k_mut = 0
    for k, cons in enumerate(fams):
        used = 0
        age = 0.05 + 0.20 * rng.random()  # a family's copies share an age: divergence around it
        while used < per_family[k]:
            ln = int(min(cons.size, max(100, rng.exponential(cons.size * 0.35))))
            a = int(rng.integers(0, cons.size - ln + 1))
            d = float(np.clip(rng.normal(age, 0.03), 0.02, 0.30))
            frag = mutate(cons[a:a + ln], seed * 1000003 + k_mut, d, indel_every=0 if ln < 400 else 300)
            k_mut += 1
            if rng.random() < 0.5:
                frag = reverse_complement(frag)
            _overlay(seq, _lumpy_pos(rng, seq.size, hot), frag, rng.random() < masked)
            used += ln
    return seq


_MS_UNITS = [b"A", b"A", b"T", b"AT", b"AT", b"TA", b"AG", b"CT", b"AC", b"GT", b"AAT", b"ATT", b"AAG", b"CTT", b"AAC", b"CAG", b"AAAT", b"ATTT", b"AGAT",
             b"AAGG", b"AAAAT", b"AATAT", b"AAAAG", b"AAAAAT", b"AGATAT", b"TTAGGC"]


def add_microsatellites(seq, seed, every=20000, masked=0.93):
    """One simple repeat (1-6 bp unit, 50-500 bp -- mostly short: 50 + an exponential tail of mean 60 -- ~3 % impure) about every
    `every` bases; `masked` of them lower case, as TRF / RepeatMasker leave an assembly (the unmasked rest is what makes buckets of
    tens of thousands of entries, and -- above ~100 bp, where the score passes 3 x hspthresh and the entropy rule no longer applies
    (src/seed_filter.cu:608) -- one HSP per diagonal of every pair of runs with the same unit)."""
    rng = np.random.default_rng(seed)
    n = max(1, seq.size // every)
    starts = [_lumpy_pos(rng, seq.size, []) for _ in range(n)]
    for i, p in enumerate(starts):
        unit = np.frombuffer(_MS_UNITS[int(rng.integers(0, len(_MS_UNITS)))], dtype=np.uint8)
        ln = int(min(500, 50 + rng.exponential(60)))
        run = np.tile(unit, ln // unit.size + 1)[:ln].copy()
        bad = rng.random(ln) < 0.03
        run[bad] = _ACGT[rng.integers(0, 4, int(bad.sum()))]
        _overlay(seq, int(p), run, rng.random() < masked)
    return seq


def add_segmental_duplications(seq, seed, count=5):
    rng = np.random.default_rng(seed)
    for i in range(count):
        ln = int(rng.integers(50_000, 200_001))
        if seq.size < 4 * ln:
            ln = max(1000, seq.size // 8)
        src = int(rng.integers(0, seq.size - ln))
        dst = int(rng.integers(0, seq.size - ln))
        dup = mutate(seq[src:src + ln], seed * 7919 + i, 0.01 + 0.02 * rng.random(), indel_every=2000)
        if i % 2:
            dup = reverse_complement(dup)
        _overlay(seq, dst, dup, False)
    return seq


def add_gaps(seq, seed, count=10):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        ln = int(rng.choice([100, 1000, 10000, 50000]))
        p = int(rng.integers(0, max(seq.size - ln, 1)))
        seq[p:p + min(ln, seq.size - p)] = ord("N")
    return seq


def diverge_realistic(seq, seed, conserved_frac=0.35):
    """Diverged copy with conservation structure: islands of 100-1500 bp at 4-10 % substitutions (sparse indels) inside neutral
    sequence at 25-35 % (beyond what 12of19 seeds and an ungapped extension recover); case is kept."""
    rng = np.random.default_rng(seed)
    n = seq.size
    rate = np.empty(n, dtype=np.float32)
    p = 0
    while p < n:
        island = rng.random() < conserved_frac / (conserved_frac + (1 - conserved_frac) * 800 / 2500)  # island mean 800, desert mean 2500
        ln = int(rng.integers(100, 1501)) if island else int(rng.integers(500, 4501))
        rate[p:p + ln] = (0.04 + 0.06 * rng.random()) if island else (0.25 + 0.10 * rng.random())
        p += ln
    lower = (seq >= 97) & (seq <= 122)
    up = np.where(lower, seq - 32, seq)
    idx = np.clip(np.searchsorted(_ACGT, up), 0, 3)
    sub = (rng.random(n) < rate) & (up != ord("N"))
    new = _ACGT[(idx + rng.integers(1, 4, size=n)) % 4]
    out = np.where(sub, new, up)
    out = np.where(lower, out | 0x20, out).astype(np.uint8)
    # sparse indels (one per ~1.5 kb): deletions / insertions of 1-10 bp
    pieces, q = [], 0
    while q < n:
        step = int(rng.integers(700, 2300))
        e = min(n, q + step)
        pieces.append(out[q:e])
        k = int(rng.integers(1, 11))
        if rng.random() < 0.5:
            pieces.append(_ACGT[rng.integers(0, 4, size=k)])
            q = e
        else:
            q = e + k
    return np.concatenate(pieces)


def make_realistic(target_len, seed_t=11, seed_q=12, records=7, repeat_frac=0.12, ms_every=10000):
    """(target_ascii, query_ascii): the lumpy ce11 x cb4 stand-in described above.  Deterministic."""
    fams = repeat_families(1000 + seed_t)
    per = target_len // records
    t_recs, q_recs = [], []
    for i in range(records):
        t = markov_dna(per, seed_t + 1000 * i)
        t = add_repeats(t, seed_t + 1000 * i + 1, fams, repeat_frac)
        t = add_segmental_duplications(t, seed_t + 1000 * i + 2, count=max(1, 5 // records + (1 if i < 5 % records else 0)))
        t = add_microsatellites(t, seed_t + 1000 * i + 3, ms_every)
        t = add_gaps(t, seed_t + 1000 * i + 4, count=2)
        q = diverge_realistic(t, seed_q + 1000 * i)
        q = invert_blocks(q, seed_q + 1000 * i + 5, 100_000, 0.3)
        q = add_repeats(q, seed_q + 1000 * i + 1, fams, repeat_frac * 0.4)   # lineage-specific insertions of the same families
        q = add_microsatellites(q, seed_q + 1000 * i + 3, ms_every * 2)
        t_recs.append(t)
        q_recs.append(q)
    return join_records(t_recs), join_records(q_recs)


def human_target_block(tlen, idx):
    """One target block of the human-scale stand-in (BASELINE configs[2]): `tlen` bases of uniform DNA, 30 % in soft-masked runs of
    0.2-2 kb, as 4 records; block `idx` of a genome (its own random stream)."""
    t = random_dna(tlen, 5 + 100 * idx)
    t = soft_mask(t, 6 + 100 * idx, 0.3, 200, 2000)
    per = tlen // 4
    return join_records([t[i * per:(i + 1) * per] for i in range(4)])


def human_query_block(target, qlen, idx):
    """The query block that goes with target block `idx`: 1-10 Mbp pieces of it, 1.2 % diverged, shuffled, every third piece
    inverted, `qlen` bases in all."""
    import numpy as np
    rng = np.random.default_rng(7 + idx)
    pieces, total, i = [], 0, 0
    while total < qlen:
        n = int(rng.integers(1_000_000, 10_000_001))
        n = min(n, qlen - total)
        p = int(rng.integers(0, target.size - n))
        seg = mutate(target[p:p + n], 1000 + i + 100 * idx, 0.012)
        pieces.append(reverse_complement(seg) if i % 3 == 0 else seg)
        total += n
        i += 1
    return np.concatenate(pieces)


def human_block_pair(tlen, qlen, idx=0):
    target = human_target_block(tlen, idx)
    return target, human_query_block(target, qlen, idx)
