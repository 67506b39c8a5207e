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
